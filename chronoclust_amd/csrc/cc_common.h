// cc_common.h — shared device/host declarations of the gfx950 ChronoClust hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cc_host.h"

#define CC_KIND_PCORE 0
#define CC_KIND_OUTLIER 1
#define CC_KIND_DEAD 2

#define CC_MAX_ROUNDS 8
#define CC_T_UNKNOWN (-2)
#define CC_T_NONE (-3)  // relaxed multi-GPU mode: the point is set aside (no MC absorbs it and none may be created)
#define CC_IDX_INF 0x7fffffff

// One candidate of a per-point argmin: (distance, list-order key) ordered lexicographically
// (strict `<` + "first in list order wins", hddstream.py:326/373), plus the row it refers to.
// slot -1: no candidate.  slot CC_SLOT_BOUND: not a row but a BOUND left by the pruned snapshot scan (k_scan_p): every
// row of the kind that is not listed before it has a distance >= dist.  Its key is -1, so that on an exact tie of
// distances the bound comes first: a bound in second place says "the second-best row is unknown, but not closer than
// this", and the decision procedure treats it like a second-best candidate that is dirty (k_decide, state 2).
#define CC_SLOT_BOUND (-2)
struct __attribute__((aligned(16))) Cand {
    double dist;
    int key;
    int slot;
};

// Microcluster table in HBM, structure of arrays, row r = one microcluster
// (objects/microcluster.py:71-81: CF1, CF2, cumulative_weight, cluster_centroids, preferred_dimension_vector).
struct Table {
    double* cf1;   // [cap, d]
    double* cf2;   // [cap, d]
    double* cen;   // [cap, d]  stored centroid (NOT recomputed after decay, hddstream.py:283-286)
    double* pref;  // [cap, d]  1.0 or k
    double* scl;   // [cap, d]  the operand the distance uses: 1/pref when k is a power of two, else pref
    double* w;     // [cap]
    int* kind;     // [cap]  CC_KIND_*
    int* key;      // [cap]  position in its Python list (pcore_MC or outlier_MC) as an order key
    long long* id;   // [cap]  current id (pcore id or outlier id)
    long long* uid;  // [cap]  prev_outlier_id = creation number
    // Validation stamps, written by k_decide with atomic max for the round that follows it:
    //   touch = stamp << 20 | (0xFFFFF - first window point that targets this MC)
    //   last  = stamp << 20 | last window point that targets this MC
    // Two copies each, indexed by round parity: round r reads copy r & 1 while k_decide fills copy (r + 1) & 1.
    unsigned long long* touch;  // [2][cap]
    unsigned long long* last;   // [2][cap]
    // Lookahead (see Carry): carry_of = window_seq << 20 | carried index, for the rows the previous window changed
    unsigned long long* carry_of;  // [cap]
    // Chain members.  k_decide appends every window point to the list of the MC it targets (arrival order of the
    // atomics, i.e. unordered): cnt = stamp << 24 | members, memb[CC_CHAIN_MEMB * row + i] = the first CC_CHAIN_MEMB of them.  The point
    // that heads the chain sorts them (k_chain); longer chains are found by scanning the claims.
    unsigned long long* cnt;  // [cap]
    int* memb;                // [cap, CC_CHAIN_MEMB]
    int* clen;                // [cap] members of the MC's chain as k_chain walked it (valid while `touch` carries the round's stamp)
    // Heavy rows (skewed populations: a MC that takes a large share of a window's points).  Per claimant k_decide issues
    // three atomics on the MC's words (touch, last, cnt); thousands of claimants of one MC serialise there (~20 ns each).
    // A row whose chain was long (> CC_CHAIN_MEMB claimants) is marked heavy at the end of that window (k_commit_a:
    // Ctl::heavy_list, at most CC_HEAVY_CAP rows); from the next batch on its claimants skip the atomics and
    // k_claims_heavy - one workgroup per heavy row, behind every k_decide whose claims are replayed - gathers first /
    // last claimant, count and members from the claims, in the formats k_decide writes.  0: not heavy, 1: heavy,
    // 2: heavy, to be dropped at the end of this window (its chain was short again).
    int* heavy;               // [cap]
    size_t cap;                 // rows allocated (offset of the second copy)
};
#define CC_HEAVY_CAP 64   // heavy rows at a time (= workgroups of a k_claims_heavy launch)
#define CC_HEAVY_NEW 32   // candidates a window may nominate

// Candidate slots >= CC_CAR_BASE refer to carried rows (index = slot - CC_CAR_BASE), below to version rows.
#define CC_CAR_BASE (1 << 24)
// Largest window (points validated together): k_commit_a ranks a window's creations and promotions in 16-bit counts and
// stages one flag byte per point in LDS.
#define CC_MAX_WINDOW 49152
#define CC_CHAIN_MEMB 32
// Chains longer than the member list: on a table of more than 1 024 rows (no k_claims) the claimant that finds the list
// full enters the MC in a list of long chains (at most CC_LONG_CAP per round) and marks the counter word with
// CC_LONG_LISTED; k_chain leaves such a chain to k_chain_long, which replays it in batches of 128 steps instead of one
// step after the other.
#define CC_LONG_CAP 512
#define CC_LONG_LISTED (1ull << 23)
#ifndef CC_CHAIN_AHEAD
#define CC_CHAIN_AHEAD 4  // k_chain: points of a chain requested ahead of the step that absorbs them
#endif

// Lookahead.  While window W is validated, the snapshot scan of window W + 1 already runs - against the table as
// it is before W's commit.  What that scan could not see is exactly the set of rows W's commit changes or adds:
// the carry set.  For W + 1 they are treated like version rows that precede its first point: a carried row is
// the live state of its MC until a point of W + 1 targets it.  Indexed by W's point index (only the last version
// of every chain is alive, the rest is CC_KIND_DEAD).  Written by k_commit_b.
struct Carry {
    double* cf1;   // [B, d] state as committed
    double* cf2;
    double* cen;
    double* pref;
    double* scl;
    double* w;     // [B]
    int* kind;
    int* key;      // final list-order key
    int* slot;     // table row
    double* c0;    // [B, d] centroid of the row before the commit (= in the snapshot W + 1 was scanned against)
    double* w0;    // [B, d] 1 / pref of the row before the commit
    int* kind0;    // kind before the commit; CC_KIND_DEAD: the row did not exist
    double* dsq;   // [B] squared displacement of the committed centroid from c0 in the w0 metric (see Versions::dsq)
    unsigned long long* tile_dsq;  // [B / 16, CC_DSQ_STRIDE] per 16 rows and class: max of dsq (cc_dsq_code: 0 = no such row)
};

// Version rows of the current window: row j = state of point j's target microcluster right after point j
// was processed (speculatively).  Same column meaning as Table.
struct Versions {
    double* cf1;
    double* cf2;
    double* cen;
    double* pref;
    double* scl;
    double* w;
    int* kind;
    int* key;
    int* next;  // next point of the window with the same target (CC_IDX_INF if none; == own index if row unused)
    int* upg;   // index of the window point whose add promoted the microcluster, or -1
    int* acc;   // 1 if point j was absorbed (radius test passed or new microcluster)
    int* tgt;   // target id of the chain this row belongs to
    // squared displacement of the version's centroid from its MC's window-start centroid, in the window-start
    // metric; +inf for MCs created inside the window or whose preferred dimensions changed (k_chain); negated for
    // MCs promoted since the snapshot (class 2, see CC_DSQ_STRIDE).
    // tile_dsq[(i / 16) * CC_DSQ_STRIDE + class] = max over the rows of that class among 16 (cc_dsq_code: 0 = none).
    double* dsq;
    unsigned long long* tile_dsq;
    double* tau;  // [B, CC_TAU_STRIDE] per window point and class: a version of the class with sqrt(dsq) below this cannot matter to it (k_dseed)
    // per 64-point tile: 1 if no version row (skip) / no carried row (skip_car) can matter to any of its points,
    // so the dirty scan of the tile is not run and k_decide takes the seeds (k_dseed)
    int* skip;
    int* skip_car;
    // per window point: CC_FLAG_* (k_dseed).  Which kinds of version / carried rows could matter to it beyond the seeds:
    // k_decide refuses to decide a point (CC_T_UNKNOWN: the window commits up to it) when a stage it has to evaluate
    // needs rows no dirty scan covered - the tile's scan was skipped, or the host has stopped launching them
    int* unsafe;
};

// Read-only view the scan kernel walks (either the table or the version rows).
struct Rows {
    const double* cen;
    const double* scl;   // distance operand per dimension (see Table::scl)
    const double* pref;
    const double* cf1;
    const double* cf2;
    const double* w;
    const int* kind;
    const int* key;
    const int* next;  // only for version rows
    const unsigned long long* tile_dsq;  // only for version / carried rows (see Versions): [rows / 16, 2]
    const double* dsq;                   // only for version / carried rows: per-row squared displacement
    const double* tau;                   // only for version rows
    const int* skip;                     // dirty scans: per 64-point tile, 1 = nothing to do (Versions::skip)
    const int* plist;                    // sparse dirty scans: the window points to scan, compacted (Ctl::n_sparse of them); else null
    const int* slot;                     // only for carried rows: table row
    const unsigned long long* touch;     // only for carried rows: Table::touch
    size_t cap;                          //                         Table::cap
};

struct Ctl {
    long long cursor;    // next point of the resident buffer to be committed
    long long n_points;  // end of the range this call clusters
    long long xt_stride; // rows of the resident buffer (column stride of its dimension-major copy)
    int no_create;       // 1: points that no MC absorbs are set aside instead of creating one (CC_T_NONE)
    int pad1;
    int d;
    int m_rows;          // table rows in use
    int n_pkeys, n_okeys;
    long long pcore_last_id, outlier_last_id;  // hddstream.py:63-64
    unsigned long long window_seq;
    int win_cfg;         // configured window size
    int win_b;           // size of the current window (0: nothing left)
    int max_rounds;
    int last_round;      // last validation round that ran for the current window
    int fc[CC_MAX_ROUNDS + 2];  // fc[r]: first point whose decision changed in round r (>= win_b: converged)
    // any_new[r]: some decision of round r creates a MC (k_decide); any_up[r]: some add replayed in round r promotes
    // one (k_chain).  Plain flags; while both are clear k_commit_a has nothing to rank.
    int any_new[CC_MAX_ROUNDS + 2], any_up[CC_MAX_ROUNDS + 2];
    // n_long[r]: chains of existing MCs with more than CC_CHAIN_MEMB claimants that round r replays with k_chain_long
    // (k_decide of round r - 1 lists them, see CC_LONG_LISTED); stat_long: how many such chains the call has seen
    int n_long[CC_MAX_ROUNDS + 2];
    long long stat_long;
    // parameters (cc_params, see include/chronoclust_hip.h)
    double eps_sq, delta_sq, k, inv_k, beta_mu, mu, omicron;
    int pi;
    int filter;  // pi < d: the pdim filter of hddstream.py:317-321 is not vacuous
    int pow2;    // k is a power of two: x / k == x * (1/k) bit for bit
    int pad0;
    // lookahead
    int mode;    // current window: 0 = its snapshot scan saw the table as it is (fresh), 1 = as it was one commit earlier
    int car_n;   // rows of the carry set (= size of the previous window) when mode == 1, else 0
    int la_on;   // set by the host: lookahead scans are being enqueued
    // In a lookahead batch only the first window can be scanned in place (the host enqueues that launch when it
    // knows the window is a fresh one).  If a later window cannot use its lookahead scan (the previous one stopped
    // short), the rest of the batch idles: win_b = 0 and the withheld size waits here for the host.
    int stall_b;
    // the window after the current one, per parity of its window_seq: what its lookahead scan covers
    long long la_cursor[2];
    int la_b[2];
    int la_rows[2];
    // statistics
    long long stat_windows, stat_rounds, stat_truncated;
    long long stat_table_rows;  // sum over windows of the table rows scanned
    double stat_pair_rows;      // sum over windows of (window points x table rows)
    long long round_hist[CC_MAX_ROUNDS + 2];  // windows by the validation round they ended in
    long long stat_lookahead;  // windows whose snapshot scan ran ahead (mode 1)
    long long stat_trunc_unknown;  // truncated windows that stopped at a point whose decision could not be made (the rest: one more round needed)
    long long stat_unprovable, stat_unsafe;  // points whose live versions k_dseed could not locate / that needed the dirty scans
    long long stat_tiles, stat_dirty_tiles;  // 64-point tiles validated / of those, tiles whose dirty scan had to run
    long long stat_seq_points;               // points taken by the sequential kernel (k_seq)
    long long stat_seq_clk, stat_seq_wall;   // ... its shader-clock cycles / constant 100 MHz ticks (trace)
    // pruned snapshot scans (k_scan_p): (wave, row) pairs visited / of those, pairs whose distance was evaluated in full
    // (a sample: the waves of every window's first point tile)
    unsigned long long stat_prune_rows, stat_prune_full;
    // the round's list of points for the sparse dirty scans (k_dseed fills it, k_chain of the round resets the count)
    int n_sparse;
    int pad2;
    // Pruned snapshot scans with a table-wide threshold (k_scan_p, GUESS; cc_scan.h): tg[q][K] = mean snapshot distance of
    // the points of an earlier window that joined a MC of kind K (k_commit_a of window V writes slot V & 1; the scan of
    // window U reads slot U & 1, i.e. what window U - 2 left: no commit writes a slot while a scan may read it),
    // tg_ok[q][K] = 1 once such a mean exists.  n_missed[q]: points of the window of parity q whose own MC the guessed
    // threshold missed (k_missed lists them, the seeded chain then runs for them alone); coord_bound: bits of a double
    // >= every |coordinate| of the resident points and of the table's centroids (k_check_finite / k_rebuild_scl).
    double tg[2][2];
    int tg_ok[2][2];
    int n_missed[2];
    unsigned long long cen_absmax;   // bits of the largest |centroid coordinate| in the table at the start of the call
    double x_absmax;                 // the largest |coordinate| of the resident points (host, from the upload)
    long long stat_missed;
    // What the scan kernels count per window parity - they may run on the lookahead stream, whose progress the host does
    // not wait for before it reads this block - and k_decide (round 0 of that window, main stream) moves into the counters
    // the window policy reads, so that those are a function of the windows validated so far and nothing else:
    // pstat[q] = {(wave, row) pairs visited, pairs completed} of the window's pruned scan (k_scan_p's sample; on the exact
    // multi-GPU path it travels with the rank's records instead, k_merge_partials), n_missed_all[q] = points its guessed
    // thresholds missed (k_missed / k_missed_g; not cut at the list's capacity).
    unsigned long long pstat[2][2];
    int n_missed_all[2];
    int seq_rest;  // k_seq_r -> k_seq: points of the stint the register kernel left (its capacity was reached, or the table did not fit)
    int pad3;
    long long stat_seq_r_points;  // of stat_seq_points: taken by k_seq_r
    // seed_at: absolute index of the point a window was cut short at because it could not be decided (k_commit_a; -1
    // otherwise).  The next window starts there; if its scan runs with guessed thresholds, k_missed / k_missed_g put that
    // first point on the list of the seeded chain whatever its marks say: a point whose pcore list was resolved by the guess
    // but whose OUTLIER list starts with a bound (its pcore stage fails, no outlier MC lies within the guess - the first
    // point of a new population beside an existing one) would otherwise be refused again and again.
    long long seed_at;
    // heavy rows (Table::heavy): the list k_claims_heavy works on, and this window's nominations (k_decide: the claimant
    // that finds a chain's member list full); k_commit_a merges the nominations into the list and drops rows marked 2
    // rdiff[r]: some decision of round r may differ from the claim it validates (k_dseed of round r; reset by k_chain of
    // round r).  0 after k_dseed: k_decide of round r would repeat every claim - it returns at once (a "quiet" round).
    int rdiff[CC_MAX_ROUNDS + 2];
    int n_heavy, n_heavy_new;
    int heavy_list[CC_HEAVY_CAP];
    int heavy_new[CC_HEAVY_NEW];
#ifdef CC_LONG_TIMERS
    unsigned long long dbg_long[8];  // build variant: shader cycles per phase of k_chain_long (workgroup 0 of each launch)
#endif
#ifdef CC_ROUND_DEBUG
    // build variant: decisions of validation round r that differ from the claim they validate, by what changed -
    // [r][0] refused (CC_T_UNKNOWN), [r][1] "create" -> joins a MC created inside the window, [r][2] "create" -> joins a table
    // row, [r][3] joins -> "create", [r][4] another MC, [r][5] all decisions of the round
    unsigned long long dbg_round[CC_MAX_ROUNDS + 2][6];
#endif
};

// Displacement classes of a version row / carried row relative to the snapshot its window was scanned against
// (Versions::tile_dsq, Carry::tile_dsq: CC_DSQ_STRIDE words per 16 rows):
//   0  pcore now and in the snapshot       competes in a point's pcore list, bounded through that list's second-best
//   1  outlier now and in the snapshot     ... outlier list ...
//   2  pcore now, OUTLIER in the snapshot  (promoted since: hddstream.py:416-430) competes in the pcore list, but its
//      snapshot distance is bounded through the point's OUTLIER list - its dsq is stored negated (see cc_dsq_class)
// Rows without a bound (new microclusters, changed preferred dimensions) carry dsq = +inf in class 0 / 1.
#define CC_MISSED_CAP 2048  // points per window the seeded chain may run for after a guessed-threshold scan (k_missed)
#define CC_DSQ_STRIDE 4
#define CC_TAU_STRIDE 4   // Versions::tau: thresholds of classes 0, 1, 2 per window point (+ 1 pad)
// Versions::unsafe, per window point (k_dseed -> k_decide):
#define CC_FLAG_U0 1       // some version row of class 0 / 2 could matter to the point beyond the seeds
#define CC_FLAG_U1 2       // ... of class 1
#define CC_FLAG_C0 4       // the same for the carried rows
#define CC_FLAG_C1 8
#define CC_FLAG_N1SKIP 32  // the outlier-kind threshold was withheld (the point is expected to join a pcore MC): a dirty
                           // scan that ran does not cover the outlier kind for this point
#define CC_FLAG_SPARSE 64  // the point is on the round's list of points whose dirty scans run point by point (the sparse
                           // dirty scans: compacted tiles of such points, both kinds covered) although its tile's did not

// (cc_shard_share / cc_shard_range: cc_host.h - plain C++, shared with the host-only sanitizer build)
__host__ __device__ inline bool cand_less(double ad, int ak, double bd, int bk)
{
    return ad < bd || (ad == bd && ak < bk);
}
