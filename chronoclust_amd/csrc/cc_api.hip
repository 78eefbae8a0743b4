// cc_api.hip — host side of the C-ABI declared in include/chronoclust_hip.h.
// Owns the HBM-resident state (microcluster table, window buffers, points, labels) and enqueues the gfx950
// kernels of cc_online.h / cc_offline.h on one HIP stream.  No CPU fallback exists for any kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <limits>
#include <optional>
#include <vector>

#include "../../include/chronoclust_hip.h"
#include "cc_common.h"
#include "cc_online.h"
#include "cc_offline.h"
#include "cc_comm.h"
#include "cc_csv.h"
#include "cc_policy.h"

namespace {

struct HipErr {
    hipError_t e;
    const char* what;
};

#define HIPCHK(call)                                  \
    do {                                              \
        hipError_t _e = (call);                       \
        if (_e != hipSuccess) throw HipErr{_e, #call}; \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void ensure(size_t count)
    {
        if (count <= n && p) return;
        release();
        size_t want = std::max<size_t>(count, 1);
        if (hipMalloc((void**)&p, want * sizeof(T)) != hipSuccess) {
            p = nullptr;
            throw HipErr{hipErrorOutOfMemory, "hipMalloc"};
        }
        n = want;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

struct TableStore {
    DevBuf<double> cf1, cf2, cen, pref, scl, w;
    DevBuf<int> kind, key;
    DevBuf<long long> id, uid;
    DevBuf<unsigned long long> touch, last, carry_of, cnt;
    DevBuf<int> memb, clen, heavy;
    size_t cap = 0;
    int d = 0;
    void alloc(size_t rows, int dim)
    {
        cf1.ensure(rows * dim); cf2.ensure(rows * dim); cen.ensure(rows * dim); pref.ensure(rows * dim);
        scl.ensure(rows * dim);
        w.ensure(rows); kind.ensure(rows); key.ensure(rows); id.ensure(rows); uid.ensure(rows); touch.ensure(2 * rows); last.ensure(2 * rows);
        carry_of.ensure(rows); cnt.ensure(rows); memb.ensure(rows * CC_CHAIN_MEMB); clen.ensure(rows);
        heavy.ensure(rows);
        cap = rows;
        d = dim;
    }
    Table view() const { return Table{cf1.p, cf2.p, cen.p, pref.p, scl.p, w.p, kind.p, key.p, id.p, uid.p, touch.p, last.p, carry_of.p, cnt.p, memb.p, clen.p, heavy.p, cap}; }
    void swap(TableStore& o)
    {
        std::swap(cf1.p, o.cf1.p); std::swap(cf1.n, o.cf1.n); std::swap(cf2.p, o.cf2.p); std::swap(cf2.n, o.cf2.n);
        std::swap(cen.p, o.cen.p); std::swap(cen.n, o.cen.n); std::swap(pref.p, o.pref.p); std::swap(pref.n, o.pref.n);
        std::swap(scl.p, o.scl.p); std::swap(scl.n, o.scl.n);
        std::swap(w.p, o.w.p); std::swap(w.n, o.w.n); std::swap(kind.p, o.kind.p); std::swap(kind.n, o.kind.n);
        std::swap(key.p, o.key.p); std::swap(key.n, o.key.n); std::swap(id.p, o.id.p); std::swap(id.n, o.id.n);
        std::swap(uid.p, o.uid.p); std::swap(uid.n, o.uid.n); std::swap(touch.p, o.touch.p); std::swap(touch.n, o.touch.n);
        std::swap(last.p, o.last.p); std::swap(last.n, o.last.n);
        std::swap(carry_of.p, o.carry_of.p); std::swap(carry_of.n, o.carry_of.n);
        std::swap(cnt.p, o.cnt.p); std::swap(cnt.n, o.cnt.n); std::swap(memb.p, o.memb.p); std::swap(memb.n, o.memb.n);
        std::swap(clen.p, o.clen.p); std::swap(clen.n, o.clen.n);
        std::swap(heavy.p, o.heavy.p); std::swap(heavy.n, o.heavy.n);
        std::swap(cap, o.cap); std::swap(d, o.d);
    }
};

// The clusters of the last offline phase, flat: cluster c = members [off[c], off[c + 1]) of `mem`, pcore list positions in
// merge order (a vector per cluster cost 5 000 allocations per call at C2: 350 us of the ordered expansion's 380)
struct HostClusters {
    std::vector<int> mem, off{0};
    size_t size() const { return off.size() - 1; }
    void clear() { mem.clear(); off.assign(1, 0); }
};

}  // namespace

// Page-locked host scratch for the small read-backs and uploads of a call (offline phase: row order, flags, counts, lists):
// a copy to or from pageable memory is driven by the host thread - it first waits for the stream to drain -, one to or from
// page-locked memory is a stream operation; six to ten of them per cc_offline call were 300 us of idle device.  Bump
// allocation per call (`reset`), never freed in between; growing it (rare) drains the stream first.
struct PinArena {
    char* p = nullptr;
    size_t cap = 0, used = 0;
    void reset() { used = 0; }
    void reserve(size_t bytes)
    {
        if (bytes <= cap) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(bytes, (size_t)1 << 20);
        if (hipHostMalloc((void**)&p, want, hipHostMallocDefault) != hipSuccess) {
            p = nullptr;
            throw HipErr{hipErrorOutOfMemory, "hipHostMalloc"};
        }
        cap = want;
    }
    template <typename T>
    T* take(size_t n)  // (within what reserve() was given)
    {
        used = (used + 63) & ~(size_t)63;
        T* r = reinterpret_cast<T*>(p + used);
        used += std::max<size_t>(n, 1) * sizeof(T);
        if (used > cap) throw HipErr{hipErrorOutOfMemory, "page-locked scratch exhausted"};
        return r;
    }
    ~PinArena() { if (p) (void)hipHostFree(p); }
};

struct cc_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // lookahead scans
    std::string err;
    cc_params par{};
    bool have_par = false;
    cc_tuning tun{};
    cc_stats stats{};

    int d = 0;
    TableStore tab, tab2;
    Ctl hc{};  // host mirror of the device control block
    Ctl* hc_pin = nullptr;  // two page-locked staging blocks for it (read-back between batches / restart push)
    DevBuf<Ctl> ctl;
    bool tainted = false;  // a pref value outside {1, k} may be present -> never take the x * (1/k) shortcut
    int adapt_win = 0;      // window size the last call settled at (0: none yet)
    int clean_batches = 0;  // consecutive batches without a truncated window
    int since_shrink = 1000;  // batches since the window was last shrunk
    // threads per workgroup of the validation kernels (32-lane groups x 32).  k_chain in workgroups of one wave: beside a
    // lookahead scan a small workgroup finds room on a single SIMD (measured: 4 % per window in steady state)
    int chain_threads = 64, decide_threads = 256, commit_threads = 256;
    bool allow_scan_u = true;  // k_scan_u where it applies (CHRONOCLUST_HIP_SCANU=0: always k_scan)
    // the pruned snapshot scan (k_seed / k_seed_merge / k_scan_p) where k_scan_u applies and d > 8:
    // CHRONOCLUST_HIP_PRUNE = 0 never, 1 (default) while it pays (the device counts the rows it still evaluates in
    // full), 2 always; CHRONOCLUST_HIP_PRUNE_F = threshold factor (default 16)
    int prune_mode = 1;
    double prune_F = 16.0;
    bool prune_now = false;   // this batch's snapshot scans are pruned ones (set per batch by online_range)
    bool group_guess_now = false;     // ... and the missed points derived from the gathered records (k_missed_g), see timed_scan
    bool group_guess_always = false;  // CHRONOCLUST_HIP_GROUP_GUESS=1: also in a group of one rank
    bool lean_now = false;    // ... guessed thresholds without k_missed / the seeded chain for missed points (cc_policy_decision::prune == 3)
    bool guess_now = false;   // ... with guessed thresholds (k_scan_p + k_missed + the seeded chain for the missed points)
    bool probe_now = false;   // the next plain scan also runs the pruned chain on 128 points (cc_policy_decision::probe)
    DevBuf<Cand> probe_part;  // ... into these scratch partials
    bool allow_guess = true;  // CHRONOCLUST_HIP_GUESS=0: seeded thresholds only
    bool allow_lean = true;   // CHRONOCLUST_HIP_LEAN=0: guessed scans always list and rescan the points they missed
    int force_prune_rows = 0;  // CHRONOCLUST_HIP_FORCE_PRUNE_ROWS: pruned scans whatever the phase from this many table rows on (0, the default: never forced - measured: it pays nowhere yet)
    bool allow_seed16 = false;  // CHRONOCLUST_HIP_SEED16=1: the seeds of a seeded pruned chain from the matrix cores (k_seed16) with the tight threshold, not from k_seed (eight-dimension prefix scores, F x the nearest) - measured a wash at C2's shapes, DESIGN section 9
    bool allow_prune_general = true;  // CHRONOCLUST_HIP_PRUNE_GENERAL=0: no pruned scans where the pdim filter is on or k is not a power of two
    bool la_pruned = false;   // CHRONOCLUST_HIP_LA_PRUNED=1: lookahead scans also while the scans are pruned chains on one GPU
    bool allow_probe = true;  // CHRONOCLUST_HIP_PROBE=0: pruned scans are retried blindly after a stretch of points
    DevBuf<unsigned long long> found;  // [2][CC_MAX_WINDOW / 64] per point tile: the points a guessed-threshold scan found a pcore MC for
    DevBuf<int> missed;                // [2][CC_MISSED_CAP] the others, listed by k_missed (two window parities)
    double x_absmax = 0.0;             // the largest |coordinate| of the resident points (k_check_finite)
    int prune_rounds4 = 0;    // workgroups per CU a pruned scan is split into (CHRONOCLUST_HIP_PRUNE_WGS; 0: by width, see S)
    DevBuf<SeedCand> spart;   // [2][window, S, 2]  prefix-score winners per workgroup sub-range and kind (two window parities)
    DevBuf<double> thr;       // [2][window, 2]     abandon thresholds per point and kind
    DevBuf<float> thr32;      // [2][window, 2]     ... and what phase A's single-precision prefix sums are compared with
    DevBuf<unsigned long long> cmax;  // [2]        largest |centroid coordinate| of the scanned prefixes (bits of a double)
    DevBuf<cc_h8> a16;        // [2][(table capacity + 64) x 4]  k_prefix16: the table rows as half-precision operands of the MFMA prefix test (two window parities)
    DevBuf<Prefix16Hdr> hdr16;  // [2]              ... origin and scale they were converted with
    size_t a16_stride = 0;
    DevBuf<unsigned> masks;   // [2][tiles of 128 points, sub-ranges, words per sub-range]  k_scan_a's survivor masks (two window parities)
    size_t mask_stride = 0;
    // CHRONOCLUST_HIP_SCANA: 0 phase A inside k_scan_p (one point per lane, the round-3 form), 2 always as a kernel of its own
    // (k_scan_a: two points per lane), 1 (default) k_scan_a from 10 000 table rows on: at 5 000 rows the second launch and
    // phase B's own prologue cost what the cheaper phase A saves, at 50 000 the scan launch is 21 % shorter
    // (profiles/r05_tool_scan_a.txt)
    int split_a_mode = 1;
    // a pruned scan's sample {rows visited, rows completed} per window parity: Ctl::pstat, as the kernels take it
    unsigned long long* pstat_p() const { return (unsigned long long*)((char*)ctl.p + offsetof(Ctl, pstat)); }
    size_t spart_stride = 0, thr_stride = 0;
    bool trace = false;     // CHRONOCLUST_HIP_TRACE=1: one stderr line per batch of windows
    bool allow_nodirty = true;  // CHRONOCLUST_HIP_NODIRTY=0: always launch the dirty scans
    bool allow_claims = true;   // CHRONOCLUST_HIP_CLAIMS=0: k_decide's atomics whatever the table size
    bool allow_long = true;     // CHRONOCLUST_HIP_LONGCHAINS=0: every chain replayed by k_chain
    bool allow_quiet = true;    // CHRONOCLUST_HIP_QUIET=0: k_decide re-derives every decision of a validation round even when k_dseed has shown that all of them repeat their claims
    bool allow_missed_plain = true;  // CHRONOCLUST_HIP_MISSED_PLAIN=0: the points a guessed threshold missed go through the seeded chain, not k_scan_u
    int p3_listed_rows = 10000;  // CHRONOCLUST_HIP_P3_LISTED=<rows>: k_scan_p3 lists the rows phase A keeps from this many table rows on
    bool allow_scan_p3 = true;   // CHRONOCLUST_HIP_SCANP3=0: the prefix test of the window's pruned scan on the VALU (k_scan_p2), not the matrix cores
    bool allow_scan_p2 = true;  // CHRONOCLUST_HIP_SCANP2=0: the pruned scan of a window as k_scan_p (one point per lane) instead of k_scan_p2
    bool allow_link = true;     // CHRONOCLUST_HIP_LINK=0: round 0 does not link the points that decide "create" among themselves (cc_link.h)
    DevBuf<int> link_near;      // [window] k_link_scan: per window point that decided "create", the earliest such point before it that would absorb it
    bool allow_heavy = true;    // CHRONOCLUST_HIP_HEAVY=0: k_decide's atomics also for rows that take a large share of a window
    bool allow_seq_r = true;    // CHRONOCLUST_HIP_SEQR=0: the sequential kernel with the table in LDS whatever d
    bool allow_seq_g = true;    // CHRONOCLUST_HIP_SEQG=0: no sequential kernel beyond the LDS image (k_seq_g, the table in HBM)
    DevBuf<int> seq_lists;      // [2][table capacity] k_seq_g: rows of the pcore MCs / of the outlier MCs
    DevBuf<double> seq_img;     // [4][d][table capacity] k_seq_g: dimension-major copy of the rows it scans (centroid, operand, CF1, CF2)
    int allow_sparse = 128;     // CHRONOCLUST_HIP_SPARSE=0: no sparse dirty scans (the tiles' scans or none); N: while at most one point in N needs them
    DevBuf<int> sp_list;        // [window] the round's list of points for the sparse dirty scans
    bool seq_sticky = false;    // the last call ended on the sequential kernel (k_seq): the next one starts there
    int n_cus = 256;            // compute units of the device (hipDeviceProp_t::multiProcessorCount)

    // points + labels of the current call
    DevBuf<double> X, Xt;
    long long n_points = 0;
    DevBuf<long long> lab_uid;
    DevBuf<int8_t> lab_path;
    DevBuf<int> badflag;
    DevBuf<double> scr, scr2;  // scaler scratch

    // window buffers
    int win_alloc = 0, seg_alloc = 0, d_alloc = 0;
    DevBuf<double> v_cf1, v_cf2, v_cen, v_pref, v_scl, v_w, v_dsq, v_tau;
    DevBuf<unsigned long long> v_tile_dsq;
    DevBuf<int> v_kind, v_key, v_next, v_upg, v_acc, v_tgt, v_skip, v_skip_car, v_unsafe;
    DevBuf<Cand> part, clean, dpart, dpart2, dseed;  // part: two copies (window parity), dpart2: carry-set scan
    size_t part_stride = 0;
    // carry set of the previous window (lookahead)
    DevBuf<double> c_cf1v, c_cf2v, c_cenv, c_prefv, c_sclv, c_wv, c_c0, c_w0, c_dsq;
    DevBuf<int> c_kind, c_key, c_slot, c_kind0;
    DevBuf<unsigned long long> c_tile_dsq;
    DevBuf<int> T0, T1, rk;
    DevBuf<int> long_list;    // [2][CC_LONG_CAP] MCs whose chain k_chain_long replays (tables beyond k_claims' reach)
    DevBuf<unsigned long long> lstat, lprev;  // [2][CC_LSTAT_ROWS] / [CC_MAX_WINDOW]: long chains laid out ahead of k_chain (k_chain_long<.., true>)
    bool prep_launched = false;  // (this call: the counters behind lstat are worth reading)
    bool allow_prep = true;   // CHRONOCLUST_HIP_LONGPREP=0: long chains replayed by one workgroup each, as before round 5
    DevBuf<CommitRec> rec;
    // scan copy of the table for lookahead scans (see ScanCopy)
    DevBuf<double> sh_cen[2], sh_scl[2], sh_cf1[2], sh_cf2[2], sh_w[2];
    DevBuf<int> sh_kind[2], sh_key[2];
    DevBuf<int8_t> dpath;

    // offline results
    DevBuf<double> pv_cf1, pv_cf2, pv_cen, pv_pref, pv_w, wvec;
    DevBuf<long long> pv_id;
    DevBuf<int> prow, nn, pdim, mem_dev, off_dev, nw_cnt, nw_nbr;
    DevBuf<long long> nw_off;
    DevBuf<int8_t> core;
    DevBuf<unsigned long long> adj, adjw;
    DevBuf<double> c_cf1, c_cf2, c_cen, c_pref, c_w;
    HostClusters clusters;
    PinArena pin;  // page-locked scratch of the current call
    std::vector<long long> pcore_ids_host, pcore_uid_host;  // ids / creation numbers of the pcores, list order
    DevBuf<int32_t> pc_map, pc_out;                         // cc_point_clusters: creation number -> cluster, result
    int n_core = 0;

    // association scratch
    DevBuf<double> a_cur_cen, a_cur_pref, a_prev_cen, a_dist, a_pdist;
    DevBuf<int> a_idx, a_pidx;
    DevBuf<int> flags;

    std::vector<hipEvent_t> ev_pool, sync_pool;
    bool light_sync_events = true;

    // exact multi-GPU path: this handle is rank comm.rank of comm.world replicas of one stream
    cc::Comm comm;
    // a snapshot scan is split over the ranks when the table holds at least this many (row, dim) entries
    // (below that a window's scan is shorter than the all-gather that would follow it)
    long long shard_min_row_dims = 400000;
    long long shard_min_row_dims_pruned = 0;  // ... while the scans are pruned chains (0: the same; set by cc_comm_calibrate)
    double calib_ag_us = 0.0, calib_scan_ns = 0.0;  // what cc_comm_calibrate measured (group maxima)
    int offline_shard_min_rows = 8192;  // the pair matrices of the offline phase / association tracker likewise
    DevBuf<Cand> gsend, gpart;  // one merged record per window point (two parities) / the gathered records of all ranks
    DevBuf<Cand> gsend2, gpart2;  // guessed thresholds in a group: the missed points' new records, compact / gathered
    size_t gsend_stride = 0, gpart_stride = 0;
    DevBuf<int> g_i32;          // gather scratch of the offline phase

    // cc_points_prefetch: the next timepoint's points, uploaded by a worker thread through page-locked staging
    struct Prefetch {
        std::thread worker;
        bool active = false;            // a worker was started and has not been adopted / discarded yet
        const double* x = nullptr;      // what it uploads: pointer, shape, scaling (compared by the adopting upload)
        long long n = 0;
        int d = 0;
        bool scaled = false;
        std::vector<double> scale, mn;
        DevBuf<double> X, Xt, sm;       // destination buffers (swapped with the handle's on adoption), scale / min
        DevBuf<int> bad;
        hipStream_t stream = nullptr;
        void* pin[2] = {nullptr, nullptr};
        size_t pin_bytes = 0;
        int bad_host[4] = {0, 0, 0, 0};  // k_check_finite's words: [0] non-finite flag, [2..3] bits of the largest |value|
        int rc = 0;                     // hipError_t of the worker (0: fine)
        const char* what = "";
    } pf;

    // relaxed multi-GPU mode (events sharded over the ranks): points per rank and super-step (0: the exact path)
    int relaxed_minibatch = 0;
    bool shard_suspended = false;  // inside a relaxed super-step the ranks cluster different points: no split scans
    DevBuf<double> rs_cf1, rs_cf2, rs_w, r_delta, r_gather;   // snapshot of the shared table, deltas, all-reduce scratch
    DevBuf<int> rs_kind, rs_key, r_didx, r_didx_all, r_cnt_all;
    DevBuf<long long> rs_id;
    DevBuf<double> rg_X, rg_Xt;                               // the set-aside points of a super-step, gathered
    DevBuf<long long> rg_uid;
    DevBuf<int8_t> rg_path;
    cc_relaxed_stats rstats{};
};

namespace {

// CHRONOCLUST_HIP_POLICY_TRACE=<file>: the observations and decisions of the window policy, one JSON object per line
struct PolicyTrace {
    FILE* f = nullptr;
    static void obs_json(FILE* f, const cc_policy_obs& o)
    {
        fprintf(f, "{\"cursor\": %lld, \"m_rows\": %d, \"stall_b\": %d, \"stat_windows\": %lld, \"stat_truncated\": %lld, "
                   "\"stat_trunc_unknown\": %lld, \"stat_tiles\": %lld, \"stat_dirty_tiles\": %lld, \"stat_unsafe\": %lld, \"stat_missed\": %lld, \"tg_ok\": %d, \"round_hist\": [",
                (long long)o.cursor, o.m_rows, o.stall_b, (long long)o.stat_windows, (long long)o.stat_truncated,
                (long long)o.stat_trunc_unknown, (long long)o.stat_tiles, (long long)o.stat_dirty_tiles, (long long)o.stat_unsafe,
                (long long)o.stat_missed, o.tg_ok);
        for (int r = 0; r < CC_POLICY_MAX_ROUNDS + 2; ++r) fprintf(f, "%s%lld", r ? ", " : "", (long long)o.round_hist[r]);
        fprintf(f, "], \"prune_rows\": %llu, \"prune_full\": %llu, \"after_sequential\": %d}", (unsigned long long)o.prune_rows,
                (unsigned long long)o.prune_full, o.after_sequential);
    }
    static void dec_json(FILE* f, const cc_policy_decision& d)
    {
        fprintf(f, "{\"win_cfg\": %d, \"want\": %d, \"rounds\": %d, \"batch_windows\": %d, \"lookahead\": %d, \"nodirty\": %d, "
                   "\"prune\": %d, \"shard\": %d, \"restart\": %d, \"bad\": %d, \"stalled\": %d, \"sparse\": %d, \"probe\": %d}",
                d.win_cfg, d.want, d.rounds, d.batch_windows, d.lookahead, d.nodirty, d.prune, d.shard, d.restart, d.bad, d.stalled, d.sparse, d.probe);
    }
    // rank >= 0: the handle is rank `rank` of a group and writes <file>.rank<rank>
    PolicyTrace(const cc_policy_config& c, const cc_policy_carry& k, long long cursor, int rows, const cc_policy_decision& d0,
                int rank)
    {
        const char* path = getenv("CHRONOCLUST_HIP_POLICY_TRACE");
        if (!path || !path[0]) return;
        const std::string name = rank >= 0 ? std::string(path) + ".rank" + std::to_string(rank) : std::string(path);
        f = fopen(name.c_str(), "a");
        if (!f) return;
        fprintf(f, "{\"call\": {\"config\": {\"window\": %d, \"rounds_max\": %d, \"windows_per_sync\": %d, \"early_window\": %d, "
                   "\"lookahead\": %d, \"allow_nodirty\": %d, \"prune_mode\": %d, \"prune_applicable\": %d, \"can_shard\": %d, \"d\": %d, "
                   "\"resume\": %d, \"allow_sparse\": %d, \"allow_guess\": %d, \"allow_probe\": %d, \"shard_min_row_dims\": %lld, \"n_end\": %lld, \"shard_min_row_dims_pruned\": %lld, \"lookahead_pruned\": %d, \"force_prune_rows\": %d}, \"carry\": [%d, %d, %d], \"start\": [%lld, %d], \"dec\": ",
                c.window, c.rounds_max, c.windows_per_sync, c.early_window, c.lookahead, c.allow_nodirty, c.prune_mode,
                c.prune_applicable, c.can_shard, c.d, c.resume, c.allow_sparse, c.allow_guess, c.allow_probe, (long long)c.shard_min_row_dims, (long long)c.n_end, (long long)c.shard_min_row_dims_pruned, c.lookahead_pruned, c.force_prune_rows,
                k.adapt_win, k.clean_batches, k.since_shrink, cursor, rows);
        dec_json(f, d0);
        fprintf(f, "}}\n");
    }
    void batch(const cc_policy_obs& o, const cc_policy_decision& d)
    {
        if (!f) return;
        fprintf(f, "{\"obs\": ");
        obs_json(f, o);
        fprintf(f, ", \"dec\": ");
        dec_json(f, d);
        fprintf(f, "}\n");
    }
    void sequential(long long cursor, int rows, const cc_policy_decision& d)
    {
        cc_policy_obs o{};
        o.cursor = cursor;
        o.m_rows = rows;
        o.after_sequential = 1;
        batch(o, d);
    }
    ~PolicyTrace()
    {
        if (f) fclose(f);
    }
    PolicyTrace(const PolicyTrace&) = delete;
    PolicyTrace& operator=(const PolicyTrace&) = delete;
};

int fail(cc_handle* h, int code, const std::string& msg)
{
    if (h) h->err = msg;
    return code;
}

// Every non-OK way out of a call made while the handle belongs to a group leaves the peers waiting for a rank that
// will not come: the group is given up (in-process peers are released, RCCL communicators aborted) and the peers
// get CC_ERR_COMM instead of hanging.
void group_lost(cc_handle* h)
{
    if (h && h->comm.active()) h->comm.fail_group();
}

template <typename F>
int guarded(cc_handle* h, F&& f)
{
    try {
        if (h) HIPCHK(hipSetDevice(h->device));
        const int rc = f();
        if (rc < 0) group_lost(h);
        return rc;
    } catch (const HipErr& e) {
        char buf[512];
        snprintf(buf, sizeof buf, "HIP error %d (%s) in %s", (int)e.e, hipGetErrorString(e.e), e.what);
        group_lost(h);
        return fail(h, e.e == hipErrorOutOfMemory ? CC_ERR_OOM : CC_ERR_NO_DEVICE, buf);
    } catch (const cc::CommErr& e) {
        group_lost(h);
        return fail(h, CC_ERR_COMM, "exchange between ranks failed: " + e.what);
    } catch (const std::bad_alloc&) {
        group_lost(h);
        return fail(h, CC_ERR_OOM, "host allocation failed");
    }
}

// hipStreamSynchronize of a stream that may hold a collective of the handle's group: bounded (cc::Comm::wait_stream)
void sync_stream(cc_handle* h, hipStream_t st)
{
    if (h->comm.rccl() || h->comm.broken) h->comm.wait_stream(st);
    else HIPCHK(hipStreamSynchronize(st));
}

bool is_pow2(double k)
{
    if (!(k > 0.0) || !std::isfinite(k)) return false;
    int e;
    double m = std::frexp(k, &e);
    return m == 0.5 && e > -1000 && e < 1000;
}

void refresh_ctl_params(cc_handle* h)
{
    Ctl& c = h->hc;
    const cc_params& p = h->par;
    c.eps_sq = p.eps_sq;
    c.delta_sq = p.delta_sq;
    c.k = p.k;
    c.pow2 = (is_pow2(p.k) && !h->tainted) ? 1 : 0;
    c.inv_k = c.pow2 ? 1.0 / p.k : 0.0;
    c.beta_mu = p.beta * p.mu;  // hddstream.py:416, 529
    c.mu = p.mu;
    c.omicron = p.omicron;
    c.pi = p.pi;
    c.filter = (h->d > 0 && p.pi < h->d) ? 1 : 0;
    c.d = h->d;
}

// (Every push opens a fresh window chain - start of a call, back from the sequential kernel, a restart -: whatever scans
// left per window parity belongs to windows that will be scanned again, and the host's copy of it may be a half-summed one.)
void push_ctl(cc_handle* h)
{
    memset(h->hc.pstat, 0, sizeof(h->hc.pstat));
    h->hc.n_missed_all[0] = h->hc.n_missed_all[1] = 0;
    HIPCHK(hipMemcpyAsync(h->ctl.p, &h->hc, sizeof(Ctl), hipMemcpyHostToDevice, h->stream));
}
void pull_ctl(cc_handle* h)
{
    HIPCHK(hipMemcpyAsync(&h->hc, h->ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, h->stream));
    sync_stream(h, h->stream);
}
// The same between the batches of a call, through page-locked staging blocks: a copy to or from pageable memory is driven
// by the host - it waits for the stream to drain and only then starts the copy (30 us of idle device before the copy
// kernel at every read-back, profiles/r06_tool_startup_gaps_before.txt) -, one from page-locked memory is a stream
// operation like any other.  pull: the block is read once the stream has drained.  push: only ever called right after a
// pull (the stream is idle, the previous push's copy has completed), so one block serves.
void pull_ctl_pinned(cc_handle* h)
{
    if (!h->hc_pin) { pull_ctl(h); return; }
    HIPCHK(hipMemcpyAsync(h->hc_pin, h->ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, h->stream));
    sync_stream(h, h->stream);
    memcpy(&h->hc, h->hc_pin, sizeof(Ctl));
}
void push_ctl_pinned(cc_handle* h)
{
    if (!h->hc_pin) { push_ctl(h); return; }
    memset(h->hc.pstat, 0, sizeof(h->hc.pstat));
    h->hc.n_missed_all[0] = h->hc.n_missed_all[1] = 0;
    memcpy(h->hc_pin + 1, &h->hc, sizeof(Ctl));
    HIPCHK(hipMemcpyAsync(h->ctl.p, h->hc_pin + 1, sizeof(Ctl), hipMemcpyHostToDevice, h->stream));
}

// grow the table to at least `rows` rows, keeping the first m_rows rows
void ensure_table(cc_handle* h, size_t rows)
{
    if (h->tab.cap >= rows && h->tab.d == h->d) return;
    size_t want = std::max<size_t>(rows, std::max<size_t>(1024, h->tab.cap * 2));
    TableStore nt;
    nt.alloc(want, h->d);
    const size_t m = (size_t)h->hc.m_rows, d = (size_t)h->d;
    if (m > 0) {
        const TableStore& o = h->tab;
        HIPCHK(hipMemcpyAsync(nt.cf1.p, o.cf1.p, m * d * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(nt.cf2.p, o.cf2.p, m * d * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(nt.cen.p, o.cen.p, m * d * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(nt.pref.p, o.pref.p, m * d * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(nt.scl.p, o.scl.p, m * d * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(nt.w.p, o.w.p, m * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(nt.kind.p, o.kind.p, m * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(nt.key.p, o.key.p, m * 4, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(nt.id.p, o.id.p, m * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(nt.uid.p, o.uid.p, m * 8, hipMemcpyDeviceToDevice, h->stream));
    }
    HIPCHK(hipMemsetAsync(nt.touch.p, 0, 2 * want * 8, h->stream));
    HIPCHK(hipMemsetAsync(nt.last.p, 0, 2 * want * 8, h->stream));
    HIPCHK(hipMemsetAsync(nt.carry_of.p, 0, want * 8, h->stream));
    // the carry marks of the last commit are live state: the next window may be a lookahead window
    if (m > 0) HIPCHK(hipMemcpyAsync(nt.carry_of.p, h->tab.carry_of.p, m * 8, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemsetAsync(nt.cnt.p, 0, want * 8, h->stream));
    HIPCHK(hipMemsetAsync(nt.clen.p, 0, want * 4, h->stream));
    // (heavy marks index rows like the list in the control block: they move with the table)
    HIPCHK(hipMemsetAsync(nt.heavy.p, 0, want * 4, h->stream));
    if (m > 0 && h->tab.heavy.p) HIPCHK(hipMemcpyAsync(nt.heavy.p, h->tab.heavy.p, m * 4, hipMemcpyDeviceToDevice, h->stream));
    sync_stream(h, h->stream);
    h->tab.swap(nt);
}

int set_dim(cc_handle* h, int d)
{
    if (d <= 0 || d > CC_MAX_DIM) return fail(h, CC_ERR_BAD_ARG, "d must be in 1.." + std::to_string(CC_MAX_DIM));
    if (h->d == 0) h->d = d;
    if (h->d != d) {
        if (h->hc.m_rows == 0) h->d = d;
        else return fail(h, CC_ERR_BAD_ARG, "dimensionality differs from the microclusters already held");
    }
    return CC_OK;
}

void ensure_window_buffers(cc_handle* h, int win, int seg)
{
    if (win <= h->win_alloc && seg <= h->seg_alloc && h->d <= h->d_alloc) return;
    win = std::max(win, h->win_alloc);
    seg = std::max(seg, h->seg_alloc);
    const size_t w = (size_t)win, d = (size_t)std::max(h->d, h->d_alloc);
    h->v_cf1.ensure(w * d); h->v_cf2.ensure(w * d); h->v_cen.ensure(w * d); h->v_pref.ensure(w * d); h->v_scl.ensure(w * d); h->v_w.ensure(w);
    h->v_kind.ensure(w); h->v_key.ensure(w); h->v_next.ensure(w); h->v_upg.ensure(w); h->v_acc.ensure(w);
    h->v_dsq.ensure(w); h->v_tau.ensure(CC_TAU_STRIDE * w); h->v_tile_dsq.ensure(CC_DSQ_STRIDE * (w / 16 + 2));
    h->v_tgt.ensure(w);
    h->v_skip.ensure(w / 64 + 2); h->v_skip_car.ensure(w / 64 + 2); h->v_unsafe.ensure(w);
    h->part_stride = w * seg * 4;
    h->spart_stride = w * seg * 2;
    h->thr_stride = w * 2;
    h->spart.ensure(2 * h->spart_stride);
    h->thr.ensure(2 * h->thr_stride);
    h->thr32.ensure(2 * h->thr_stride);
    h->cmax.ensure(2);
    h->found.ensure(2 * (CC_MAX_WINDOW / 64));
    h->missed.ensure(2 * CC_MISSED_CAP);
    h->part.ensure(2 * h->part_stride); h->dpart.ensure(w * seg * 2); h->dpart2.ensure(w * seg * 2);
    h->clean.ensure(w * 4); h->dseed.ensure(w * 4);
    h->c_cf1v.ensure(w * d); h->c_cf2v.ensure(w * d); h->c_cenv.ensure(w * d); h->c_prefv.ensure(w * d);
    h->c_sclv.ensure(w * d); h->c_wv.ensure(w); h->c_c0.ensure(w * d); h->c_w0.ensure(w * d);
    h->c_kind.ensure(w); h->c_key.ensure(w); h->c_slot.ensure(w); h->c_kind0.ensure(w); h->c_dsq.ensure(w); h->c_tile_dsq.ensure(CC_DSQ_STRIDE * (w / 16 + 2));
    h->T0.ensure(w + 128); h->T1.ensure(w + 128);  // k_chain reads the claims in 128-entry blocks
    h->long_list.ensure(2 * CC_LONG_CAP);
    h->lstat.ensure(2 * CC_LSTAT_ROWS + 2);
    h->lprev.ensure(CC_MAX_WINDOW);
    h->dpath.ensure(w); h->rk.ensure(w); h->rec.ensure(1); h->sp_list.ensure(w); h->link_near.ensure(w);
    h->win_alloc = win; h->seg_alloc = seg; h->d_alloc = (int)d;
}

Carry carry_view(cc_handle* h)
{
    return Carry{h->c_cf1v.p, h->c_cf2v.p, h->c_cenv.p, h->c_prefv.p, h->c_sclv.p, h->c_wv.p, h->c_kind.p, h->c_key.p,
                 h->c_slot.p, h->c_c0.p, h->c_w0.p, h->c_kind0.p, h->c_dsq.p, h->c_tile_dsq.p};
}

// bring both scan copies in line with the table (rows [0, m_rows)); enqueued on the main stream
void scan_copy_sync(cc_handle* h, ScanCopy (&out)[2])
{
    const size_t cap = h->tab.cap, d = (size_t)h->d, m = (size_t)h->hc.m_rows;
    const bool filter = h->hc.filter != 0;
    const TableStore& t = h->tab;
    for (int q = 0; q < 2; ++q) {
        h->sh_cen[q].ensure(cap * d); h->sh_scl[q].ensure(cap * d); h->sh_kind[q].ensure(cap); h->sh_key[q].ensure(cap);
        if (filter) { h->sh_cf1[q].ensure(cap * d); h->sh_cf2[q].ensure(cap * d); h->sh_w[q].ensure(cap); }
        if (m > 0) {
            HIPCHK(hipMemcpyAsync(h->sh_cen[q].p, t.cen.p, m * d * 8, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->sh_scl[q].p, t.scl.p, m * d * 8, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->sh_kind[q].p, t.kind.p, m * 4, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->sh_key[q].p, t.key.p, m * 4, hipMemcpyDeviceToDevice, h->stream));
            if (filter) {
                HIPCHK(hipMemcpyAsync(h->sh_cf1[q].p, t.cf1.p, m * d * 8, hipMemcpyDeviceToDevice, h->stream));
                HIPCHK(hipMemcpyAsync(h->sh_cf2[q].p, t.cf2.p, m * d * 8, hipMemcpyDeviceToDevice, h->stream));
                HIPCHK(hipMemcpyAsync(h->sh_w[q].p, t.w.p, m * 8, hipMemcpyDeviceToDevice, h->stream));
            }
        }
        out[q] = ScanCopy{h->sh_cen[q].p, h->sh_scl[q].p, h->sh_cf1[q].p, h->sh_cf2[q].p, h->sh_w[q].p, h->sh_kind[q].p,
                          h->sh_key[q].p};
    }
}

Versions versions_view(cc_handle* h)
{
    return Versions{h->v_cf1.p, h->v_cf2.p, h->v_cen.p, h->v_pref.p, h->v_scl.p, h->v_w.p, h->v_kind.p,
                    h->v_key.p, h->v_next.p, h->v_upg.p, h->v_acc.p, h->v_tgt.p, h->v_dsq.p, h->v_tile_dsq.p,
                    h->v_tau.p, h->v_skip.p, h->v_skip_car.p, h->v_unsafe.p};
}

// ---- scan dispatch over the padded dimensionality ---------------------------------

// does the snapshot scan of this handle's stream run as k_scan_u? (decided per launch by the same test)
bool scan_u_applies(const cc_handle* h, int DP) { return h->allow_scan_u && h->hc.filter == 0 && h->hc.pow2 && h->d == DP; }

// the table rows as half-precision operands of the MFMA prefix test (k_prefix16): two window parities, whole tiles of 32 rows
// (grown between batches only: a scan in flight on the other stream may be reading it)
void ensure_prefix16(cc_handle* h)
{
    const size_t a16_rows = h->tab.cap + 2 * CC_P16_TM;
    if (h->a16_stride >= a16_rows * 4) return;
    sync_stream(h, h->stream);
    sync_stream(h, h->stream2);
    h->a16.ensure(2 * a16_rows * 4);
    h->a16_stride = a16_rows * 4;
    h->hdr16.ensure(2);
}

// does a PRUNED snapshot scan of this handle run as k_scan_p3<GENERAL> - the pdim filter on and / or k not a power of two?
// (the plain scans of these cases stay with k_scan<FILTER, POW2>)
bool scan_p3_general_applies(const cc_handle* h, int DP)
{
    return h->allow_scan_p3 && h->allow_prune_general && h->d == DP && DP > 8 && DP <= 40 && (h->hc.filter != 0 || !h->hc.pow2);
}

template <int DP, bool DIRTY>
void launch_scan_dp(cc_handle* h, hipStream_t st, int win, Rows rows, const Cand* clean, Cand* part, int S, int round,
                    int mode, int shard_rank, int shard_world, int phase)
{
    constexpr int NW = ScanShape<DP, DIRTY>::NW;
    const dim3 block(64 * NW);
    constexpr int TILE = 64 * ScanShape<DP, DIRTY>::PT;  // window points per workgroup
    const dim3 grid((win + TILE - 1) / TILE, S);
    // the clean scan is compiled without the pdim filter for the common case pi >= d; the dirty scan (few rows
    // survive its pruning) tests the flag at run time
    const bool filter = DIRTY || h->hc.filter != 0;
#define CC_LAUNCH_SCAN(F, P)                                                                                   \
    hipLaunchKernelGGL((k_scan<DP, F, P, DIRTY, NW>), grid, block, 0, st, h->ctl.p, h->X.p, h->Xt.p, rows, clean, \
                       part, round, mode, h->part_stride, shard_rank, shard_world)
    if constexpr (!DIRTY) {
        // the common case (k a power of two, no pdim filter, no padded dimensions): rows as scalar operands
        static_assert(ScanShape<DP, false>::PT == 1, "k_scan_u holds one window point per lane");
        if (scan_u_applies(h, DP)) {
            ++h->stats.scan_u_launches;
            if constexpr (DP > 8) {
                // prefix scores -> thresholds -> the scan that abandons rows whose partial sums pass them; for the
                // window's points (plist == nullptr) or for the ones a guessed threshold missed
                // k_scan_p for the window's points: phase A as a kernel of its own (two points per lane, survivor masks), phase
                // B behind it; lists of points (the ones a guessed threshold missed, probes) keep the one-kernel form
                auto scan_p_window = [&](int srank, int sworld, double gF, unsigned long long* found_, bool have_prefix16 = false) {
                    const bool one_kernel = h->split_a_mode == 0 || (h->split_a_mode == 1 && h->hc.m_rows < 10000);
                    if (one_kernel || (h->allow_scan_p3 && h->split_a_mode == 1 && DP <= 40)) {
                        if constexpr (DP <= 40) {
                            // one kernel, phase A on the matrix cores (cc_scan16.h): the rows as half-precision operands first
                            // (at any table size: CHRONOCLUST_HIP_SCANA=2 keeps the two-kernel form of large tables)
                            if (h->allow_scan_p3) {
                                ensure_prefix16(h);
                                const size_t a16_rows = h->tab.cap + 2 * CC_P16_TM;
                                if (!have_prefix16)
                                    hipLaunchKernelGGL((k_prefix16<DP>), dim3((unsigned)((a16_rows + 255) / 256)), dim3(256), 0, st, (const Ctl*)h->ctl.p,
                                                       rows.cen, rows.kind, h->a16.p, h->hdr16.p, h->a16_stride, round, mode);
                                // (unused dynamic LDS caps the workgroups a CU holds: CHRONOCLUST_HIP_SCAN_LDS_KB, an experiment knob)
                                static const unsigned p3_lds = []() { const char* e = getenv("CHRONOCLUST_HIP_SCAN_LDS_KB"); return e ? (unsigned)atoi(e) * 1024u : 0u; }();
                                // (kept rows listed per wave and walked with their operands prefetched - LISTED - from p3_listed_rows table rows on)
                                if (h->hc.m_rows >= h->p3_listed_rows)
                                    hipLaunchKernelGGL((k_scan_p3<DP, NW, true, false>), dim3((win + 127) / 128, S), block, p3_lds, st, h->ctl.p, h->Xt.p, rows.cen, rows.scl,
                                                       rows.kind, rows.key, h->thr.p, h->thr_stride, part, round, mode, h->part_stride, srank, sworld,
                                                       h->pstat_p(), gF, found_, (const cc_h8*)h->a16.p, (const Prefix16Hdr*)h->hdr16.p, h->a16_stride,
                                                       (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr);
                                else
                                    hipLaunchKernelGGL((k_scan_p3<DP, NW, false, false>), dim3((win + 127) / 128, S), block, p3_lds, st, h->ctl.p, h->Xt.p, rows.cen, rows.scl,
                                                       rows.kind, rows.key, h->thr.p, h->thr_stride, part, round, mode, h->part_stride, srank, sworld,
                                                       h->pstat_p(), gF, found_, (const cc_h8*)h->a16.p, (const Prefix16Hdr*)h->hdr16.p, h->a16_stride,
                                                       (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr);
                                ++h->stats.scan_p2_launches;
                                return;
                            }
                            // one kernel, two points per lane in phase A, phase B from the same residency (k_scan_p2)
                            if (h->allow_scan_p2) {
                                hipLaunchKernelGGL((k_scan_p2<DP, NW>), dim3((win + 127) / 128, S), block, 0, st, h->ctl.p, h->Xt.p, rows.cen, rows.scl,
                                                   rows.kind, rows.key, h->thr.p, h->thr_stride, part, round, mode, h->part_stride, srank, sworld,
                                                   h->pstat_p(), gF, found_);
                                ++h->stats.scan_p2_launches;
                                return;
                            }
                        }
                        hipLaunchKernelGGL((k_scan_p<DP, NW, false>), grid, block, 0, st, h->ctl.p, h->Xt.p, rows.cen, rows.scl, rows.kind,
                                           rows.key, h->thr.p, h->thr32.p, h->thr_stride, part, round, mode, h->part_stride, srank, sworld,
                                           h->pstat_p(), (const int*)nullptr, gF, found_, (const unsigned*)nullptr, (size_t)0, 0, 1);
                        return;
                    }
                    const int nsub = S * NW;
                    const int tps = cc_mask_tiles_per_sub((int)std::min<size_t>(h->tab.cap, (size_t)INT_MAX - 64), nsub);
                    const size_t need = (size_t)((win + 127) / 128) * (size_t)nsub * (size_t)tps;
                    if (need > h->mask_stride) {
                        h->masks.ensure(2 * need);
                        h->mask_stride = need;
                    }
                    hipLaunchKernelGGL((k_scan_a<DP, NW>), dim3((win + 127) / 128, S), block, 0, st, (const Ctl*)h->ctl.p,
                                       (const double*)h->Xt.p, rows.cen, rows.kind, (const double*)h->thr.p, h->thr_stride,
                                       h->masks.p, h->mask_stride, tps, round, mode, srank, sworld, gF);
                    // phase B: about one and a half rounds of the resident workgroups, i.e. q of phase A's sub-ranges per wave
                    // (its waves live on chains of memory round trips - prologue, masks, a few rows, merge -, not on arithmetic:
                    // with phase A's own split - eight rounds at the C5 shape - the prologues dominate, with one round every
                    // wave walks q times the rows; 2 M x 40, 50 000 rows: q = 1 / 2 / 3 / 4 / 6 / 12 -> 39.6 / 40.6 / 40.7 /
                    // 41.2 / 39.9 / 39.8 M points/s.  CHRONOCLUST_HIP_SCANB_Q overrides.)
                    int q = 1;
                    {
                        const int tiles = (win + 63) / 64;
                        const int resident = h->n_cus * (DP <= 20 ? CC_SCANP_WGS20 : (DP <= 40 ? 3 : 2));
                        for (int c = 1; c <= S; ++c)
                            if (S % c == 0 && 2 * tiles * (S / c) >= 3 * resident) q = c;
                        static const int q_env = []() { const char* e = getenv("CHRONOCLUST_HIP_SCANB_Q"); return e ? atoi(e) : 0; }();
                        if (q_env > 0 && S % q_env == 0) q = q_env;
                    }
                    hipLaunchKernelGGL((k_scan_p<DP, NW, true>), dim3(grid.x, S / q), block, 0, st, h->ctl.p, h->Xt.p, rows.cen, rows.scl,
                                       rows.kind, rows.key, h->thr.p, h->thr32.p, h->thr_stride, part, round, mode, h->part_stride, srank,
                                       sworld, h->pstat_p(), (const int*)nullptr, gF, found_, (const unsigned*)h->masks.p,
                                       h->mask_stride, tps, q);
                };
                auto seeded_chain = [&](int n_pts, const int* plist, Cand* part, size_t part_stride, int S) {
                        const bool whole_window = plist == nullptr && n_pts == win && part_stride == h->part_stride && S == (int)grid.y;
                        // seeds from the matrix cores and the tight threshold they allow (k_seed16; round 6) where the window's scan is
                        // k_scan_p3: the rows' half-precision records first, they serve both kernels
                        bool s16 = false;
                        if constexpr (DP <= 40) s16 = whole_window && h->allow_seed16 && h->allow_scan_p3 && h->split_a_mode != 2;
                        if (s16) {
                            if constexpr (DP <= 40) {
                                ensure_prefix16(h);
                                const size_t a16_rows = h->tab.cap + 2 * CC_P16_TM;
                                hipLaunchKernelGGL((k_prefix16<DP>), dim3((unsigned)((a16_rows + 255) / 256)), dim3(256), 0, st, (const Ctl*)h->ctl.p,
                                                   rows.cen, rows.kind, h->a16.p, h->hdr16.p, h->a16_stride, round, mode);
                                hipLaunchKernelGGL((k_seed16<DP, NW>), dim3((n_pts + 127) / 128, S), block, 0, st, (const Ctl*)h->ctl.p, (const double*)h->Xt.p,
                                                   rows.cen, h->spart.p, round, mode, h->spart_stride, (const cc_h8*)h->a16.p,
                                                   (const Prefix16Hdr*)h->hdr16.p, h->a16_stride);
                                ++h->stats.seed16_launches;
                            }
                        } else {
                            // (k_seed holds two points per lane: point tiles of 128)
                            hipLaunchKernelGGL((k_seed<DP, NW>), dim3((n_pts + 127) / 128, S), block, 0, st, h->ctl.p, h->Xt.p, rows.cen,
                                               rows.kind, h->spart.p, round, mode, h->spart_stride, h->cmax.p, plist);
                        }
                        hipLaunchKernelGGL((k_seed_merge<DP>), dim3((2 * n_pts + 63) / 64), dim3(64), 0, st, h->ctl.p, h->X.p, rows.cen,
                                           rows.scl, h->spart.p, h->spart_stride, S, h->thr.p, h->thr32.p, h->thr_stride,
                                           s16 ? 0.0 : h->prune_F, round, mode, h->cmax.p, h->pstat_p(), plist);
                        // (split over the ranks of a group: seeds and thresholds over ALL rows on every rank - replicated, so
                        // that every rank abandons against the same T -, phases A / B over the rank's own rows)
                        if (whole_window) {
                            scan_p_window(shard_rank, shard_world, 0.0, (unsigned long long*)nullptr, s16);
                            return;
                        }
                        hipLaunchKernelGGL((k_scan_p<DP, NW, false>), dim3((n_pts + 63) / 64, S), block, 0, st, h->ctl.p, h->Xt.p, rows.cen,
                                           rows.scl, rows.kind, rows.key, h->thr.p, h->thr32.p, h->thr_stride, part, round, mode,
                                           part_stride, shard_rank, shard_world, h->pstat_p(), plist, 0.0,
                                           (unsigned long long*)nullptr, (const unsigned*)nullptr, (size_t)0, 0, 1);
                };
                // the points a guessed threshold missed: the plain scan over their list (k_scan_u's header says why); the seeded
                // chain on request (CHRONOCLUST_HIP_MISSED_PLAIN=0)
                auto missed_scan = [&](const int* list) {
                    if (!h->allow_missed_plain) {
                        seeded_chain(CC_MISSED_CAP, list, part, h->part_stride, S);
                        return;
                    }
                    ++h->stats.missed_plain_launches;
                    hipLaunchKernelGGL((k_scan_u<DP, NW>), dim3((CC_MISSED_CAP + 63) / 64, S), block, 0, st, h->ctl.p, h->Xt.p, rows.cen,
                                       rows.scl, rows.kind, rows.key, part, round, mode, h->part_stride, shard_rank, shard_world, list);
                };
                if (phase == 1) {
                    // guessed thresholds on the exact multi-GPU path, after the ranks' records were gathered: the points
                    // whose merged pcore list starts with a bound (k_missed_g: the same list on every rank) go through the
                    // seeded chain - seeds over all rows on every rank, phases A / B over the rank's rows
                    hipLaunchKernelGGL(k_missed_g, dim3(1), dim3(1024), 0, st, h->ctl.p, (const Cand*)h->gpart.p, h->gpart_stride,
                                       (size_t)win * 4 + 4, shard_world, h->missed.p, CC_MISSED_CAP, round, mode, h->found.p);
                    missed_scan(h->missed.p);
                    return;
                }
                if (h->prune_now) {
                    ++h->stats.scan_p_launches;
                    if (h->guess_now && h->group_guess_now) {
                        // (split over ranks: every rank scans its rows against the same guess; who was missed is only known
                        // once the records are gathered - phase 1, enqueued by the caller behind the all-gather)
                        ++h->stats.scan_g_launches;
                        hipLaunchKernelGGL(k_pstat_zero, dim3(1), dim3(2), 0, st, (const Ctl*)h->ctl.p, h->pstat_p(), round, mode);
                        scan_p_window(shard_rank, shard_world, h->prune_F, h->lean_now ? (unsigned long long*)nullptr : h->found.p);
                        return;
                    }
                    if (h->guess_now) {
                        // guessed thresholds: one scan, the list of the points it missed, the seeded chain for those
                        // (list of the window's parity: the lookahead scan of the next window fills the other one)
                        ++h->stats.scan_g_launches;
                        int* const list = h->missed.p;  // (the kernels take the half of the window's parity)
                        scan_p_window(0, 1, h->prune_F, h->lean_now ? (unsigned long long*)nullptr : h->found.p);
                        if (h->lean_now) {  // (nobody is expected to be missed: see cc_policy.h)
                            ++h->stats.scan_lean_launches;
                            return;
                        }
                        hipLaunchKernelGGL(k_missed, dim3(1), dim3(1024), 0, st, h->ctl.p, h->found.p, list, CC_MISSED_CAP, round, mode);
                        missed_scan(list);
                        return;
                    }
                    seeded_chain(win, nullptr, part, h->part_stride, S);
                    return;
                }
                hipLaunchKernelGGL((k_scan_u<DP, NW>), grid, block, 0, st, h->ctl.p, h->Xt.p, rows.cen, rows.scl, rows.kind,
                                   rows.key, part, round, mode, h->part_stride, shard_rank, shard_world);
                if (h->probe_now) {
                    // the probe: the pruned chain on the window's first 128 points, over all rows, into scratch partials -
                    // only its sample of completed rows (Ctl::stat_prune_*) is used, by the window policy
                    ++h->stats.probe_launches;
                    const int keep_rank = shard_rank, keep_world = shard_world;
                    shard_rank = 0; shard_world = 1;  // (the whole table on every rank: the same sample everywhere)
                    // (few points, so many sub-ranges of rows: while the table fills most rows are completed, at the
                    // latency of scalar loads - 128 workgroups per point tile keep that to a hundred rows per wave)
                    const int Sp = (int)std::max<size_t>(1, std::min<size_t>(128, h->spart_stride / 2 / 128));  // (what the seed buffer holds for 128 points)
                    h->probe_part.ensure((size_t)2 * 128 * Sp * 4);
                    seeded_chain(std::min(win, 128), nullptr, h->probe_part.p, (size_t)128 * Sp * 4, Sp);
                    shard_rank = keep_rank; shard_world = keep_world;
                }
                return;
            }
            hipLaunchKernelGGL((k_scan_u<DP, NW>), grid, block, 0, st, h->ctl.p, h->Xt.p, rows.cen, rows.scl, rows.kind,
                               rows.key, part, round, mode, h->part_stride, shard_rank, shard_world);
            return;
        }
    }
    if constexpr (!DIRTY && DP > 8 && DP <= 40) {
        // the pruned chain where the pdim filter is on or k is not a power of two (round 6): seeds and thresholds as ever (any
        // threshold is a valid one), then k_scan_p3<GENERAL>.  Seeded thresholds only: the points a guessed threshold misses
        // would need a plain scan over a point list, which exists for the common case alone (k_scan_u).
        if (h->prune_now && scan_p3_general_applies(h, DP) && phase == 0) {
            ++h->stats.scan_p_launches;
            ++h->stats.scan_p2_launches;
            ensure_prefix16(h);
            const size_t a16_rows = h->tab.cap + 2 * CC_P16_TM;
            hipLaunchKernelGGL((k_prefix16<DP>), dim3((unsigned)((a16_rows + 255) / 256)), dim3(256), 0, st, (const Ctl*)h->ctl.p, rows.cen,
                               rows.kind, h->a16.p, h->hdr16.p, h->a16_stride, round, mode);
            if (h->allow_seed16) {
                hipLaunchKernelGGL((k_seed16<DP, NW>), dim3((win + 127) / 128, S), block, 0, st, (const Ctl*)h->ctl.p, (const double*)h->Xt.p, rows.cen,
                                   h->spart.p, round, mode, h->spart_stride, (const cc_h8*)h->a16.p, (const Prefix16Hdr*)h->hdr16.p, h->a16_stride);
                ++h->stats.seed16_launches;
            } else
                hipLaunchKernelGGL((k_seed<DP, NW>), dim3((win + 127) / 128, S), block, 0, st, h->ctl.p, h->Xt.p, rows.cen, rows.kind,
                                   h->spart.p, round, mode, h->spart_stride, h->cmax.p, (const int*)nullptr);
            hipLaunchKernelGGL((k_seed_merge<DP>), dim3((2 * win + 63) / 64), dim3(64), 0, st, h->ctl.p, h->X.p, rows.cen, rows.scl,
                               h->spart.p, h->spart_stride, S, h->thr.p, h->thr32.p, h->thr_stride, h->allow_seed16 ? 0.0 : h->prune_F, round,
                               mode, h->cmax.p, h->pstat_p(), (const int*)nullptr);
            if (h->hc.m_rows >= h->p3_listed_rows)
                hipLaunchKernelGGL((k_scan_p3<DP, NW, true, true>), dim3((win + 127) / 128, S), block, 0, st, h->ctl.p, h->Xt.p, rows.cen,
                                   rows.scl, rows.kind, rows.key, h->thr.p, h->thr_stride, part, round, mode, h->part_stride, shard_rank,
                                   shard_world, h->pstat_p(), 0.0, (unsigned long long*)nullptr, (const cc_h8*)h->a16.p,
                                   (const Prefix16Hdr*)h->hdr16.p, h->a16_stride, (const double*)h->X.p, rows.cf1, rows.cf2, rows.w);
            else
                hipLaunchKernelGGL((k_scan_p3<DP, NW, false, true>), dim3((win + 127) / 128, S), block, 0, st, h->ctl.p, h->Xt.p, rows.cen,
                                   rows.scl, rows.kind, rows.key, h->thr.p, h->thr_stride, part, round, mode, h->part_stride, shard_rank,
                                   shard_world, h->pstat_p(), 0.0, (unsigned long long*)nullptr, (const cc_h8*)h->a16.p,
                                   (const Prefix16Hdr*)h->hdr16.p, h->a16_stride, (const double*)h->X.p, rows.cf1, rows.cf2, rows.w);
            return;
        }
    }
    if (filter) {
        if (h->hc.pow2) CC_LAUNCH_SCAN(true, true);
        else CC_LAUNCH_SCAN(true, false);
    } else if constexpr (!DIRTY) {
        if (h->hc.pow2) CC_LAUNCH_SCAN(false, true);
        else CC_LAUNCH_SCAN(false, false);
    }
#undef CC_LAUNCH_SCAN
}

// S = partials per point (workgroups per point tile); sub-ranges per tile = S * waves per workgroup.
// Clean scan: mode 0 = current window, 1 = lookahead (round = parity of that window's sequence number);
// dirty scan: mode 0 = version rows, 1 = carry set.
template <bool DIRTY>
void launch_scan(cc_handle* h, hipStream_t st, int win, Rows rows, const Cand* clean, Cand* part, int S, int round,
                 int mode, int shard_rank = 0, int shard_world = 1, int phase = 0)
{
    const int d = h->d;
#define CC_SCAN_DP(DP) launch_scan_dp<DP, DIRTY>(h, st, win, rows, clean, part, S, round, mode, shard_rank, shard_world, phase)
    // (padded dimensions cost full distance terms: the ladder follows the shapes of BASELINE.json - d = 14, 20, 40)
    if (d <= 4) CC_SCAN_DP(4);
    else if (d <= 8) CC_SCAN_DP(8);
    else if (d <= 14) CC_SCAN_DP(14);
    else if (d <= 16) CC_SCAN_DP(16);
    else if (d <= 20) CC_SCAN_DP(20);
    else if (d <= 32) CC_SCAN_DP(32);
    else if (d <= 40) CC_SCAN_DP(40);
    else CC_SCAN_DP(64);
#undef CC_SCAN_DP
}

// workgroups of the clean scan that are resident at once (compute units x workgroups per CU it is compiled for)
template <int DP>
int scan_u_wgs_per_cu()
{
    static const int n = []() {
        int blocks = 0;
        constexpr int NW = ScanShape<DP, false>::NW;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_scan_u<DP, NW>, 64 * NW, 0) != hipSuccess || blocks < 1)
            blocks = ScanUShape<DP>::WGS;
        return blocks;
    }();
    return n;
}

int scan_resident_wgs(const cc_handle* h, int n_cus)
{
    const int d = h->d;
    if (d == 4 && scan_u_applies(h, 4)) return n_cus * scan_u_wgs_per_cu<4>();
    if (d == 8 && scan_u_applies(h, 8)) return n_cus * scan_u_wgs_per_cu<8>();
    if (d == 14 && scan_u_applies(h, 14)) return n_cus * scan_u_wgs_per_cu<14>();
    if (d == 16 && scan_u_applies(h, 16)) return n_cus * scan_u_wgs_per_cu<16>();
    if (d == 20 && scan_u_applies(h, 20)) return n_cus * scan_u_wgs_per_cu<20>();
    if (d == 32 && scan_u_applies(h, 32)) return n_cus * scan_u_wgs_per_cu<32>();
    if (d == 40 && scan_u_applies(h, 40)) return n_cus * scan_u_wgs_per_cu<40>();
    if (d == 64 && scan_u_applies(h, 64)) return n_cus * scan_u_wgs_per_cu<64>();
    int per_cu;
    if (d <= 4) per_cu = ScanShape<4, false>::WGS;
    else if (d <= 8) per_cu = ScanShape<8, false>::WGS;
    else if (d <= 14) per_cu = ScanShape<14, false>::WGS;
    else if (d <= 16) per_cu = ScanShape<16, false>::WGS;
    else if (d <= 20) per_cu = ScanShape<20, false>::WGS;
    else if (d <= 32) per_cu = ScanShape<32, false>::WGS;
    else if (d <= 40) per_cu = ScanShape<40, false>::WGS;
    else per_cu = ScanShape<64, false>::WGS;
    return n_cus * per_cu;
}

// Partials per point for a batch whose windows have `tiles` point tiles: at most S, not less than S / 2, chosen so that
// the launch (tiles x S' workgroups) fills whole rounds of the resident workgroups - 1 024 workgroups on a machine that
// holds 768 at once (d = 40) run as long as 1 536 would.
int scan_partials_for(int tiles, int S, int resident)
{
    int best = S;
    double best_eff = 0.0;
    for (int s = S; s >= std::max(1, S / 2); --s) {
        const double x = (double)tiles * s / (double)resident;
        const double eff = x / std::ceil(x);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = s; }
    }
    return best;
}

// waves per workgroup of the scans at dimensionality d (the host turns `segments` sub-ranges into partials per point)
int scan_waves_for_dim(int d, bool dirty)
{
    if (dirty) return 4;
    return (d > 16 && d <= 20) ? ScanShape<20, false>::NW : 4;
}

hipEvent_t get_event(cc_handle* h, size_t i)
{
    while (h->ev_pool.size() <= i) {
        hipEvent_t e;
        // (timestamps are all the host reads from these: no system-scope release when one is recorded)
        HIPCHK(hipEventCreateWithFlags(&e, h->light_sync_events ? hipEventReleaseToDevice : hipEventDefault));
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[i];
}

// events that only order the two streams (never read back): no timestamp, device-scope release
hipEvent_t get_sync_event(cc_handle* h, size_t i)
{
    if (!h->light_sync_events) return get_event(h, i);
    while (h->sync_pool.size() <= i) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice));
        h->sync_pool.push_back(e);
    }
    return h->sync_pool[i];
}

// k_link_scan over the padded dimensionality (the scans' ladder)
void launch_link_scan(cc_handle* h, hipStream_t st, int win)
{
    const dim3 grid((win + 63) / 64, (win + 4 * CC_LINK_SUB - 1) / (4 * CC_LINK_SUB)), block(256);
    const int d = h->d;
#define CC_LINK_DP(DP) hipLaunchKernelGGL((k_link_scan<DP>), grid, block, 0, st, (const Ctl*)h->ctl.p, (const double*)h->X.p, \
                                          (const double*)h->Xt.p, (const int*)h->T0.p, h->link_near.p)
    if (d <= 4) CC_LINK_DP(4);
    else if (d <= 8) CC_LINK_DP(8);
    else if (d <= 14) CC_LINK_DP(14);
    else if (d <= 16) CC_LINK_DP(16);
    else if (d <= 20) CC_LINK_DP(20);
    else if (d <= 32) CC_LINK_DP(32);
    else if (d <= 40) CC_LINK_DP(40);
    else CC_LINK_DP(64);
#undef CC_LINK_DP
}

struct RowList {
    std::vector<int> pcore, outlier;  // table rows in Python list order
};

// list order = ascending key within a kind
RowList list_order(cc_handle* h, std::vector<int>* kind_out = nullptr, std::vector<int>* key_out = nullptr)
{
    const int m = h->hc.m_rows;
    std::vector<int> kind(m), key(m);
    if (m) {
        HIPCHK(hipMemcpyAsync(kind.data(), h->tab.kind.p, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(key.data(), h->tab.key.p, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
        sync_stream(h, h->stream);
    }
    RowList rl;
    for (int r = 0; r < m; ++r) {
        if (kind[r] == CC_KIND_PCORE) rl.pcore.push_back(r);
        else if (kind[r] == CC_KIND_OUTLIER) rl.outlier.push_back(r);
    }
    auto by_key = [&](int a, int b) { return key[a] < key[b]; };
    std::sort(rl.pcore.begin(), rl.pcore.end(), by_key);
    std::sort(rl.outlier.begin(), rl.outlier.end(), by_key);
    if (kind_out) *kind_out = kind;
    if (key_out) *key_out = key;
    return rl;
}

}  // namespace

// =====================================================================================
// C-ABI
// =====================================================================================

extern "C" {

int cc_create(int device, cc_handle** out)
{
    if (!out) return CC_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CC_ERR_NO_DEVICE;
    cc_handle* h = new (std::nothrow) cc_handle();
    if (!h) return CC_ERR_OOM;
    h->device = device;
    int rc = guarded(h, [&]() {
        // the validation kernels (first stream) are short latency chains, the lookahead scans (second stream) fill the
        // machine: when both have workgroups pending the validation ones go first
        int prio_lo = 0, prio_hi = 0;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_cus = cus;
        HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIPCHK(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi));
        HIPCHK(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_lo));
        h->ctl.ensure(1);
        if (hipHostMalloc((void**)&h->hc_pin, 2 * sizeof(Ctl), hipHostMallocDefault) != hipSuccess) h->hc_pin = nullptr;
        h->badflag.ensure(4);
        memset(&h->hc, 0, sizeof(Ctl));
        // (round 6: the largest window - with the scans in place, what a window costs whatever its size (launches, prologues, the
        // validation kernels' chains of round trips: ~100 us) is a larger share of a shorter one: C2 13.98 -> 13.32 ms, the C5 shape
        // 75 -> 85 M points/s; the policy holds it at 32 768 while the table is small - cc_policy.h, `win`)
        h->tun.window = CC_MAX_WINDOW;
        h->tun.rounds = 3;
        h->tun.segments = 64;
        h->tun.windows_per_sync = 16;
        h->tun.time_kernels = 0;
        const char* tr = getenv("CHRONOCLUST_HIP_TRACE");
        h->trace = tr && tr[0] == '1';
        const char* nd = getenv("CHRONOCLUST_HIP_NODIRTY");
        h->allow_nodirty = !(nd && nd[0] == '0');
        const char* cl = getenv("CHRONOCLUST_HIP_CLAIMS");
        h->allow_claims = !(cl && cl[0] == '0');
        const char* ct = getenv("CHRONOCLUST_HIP_CHAIN_THREADS");
        if (ct && (atoi(ct) == 64 || atoi(ct) == 128 || atoi(ct) == 256)) h->chain_threads = atoi(ct);
        const char* dt = getenv("CHRONOCLUST_HIP_DECIDE_THREADS");
        if (dt && (atoi(dt) == 64 || atoi(dt) == 128 || atoi(dt) == 256)) h->decide_threads = atoi(dt);
        const char* mt = getenv("CHRONOCLUST_HIP_COMMIT_THREADS");
        if (mt && (atoi(mt) == 64 || atoi(mt) == 128 || atoi(mt) == 256)) h->commit_threads = atoi(mt);
        const char* se = getenv("CHRONOCLUST_HIP_LIGHT_EVENTS");
        if (se) h->light_sync_events = atoi(se) != 0;
        const char* su = getenv("CHRONOCLUST_HIP_SCANU");
        if (su) h->allow_scan_u = atoi(su) != 0;
        const char* pr = getenv("CHRONOCLUST_HIP_PRUNE");
        if (pr && atoi(pr) >= 0 && atoi(pr) <= 2) h->prune_mode = atoi(pr);
        const char* pw = getenv("CHRONOCLUST_HIP_PRUNE_WGS");
        if (pw && atoi(pw) >= 1 && atoi(pw) <= 64) h->prune_rounds4 = atoi(pw);
        const char* pf = getenv("CHRONOCLUST_HIP_PRUNE_F");
        if (pf && atof(pf) >= 1.0) h->prune_F = atof(pf);
        const char* lc = getenv("CHRONOCLUST_HIP_LONGCHAINS");
        h->allow_long = !(lc && lc[0] == '0');
        const char* lpp = getenv("CHRONOCLUST_HIP_LONGPREP");
        h->allow_prep = !(lpp && lpp[0] == '0');
        const char* gg = getenv("CHRONOCLUST_HIP_GROUP_GUESS");
        h->group_guess_always = gg && gg[0] == '1';
        const char* qt = getenv("CHRONOCLUST_HIP_QUIET");
        h->allow_quiet = !(qt && qt[0] == '0');
        const char* hv = getenv("CHRONOCLUST_HIP_HEAVY");
        h->allow_heavy = !(hv && hv[0] == '0');
        const char* sr = getenv("CHRONOCLUST_HIP_SEQR");
        h->allow_seq_r = !(sr && sr[0] == '0');
        const char* sg = getenv("CHRONOCLUST_HIP_SEQG");
        h->allow_seq_g = !(sg && sg[0] == '0');
        const char* pb = getenv("CHRONOCLUST_HIP_PROBE");
        h->allow_probe = !(pb && pb[0] == '0');
        const char* gs = getenv("CHRONOCLUST_HIP_GUESS");
        h->allow_guess = !(gs && gs[0] == '0');
        const char* sa = getenv("CHRONOCLUST_HIP_SCANA");
        if (sa) h->split_a_mode = std::max(0, std::min(2, atoi(sa)));
        const char* ln = getenv("CHRONOCLUST_HIP_LEAN");
        h->allow_lean = !(ln && ln[0] == '0');
        const char* mpl = getenv("CHRONOCLUST_HIP_MISSED_PLAIN");
        if (mpl && atoi(mpl) == 0) h->allow_missed_plain = false;
        const char* fpr = getenv("CHRONOCLUST_HIP_FORCE_PRUNE_ROWS");
        if (fpr) h->force_prune_rows = atoi(fpr);
        const char* s16 = getenv("CHRONOCLUST_HIP_SEED16");
        if (s16) h->allow_seed16 = atoi(s16) != 0;
        const char* pg = getenv("CHRONOCLUST_HIP_PRUNE_GENERAL");
        if (pg && atoi(pg) == 0) h->allow_prune_general = false;
        const char* lap = getenv("CHRONOCLUST_HIP_LA_PRUNED");
        if (lap && atoi(lap) != 0) h->la_pruned = true;
        const char* p3l = getenv("CHRONOCLUST_HIP_P3_LISTED");
        if (p3l) h->p3_listed_rows = atoi(p3l);
        const char* p3 = getenv("CHRONOCLUST_HIP_SCANP3");
        if (p3 && atoi(p3) == 0) h->allow_scan_p3 = false;
        const char* p2 = getenv("CHRONOCLUST_HIP_SCANP2");
        h->allow_scan_p2 = !(p2 && p2[0] == '0');
        const char* lk = getenv("CHRONOCLUST_HIP_LINK");
        h->allow_link = !(lk && lk[0] == '0');
        const char* sp = getenv("CHRONOCLUST_HIP_SPARSE");
        if (sp && atoi(sp) >= 0) h->allow_sparse = atoi(sp);
        push_ctl(h);
        sync_stream(h, h->stream);
        return CC_OK;
    });
    if (rc != CC_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return CC_OK;
}

void cc_destroy(cc_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->pf.worker.joinable()) h->pf.worker.join();
    if (h->pf.stream) (void)hipStreamDestroy(h->pf.stream);
    if (h->hc_pin) (void)hipHostFree(h->hc_pin);
    for (int q = 0; q < 2; ++q)
        if (h->pf.pin[q]) (void)hipHostFree(h->pf.pin[q]);
    // (a pending collective whose peer is gone must not hang the destructor: bounded wait, then abort)
    auto drain = [&](hipStream_t st) {
        if (!st) return;
        try {
            if (h->comm.rccl() && !h->comm.broken) h->comm.wait_stream(st);
            else (void)hipStreamSynchronize(st);
        } catch (const cc::CommErr&) {
        }
    };
    drain(h->stream);
    drain(h->stream2);
    h->comm.destroy();
    if (h->stream) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamDestroy(h->stream);
    }
    if (h->stream2) {
        (void)hipStreamSynchronize(h->stream2);
        (void)hipStreamDestroy(h->stream2);
    }
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->sync_pool) (void)hipEventDestroy(e);
    delete h;
}

const char* cc_last_error(const cc_handle* h) { return h ? h->err.c_str() : "null handle"; }

int cc_set_tuning(cc_handle* h, const cc_tuning* t)
{
    if (!h || !t) return CC_ERR_BAD_ARG;
    if (t->window > 0) h->tun.window = std::min(t->window, CC_MAX_WINDOW);
    if (t->rounds > 0) h->tun.rounds = std::min(t->rounds, CC_MAX_ROUNDS);
    if (t->segments > 0) h->tun.segments = std::min(t->segments, 1024);
    if (t->windows_per_sync > 0) h->tun.windows_per_sync = t->windows_per_sync;
    h->tun.time_kernels = t->time_kernels;
    if (t->dirty_segments > 0) h->tun.dirty_segments = std::min(t->dirty_segments, 1024);
    h->tun.lookahead = t->lookahead;
    h->tun.sequential = t->sequential;
    if (t->early_window > 0) h->tun.early_window = t->early_window;
    return CC_OK;
}

int cc_reset(cc_handle* h)
{
    if (!h) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        Ctl& c = h->hc;
        c.m_rows = 0;
        c.n_pkeys = c.n_okeys = 0;
        c.pcore_last_id = c.outlier_last_id = 0;
        c.cursor = 0;
        h->tainted = false;
        h->adapt_win = 0;  // an empty table starts with small windows again
        h->seq_sticky = false;
        h->clean_batches = 0;
        h->since_shrink = 1000;
        h->clusters.clear();
        h->n_core = 0;
        refresh_ctl_params(h);
        push_ctl(h);
        sync_stream(h, h->stream);
        return (int)CC_OK;
    });
}

int cc_set_params(cc_handle* h, const cc_params* p)
{
    if (!h || !p) return CC_ERR_BAD_ARG;
    if (h->have_par && h->hc.m_rows > 0 && p->k != h->par.k) h->tainted = true;  // old rows keep their old k
    h->par = *p;
    h->have_par = true;
    refresh_ctl_params(h);
    return CC_OK;
}

int cc_dim(cc_handle* h) { return h ? h->d : CC_ERR_BAD_ARG; }

int cc_counters(cc_handle* h, int64_t* pcore_last_id, int64_t* outlier_last_id)
{
    if (!h) return CC_ERR_BAD_ARG;
    if (pcore_last_id) *pcore_last_id = h->hc.pcore_last_id;
    if (outlier_last_id) *outlier_last_id = h->hc.outlier_last_id;
    return CC_OK;
}

int cc_set_counters(cc_handle* h, int64_t pcore_last_id, int64_t outlier_last_id)
{
    if (!h || pcore_last_id < 0 || outlier_last_id < 0) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        h->hc.pcore_last_id = pcore_last_id;
        h->hc.outlier_last_id = outlier_last_id;
        push_ctl(h);
        sync_stream(h, h->stream);
        return (int)CC_OK;
    });
}

static int upload_points(cc_handle* h, const double* x, int64_t n, int32_t d, const double* scale, const double* mn);

int cc_points_upload(cc_handle* h, const double* x, int64_t n, int32_t d)
{
    if (!h || (!x && n > 0) || n < 0) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() { return upload_points(h, x, n, d, nullptr, nullptr); });
}

// MinMax scaling on the device (scaling/scaler.py:27-47).  cc_col_minmax: per-column min / max of a host buffer,
// NaN ignored (what MinMaxScaler.partial_fit takes from one file); cc_points_upload_scaled: cc_points_upload of
// x * scale + min_; cc_points_download_unscaled: (resident points - min_) / scale back to the host.
int cc_col_minmax(cc_handle* h, const double* x, int64_t n, int32_t d, double* out_min, double* out_max)
{
    if (!h || !x || n <= 0 || d <= 0 || d > CC_MAX_DIM || !out_min || !out_max) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        h->scr.ensure((size_t)n * d);
        const int chunks = (int)std::min<long long>(1024, (n + 255) / 256);
        h->scr2.ensure((size_t)2 * chunks * d);
        HIPCHK(hipMemcpyAsync(h->scr.p, x, (size_t)n * d * 8, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_col_minmax, dim3(chunks), dim3(256), 0, h->stream, h->scr.p, (long long)n, (int)d, h->scr2.p,
                           chunks);
        std::vector<double> part((size_t)2 * chunks * d);
        HIPCHK(hipMemcpyAsync(part.data(), h->scr2.p, part.size() * 8, hipMemcpyDeviceToHost, h->stream));
        sync_stream(h, h->stream);
        for (int c = 0; c < d; ++c) {
            double mn = std::numeric_limits<double>::infinity(), mx = -mn;
            for (int b = 0; b < chunks; ++b) {
                mn = std::fmin(mn, part[(size_t)b * d + c]);
                mx = std::fmax(mx, part[(size_t)(chunks + b) * d + c]);
            }
            out_min[c] = mn;
            out_max[c] = mx;
        }
        return (int)CC_OK;
    });
}

// waits for a running prefetch; returns true if it finished without an error
static bool prefetch_join(cc_handle* h)
{
    if (h->pf.worker.joinable()) h->pf.worker.join();
    return h->pf.active && h->pf.rc == 0;
}

static void prefetch_discard(cc_handle* h)
{
    (void)prefetch_join(h);
    h->pf.active = false;
}

// the largest |value| k_check_finite saw (words 2..3 of its flag buffer)
static double absmax_of(const int* flag_words)
{
    double m;
    memcpy(&m, flag_words + 2, 8);
    return m;
}

static int upload_points(cc_handle* h, const double* x, int64_t n, int32_t d, const double* scale, const double* mn)
{
    int rc = set_dim(h, d);
    if (rc != CC_OK) return rc;
    if (h->pf.active) {
        // the points may already be on their way (cc_points_prefetch): adopt them if it is this very upload
        cc_handle::Prefetch& pf = h->pf;
        bool same = pf.x == x && pf.n == n && pf.d == d && pf.scaled == (scale != nullptr);
        for (int i = 0; same && scale && i < d; ++i) same = pf.scale[i] == scale[i] && pf.mn[i] == mn[i];
        const bool ok = prefetch_join(h);
        pf.active = false;
        if (same && ok) {
            std::swap(h->X.p, pf.X.p); std::swap(h->X.n, pf.X.n);
            std::swap(h->Xt.p, pf.Xt.p); std::swap(h->Xt.n, pf.Xt.n);
            h->lab_uid.ensure((size_t)n);
            h->lab_path.ensure((size_t)n);
            h->n_points = n;
            if (pf.bad_host[0]) {
                h->n_points = 0;
                return fail(h, CC_ERR_NONFINITE, "input points contain NaN or Inf");
            }
            h->x_absmax = absmax_of(pf.bad_host);
            return (int)CC_OK;
        }
    }
    h->X.ensure((size_t)n * d);
    h->Xt.ensure((size_t)n * d);
    h->lab_uid.ensure((size_t)n);
    h->lab_path.ensure((size_t)n);
    h->n_points = n;
    if (n == 0) return (int)CC_OK;
    HIPCHK(hipMemcpyAsync(h->X.p, x, (size_t)n * d * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->badflag.p, 0, 16, h->stream));
    const long long tot = (long long)n * d;
    if (scale) {
        h->scr2.ensure((size_t)2 * d);
        HIPCHK(hipMemcpyAsync(h->scr2.p, scale, (size_t)d * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->scr2.p + d, mn, (size_t)d * 8, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_scale_points, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->X.p, tot, (int)d,
                           h->scr2.p, h->scr2.p + d);
    }
    int blocks = (int)std::min<long long>((tot + 255) / 256, 4096);
    hipLaunchKernelGGL(k_check_finite, dim3(blocks), dim3(256), 0, h->stream, h->X.p, tot, h->badflag.p);
    hipLaunchKernelGGL(k_transpose_points, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->X.p,
                       h->Xt.p, (long long)n, (int)d);
    int bad[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(bad, h->badflag.p, 16, hipMemcpyDeviceToHost, h->stream));
    sync_stream(h, h->stream);
    if (bad[0]) {
        h->n_points = 0;
        return fail(h, CC_ERR_NONFINITE, "input points contain NaN or Inf");
    }
    h->x_absmax = absmax_of(bad);
    return (int)CC_OK;
}

int cc_points_prefetch(cc_handle* h, const double* x, int64_t n, int32_t d, const double* scale, const double* min_)
{
    if (!h || !x || n <= 0 || d <= 0 || d > CC_MAX_DIM || ((scale == nullptr) != (min_ == nullptr))) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        prefetch_discard(h);
        cc_handle::Prefetch& pf = h->pf;
        pf.x = x; pf.n = n; pf.d = d; pf.scaled = scale != nullptr;
        pf.scale.assign(scale ? scale : x, scale ? scale + d : x);
        pf.mn.assign(min_ ? min_ : x, min_ ? min_ + d : x);
        pf.rc = 0; pf.what = ""; pf.bad_host[0] = pf.bad_host[1] = pf.bad_host[2] = pf.bad_host[3] = 0;
        if (!pf.stream) HIPCHK(hipStreamCreateWithFlags(&pf.stream, hipStreamNonBlocking));
        const size_t chunk = (size_t)16 << 20;
        if (pf.pin_bytes < chunk) {
            for (int q = 0; q < 2; ++q) {
                if (pf.pin[q]) (void)hipHostFree(pf.pin[q]);
                pf.pin[q] = nullptr;
                HIPCHK(hipHostMalloc(&pf.pin[q], chunk, hipHostMallocDefault));
            }
            pf.pin_bytes = chunk;
        }
        pf.X.ensure((size_t)n * d); pf.Xt.ensure((size_t)n * d); pf.sm.ensure((size_t)2 * d); pf.bad.ensure(4);
        pf.active = true;
        const int device = h->device;
        pf.worker = std::thread([&pf, device, chunk]() {
            auto chk = [&](hipError_t e, const char* what) {
                if (e != hipSuccess && pf.rc == 0) { pf.rc = (int)e; pf.what = what; }
                return e == hipSuccess;
            };
            if (!chk(hipSetDevice(device), "hipSetDevice")) return;
            const size_t bytes = (size_t)pf.n * pf.d * 8;
            hipEvent_t ev[2] = {nullptr, nullptr};
            for (int q = 0; q < 2; ++q)
                if (!chk(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming), "hipEventCreate")) return;
            int k = 0;
            for (size_t off = 0; off < bytes && pf.rc == 0; off += chunk, k ^= 1) {
                const size_t len = std::min(chunk, bytes - off);
                if (off >= 2 * chunk) chk(hipEventSynchronize(ev[k]), "hipEventSynchronize");  // the staging buffer is free again
                memcpy(pf.pin[k], (const char*)pf.x + off, len);
                chk(hipMemcpyAsync((char*)pf.X.p + off, pf.pin[k], len, hipMemcpyHostToDevice, pf.stream), "hipMemcpyAsync");
                chk(hipEventRecord(ev[k], pf.stream), "hipEventRecord");
            }
            const long long tot = pf.n * (long long)pf.d;
            if (pf.rc == 0) {
                chk(hipMemsetAsync(pf.bad.p, 0, 16, pf.stream), "hipMemsetAsync");
                if (pf.scaled) {
                    chk(hipMemcpyAsync(pf.sm.p, pf.scale.data(), (size_t)pf.d * 8, hipMemcpyHostToDevice, pf.stream), "hipMemcpyAsync");
                    chk(hipMemcpyAsync(pf.sm.p + pf.d, pf.mn.data(), (size_t)pf.d * 8, hipMemcpyHostToDevice, pf.stream), "hipMemcpyAsync");
                    hipLaunchKernelGGL(k_scale_points, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, pf.stream, pf.X.p, tot,
                                       pf.d, pf.sm.p, pf.sm.p + pf.d);
                }
                const int blocks = (int)std::min<long long>((tot + 255) / 256, 4096);
                hipLaunchKernelGGL(k_check_finite, dim3(blocks), dim3(256), 0, pf.stream, pf.X.p, tot, pf.bad.p);
                hipLaunchKernelGGL(k_transpose_points, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, pf.stream, pf.X.p, pf.Xt.p,
                                   pf.n, pf.d);
                chk(hipMemcpyAsync(pf.bad_host, pf.bad.p, 16, hipMemcpyDeviceToHost, pf.stream), "hipMemcpyAsync");
                chk(hipGetLastError(), "kernel launch");
            }
            chk(hipStreamSynchronize(pf.stream), "hipStreamSynchronize");
            for (int q = 0; q < 2; ++q) (void)hipEventDestroy(ev[q]);
        });
        return (int)CC_OK;
    });
}

int cc_points_upload_scaled(cc_handle* h, const double* x, int64_t n, int32_t d, const double* scale, const double* min_)
{
    if (!h || (!x && n > 0) || n < 0 || !scale || !min_) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() { return upload_points(h, x, n, d, scale, min_); });
}

int cc_points_download(cc_handle* h, double* out, const double* scale, const double* min_)
{
    if (!h || !out || ((scale == nullptr) != (min_ == nullptr))) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        const long long tot = h->n_points * (long long)h->d;
        if (tot == 0) return (int)CC_OK;
        const int d = h->d;
        if (!scale) {
            HIPCHK(hipMemcpyAsync(out, h->X.p, (size_t)tot * 8, hipMemcpyDeviceToHost, h->stream));
        } else {
            h->scr.ensure((size_t)tot);
            h->scr2.ensure((size_t)2 * d);
            HIPCHK(hipMemcpyAsync(h->scr2.p, scale, (size_t)d * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->scr2.p + d, min_, (size_t)d * 8, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_unscale_points, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->X.p, h->scr.p,
                               tot, d, h->scr2.p, h->scr2.p + d);
            HIPCHK(hipMemcpyAsync(out, h->scr.p, (size_t)tot * 8, hipMemcpyDeviceToHost, h->stream));
        }
        sync_stream(h, h->stream);
        return (int)CC_OK;
    });
}

}  // extern "C"

namespace {

int online_range(cc_handle* h, long long range_a, long long range_e, bool no_create, bool resume);

// sum over the ranks of buf[0 .. count), the same result on every rank, ordered on `st`
void comm_all_reduce_sum(cc_handle* h, double* buf, size_t count, hipStream_t st)
{
    cc::Comm& cm = h->comm;
    // (fail_group() drops the communicators, so a broken group no longer looks like an RCCL one: ask first)
    if (cm.broken) throw cc::CommErr{"the group has failed earlier"};
    if (cm.rccl()) {
        cm.check(cc::RcclApi::get().AllReduce(buf, buf, count, ncclDouble, ncclSum, cm.lane(0), st), "ncclAllReduce");
        return;
    }
    if (!cm.local || cm.world == 1) return;
    h->r_gather.ensure((size_t)cm.world * count);
    cm.all_gather(buf, h->r_gather.p, count * 8, st);
    hipLaunchKernelGGL(k_sum_ranks, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, h->r_gather.p, cm.world, count, buf);
}

// Relaxed multi-GPU mode: the points of the timepoint are sharded over the ranks in contiguous blocks; per super-step
// every rank clusters `relaxed_minibatch` of its points against the shared table (exact path, no MC creation), the CF
// changes are all-reduced, and the set-aside points of all ranks are clustered redundantly on every rank (exact path).
int online_relaxed(cc_handle* h)
{
    const long long N = h->n_points;
    const int W = h->comm.world, rank = h->comm.rank, d = h->d;
    if (N == 0) return (int)CC_OK;
    if (d == 0) return fail(h, CC_ERR_BAD_ARG, "no points uploaded");
    const long long L = (N + W - 1) / W;  // shard length
    const long long a0 = std::min(N, (long long)rank * L), e0 = std::min(N, a0 + L);
    const long long b = h->relaxed_minibatch;
    // Mini-batches grow from 2 048 points per rank by doubling: while the table is (nearly) empty every point is set
    // aside and clustered by all ranks redundantly, so the first super-steps are kept small; the schedule depends on
    // nothing but the shard length, hence is the same on every rank.
    std::vector<long long> starts(1, 0);
    for (long long sz = std::min<long long>(b, 2048); starts.back() < L; sz = std::min(b, sz * 2)) starts.push_back(std::min(L, starts.back() + sz));
    const long long steps = (long long)starts.size() - 1;
    hipStream_t st = h->stream;
    memset(&h->rstats, 0, sizeof(h->rstats));
    struct Suspend {  // (restored on every way out)
        cc_handle* h;
        explicit Suspend(cc_handle* hh) : h(hh) { h->shard_suspended = true; }
        ~Suspend() { h->shard_suspended = false; }
    } suspend(h);
    // labels: room for every rank's padded shard (the final all-gather is in place); -1 = not clustered yet
    if (h->lab_uid.n < (size_t)(L * W)) { h->lab_uid.ensure((size_t)(L * W)); h->lab_path.ensure((size_t)(L * W)); }
    HIPCHK(hipMemsetAsync(h->lab_uid.p, 0xFF, (size_t)(L * W) * 8, st));
    HIPCHK(hipMemsetAsync(h->lab_path.p, 0, (size_t)(L * W), st));
    h->r_didx.ensure((size_t)b + 1);
    h->r_didx_all.ensure((size_t)W * (b + 1));
    std::vector<int> didx_host((size_t)W * (b + 1)), list;
    for (long long sidx = 0; sidx < steps; ++sidx) {
        const long long a = std::min(e0, a0 + starts[sidx]), e = std::min(e0, a0 + starts[sidx + 1]);
        // ---- snapshot of the table all ranks share ----
        refresh_ctl_params(h);
        const int M = h->hc.m_rows;
        const int n_pkeys0 = h->hc.n_pkeys;
        const long long pid0 = h->hc.pcore_last_id;
        const size_t md = (size_t)M * d, dl = (size_t)M * (2 * d + 1);
        if (M > 0) {
            h->rs_cf1.ensure(md); h->rs_cf2.ensure(md); h->rs_w.ensure(M); h->rs_kind.ensure(M); h->rs_key.ensure(M);
            h->rs_id.ensure(M); h->r_delta.ensure(dl);
            HIPCHK(hipMemcpyAsync(h->rs_cf1.p, h->tab.cf1.p, md * 8, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(h->rs_cf2.p, h->tab.cf2.p, md * 8, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(h->rs_w.p, h->tab.w.p, (size_t)M * 8, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(h->rs_kind.p, h->tab.kind.p, (size_t)M * 4, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(h->rs_key.p, h->tab.key.p, (size_t)M * 4, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(h->rs_id.p, h->tab.id.p, (size_t)M * 8, hipMemcpyDeviceToDevice, st));
        }
        // ---- A: this rank's mini-batch, no MC creation ----
        const auto tA0 = std::chrono::steady_clock::now();
        if (h->trace) sync_stream(h, st);
        const auto tA1 = std::chrono::steady_clock::now();
        int rc = online_range(h, a, e, true, sidx > 0);
        const auto tA2 = std::chrono::steady_clock::now();
        if (rc != CC_OK) return rc;
        if (h->hc.m_rows != M) return fail(h, CC_ERR_INTERNAL, "relaxed mode: a mini-batch created microclusters");
        // ---- M: merge the changes of the existing rows ----
        if (M > 0) {
            const Table tab = h->tab.view();  // (online_range may have moved the table)
            hipLaunchKernelGGL(k_rel_delta, dim3((unsigned)((md + 255) / 256)), dim3(256), 0, st, tab, h->rs_cf1.p, h->rs_cf2.p,
                               h->rs_w.p, M, d, h->r_delta.p);
            comm_all_reduce_sum(h, h->r_delta.p, dl, st);
            hipLaunchKernelGGL(k_rel_merge, dim3((unsigned)((md + 255) / 256)), dim3(256), 0, st, tab, h->rs_cf1.p, h->rs_cf2.p,
                               h->rs_w.p, h->rs_kind.p, h->rs_key.p, h->rs_id.p, M, d, h->r_delta.p, h->hc.delta_sq, h->hc.k,
                               h->hc.pow2, h->hc.inv_k);
            hipLaunchKernelGGL(k_rel_promote, dim3(1), dim3(1024), 0, st, h->ctl.p, tab, M, d, h->r_delta.p, h->hc.beta_mu,
                               h->hc.pi, n_pkeys0, pid0);
        }
        // ---- B: the set-aside points of all ranks, in rank order, on every rank ----
        hipLaunchKernelGGL(k_rel_collect, dim3(1), dim3(1024), 0, st, h->lab_uid.p, a, e, h->r_didx.p);
        // their numbers first (4 bytes per rank); the index lists only travel when there are any - in the steady state
        // there are none
        h->r_cnt_all.ensure((size_t)W);
        h->comm.all_gather(h->r_didx.p, h->r_cnt_all.p, 4, st);
        std::vector<int> cnt_host((size_t)W);
        HIPCHK(hipMemcpyAsync(cnt_host.data(), h->r_cnt_all.p, (size_t)W * 4, hipMemcpyDeviceToHost, st));
        pull_ctl(h);  // (synchronises the stream; the counters k_rel_promote left)
        long long total = 0;
        for (int r = 0; r < W; ++r) total += cnt_host[r];
        list.clear();
        if (total > 0) {  // (the same decision on every rank: the counts are the gathered ones)
            h->comm.all_gather(h->r_didx.p, h->r_didx_all.p, (size_t)(b + 1) * 4, st);
            HIPCHK(hipMemcpyAsync(didx_host.data(), h->r_didx_all.p, didx_host.size() * 4, hipMemcpyDeviceToHost, st));
            sync_stream(h, st);
            for (int r = 0; r < W; ++r) {
                const int* blk = didx_host.data() + (size_t)r * (b + 1);
                list.insert(list.end(), blk + 1, blk + 1 + blk[0]);
            }
        }
        if (h->trace) {
            const auto tA3 = std::chrono::steady_clock::now();
            auto ms = [](auto x, auto y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
            fprintf(stderr, "[cc] relaxed super-step %lld: %lld points | snapshot %.3f ms, sharded half %.3f ms, merge + collect %.3f ms, set aside %lld\n",
                    sidx, e - a, ms(tA0, tA1), ms(tA1, tA2), ms(tA2, tA3), total);
        }
        h->rstats.super_steps += 1;
        h->rstats.minibatch_points += e - a;
        const long long K = (long long)list.size();
        if (K > 0) {
            h->rstats.deferred_points += K;
            h->r_didx_all.ensure((size_t)std::max<long long>((long long)W * (b + 1), K));
            HIPCHK(hipMemcpyAsync(h->r_didx_all.p, list.data(), (size_t)K * 4, hipMemcpyHostToDevice, st));
            h->rg_X.ensure((size_t)K * d); h->rg_Xt.ensure((size_t)K * d); h->rg_uid.ensure((size_t)K); h->rg_path.ensure((size_t)K);
            hipLaunchKernelGGL(k_rel_gather_points, dim3((unsigned)(((size_t)K * d + 255) / 256)), dim3(256), 0, st, h->X.p,
                               h->r_didx_all.p, (int)K, d, h->rg_X.p);
            hipLaunchKernelGGL(k_transpose_points, dim3((unsigned)(((size_t)K * d + 255) / 256)), dim3(256), 0, st, h->rg_X.p,
                               h->rg_Xt.p, K, d);
            // the gathered points take the place of the resident ones for one exact run
            auto swap_in = [&]() {
                std::swap(h->X.p, h->rg_X.p); std::swap(h->X.n, h->rg_X.n);
                std::swap(h->Xt.p, h->rg_Xt.p); std::swap(h->Xt.n, h->rg_Xt.n);
                std::swap(h->lab_uid.p, h->rg_uid.p); std::swap(h->lab_uid.n, h->rg_uid.n);
                std::swap(h->lab_path.p, h->rg_path.p); std::swap(h->lab_path.n, h->rg_path.n);
            };
            swap_in();
            h->n_points = K;
            // (the window policy this rank's mini-batches settled on is not the business of the replicated half)
            const int keep_win = h->adapt_win, keep_clean = h->clean_batches, keep_shrink = h->since_shrink;
            auto restore_policy = [&]() { h->adapt_win = keep_win; h->clean_batches = keep_clean; h->since_shrink = keep_shrink; };
            try {
                rc = online_range(h, 0, K, false, false);
            } catch (...) {
                swap_in();
                h->n_points = N;
                restore_policy();
                throw;
            }
            swap_in();
            h->n_points = N;
            restore_policy();
            if (rc != CC_OK) return rc;
            hipLaunchKernelGGL(k_rel_scatter_labels, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st, h->rg_uid.p,
                               h->rg_path.p, h->r_didx_all.p, (int)K, h->lab_uid.p, h->lab_path.p);
        }
    }
    // every rank's shard of the labels to every rank (in place, shards padded to the same length)
    h->comm.all_gather(h->lab_uid.p + (size_t)rank * L, h->lab_uid.p, (size_t)L * 8, st);
    h->comm.all_gather(h->lab_path.p + (size_t)rank * L, h->lab_path.p, (size_t)L, st);
    sync_stream(h, st);
    HIPCHK(hipGetLastError());
    h->stats.points = N;
    return (int)CC_OK;
}

// One call of the exact windowed online phase over the resident points [range_a, range_e): the state that lives across its
// batches of windows, and what happens to it - prepare(), then per iteration either a stint of the sequential kernel or a
// batch of windows enqueued (enqueue_batch) and read back (after_batch: the policy's decision for the next one) -, finish().
// online_range() below is its only user.
struct OnlineRun {
    cc_handle* const h;
    const long long range_a, N;
    const bool no_create, resume;
    Ctl& c;  // the host mirror of the control block (h->hc)

    // ---- constants of the call ----
    int win = 0, R = 0;
    int S_cfg = 1, Sd_full = 1, Sd = 1;  // partials per point: clean scans (refined per batch) / dirty scans
    int world = 1, myrank = 0;
    bool grouped = false;
    size_t batch_max = 2;
    bool timing = false;
    int seq_mode = 0, seq_cap = 0;
    Versions ver{};
    Carry car{};
    hipStream_t sA = nullptr, sB = nullptr;
    static constexpr size_t ev_base = 4;

    // ---- the window policy and its current decision ----
    cc_policy_config pcfg{};
    std::optional<cc::WindowPolicy> policy;
    std::optional<PolicyTrace> ptrace;
    cc_policy_decision dec{};
    // While k_dseed rules the dirty scans out for every tile they are not launched at all (beside a lookahead scan
    // even a launch whose workgroups all return at once waits for registers until the scan has dispatched its last
    // workgroup); k_decide then refuses points that would have needed them, the device idles the rest of the batch
    // if that stops a window at its first point, and the next batch launches them again.
    bool nodirty = false;
    bool sparse_now = false;  // with nodirty: sparse dirty scans for the round's list of points
    bool la_on = false;       // lookahead scans are being enqueued
    bool shard_on = false;    // snapshot scans are split over the ranks of the group
    bool link_now = false;    // round 0 of this batch's windows links the points that decide "create" (cc_link.h)
    int Rcur = 1;             // validation rounds enqueued per window of the batch
    int batch_windows = 2;

    // ---- progress ----
    long long done = 0;       // the device's cursor as last read back
    int m_known = 0;          // table rows as last read back
    unsigned long long seq_host = 0;  // sequence number of the window the next iteration validates
    long long cursor_prev = 0;
    double batch_t0 = 0.0;

    // ---- events, timing, statistics ----
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evCommit = nullptr, evScan = nullptr;
    size_t ev_used = 2, ev_sync = ev_base;
    std::vector<std::pair<size_t, double>> timed;  // (event index, 1.0 for a pruned chain)
    std::vector<size_t> timed_comm;                // event index of every timed merge + all-gather
    double pair_rows_eff = 0.0, pair_rows_prev = 0.0;  // (window points x table rows) this rank's scans covered
    double pair_rows_pruned = 0.0;                     // ... of those, by pruned chains
    long long sharded_windows = 0;

    // ---- the sequential kernel's wall-clock rule (off inside a group) ----
    bool seq_on = false;
    int bad_batches = 0;              // consecutive batches of short, truncated windows
    long long seq_stint_len = 32768, seq_stint_left = 32768;
    bool seq_probe = false;           // the batch of windows in flight is a probe after a sequential stint
    double win_rate = 0.0, seq_rate_last = 0.0;  // points per millisecond (wall clock) of the last batch / chunk

    // ---- long chains (k_chain_long over the list k_decide keeps) ----
    long long long_prev = 0;   // Ctl::stat_long at the end of the previous batch
    bool long_seen = false;    // ... and whether that batch added to it
    bool long_few = true;      // ... by no more than 64 chains per validation round
    long long rounds_prev = 0, long_avg = 1;
    long long long_launches = 0;

    OnlineRun(cc_handle* handle, long long a, long long e, bool no_create_, bool resume_)
        : h(handle), range_a(a), N(e), no_create(no_create_), resume(resume_), c(handle->hc) {}

    static double now_ms()
    {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    // points per ms the sequential kernel is assumed to manage before it has been measured in this call (k_seq on its LDS
    // image: ~0.9 us per point; k_seq_r, rows in registers, d <= 4: ~0.6 us)
    bool seq_r_applies() const { return h->allow_seq_r && h->d >= 2 && h->d <= 4; }
    // (k_seq_g, beyond the LDS image: 3-5 us per point at a few hundred rows)
    // d > CC_WINDOW_MAX_DIM: the windowed path (two dimensions per lane of a 32-lane group, 2 d registers per point in the scans)
    // does not take such points; k_seq_g does, from the first one on
    bool wide() const { return h->d > CC_WINDOW_MAX_DIM; }
    bool seq_g_applies() const { return h->allow_seq_g && h->hc.m_rows >= seq_cap; }
    double seq_rate_guess() const { return cc::seq_rate_guess(h->d, h->hc.m_rows, seq_cap, h->allow_seq_r, h->allow_seq_g); }
    // (never in a group - every rank has to take the same path, and wall-clock measurements differ -, never with no_create:
    // the sequential kernels know the reference's loop only)
    bool seq_possible() const { return seq_mode != 1 && !h->comm.active() && !no_create && (h->hc.m_rows < seq_cap || h->allow_seq_g); }

    // lookahead (re)start: the current window is a fresh one (scanned in place), the lookahead scan enqueued next covers
    // the one after it
    void set_lookahead(bool on)
    {
        // (re)start: the current window is a fresh one, the lookahead scan enqueued next covers the one after it
        la_on = on;
        c.la_on = on ? 1 : 0;
        c.stall_b = 0;
        c.mode = 0;
        c.car_n = 0;
        const int q = (int)((c.window_seq + 1ull) & 1ull);
        const long long c1 = c.cursor + c.win_b;
        const long long left1 = c.n_points - c1;
        c.la_cursor[q] = c1;
        c.la_b[q] = (on && left1 > 0) ? (int)std::min<long long>(left1, c.win_cfg) : 0;
        c.la_rows[q] = c.m_rows;
        c.la_cursor[q ^ 1] = 0;
        c.la_b[q ^ 1] = 0;
        c.la_rows[q ^ 1] = 0;
    }

    // buffers, control block, policy: everything before the first batch
    void prepare()
    {
        win = h->tun.window; R = h->tun.rounds;
        // `segments` MC sub-ranges per point tile = S workgroups of 4 waves -> S partials per point
        S_cfg = std::max(1, h->tun.segments / scan_waves_for_dim(h->d, false));
        Sd_full = std::max(1, (h->tun.dirty_segments > 0 ? h->tun.dirty_segments : h->tun.segments) /
                                            scan_waves_for_dim(h->d, true));
        // while the dirty scans are ruled out tile by tile (k_dseed) their launches only have to be scheduled: a
        // few workgroups per point tile then, the full split while they really run (set per batch below)
        Sd = Sd_full;
        // While k_dseed rules the dirty scans out for every tile they are not launched at all (beside a lookahead scan
        // even a launch whose workgroups all return at once waits for registers until the scan has dispatched its last
        // workgroup); k_decide then refuses points that would have needed them, the device idles the rest of the batch
        // if that stops a window at its first point, and the next batch launches them again.
        nodirty = false;
        refresh_ctl_params(h);
        ensure_window_buffers(h, win, std::max(S_cfg, Sd_full));
        // Exact multi-GPU path: while the table is large enough, every rank scans its share of the table rows and
        // the ranks all-gather one merged candidate record per window point; the rest of the window runs replicated.
        // All ranks take the same decision: it depends on the row count only, which is the same everywhere.
        world = h->comm.world; myrank = h->comm.rank;
        // (a communicator of one rank takes the same path: that is how the RCCL calls are exercised on one GPU)
        grouped = h->comm.active();
        if (grouped) {
            // (+ 4: the record behind the last point's carries the rank's pruned-scan sample, see k_merge_partials)
            // (grids cover at least 64 points, see gw below: the blocks are sized for that even when the window is smaller)
            const size_t gmax = (size_t)std::max(64, h->win_alloc);
            h->gsend_stride = gmax * 4 + 4;
            h->gpart_stride = (size_t)world * (gmax * 4 + 4);
            h->gsend.ensure(2 * h->gsend_stride);
            h->gpart.ensure(2 * h->gpart_stride);
            h->gsend2.ensure((size_t)2 * CC_MISSED_CAP * 4);  // (per window parity, like the lists they serve)
            h->gpart2.ensure((size_t)2 * world * CC_MISSED_CAP * 4);
        }
        // every window of a batch may create one MC per point: rows for the largest batch that can be enqueued
        batch_max = (size_t)std::max(2, h->tun.windows_per_sync);
        ensure_table(h, (size_t)h->hc.m_rows + (size_t)win * batch_max + 1);

        c.cursor = range_a;
        c.n_points = N;
        c.xt_stride = h->n_points;
        c.no_create = no_create ? 1 : 0;
        // How the batches of windows run - window size, validation rounds, windows per batch, lookahead, dirty scans,
        // pruned or plain scans, split over the ranks - is decided by cc::WindowPolicy (cc_policy.h) from the device
        // counters alone; this function carries the decisions out.
        pcfg.window = win;
        pcfg.rounds_max = R;
        pcfg.windows_per_sync = h->tun.windows_per_sync;
        pcfg.early_window = h->tun.early_window;
        pcfg.lookahead = h->tun.lookahead;
        pcfg.allow_nodirty = h->allow_nodirty ? 1 : 0;
        pcfg.prune_mode = h->prune_mode;
        const bool width_ok = h->d == 14 || h->d == 16 || h->d == 20 || h->d == 32 || h->d == 40;
        // (the pdim filter on and / or k not a power of two: k_scan_p3<GENERAL> behind seeded thresholds - no guesses, no probes)
        const bool prune_general = width_ok && scan_p3_general_applies(h, h->d);
        pcfg.prune_applicable = ((h->d > 8 && h->allow_scan_u && h->hc.filter == 0 && h->hc.pow2 != 0 && (width_ok || h->d == 64)) || prune_general) ? 1 : 0;
        pcfg.can_shard = (grouped && !h->shard_suspended) ? 1 : 0;
        pcfg.d = h->d;
        pcfg.resume = resume ? 1 : 0;
        pcfg.allow_sparse = h->allow_sparse;
        pcfg.allow_guess = (h->allow_guess && !prune_general) ? (h->allow_lean ? 1 : 2) : 0;  // (2: guessed thresholds, never lean)
        pcfg.allow_probe = (h->allow_probe && !prune_general) ? 1 : 0;
        pcfg.lookahead_pruned = h->la_pruned ? 1 : 0;
        // (pruned scans from this many rows on whatever the phase: where the seeded chain runs behind k_seed16)
        pcfg.force_prune_rows = (pcfg.prune_applicable && h->allow_seed16 && h->allow_scan_p3 && h->d <= 40 && h->split_a_mode != 2) ? h->force_prune_rows : 0;
        pcfg.shard_min_row_dims = h->shard_min_row_dims;
        pcfg.shard_min_row_dims_pruned = h->shard_min_row_dims_pruned;
        pcfg.n_end = N;
        const cc_policy_carry pcarry{h->adapt_win, h->clean_batches, h->since_shrink, 0};
        policy.emplace(pcfg, pcarry);
        dec = policy->start(range_a, c.m_rows);
        ptrace.emplace(pcfg, pcarry, range_a, c.m_rows, dec, grouped ? myrank : -1);
        c.win_cfg = dec.win_cfg;
        c.win_b = (int)std::min<long long>(c.win_cfg, N - range_a);
        c.max_rounds = R;
        c.last_round = 0;
        c.fc[0] = 0;
        for (int i = 1; i < CC_MAX_ROUNDS + 2; ++i) c.fc[i] = CC_IDX_INF;
        c.stat_windows = c.stat_rounds = c.stat_truncated = 0;
        c.stat_lookahead = 0;
        c.stat_tiles = c.stat_dirty_tiles = 0;
        c.stat_unprovable = c.stat_unsafe = 0;
        c.stat_long = 0;
        for (int i = 0; i < CC_MAX_ROUNDS + 2; ++i) c.n_long[i] = 0;
        c.stat_trunc_unknown = 0;
        c.stat_table_rows = 0;
        c.stat_seq_points = 0;
        c.stat_seq_r_points = 0;
#ifdef CC_LONG_TIMERS
        for (int i = 0; i < 8; ++i) c.dbg_long[i] = 0;
#endif
#ifdef CC_ROUND_DEBUG
        for (int i = 0; i < CC_MAX_ROUNDS + 2; ++i)
            for (int q = 0; q < 6; ++q) c.dbg_round[i][q] = 0;
#endif
        c.n_heavy = c.n_heavy_new = 0;  // (rows are renumbered between calls: the marks of the last call are void)
        HIPCHK(hipMemsetAsync(h->tab.heavy.p, 0, h->tab.cap * sizeof(int), h->stream));
        c.stat_seq_clk = c.stat_seq_wall = 0;
        c.stat_prune_rows = c.stat_prune_full = 0;
        c.stat_missed = 0;
        c.seed_at = -1;
        for (int q = 0; q < 2; ++q) {
            c.n_missed[q] = 0;
            for (int K = 0; K < 2; ++K) { c.tg[q][K] = 0.0; c.tg_ok[q][K] = 0; }
        }
        c.cen_absmax = 0ull;            // (k_rebuild_scl takes the table's maximum into it)
        c.x_absmax = h->x_absmax;
        set_lookahead(dec.lookahead != 0);
        c.stat_pair_rows = 0.0;
        for (int i = 0; i < CC_MAX_ROUNDS + 2; ++i) c.round_hist[i] = 0;
        push_ctl(h);

        // no carry set yet: the commit record of an earlier call describes rows that may have moved since
        HIPCHK(hipMemsetAsync(h->rec.p, 0, sizeof(CommitRec), h->stream));
        HIPCHK(hipMemsetAsync(h->cmax.p, 0, 2 * sizeof(unsigned long long), h->stream));  // (k_seed takes maxima into it)
        HIPCHK(hipMemsetAsync(h->found.p, 0, h->found.n * sizeof(unsigned long long), h->stream));
        // (marks of long chains laid out in an earlier call - another table, perhaps another numbering of the windows)
        HIPCHK(hipMemsetAsync(h->lstat.p, 0, h->lstat.n * sizeof(unsigned long long), h->stream));
        HIPCHK(hipMemsetAsync(h->lprev.p, 0, h->lprev.n * sizeof(unsigned long long), h->stream));
        if (c.m_rows > 0)
            hipLaunchKernelGGL(k_rebuild_scl, dim3((c.m_rows * h->d + 255) / 256), dim3(256), 0, h->stream, h->ctl.p, h->tab.view(),
                               c.m_rows, h->d, c.pow2, c.inv_k);
        ev0 = get_event(h, 0); ev1 = get_event(h, 1);
        HIPCHK(hipEventRecord(ev0, h->stream));
        ev_used = 2;
        timing = h->tun.time_kernels != 0;
        shard_on = dec.shard != 0;

        ver = versions_view(h);
        car = carry_view(h);
        sA = h->stream; sB = h->stream2;
        // cross-stream hand-offs: a fresh event per hand-off (the pool is reused from batch to batch)
        evCommit = get_event(h, 2); evScan = nullptr;
        ev_sync = ev_base;
        ev_used = ev_base + 3 * (batch_max + 2);

        done = range_a;
        m_known = c.m_rows;
        // The sequential kernel (k_seq) for streams on which speculation does not pay: used while the table fits its
        // LDS image and either the caller forces it or (default) the windows keep being cut short and it measures
        // faster than they do.  Never inside a multi-GPU group (every rank has to take the same path, and wall-clock
        // measurements differ between ranks).
        seq_mode = wide() ? 2 : h->tun.sequential;
        seq_cap = wide() ? 0 : cc_seq_cap_rows(h->d);
        seq_on = seq_possible() && (seq_mode == 2 || h->seq_sticky);
        // default policy: the sequential kernel takes over after two batches in a row whose windows were cut short
        // at a few hundred points; it works in stints (32 k points, doubling), after each of which one batch of
        // windows is run again and the two measured rates decide who continues
        Rcur = dec.rounds;
        batch_windows = dec.batch_windows;
        h->prune_now = dec.prune != 0;
        h->guess_now = dec.prune >= 2;
        h->lean_now = dec.prune == 3;
        nodirty = dec.nodirty != 0;
        sparse_now = dec.sparse != 0;
        cursor_prev = range_a;
        seq_host = c.window_seq;
        // (a call starts with whatever is new since the last one: the first batch links, the later ones while the table grows)
        link_now = h->allow_link && !no_create && !wide();
    }

    // a stint of the sequential kernel (k_seq): one chunk of points, then back to the windows if the table outgrew its LDS
    // image or the stint is over
    void sequential_stint(const Table& tab)
    {
        const int chunk = 8192;
        const double t0 = now_ms();
        const bool use_g = seq_g_applies();
        if (use_g) {
            // the table has outgrown the LDS image: the same loop on the table where it lies, one workgroup
            const bool f = h->hc.filter != 0, p2 = h->hc.pow2 != 0;
            const int list_cap = (int)std::min<size_t>(h->tab.cap, (size_t)INT_MAX / 2);
            h->seq_lists.ensure(2 * (size_t)list_cap);
            h->seq_img.ensure(4 * (size_t)list_cap * (size_t)h->d);
#define CC_SEQG(F, P) hipLaunchKernelGGL((k_seq_g<F, P>), dim3(1), dim3(CC_SEQG_THREADS), 0, sA, h->ctl.p, h->X.p, tab, h->lab_uid.p, h->lab_path.p, chunk, h->seq_lists.p, list_cap, h->seq_img.p)
            if (f && p2) CC_SEQG(true, true);
            else if (f) CC_SEQG(true, false);
            else if (p2) CC_SEQG(false, true);
            else CC_SEQG(false, false);
#undef CC_SEQG
        } else {
            const bool f = h->hc.filter != 0, p2 = h->hc.pow2 != 0;
            // d <= 4: the register-resident kernel first; what it cannot hold (Ctl::seq_rest) is left to the LDS kernel
            int follow = 0;
#define CC_SEQR(D) do { \
                if (p2) hipLaunchKernelGGL((k_seq_r<D, true>), dim3(1), dim3(64), 0, sA, h->ctl.p, h->X.p, tab, h->lab_uid.p, h->lab_path.p, chunk); \
                else hipLaunchKernelGGL((k_seq_r<D, false>), dim3(1), dim3(64), 0, sA, h->ctl.p, h->X.p, tab, h->lab_uid.p, h->lab_path.p, chunk); \
                follow = 1; } while (0)
            if (h->allow_seq_r) {
                if (h->d == 2) CC_SEQR(2);
                else if (h->d == 3) CC_SEQR(3);
                else if (h->d == 4) CC_SEQR(4);
            }
#undef CC_SEQR
#define CC_SEQ(F, P) hipLaunchKernelGGL((k_seq<F, P>), dim3(1), dim3(64), 0, sA, h->ctl.p, h->X.p, tab, h->lab_uid.p, h->lab_path.p, chunk, follow)
            if (f && p2) CC_SEQ(true, true);
            else if (f) CC_SEQ(true, false);
            else if (p2) CC_SEQ(false, true);
            else CC_SEQ(false, false);
#undef CC_SEQ
        }
        HIPCHK(hipGetLastError());
        pull_ctl(h);
        const double dt = now_ms() - t0;
        const long long got = h->hc.cursor - done;
        done = h->hc.cursor;
        m_known = h->hc.m_rows;
        seq_host = h->hc.window_seq;
        cursor_prev = h->hc.cursor;
        const double seq_rate = got > 0 ? (double)got / std::max(dt, 1e-3) : 0.0;
        if (h->trace)
            fprintf(stderr, "[cc] done %lld rows %d | sequential kernel: %lld points in %.3f ms (so far %lld shader cycles, %.3f ms of kernel time)\n",
                    done, h->hc.m_rows, got, dt, (long long)h->hc.stat_seq_clk, (double)h->hc.stat_seq_wall / 1e5);
        if (got >= 1024) seq_rate_last = seq_rate;
        seq_stint_left -= got;
        if (use_g) h->stats.seq_g_points += got;
        // (k_seq hands back early when its image is full: k_seq_g continues the stint; k_seq_g itself only when the table's
        // capacity is used up - the windows' loop makes room)
        const bool full = !seq_possible() || (got < chunk && done < N && (use_g || !h->allow_seq_g));
        const bool stint_over = seq_mode != 2 && seq_stint_left <= 0;
        if ((full || stint_over) && done < N && !wide()) {
            // back to the windows: a fresh window at the cursor, no carry set, no pending lookahead scan
            seq_on = false;
            seq_probe = stint_over && !full;
            bad_batches = 0;
            HIPCHK(hipMemsetAsync(h->rec.p, 0, sizeof(CommitRec), h->stream));
            h->hc.win_b = (int)std::min<long long>(h->hc.win_cfg, N - done);
            dec = policy->after_sequential(h->hc.cursor, h->hc.m_rows);
            ptrace->sequential(h->hc.cursor, h->hc.m_rows, dec);
            nodirty = dec.nodirty != 0;
            sparse_now = dec.sparse != 0;
            set_lookahead(dec.lookahead != 0);
            push_ctl(h);
        }
    }

    // one batch of windows: per window the snapshot scan (in place or one window ahead on the second stream), round 0 of
    // the decisions, the validation rounds, the commit - all enqueued without a host round-trip
    void enqueue_batch(const Table& tab)
    {
        batch_t0 = now_ms();
        // pruned snapshot scans for this batch?  (a function of device counters only: every rank decides alike)
        // (h->prune_now was set for this batch at the end of the previous one, together with the lookahead restart a
        // change of it needs: a pruned scan leaves fewer partials per point than a plain one)
        const Rows trows{tab.cen, tab.scl, tab.pref, tab.cf1, tab.cf2, tab.w, tab.kind, tab.key, nullptr, nullptr, nullptr,
                         nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        const Rows vrows{ver.cen, ver.scl, ver.pref, ver.cf1, ver.cf2, ver.w, ver.kind, ver.key, ver.next,
                         ver.tile_dsq, ver.dsq, ver.tau, ver.skip, nullptr, nullptr, nullptr, 0};
        const Rows crows{car.cen, car.scl, car.pref, car.cf1, car.cf2, car.w, car.kind, car.key, nullptr,
                         car.tile_dsq, car.dsq, ver.tau, ver.skip_car, nullptr, car.slot, tab.touch, tab.cap};
        // the sparse dirty scans: the same rows for the round's list of points instead of the window's tiles
        Rows vrows_sp = vrows, crows_sp = crows;
        vrows_sp.skip = nullptr; vrows_sp.plist = h->sp_list.p;
        crows_sp.skip = nullptr; crows_sp.plist = h->sp_list.p;
        ev_sync = ev_base;
        // grids cover the window size of this batch (no window of the batch is larger), not the configured maximum
        const int gw = std::max(64, std::min(win, h->hc.win_cfg));
        // partials per point of this batch's clean scans (a pending lookahead scan was launched with the same value:
        // it only depends on the window size, and a change of that restarts the lookahead chain)
        // A pruned scan spends a few VALU instructions per row, so a wave must own many rows for its fixed costs
        // (points, thresholds, tile pipeline, candidate merge: microseconds) not to dominate: as few sub-ranges as fill
        // the machine once (about a fifth of the plain scan's partials at the full window).
        const int scan_cus = h->n_cus;
        // (measured, `profiles/r03_tool_prune_split.txt`: four rounds of the resident workgroups at d <= 20 - k_scan_p is
        // compiled for four per CU there -, eight of the three per CU beyond)
        // (round 6, k_scan_p3: the prefix test costs next to nothing on the matrix cores, what is left of a workgroup's time is its
        // prologue - the points staged, their constants - and the rows it completes: two rounds of the resident workgroups at
        // d <= 20, 54 against 63 us per C2 window running alone, half the partials per point)
        const bool p3 = h->allow_scan_p3 && h->d <= 40 && h->split_a_mode != 2;
        const int prune_wgs = h->prune_rounds4 > 0 ? h->prune_rounds4 : (h->d <= 20 ? (p3 ? 8 : 16) : 24);
        const int S = h->prune_now ? std::max(1, std::min(S_cfg, (scan_cus * prune_wgs) / std::max(1, (gw + 63) / 64)))
                                   : scan_partials_for((gw + 63) / 64, S_cfg, scan_resident_wgs(h, scan_cus));
        // capacity of the round's list for the sparse dirty scans: a sixteenth of the window (the policy's bound on
        // the batch's average), in whole tiles
        const int sparse_cap = std::min(CC_MAX_WINDOW / 16, std::max(64, ((gw / 16 + 63) / 64) * 64));
        const int decide_threads = h->decide_threads;
        const int dblocks = (gw + decide_threads / 32 - 1) / (decide_threads / 32);   // one 32-lane group per point
        const int chain_threads = h->chain_threads;  // 32-lane groups of k_chain per workgroup x 32
        const int cblocks = (gw + chain_threads / 32 - 1) / (chain_threads / 32);
        const int commit_threads = h->commit_threads;
        const int rblocks = std::min((gw + commit_threads / 32 - 1) / (commit_threads / 32), 1024 * (256 / commit_threads));
        // few MCs: the claims of a window are gathered per MC by k_claims (rows beyond scan_rows, e.g. rows created
        // during the batch, keep k_decide's atomics)
        const int scan_rows = (h->allow_claims && h->hc.m_rows > 0 && h->hc.m_rows <= 1024) ? h->hc.m_rows : 0;
        // ... and their long chains (more than CC_CHAIN_MEMB claimants; k_claims leaves the exact count) are replayed
        // by k_chain_long, one workgroup per MC, instead of one point after the other
        const int long_rows = h->allow_long ? scan_rows : 0;
        // On a larger table long chains are rare on evenly spread data and the rule on skewed data (one population
        // that takes a third of the events): k_chain_long is launched, over the list k_decide keeps, in the batches
        // that follow one in which such chains were seen (a function of device counters: every rank decides alike)
        const bool long_listed = h->allow_long && scan_rows == 0 && long_seen;
        // heavy rows: their claims are gathered by k_claims_heavy instead of k_decide's atomics from the batch after the
        // one that marked them (the marks change between windows, on the device; what the host saw at the last sync
        // decides for the whole batch whether the gathering kernel is launched - k_decide is told the same)
        const bool heavy_on = h->allow_heavy && scan_rows == 0 && h->hc.n_heavy > 0;
        int* const long_list = long_listed ? h->long_list.p : nullptr;
        // workgroups of its launches = entries k_decide may list per round: a few more than the previous batch's
        // average when that was small (a launch of hundreds of workgroups that return at once is not free)
        const int long_cap = long_few ? (int)std::min<long long>(CC_LONG_CAP, 2 * long_avg + 8) : CC_LONG_CAP;
        // lookahead scans read a scan copy of the table (see ScanCopy): both in line with the table at the start of
        // a batch, then kept up commit by commit
        ScanCopy scopy[2] = {ScanCopy{}, ScanCopy{}};
        if (la_on) scan_copy_sync(h, scopy);
        Rows srows[2];
        for (int q = 0; q < 2; ++q)
            srows[q] = Rows{scopy[q].cen, scopy[q].scl, nullptr, scopy[q].cf1, scopy[q].cf2, scopy[q].w, scopy[q].kind,
                            scopy[q].key, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        evScan = nullptr;  // the scan of the batch's first window is complete (the second stream was drained)
        if (la_on) HIPCHK(hipEventRecord(evCommit, sA));  // everything so far (table, control block) is in place
        int probe_left = (dec.probe != 0) ? 1 : 0;
        // No window beyond the end of the range: when windows commit in full, ceil(left / window) of them finish the call (a
        // window that is cut short leaves its rest to the next batch, as anywhere else).  Every window enqueued past the end
        // is a dozen launches that find nothing to do - 50-100 us each, up to fifteen of them at the end of every call
        // (profiles/r06_tool_startup_timeline_before.txt: w36-w47).  A function of counters that are the same on every rank.
        const int windows_now = (int)std::max<long long>(1, std::min<long long>(batch_windows,
                                    (N - done + (long long)std::max(1, h->hc.win_cfg) - 1) / (long long)std::max(1, h->hc.win_cfg)));
        for (int wv = 0; wv < windows_now; ++wv, ++seq_host) {
            hipEvent_t scan_end = nullptr;  // the event recorded right behind the last timed scan (nothing after it yet)
            auto timed_scan = [&](hipStream_t st, int mode, int round) {
                const Rows& rws = (mode == 1) ? srows[round & 1] : trows;
                h->probe_now = probe_left > 0 && !h->prune_now;  // (the batch's first scan carries the probe)
                if (h->probe_now) --probe_left;
                const int srank = shard_on ? myrank : 0, sworld = shard_on ? world : 1;
                // guessed thresholds with the missed points agreed on from the gathered records: whenever the scan is split
                // over more than one rank (a group of one rank takes the same steps on request, CHRONOCLUST_HIP_GROUP_GUESS=1:
                // that is how the second all-gather is exercised over RCCL on one GPU)
                h->group_guess_now = shard_on && (sworld > 1 || h->group_guess_always);
                scan_end = nullptr;
                if (timing) {
                    hipEvent_t a = get_event(h, ev_used), b = get_event(h, ev_used + 1);
                    HIPCHK(hipEventRecord(a, st));
                    launch_scan<false>(h, st, gw, rws, nullptr, h->part.p, S, round, mode, srank, sworld);
                    HIPCHK(hipEventRecord(b, st));
                    timed.push_back({ev_used, h->prune_now ? 1.0 : 0.0});  // (second: a pruned chain or a plain scan)
                    ev_used += 2;
                    if (!shard_on) scan_end = b;
                } else {
                    launch_scan<false>(h, st, gw, rws, nullptr, h->part.p, S, round, mode, srank, sworld);
                }
                if (shard_on) {
                    // the rank's S partials per point -> one record per point -> the records of all ranks, in rank
                    // order, in the gathered buffer of the window's parity (what k_decide round 0 reads)
                    const int q = (mode == 1) ? (round & 1) : (int)(seq_host & 1ull);
                    if (timing) HIPCHK(hipEventRecord(get_event(h, ev_used), st));
                    hipLaunchKernelGGL(k_merge_partials, dim3((gw + 255) / 256), dim3(256), 0, st, h->ctl.p, h->part.p,
                                       h->part_stride, S, h->gsend.p, h->gsend_stride, round, mode,
                                       (const unsigned long long*)h->pstat_p(), gw * 4,
                                       (const int*)nullptr);
                    h->comm.all_gather(h->gsend.p + (size_t)q * h->gsend_stride, h->gpart.p + (size_t)q * h->gpart_stride,
                                       ((size_t)gw * 4 + 4) * sizeof(Cand), st, st == h->stream2 ? 1 : 0);
                    if (h->prune_now && h->guess_now && h->group_guess_now && h->lean_now) ++h->stats.scan_lean_launches;
                    if (h->prune_now && h->guess_now && h->group_guess_now && !h->lean_now) {
                        // guessed thresholds: the points no rank found a pcore MC for (a function of the gathered records:
                        // the same list everywhere) go through the seeded chain on every rank's rows, their new records
                        // are exchanged in a second, small all-gather of fixed size and take the place of the old ones
                        launch_scan<false>(h, st, gw, rws, nullptr, h->part.p, S, round, mode, srank, sworld, 1);
                        Cand* const send2 = h->gsend2.p + (size_t)q * CC_MISSED_CAP * 4;
                        Cand* const recv2 = h->gpart2.p + (size_t)q * sworld * CC_MISSED_CAP * 4;
                        hipLaunchKernelGGL(k_merge_partials, dim3((CC_MISSED_CAP + 255) / 256), dim3(256), 0, st, h->ctl.p,
                                           h->part.p, h->part_stride, S, send2, (size_t)0, round, mode,
                                           (const unsigned long long*)nullptr, 0, (const int*)h->missed.p);
                        h->comm.all_gather(send2, recv2, (size_t)CC_MISSED_CAP * 4 * sizeof(Cand), st, st == h->stream2 ? 1 : 0);
                        hipLaunchKernelGGL(k_scatter_missed, dim3((CC_MISSED_CAP * sworld + 255) / 256), dim3(256), 0, st,
                                           (const Ctl*)h->ctl.p, (const int*)h->missed.p, (const Cand*)recv2, sworld,
                                           h->gpart.p, h->gpart_stride, (size_t)gw * 4 + 4, round, mode);
                    }
                    if (timing) {
                        HIPCHK(hipEventRecord(get_event(h, ev_used + 1), st));
                        timed_comm.push_back(ev_used);
                        ev_used += 2;
                    }
                }
            };
            // where k_decide round 0 finds the snapshot candidates of a point
            const Cand* const dec_part = shard_on ? h->gpart.p : h->part.p;
            const size_t dec_stride = shard_on ? h->gpart_stride : h->part_stride;
            const int dec_S = shard_on ? world : S, dec_inner = shard_on ? 1 : S;
            const size_t dec_outer = shard_on ? (size_t)gw * 4 + 4 : 0;
            const int dec_tail = shard_on ? gw * 4 : -1;  // where each rank's pruned-scan sample sits in its block
            if (la_on) {
                // first stream: this window's snapshot scan (enqueued one iteration ago on the second stream)
                if (evScan) HIPCHK(hipStreamWaitEvent(sA, evScan, 0));
                // second stream: the snapshot scan of the window after this one, against the scan copy of its
                // parity (= the table as the previous commit left it), while this window is validated on the first
                HIPCHK(hipStreamWaitEvent(sB, evCommit, 0));
                timed_scan(sB, 1, (int)((seq_host + 1ull) & 1ull));
                if (scan_end) evScan = scan_end;  // the timing event already marks the end of the scan: no second record
                else {
                    evScan = get_sync_event(h, ev_sync++);
                    HIPCHK(hipEventRecord(evScan, sB));
                }
                // only the first window of a lookahead batch can need an in-place scan (the device idles the
                // rest of a batch whose lookahead chain breaks, see Ctl::stall_b)
                if (wv == 0 && h->hc.mode == 0) timed_scan(sA, 0, 0);
            } else {
                timed_scan(sA, 0, 0);
            }
            // the scan copy of this window's parity was last read by this window's own snapshot scan: first the rows
            // of the previous commit (cc_apply_carry, extra workgroups of this launch: before this window's commit
            // overwrites the carry set), then, in k_commit_b, this window's own
            const ScanCopy sc_now = scopy[seq_host & 1ull];
            const int ac_blocks = la_on ? rblocks : 0;
            hipLaunchKernelGGL(k_decide, dim3(dblocks + ac_blocks), dim3(decide_threads), 0, sA, h->ctl.p, h->X.p, tab, ver, car,
                               dec_part, dec_stride, h->clean.p, h->dpart.p, h->dpart2.p, h->dseed.p, (const int*)nullptr,
                               h->T0.p, h->dpath.p, dec_S, Sd, 0, 0, scan_rows, dec_inner, dec_outer,
                               (const CommitRec*)h->rec.p, sc_now, ac_blocks, long_list, long_cap, dec_tail, 0, heavy_on ? 1 : 0, 0,
                               link_now ? h->link_near.p : (int*)nullptr);
            if (link_now) {
                // the window's own creators (cc_link.h): points that decided "create" and would be absorbed by an earlier
                // such point claim the microcluster that one creates - before the first chain replay, not after two of them
                launch_link_scan(h, sA, gw);
                hipLaunchKernelGGL(k_link_apply, dim3((gw + 255) / 256), dim3(256), 0, sA, h->ctl.p, tab, h->T0.p,
                                   (const int*)h->link_near.p, h->dpath.p);
                ++h->stats.link_launches;
            }
            if (scan_rows > 0)
                hipLaunchKernelGGL(k_claims, dim3(scan_rows), dim3(256), 0, sA, h->ctl.p, tab, (const int*)h->T0.p, 0, scan_rows, 0);
            if (heavy_on && ++h->stats.heavy_launches > 0)
                hipLaunchKernelGGL(k_claims_heavy, dim3(CC_HEAVY_CAP), dim3(256), 0, sA, h->ctl.p, tab, (const int*)h->T0.p, 0,
                                   long_list, long_cap, 0);
            for (int r = 1; r <= Rcur; ++r) {
                const int* told = ((r - 1) & 1) ? h->T1.p : h->T0.p;
                int* tnew = (r & 1) ? h->T1.p : h->T0.p;
                // long chains of pcore MCs: running sums first, by one workgroup per chain; the steps themselves inside k_chain,
                // the fallback (rejected steps, outlier MCs) behind it
                // (while the chains are few and long: with 200 table rows a chain is one batch of k_chain_long and the rows' 200
                // workgroups are parallel enough - the extra launch cost 2 % there, measured)
                const bool prep = h->allow_prep && ((long_rows > 0 && long_rows <= 64) || (long_listed && long_few));
                unsigned long long* const lstat = prep ? h->lstat.p : nullptr;
                unsigned long long* const lprev = prep ? h->lprev.p : nullptr;
                if (prep) h->prep_launched = true;
                if (prep && long_rows > 0)
                    hipLaunchKernelGGL((k_chain_long<true, true>), dim3(long_rows), dim3(CC_LONG_THREADS), 0, sA, h->ctl.p, h->X.p, tab,
                                       ver, car, told, r, long_rows, (const int*)nullptr, lstat, lprev);
                else if (prep)
                    hipLaunchKernelGGL((k_chain_long<true, true>), dim3(long_cap), dim3(CC_LONG_THREADS), 0, sA, h->ctl.p, h->X.p,
                                       tab, ver, car, told, r, 0, (const int*)long_list, lstat, lprev);
                hipLaunchKernelGGL(k_chain, dim3(cblocks), dim3(chain_threads), 0, sA,
                                   h->ctl.p, h->X.p, tab, ver, car, told, r, long_rows, (const unsigned long long*)lprev, lstat);
                // k_chain_long: one workgroup per table row while k_claims serves the table, else per entry of the
                // round's list.  The large workgroups (SPLIT) while they are few - rows <= 256, or a short list, judged by
                // the previous batch's count -, the small ones (two per CU) when hundreds of chains are long
                if (long_rows > 0 && long_rows <= 256)
                    hipLaunchKernelGGL((k_chain_long<true, false>), dim3(long_rows), dim3(CC_LONG_THREADS), 0, sA, h->ctl.p, h->X.p, tab,
                                       ver, car, told, r, long_rows, (const int*)nullptr, lstat, lprev);
                else if (long_rows > 0)
                    hipLaunchKernelGGL((k_chain_long<false, false>), dim3(long_rows), dim3(256), 0, sA, h->ctl.p, h->X.p, tab, ver, car,
                                       told, r, long_rows, (const int*)nullptr, lstat, lprev);
                else if (long_listed && ++long_launches > 0) {
                    if (long_few)
                        hipLaunchKernelGGL((k_chain_long<true, false>), dim3(long_cap), dim3(CC_LONG_THREADS), 0, sA, h->ctl.p, h->X.p,
                                           tab, ver, car, told, r, 0, (const int*)long_list, lstat, lprev);
                    else
                        hipLaunchKernelGGL((k_chain_long<false, false>), dim3(long_cap), dim3(256), 0, sA, h->ctl.p, h->X.p, tab, ver,
                                           car, told, r, 0, (const int*)long_list, lstat, lprev);
                }
                const bool sparse_r = nodirty && sparse_now;
                hipLaunchKernelGGL(k_dseed, dim3((gw + 63) / 64), dim3(64), 0, sA, h->ctl.p, h->X.p, tab, ver, car,
                                   h->clean.p, h->dseed.p, told, r, (const int8_t*)h->dpath.p, h->sp_list.p,
                                   sparse_r ? sparse_cap : 0);
                if (!nodirty) {
                    launch_scan<true>(h, sA, gw, vrows, h->dseed.p, h->dpart.p, Sd, r, 0);
                    if (la_on) launch_scan<true>(h, sA, gw, crows, h->dseed.p, h->dpart2.p, Sd, r, 1);
                } else if (sparse_r) {
                    // (the grid covers the list's capacity; workgroups beyond the round's count return at once)
                    launch_scan<true>(h, sA, sparse_cap, vrows_sp, h->dseed.p, h->dpart.p, Sd, r, 0);
                    if (la_on) launch_scan<true>(h, sA, sparse_cap, crows_sp, h->dseed.p, h->dpart2.p, Sd, r, 1);
                }
                hipLaunchKernelGGL(k_decide, dim3(dblocks), dim3(decide_threads), 0, sA, h->ctl.p, h->X.p, tab, ver, car, dec_part,
                                   dec_stride, h->clean.p, h->dpart.p, h->dpart2.p, h->dseed.p, told, tnew, h->dpath.p, dec_S, Sd, r, nodirty ? (sparse_r ? 2 : 1) : 0, scan_rows,
                                   dec_inner, dec_outer, (const CommitRec*)nullptr, ScanCopy{}, 0, long_list, long_cap, -1,
                                   r == Rcur ? 1 : 0, heavy_on ? 1 : 0, h->allow_quiet ? 1 : 0, (int*)nullptr);
                // (the claims of the last round are not replayed: nothing to gather either)
                if (scan_rows > 0 && r < Rcur)
                    hipLaunchKernelGGL(k_claims, dim3(scan_rows), dim3(256), 0, sA, h->ctl.p, tab, (const int*)tnew, r, scan_rows, h->allow_quiet ? 1 : 0);
                if (heavy_on && r < Rcur && ++h->stats.heavy_launches > 0)
                    hipLaunchKernelGGL(k_claims_heavy, dim3(CC_HEAVY_CAP), dim3(256), 0, sA, h->ctl.p, tab, (const int*)tnew, r,
                                       long_list, long_cap, h->allow_quiet ? 1 : 0);
            }
            hipLaunchKernelGGL(k_commit_a, dim3(1), dim3(1024), 0, sA, h->ctl.p, tab, ver, car, h->T0.p, h->T1.p,
                               h->rk.p, h->rec.p, (const Cand*)h->clean.p, (const int8_t*)h->dpath.p);
            hipLaunchKernelGGL(k_commit_b, dim3(rblocks), dim3(commit_threads), 0, sA, h->rec.p, tab, ver, car, h->rk.p, h->dpath.p,
                               h->lab_uid.p, h->lab_path.p, h->d, sc_now, h->hc.filter);
            if (la_on) {
                evCommit = get_sync_event(h, ev_sync++);
                HIPCHK(hipEventRecord(evCommit, sA));
            }
        }
    }

    // the batch has been enqueued: wait for it, read the control block back, let the policy decide how the next one runs
    int after_batch()
    {
        HIPCHK(hipGetLastError());
        pull_ctl_pinned(h);
        if (la_on) sync_stream(h, sB);
        seq_host = h->hc.window_seq;
        done = h->hc.cursor;
        // Round 0 links the points that decide "create" among themselves (cc_link.h: two more small launches per window)
        // while the batch just read back created a microcluster per 256 points or more - a function of device counters
        // that are identical on every rank
        link_now = h->allow_link && !no_create &&
                   ((long long)(h->hc.m_rows - m_known) * 256 >= std::max<long long>(1, h->hc.cursor - cursor_prev));
        m_known = h->hc.m_rows;
        const bool shard_was = shard_on;
        {
            const double dt = now_ms() - batch_t0;
            const long long pts_b = h->hc.cursor - cursor_prev;
            if (pts_b > 0) win_rate = (double)pts_b / std::max(dt, 1e-3);
            cursor_prev = h->hc.cursor;
            long_seen = h->hc.stat_long > long_prev;
            // (long chains per window and validation round of the batch: up to 64 count as few)
            long_avg = (h->hc.stat_long - long_prev) / std::max<long long>(1, h->hc.stat_rounds - rounds_prev) + 1;
            long_few = long_avg <= 64;
            long_prev = h->hc.stat_long;
            rounds_prev = h->hc.stat_rounds;
        }
        {
            // what the device counted, and the policy's decision for the next batch
            cc_policy_obs o{};
            o.cursor = h->hc.cursor;
            o.m_rows = h->hc.m_rows;
            o.stall_b = h->hc.stall_b;
            o.stat_windows = h->hc.stat_windows;
            o.stat_truncated = h->hc.stat_truncated;
            o.stat_trunc_unknown = h->hc.stat_trunc_unknown;
            o.stat_tiles = h->hc.stat_tiles;
            o.stat_dirty_tiles = h->hc.stat_dirty_tiles;
            o.stat_unsafe = h->hc.stat_unsafe;
            o.stat_missed = h->hc.stat_missed;
            o.tg_ok = (h->hc.tg_ok[0][0] != 0 && h->hc.tg_ok[1][0] != 0) ? 1 : 0;  // (a mean for the pcore kind in both slots)
            for (int r = 0; r < CC_MAX_ROUNDS + 2; ++r) o.round_hist[r] = h->hc.round_hist[r];
            o.prune_rows = h->hc.stat_prune_rows;
            o.prune_full = h->hc.stat_prune_full;
            dec = policy->after_batch(o);
            ptrace->batch(o, dec);
            const cc_policy_carry& k = policy->carry();
            h->adapt_win = k.adapt_win; h->clean_batches = k.clean_batches; h->since_shrink = k.since_shrink;
            if (dec.stalled)
                return fail(h, CC_ERR_INTERNAL, "the online phase made no progress in five consecutive batches of windows");
            if (h->trace && dec.prune_rows > 0)
                fprintf(stderr, "[cc] pruned scans of the batch%s (sample): %lld (wave, row) pairs, %.1f %% evaluated in full; points missed by guessed thresholds so far: %lld\n",
                        h->guess_now ? ", guessed thresholds" : "", (long long)dec.prune_rows,
                        100.0 * (double)dec.prune_full / (double)dec.prune_rows, (long long)h->hc.stat_missed);
            pair_rows_eff += (h->hc.stat_pair_rows - pair_rows_prev) / (shard_was ? (double)world : 1.0);
            if (h->prune_now) pair_rows_pruned += (h->hc.stat_pair_rows - pair_rows_prev) / (shard_was ? (double)world : 1.0);
            pair_rows_prev = h->hc.stat_pair_rows;
            if (shard_was) sharded_windows += dec.wins;
            Rcur = dec.rounds;
            Sd = Sd_full;
            nodirty = dec.nodirty != 0;
            sparse_now = dec.sparse != 0;
            shard_on = dec.shard != 0;
            h->prune_now = dec.prune != 0;
            h->guess_now = dec.prune >= 2;
        h->lean_now = dec.prune == 3;
            if (dec.restart) {
                h->hc.win_cfg = dec.win_cfg;
                h->hc.win_b = (int)std::min<long long>(dec.win_cfg, N - done);
                set_lookahead(dec.lookahead != 0);
                push_ctl_pinned(h);
            }
#ifdef CC_ROUND_DEBUG
            for (int r = 1; r <= CC_MAX_ROUNDS; ++r)
                if (h->hc.dbg_round[r][5] != 0)
                    fprintf(stderr, "[cc]    round %d so far: %llu decisions, %llu refused, create->join new %llu, create->join row %llu, join->create %llu, other MC %llu | windows ended in round %d: %lld\n",
                            r, h->hc.dbg_round[r][5], h->hc.dbg_round[r][0], h->hc.dbg_round[r][1], h->hc.dbg_round[r][2], h->hc.dbg_round[r][3],
                            h->hc.dbg_round[r][4], r, (long long)h->hc.round_hist[r]);
#endif
            if (h->trace)
                fprintf(stderr, "[cc] %.2f ms done %lld rows %d | batch: %lld windows %lld points trunc %lld (%lld at an undecidable point) lookahead %lld dirty tiles %lld / %lld (points so far: %lld unlocated, %lld unsafe) | next window %d rounds %d\n",
                        now_ms() - batch_t0, done, h->hc.m_rows, (long long)dec.wins, (long long)dec.pts, (long long)dec.trunc, (long long)dec.unk, (long long)h->hc.stat_lookahead, (long long)dec.dtiles, (long long)dec.tiles,
                        (long long)h->hc.stat_unprovable, (long long)h->hc.stat_unsafe, dec.want, Rcur);
            batch_windows = dec.batch_windows;
            // windows that keep stopping short on a small table: the sequential kernel takes over (and hands back
            // if it measures slower than this batch did)
            {
                const bool bad = dec.bad != 0;
                bad_batches = bad ? bad_batches + 1 : 0;
                if (seq_mode == 0 && seq_possible() && done < N) {
                    if (seq_probe) {
                        // after a stint: back to the sequential kernel (for twice as long) only if the windows
                        // are still being cut short and were measurably slower
                        if (bad && seq_rate_last > 0.0 && win_rate < seq_rate_last) {
                            seq_on = true;
                            seq_stint_len = std::min<long long>(seq_stint_len * 2, 1 << 20);
                        } else {
                            seq_stint_len = 32768;
                        }
                    } else if (bad_batches >= (seq_r_applies() ? 1 : 2) &&
                               win_rate < (seq_rate_last > 0.0 ? seq_rate_last : seq_rate_guess())) {
                        // (seq_rate_guess(): what the sequential kernel delivers whatever the data, until it has been
                        // measured in this call; the short windows of a stream that is merely starting up run faster than
                        // that.  Where the register kernel applies one such batch is enough: a stint of it costs half of
                        // what k_seq's costs, and the streams it is built for have a few thousand points per call.)
                        seq_on = true;
                    }
                    if (seq_on) seq_stint_left = seq_stint_len;
                }
                seq_probe = false;
                if (seq_mode == 2 && seq_possible() && done < N) seq_on = true;
            }
        }
        return (int)CC_OK;
    }

    // statistics of the call
    void finish()
    {
        HIPCHK(hipEventRecord(ev1, h->stream));
        HIPCHK(hipEventSynchronize(ev1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
        h->stats.run_ms += ms;
        h->stats.points += N - range_a;
        h->stats.windows += h->hc.stat_windows;
        h->stats.rounds += h->hc.stat_rounds;
        h->stats.truncated += h->hc.stat_truncated;
        h->stats.rows = h->hc.m_rows;
        h->stats.scan_pair_dims += pair_rows_eff * (double)h->d;
        h->stats.scan_pair_dims_pruned += pair_rows_pruned * (double)h->d;
        h->stats.sharded_windows += sharded_windows;
        h->stats.seq_points += h->hc.stat_seq_points;
        h->stats.seq_r_points += h->hc.stat_seq_r_points;
#ifdef CC_SEQG_TIMERS
        fprintf(stderr, "[cc] k_seq_g (shader cycles, thread 0): scan %llu minimum %llu add %llu barrier %llu rest %llu chunk %llu | points %llu\n",
                h->hc.dbg_long[0], h->hc.dbg_long[1], h->hc.dbg_long[2], h->hc.dbg_long[3], h->hc.dbg_long[4], h->hc.dbg_long[5], h->hc.dbg_long[7]);
#endif
#ifdef CC_LONG_TIMERS
        fprintf(stderr, "[cc] k_chain_long, workgroup 0 (shader cycles): collect %llu stage %llu chains %llu step-dim %llu step %llu rows %llu state %llu | batches %llu\n",
                h->hc.dbg_long[0], h->hc.dbg_long[1], h->hc.dbg_long[2], h->hc.dbg_long[3], h->hc.dbg_long[4], h->hc.dbg_long[5], h->hc.dbg_long[6], h->hc.dbg_long[7]);
#endif
        h->seq_sticky = seq_on;
        h->stats.table_rows_scanned += h->hc.stat_table_rows;
        h->stats.lookahead_windows += h->hc.stat_lookahead;
        h->stats.pruned_scan_rows += (int64_t)h->hc.stat_prune_rows;
        h->stats.pruned_scan_full_rows += (int64_t)h->hc.stat_prune_full;
        h->stats.long_chains += (int64_t)h->hc.stat_long;
        h->stats.long_chain_launches += long_launches;
        if (h->prep_launched) {
            h->prep_launched = false;
            unsigned long long lp[2] = {0ull, 0ull};
            HIPCHK(hipMemcpy(lp, h->lstat.p + 2 * CC_LSTAT_ROWS, sizeof(lp), hipMemcpyDeviceToHost));
            h->stats.long_prepared += (int64_t)lp[0];
            h->stats.long_replayed += (int64_t)lp[1];
        }
        h->stats.tiles += h->hc.stat_tiles;
        h->stats.dirty_tiles += h->hc.stat_dirty_tiles;
        h->stats.missed_points += h->hc.stat_missed;
        h->probe_now = false;
        if (timing) {
            double tot = 0.0, tot_p = 0.0;
            int64_t n_p = 0;
            for (auto& t : timed) {
                float e = 0.f;
                HIPCHK(hipEventElapsedTime(&e, h->ev_pool[t.first], h->ev_pool[t.first + 1]));
                tot += e;
                if (t.second != 0.0) { tot_p += e; ++n_p; }
            }
            h->stats.scan_launches += (int64_t)timed.size();
            h->stats.scan_ms += tot;
            h->stats.scan_launches_pruned += n_p;
            h->stats.scan_ms_pruned += tot_p;
            double ctot = 0.0;
            for (size_t i : timed_comm) {
                float e = 0.f;
                HIPCHK(hipEventElapsedTime(&e, h->ev_pool[i], h->ev_pool[i + 1]));
                ctot += e;
            }
            h->stats.comm_launches += (int64_t)timed_comm.size();
            h->stats.comm_ms += ctot;
        }
    }

    int run()
    {
        prepare();
        while (done < N) {
            ensure_table(h, (size_t)m_known + std::max<size_t>((size_t)win * batch_max, seq_on ? 8192 : 0) + 1);
            const Table tab = h->tab.view();
            if (seq_on) {
                sequential_stint(tab);
                continue;
            }
            enqueue_batch(tab);
            const int rc = after_batch();
            if (rc != CC_OK) return rc;
        }
        finish();
        return (int)CC_OK;
    }
};

// The exact windowed online phase over the resident points [range_a, range_e), in row order.  no_create: a point that
// no MC absorbs does not create one; it is set aside (label -1, path code 8) and changes nothing - the first half of a
// super-step of the relaxed multi-GPU mode (section 6 of DESIGN.md).  Statistics are added to h->stats.
// resume: the range continues a stream this handle was clustering a moment ago (a later mini-batch of a timepoint): the
// window size carries over as it is instead of restarting small.
int online_range(cc_handle* h, long long range_a, long long range_e, bool no_create, bool resume)
{
    if (range_e <= range_a) return (int)CC_OK;
    if (h->d > CC_WINDOW_MAX_DIM && (h->comm.active() || no_create || !h->allow_seq_g))
        return fail(h, CC_ERR_BAD_ARG, "more than " + std::to_string(CC_WINDOW_MAX_DIM) + " dimensions: the online phase runs on the sequential "
                    "workgroup kernel (k_seq_g) only - not in a multi-GPU group, not with CHRONOCLUST_HIP_SEQG=0");
    if (h->d == 0) return fail(h, CC_ERR_BAD_ARG, "no points uploaded");
    OnlineRun run(h, range_a, range_e, no_create, resume);
    return run.run();
}

}  // namespace

extern "C" {

int cc_online_run(cc_handle* h)
{
    if (!h) return CC_ERR_BAD_ARG;
    if (!h->have_par) return fail(h, CC_ERR_BAD_ARG, "cc_set_params has not been called");
    return guarded(h, [&]() {
        memset(&h->stats, 0, sizeof(h->stats));
        if (h->relaxed_minibatch > 0 && h->comm.active()) return online_relaxed(h);
        return online_range(h, 0, h->n_points, false, false);
    });
}

int cc_labels_download(cc_handle* h, int64_t* out_uid, int8_t* out_path)
{
    if (!h) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        const size_t n = (size_t)h->n_points;
        if (n == 0) return (int)CC_OK;
        static_assert(sizeof(long long) == sizeof(int64_t), "int64");
        if (out_uid) HIPCHK(hipMemcpyAsync(out_uid, h->lab_uid.p, n * 8, hipMemcpyDeviceToHost, h->stream));
        if (out_path) HIPCHK(hipMemcpyAsync(out_path, h->lab_path.p, n, hipMemcpyDeviceToHost, h->stream));
        sync_stream(h, h->stream);
        return (int)CC_OK;
    });
}

int cc_online(cc_handle* h, const double* x, int64_t n, int32_t d, int64_t* out_uid, int8_t* out_path)
{
    int rc = cc_points_upload(h, x, n, d);
    if (rc != CC_OK) return rc;
    rc = cc_online_run(h);
    if (rc != CC_OK) return rc;
    return cc_labels_download(h, out_uid, out_path);
}

int cc_count(cc_handle* h, int kind)
{
    if (!h) return CC_ERR_BAD_ARG;
    int n = 0;
    int rc = guarded(h, [&]() {
        RowList rl = list_order(h);
        n = (int)(kind == CC_PCORE ? rl.pcore.size() : rl.outlier.size());
        return (int)CC_OK;
    });
    return rc == CC_OK ? n : rc;
}

int cc_export(cc_handle* h, int kind, int64_t* id, int64_t* uid, double* w, double* cf1, double* cf2, double* cen,
              double* pref)
{
    if (!h) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        RowList rl = list_order(h);
        const std::vector<int>& rows = kind == CC_PCORE ? rl.pcore : rl.outlier;
        const size_t m = (size_t)h->hc.m_rows, d = (size_t)h->d, n = rows.size();
        if (n == 0) return (int)CC_OK;
        auto fetch_vec = [&](const double* dev, double* out) {
            if (!out) return;
            std::vector<double> tmp(m * d);
            HIPCHK(hipMemcpyAsync(tmp.data(), dev, m * d * 8, hipMemcpyDeviceToHost, h->stream));
            sync_stream(h, h->stream);
            for (size_t i = 0; i < n; ++i) memcpy(out + i * d, tmp.data() + (size_t)rows[i] * d, d * 8);
        };
        fetch_vec(h->tab.cf1.p, cf1);
        fetch_vec(h->tab.cf2.p, cf2);
        fetch_vec(h->tab.cen.p, cen);
        fetch_vec(h->tab.pref.p, pref);
        if (w) {
            std::vector<double> tmp(m);
            HIPCHK(hipMemcpyAsync(tmp.data(), h->tab.w.p, m * 8, hipMemcpyDeviceToHost, h->stream));
            sync_stream(h, h->stream);
            for (size_t i = 0; i < n; ++i) w[i] = tmp[rows[i]];
        }
        auto fetch_i64 = [&](const long long* dev, int64_t* out) {
            if (!out) return;
            std::vector<long long> tmp(m);
            HIPCHK(hipMemcpyAsync(tmp.data(), dev, m * 8, hipMemcpyDeviceToHost, h->stream));
            sync_stream(h, h->stream);
            for (size_t i = 0; i < n; ++i) out[i] = tmp[rows[i]];
        };
        fetch_i64(h->tab.id.p, id);
        fetch_i64(h->tab.uid.p, uid);
        return (int)CC_OK;
    });
}

int cc_inject_mc(cc_handle* h, int kind, int32_t d, const double* cf1, const double* cf2, const double* cen,
                 const double* pref, double w, int64_t id, int64_t uid)
{
    if (!h || !cf1 || !cf2 || !cen || !pref) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        int rc = set_dim(h, d);
        if (rc != CC_OK) return rc;
        ensure_table(h, (size_t)h->hc.m_rows + 1);
        const size_t r = (size_t)h->hc.m_rows, dd = (size_t)d;
        for (int i = 0; i < d; ++i)
            if (pref[i] != 1.0 && !(h->have_par && pref[i] == h->par.k)) h->tainted = true;
        HIPCHK(hipMemcpyAsync(h->tab.cf1.p + r * dd, cf1, dd * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.cf2.p + r * dd, cf2, dd * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.cen.p + r * dd, cen, dd * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.pref.p + r * dd, pref, dd * 8, hipMemcpyHostToDevice, h->stream));
        const int knd = kind == CC_PCORE ? CC_KIND_PCORE : CC_KIND_OUTLIER;
        const int key = kind == CC_PCORE ? h->hc.n_pkeys++ : h->hc.n_okeys++;
        const long long lid = id, luid = uid;
        HIPCHK(hipMemcpyAsync(h->tab.w.p + r, &w, 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.kind.p + r, &knd, 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.key.p + r, &key, 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.id.p + r, &lid, 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.uid.p + r, &luid, 8, hipMemcpyHostToDevice, h->stream));
        sync_stream(h, h->stream);
        h->hc.m_rows += 1;
        if (kind == CC_PCORE && id >= h->hc.pcore_last_id) h->hc.pcore_last_id = id + 1;
        if (uid >= h->hc.outlier_last_id) h->hc.outlier_last_id = uid + 1;
        refresh_ctl_params(h);
        push_ctl(h);
        sync_stream(h, h->stream);
        return (int)CC_OK;
    });
}

int cc_inject_bulk(cc_handle* h, int kind, int32_t d, int32_t n, const double* cf1, const double* cf2,
                   const double* cen, const double* pref, const double* w, const int64_t* id, const int64_t* uid)
{
    if (!h || n < 0) return CC_ERR_BAD_ARG;
    if (n == 0) return CC_OK;
    if (!cf1 || !cf2 || !cen || !pref || !w || !id || !uid) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        int rc = set_dim(h, d);
        if (rc != CC_OK) return rc;
        ensure_table(h, (size_t)h->hc.m_rows + (size_t)n);
        const size_t r = (size_t)h->hc.m_rows, dd = (size_t)d, nn = (size_t)n;
        for (size_t i = 0; i < nn * dd; ++i)
            if (pref[i] != 1.0 && !(h->have_par && pref[i] == h->par.k)) { h->tainted = true; break; }
        const int knd = kind == CC_PCORE ? CC_KIND_PCORE : CC_KIND_OUTLIER;
        int& nkeys = kind == CC_PCORE ? h->hc.n_pkeys : h->hc.n_okeys;
        std::vector<int> kinds(nn, knd), keys(nn);
        for (size_t i = 0; i < nn; ++i) keys[i] = nkeys + (int)i;
        static_assert(sizeof(long long) == sizeof(int64_t), "int64");
        HIPCHK(hipMemcpyAsync(h->tab.cf1.p + r * dd, cf1, nn * dd * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.cf2.p + r * dd, cf2, nn * dd * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.cen.p + r * dd, cen, nn * dd * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.pref.p + r * dd, pref, nn * dd * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.w.p + r, w, nn * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.kind.p + r, kinds.data(), nn * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.key.p + r, keys.data(), nn * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.id.p + r, id, nn * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->tab.uid.p + r, uid, nn * 8, hipMemcpyHostToDevice, h->stream));
        sync_stream(h, h->stream);
        nkeys += n;
        h->hc.m_rows += n;
        for (size_t i = 0; i < nn; ++i) {
            if (kind == CC_PCORE && id[i] >= h->hc.pcore_last_id) h->hc.pcore_last_id = id[i] + 1;
            if (uid[i] >= h->hc.outlier_last_id) h->hc.outlier_last_id = uid[i] + 1;
        }
        refresh_ctl_params(h);
        push_ctl(h);
        sync_stream(h, h->stream);
        return (int)CC_OK;
    });
}

int cc_decay_downgrade(cc_handle* h, double factor)
{
    if (!h) return CC_ERR_BAD_ARG;
    if (!h->have_par) return fail(h, CC_ERR_BAD_ARG, "cc_set_params has not been called");
    return guarded(h, [&]() {
        const int m = h->hc.m_rows, d = h->d;
        if (m == 0) return (int)CC_OK;
        refresh_ctl_params(h);
        const Table tab = h->tab.view();
        hipLaunchKernelGGL(k_decay, dim3((m * d + 255) / 256), dim3(256), 0, h->stream, tab, m, d, factor);
        h->flags.ensure((size_t)m);
        hipLaunchKernelGGL(k_downgrade_flags, dim3((m + 255) / 256), dim3(256), 0, h->stream, tab, m, d,
                           h->hc.beta_mu, h->hc.pi, h->hc.omicron, h->flags.p);
        std::vector<int> flags(m);
        std::vector<long long> id(m), uid(m);
        HIPCHK(hipMemcpyAsync(flags.data(), h->flags.p, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(id.data(), tab.id, (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(uid.data(), tab.uid, (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
        RowList rl = list_order(h);  // synchronises the stream

        // hddstream.py:528-537 and :545-549: Python removes from the list it iterates, so the element that
        // slides into the freed position is skipped.  Integer work over the flags only.
        std::vector<int> pl = rl.pcore, ol = rl.outlier;
        std::vector<char> downgraded(m, 0);
        for (size_t i = 0; i < pl.size(); ++i) {
            const int r = pl[i];
            if (flags[r] & 1) {
                downgraded[r] = 1;
                pl.erase(pl.begin() + (long)i);
                ol.push_back(r);
            }
        }
        for (size_t i = 0; i < ol.size(); ++i) {
            if (flags[ol[i]] & 2) ol.erase(ol.begin() + (long)i);
        }
        const int np = (int)pl.size(), no = (int)ol.size(), n = np + no;
        std::vector<int> perm(n), nkind(n), nkey(n);
        std::vector<long long> nid(n);
        for (int i = 0; i < np; ++i) { perm[i] = pl[i]; nkind[i] = CC_KIND_PCORE; nkey[i] = i; nid[i] = id[pl[i]]; }
        for (int i = 0; i < no; ++i) {
            const int r = ol[i];
            perm[np + i] = r; nkind[np + i] = CC_KIND_OUTLIER; nkey[np + i] = i;
            nid[np + i] = downgraded[r] ? uid[r] : id[r];  // hddstream.py:535
        }
        if (h->tab2.cap < h->tab.cap || h->tab2.d != d) {
            h->tab2.alloc(h->tab.cap, d);
            // stamps are compared with atomic max: fresh memory must not hold anything that looks newer
            HIPCHK(hipMemsetAsync(h->tab2.touch.p, 0, 2 * h->tab2.cap * 8, h->stream));
            HIPCHK(hipMemsetAsync(h->tab2.last.p, 0, 2 * h->tab2.cap * 8, h->stream));
            HIPCHK(hipMemsetAsync(h->tab2.carry_of.p, 0, h->tab2.cap * 8, h->stream));
            HIPCHK(hipMemsetAsync(h->tab2.cnt.p, 0, h->tab2.cap * 8, h->stream));
        }
        DevBuf<int> dperm, dkind, dkey;
        DevBuf<long long> dnid;
        dperm.ensure(n); dkind.ensure(n); dkey.ensure(n); dnid.ensure(n);
        if (n) {
            HIPCHK(hipMemcpyAsync(dperm.p, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(dkind.p, nkind.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(dkey.p, nkey.data(), (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(dnid.p, nid.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_gather_rows, dim3((n * d + 255) / 256), dim3(256), 0, h->stream, tab, h->tab2.view(),
                               dperm.p, dkind.p, dkey.p, dnid.p, n, d);
        }
        sync_stream(h, h->stream);
        h->tab.swap(h->tab2);
        h->hc.m_rows = n;
        h->hc.n_pkeys = np;
        h->hc.n_okeys = no;
        push_ctl(h);
        sync_stream(h, h->stream);
        return (int)CC_OK;
    });
}

int cc_offline(cc_handle* h, int32_t* n_clusters, int8_t* out_core, int32_t* out_pdim, int32_t* out_nn,
               int32_t* out_nw)
{
    if (!h) return CC_ERR_BAD_ARG;
    if (!h->have_par) return fail(h, CC_ERR_BAD_ARG, "cc_set_params has not been called");
    return guarded(h, [&]() {
        refresh_ctl_params(h);
        h->clusters.clear();
        h->pcore_ids_host.clear();
        h->pcore_uid_host.clear();
        h->n_core = 0;
        if (n_clusters) *n_clusters = 0;
        // (CHRONOCLUST_HIP_TRACE=1: host wall time per phase of the call)
        auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double tph[8] = {now_us(), 0, 0, 0, 0, 0, 0, 0};
        // the pcore rows in list order (ascending key), from page-locked copies of the kind / key columns; creation numbers
        // of all rows beside them (cc_point_clusters joins the per-point labels to the clusters through the pcores')
        const int m_all = h->hc.m_rows;
        h->pin.reset();
        h->pin.reserve((size_t)m_all * 72 + ((size_t)1 << 16));  // (everything below but the neighbour lists: 49 B per row)
        std::vector<int> prow_host;
        const long long* uid_all = nullptr;
        if (m_all > 0) {
            int* kind = h->pin.take<int>((size_t)m_all);
            int* key = h->pin.take<int>((size_t)m_all);
            long long* uid = h->pin.take<long long>((size_t)m_all);
            HIPCHK(hipMemcpyAsync(kind, h->tab.kind.p, (size_t)m_all * 4, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(key, h->tab.key.p, (size_t)m_all * 4, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(uid, h->tab.uid.p, (size_t)m_all * 8, hipMemcpyDeviceToHost, h->stream));
            sync_stream(h, h->stream);
            uid_all = uid;
            std::vector<unsigned long long> order;  // (key, row) packed: one plain sort, no indirection
            order.reserve((size_t)m_all);
            for (int r = 0; r < m_all; ++r)
                if (kind[r] == CC_KIND_PCORE) order.push_back(((unsigned long long)((unsigned)key[r] ^ 0x80000000u) << 32) | (unsigned)r);  // (signed order)
            std::sort(order.begin(), order.end());
            prow_host.resize(order.size());
            for (size_t i = 0; i < order.size(); ++i) prow_host[i] = (int)(order[i] & 0xFFFFFFFFull);
        }
        tph[1] = now_us();
        const int mp = (int)prow_host.size(), d = h->d;
        if (mp == 0) return (int)CC_OK;
        const size_t md = (size_t)mp * d;
        const int words = (mp + 63) / 64;
        // Multi-GPU: a rank evaluates a block of p rows (whole 64-row blocks) of the M x M pair matrices and the ranks
        // all-gather what the ordered expansion needs of them: subspace preference vectors, neighbour counts, the
        // weighted-reachability bitmask.  The eps-neighbour bitmask stays local (a row is only read by its owner).
        const int world = h->comm.world, rank = h->comm.rank;
        const bool shard = h->comm.active() && mp >= h->offline_shard_min_rows;
        const int share = shard ? cc_shard_share(mp, world, 64) : words * 64;  // p rows per rank
        const size_t rows_pad = shard ? (size_t)share * world : (size_t)mp;     // buffers hold every rank's block
        int p_lo = 0, p_hi = mp;
        if (shard) cc_shard_range(mp, world, rank, 64, &p_lo, &p_hi);
        h->pv_cf1.ensure(md); h->pv_cf2.ensure(md); h->pv_cen.ensure(md); h->pv_pref.ensure(md); h->pv_w.ensure(mp);
        h->pv_id.ensure(mp); h->prow.ensure(mp); h->wvec.ensure(rows_pad * d); h->nn.ensure(rows_pad); h->pdim.ensure(mp);
        h->core.ensure(mp); h->adj.ensure(rows_pad * words); h->adjw.ensure(rows_pad * words);
        int* const prow_pin = h->pin.take<int>((size_t)mp);
        memcpy(prow_pin, prow_host.data(), (size_t)mp * 4);
        HIPCHK(hipMemcpyAsync(h->prow.p, prow_pin, (size_t)mp * 4, hipMemcpyHostToDevice, h->stream));
        PcoreView pv{h->pv_cf1.p, h->pv_cf2.p, h->pv_cen.p, h->pv_pref.p, h->pv_w.p, h->pv_id.p};
        const Ctl& c = h->hc;
        const cc_params& p = h->par;
        hipLaunchKernelGGL(k_gather_pcores, dim3((unsigned)((md + 255) / 256)), dim3(256), 0, h->stream, h->tab.view(),
                           pv, h->prow.p, mp, d);
        hipLaunchKernelGGL(k_core_flags, dim3((mp + 255) / 256), dim3(256), 0, h->stream, pv, mp, d, p.eps_sq, p.mu,
                           p.pi, p.k, c.inv_k, c.pow2, h->core.p);
        const int my_rows = p_hi - p_lo;
        if (my_rows > 0) {
            {
                // p rows per workgroup: CC_EPS_PCH on large tables; a table of a few thousand rows would be a few hundred
                // workgroups of one wave per SIMD each (5 000 rows: 400 workgroups, 141 us for 46 us of arithmetic) - whole
                // staging passes (CC_EPS_TP rows), at least ~8 workgroups per CU
                int pch = CC_EPS_PCH;
                while (pch > CC_EPS_TP && (long long)((words + 3) / 4) * ((my_rows + pch - 1) / pch) < 8ll * h->n_cus) pch /= 2;
                const dim3 grid((words + 3) / 4, (my_rows + pch - 1) / pch), block(256);
#define CC_EPS(DP) hipLaunchKernelGGL((k_eps_neighbours<DP>), grid, block, 0, h->stream, pv.cen, mp, d, p.ups_eps, h->adj.p, words, p_lo, p_hi, pch)
                if (d <= 4) CC_EPS(4);
                else if (d <= 8) CC_EPS(8);
                else if (d <= 16) CC_EPS(16);
                else if (d <= 20) CC_EPS(20);
                else if (d <= 24) CC_EPS(24);
                else if (d <= 40) CC_EPS(40);
                else if (d <= 64) CC_EPS(64);
                else CC_EPS(128);
#undef CC_EPS
            }
            hipLaunchKernelGGL(k_subspace_pref, dim3((unsigned)(((size_t)my_rows * d + 255) / 256)), dim3(256), 0, h->stream,
                               pv.cen, h->adj.p, words, mp, d, p.delta, p.k, h->wvec.p, h->nn.p, p_lo, p_hi);
        }
        if (shard) {
            // in place: rank r's block sits at r * share rows of the same buffer on every rank
            h->comm.all_gather(h->wvec.p + (size_t)rank * share * d, h->wvec.p, (size_t)share * d * 8, h->stream);
            h->comm.all_gather(h->nn.p + (size_t)rank * share, h->nn.p, (size_t)share * 4, h->stream);
        }
        hipLaunchKernelGGL(k_pdim, dim3((mp + 255) / 256), dim3(256), 0, h->stream, h->wvec.p, mp, d, h->pdim.p);
        if (my_rows > 0)
            hipLaunchKernelGGL(k_weighted_reach, dim3(my_rows), dim3(64), 0, h->stream, pv.cen, h->wvec.p, h->adj.p,
                               h->adjw.p, words, mp, d, p.ups_eps_sq, p_lo, p_hi);
        if (shard)
            h->comm.all_gather(h->adjw.p + (size_t)rank * share * words, h->adjw.p, (size_t)share * words * 8, h->stream);
        // the reachability rows as neighbour lists: counts -> offsets (host prefix sums) -> ascending positions
        h->nw_cnt.ensure(mp);
        hipLaunchKernelGGL(k_adj_counts, dim3(mp), dim3(64), 0, h->stream, h->adjw.p, words, mp, h->nw_cnt.p);
        h->pcore_uid_host.resize(mp);
        for (int i = 0; i < mp; ++i) h->pcore_uid_host[(size_t)i] = uid_all[(size_t)prow_host[(size_t)i]];
        int8_t* const core = h->pin.take<int8_t>((size_t)mp);
        int* const pdim = h->pin.take<int>((size_t)mp);
        int* const nn = h->pin.take<int>((size_t)mp);
        int* const nw_cnt = h->pin.take<int>((size_t)mp);
        long long* const ids_pin = h->pin.take<long long>((size_t)mp);
        long long* const nw_off = h->pin.take<long long>((size_t)mp + 1);
        HIPCHK(hipMemcpyAsync(core, h->core.p, mp, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(pdim, h->pdim.p, (size_t)mp * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(nn, h->nn.p, (size_t)mp * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(nw_cnt, h->nw_cnt.p, (size_t)mp * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(ids_pin, h->pv_id.p, (size_t)mp * 8, hipMemcpyDeviceToHost, h->stream));
        sync_stream(h, h->stream);
        HIPCHK(hipGetLastError());
        h->pcore_ids_host.assign(ids_pin, ids_pin + mp);
        tph[2] = now_us();
        nw_off[0] = 0;
        for (int i = 0; i < mp; ++i) nw_off[(size_t)i + 1] = nw_off[i] + nw_cnt[i];
        const long long n_edges = nw_off[mp];
        // (the lists go into a block of their own: the first one must stay where it is)
        std::vector<int> nbr_pageable;
        const int* nbr = nullptr;
        if (n_edges > 0) {
            h->nw_off.ensure((size_t)mp + 1);
            h->nw_nbr.ensure((size_t)n_edges);
            HIPCHK(hipMemcpyAsync(h->nw_off.p, nw_off, ((size_t)mp + 1) * 8, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_adj_fill, dim3(mp), dim3(64), 0, h->stream, h->adjw.p, words, mp, h->nw_off.p, h->nw_nbr.p);
            int* dst;
            if (h->pin.used + (size_t)n_edges * 4 + 128 <= h->pin.cap) dst = h->pin.take<int>((size_t)n_edges);
            else {  // (dense neighbourhoods: more edges than the scratch was sized for)
                nbr_pageable.resize((size_t)n_edges);
                dst = nbr_pageable.data();
            }
            HIPCHK(hipMemcpyAsync(dst, h->nw_nbr.p, (size_t)n_edges * 4, hipMemcpyDeviceToHost, h->stream));
            sync_stream(h, h->stream);
            HIPCHK(hipGetLastError());
            nbr = dst;
        }

        tph[3] = now_us();
        // ---- ordered expansion on the host: predecon.py:62-120, 242-267 (integer / graph work only) ----
        auto for_each_nw = [&](int q, auto&& fn) {  // the weighted neighbours of q in ascending (= dict) order
            for (long long e = nw_off[q]; e < nw_off[(size_t)q + 1]; ++e) fn(nbr[(size_t)e]);
        };
        std::vector<int8_t> cls(mp, 0);  // 0 'u', 1 'c', 2 'n'
        std::vector<int> queue;
        h->clusters.mem.reserve((size_t)mp);
        h->clusters.off.reserve((size_t)mp + 1);
        const int lam = p.pi;
        for (int seed = 0; seed < mp; ++seed) {
            if (cls[seed] != 0) continue;
            if (!core[seed]) { cls[seed] = 2; continue; }
            const size_t cl_begin = h->clusters.mem.size();
            queue.clear();
            for_each_nw(seed, [&](int x) { queue.push_back(x); });
            size_t head = 0;
            while (head < queue.size()) {
                const int q = queue[head++];
                if (!core[q]) continue;
                for_each_nw(q, [&](int x) {
                    if (pdim[x] > lam) return;
                    if (cls[x] == 0) queue.push_back(x);
                    if (cls[x] == 0 || cls[x] == 2) {
                        cls[x] = 1;
                        h->clusters.mem.push_back(x);
                    }
                });
            }
            if (h->clusters.mem.size() > cl_begin) h->clusters.off.push_back((int)h->clusters.mem.size());  // predecon.py:83 (W > 0)
        }
        for (int i = 0; i < mp; ++i) h->n_core += core[i];

        tph[4] = now_us();
        // ---- cluster CF sums in merge order + preferred dimensions on the device ----
        const int nc = (int)h->clusters.size();
        if (nc) {
            const std::vector<int>&mem = h->clusters.mem, &off = h->clusters.off;
            const size_t cd = (size_t)nc * d;
            h->mem_dev.ensure(mem.size()); h->off_dev.ensure(off.size());
            h->c_cf1.ensure(cd); h->c_cf2.ensure(cd); h->c_cen.ensure(cd); h->c_pref.ensure(cd); h->c_w.ensure(nc);
            HIPCHK(hipMemcpyAsync(h->mem_dev.p, mem.data(), mem.size() * 4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->off_dev.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_cluster_merge, dim3((unsigned)((cd + 255) / 256)), dim3(256), 0, h->stream, pv,
                               h->mem_dev.p, h->off_dev.p, nc, d, p.delta_sq, p.k, h->c_cf1.p, h->c_cf2.p, h->c_cen.p,
                               h->c_pref.p, h->c_w.p);
            sync_stream(h, h->stream);
        }
        tph[5] = now_us();
        if (h->trace)
            fprintf(stderr, "[cc] offline phase, host wall time: list order %.0f us, pair kernels + read-back %.0f, neighbour lists %.0f, "
                    "ordered expansion %.0f, cluster sums %.0f (%d pcores, %d clusters)\n", tph[1] - tph[0], tph[2] - tph[1], tph[3] - tph[2],
                    tph[4] - tph[3], tph[5] - tph[4], mp, nc);
        if (out_core) memcpy(out_core, core, mp);
        if (out_pdim) memcpy(out_pdim, pdim, (size_t)mp * 4);
        if (out_nn) memcpy(out_nn, nn, (size_t)mp * 4);
        if (out_nw) memcpy(out_nw, nw_cnt, (size_t)mp * 4);
        if (n_clusters) *n_clusters = nc;
        return (int)CC_OK;
    });
}

int cc_num_core(cc_handle* h) { return h ? h->n_core : CC_ERR_BAD_ARG; }

int cc_cluster_size(cc_handle* h, int32_t c)
{
    if (!h || c < 0 || c >= (int)h->clusters.size()) return CC_ERR_BAD_ARG;
    return h->clusters.off[(size_t)c + 1] - h->clusters.off[(size_t)c];
}

int cc_cluster_export(cc_handle* h, int32_t c, int64_t* members, double* w, double* cf1, double* cf2, double* cen,
                      double* pref)
{
    if (!h || c < 0 || c >= (int)h->clusters.size()) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        const size_t d = (size_t)h->d;
        const int a = h->clusters.off[(size_t)c], e = h->clusters.off[(size_t)c + 1];
        if (members)
            for (int i = a; i < e; ++i) members[i - a] = h->pcore_ids_host[(size_t)h->clusters.mem[(size_t)i]];
        if (w) HIPCHK(hipMemcpyAsync(w, h->c_w.p + c, 8, hipMemcpyDeviceToHost, h->stream));
        if (cf1) HIPCHK(hipMemcpyAsync(cf1, h->c_cf1.p + (size_t)c * d, d * 8, hipMemcpyDeviceToHost, h->stream));
        if (cf2) HIPCHK(hipMemcpyAsync(cf2, h->c_cf2.p + (size_t)c * d, d * 8, hipMemcpyDeviceToHost, h->stream));
        if (cen) HIPCHK(hipMemcpyAsync(cen, h->c_cen.p + (size_t)c * d, d * 8, hipMemcpyDeviceToHost, h->stream));
        if (pref) HIPCHK(hipMemcpyAsync(pref, h->c_pref.p + (size_t)c * d, d * 8, hipMemcpyDeviceToHost, h->stream));
        sync_stream(h, h->stream);
        return (int)CC_OK;
    });
}

int cc_clusters_total_members(cc_handle* h)
{
    if (!h) return CC_ERR_BAD_ARG;
    size_t tot = 0;
    tot = h->clusters.mem.size();
    return (int)tot;
}

int cc_clusters_export(cc_handle* h, int64_t* members, int32_t* offsets, double* w, double* cf1, double* cf2,
                       double* cen, double* pref)
{
    if (!h) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        const size_t nc = h->clusters.size(), d = (size_t)h->d;
        const size_t tot = h->clusters.mem.size();
        if (offsets)
            for (size_t c = 0; c <= nc; ++c) offsets[c] = (int32_t)h->clusters.off[c];
        if (members)
            for (size_t i = 0; i < tot; ++i) members[i] = h->pcore_ids_host[(size_t)h->clusters.mem[i]];
        if (nc == 0) return (int)CC_OK;
        if (w) HIPCHK(hipMemcpyAsync(w, h->c_w.p, nc * 8, hipMemcpyDeviceToHost, h->stream));
        if (cf1) HIPCHK(hipMemcpyAsync(cf1, h->c_cf1.p, nc * d * 8, hipMemcpyDeviceToHost, h->stream));
        if (cf2) HIPCHK(hipMemcpyAsync(cf2, h->c_cf2.p, nc * d * 8, hipMemcpyDeviceToHost, h->stream));
        if (cen) HIPCHK(hipMemcpyAsync(cen, h->c_cen.p, nc * d * 8, hipMemcpyDeviceToHost, h->stream));
        if (pref) HIPCHK(hipMemcpyAsync(pref, h->c_pref.p, nc * d * 8, hipMemcpyDeviceToHost, h->stream));
        sync_stream(h, h->stream);
        return (int)CC_OK;
    });
}

int cc_assoc_argmin(cc_handle* h, const double* cur_cen, const double* cur_pref, int32_t mc, const double* prev_cen,
                    int32_t mp, int32_t d, int32_t* out_idx, double* out_dist)
{
    if (!h || !cur_cen || !cur_pref || !out_idx || mc < 0 || mp < 0 || d <= 0) return CC_ERR_BAD_ARG;
    if (mp > 0 && !prev_cen) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        if (mc == 0) return (int)CC_OK;
        const size_t cd = (size_t)mc * d, pd = (size_t)mp * d;
        // multi-GPU: a rank takes a block of current pcores; indices and distances are all-gathered
        const int world = h->comm.world, rank = h->comm.rank;
        const bool shard = h->comm.active() && mc >= h->offline_shard_min_rows;
        const int share = shard ? cc_shard_share(mc, world, 1) : mc;
        int c_lo = 0, c_hi = mc;
        if (shard) cc_shard_range(mc, world, rank, 1, &c_lo, &c_hi);
        h->a_cur_cen.ensure(cd); h->a_cur_pref.ensure(cd); h->a_prev_cen.ensure(pd);
        h->a_idx.ensure(shard ? (size_t)share * world : (size_t)mc);
        h->a_dist.ensure(shard ? (size_t)share * world : (size_t)mc);
        // the distance operand per (current pcore, dim): 1 or 1/k when every preference entry is 1 or k and k is a power
        // of two (x / k == x * (1/k) bit for bit), else the preference entry itself (the kernel divides)
        const double k = h->have_par ? h->par.k : 1.0;
        bool unit = is_pow2(k);
        for (size_t i = 0; unit && i < cd; ++i) unit = cur_pref[i] == 1.0 || cur_pref[i] == k;
        std::vector<double> op(cd);
        const double inv_k = unit ? 1.0 / k : 0.0;
        for (size_t i = 0; i < cd; ++i) op[i] = unit ? (cur_pref[i] == 1.0 ? 1.0 : inv_k) : cur_pref[i];
        HIPCHK(hipMemcpyAsync(h->a_cur_cen.p, cur_cen, cd * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->a_cur_pref.p, op.data(), cd * 8, hipMemcpyHostToDevice, h->stream));
        if (pd) HIPCHK(hipMemcpyAsync(h->a_prev_cen.p, prev_cen, pd * 8, hipMemcpyHostToDevice, h->stream));
        if (c_hi > c_lo && mp > 0) {
            const int ctiles = (c_hi - c_lo + 255) / 256;  // workgroups of 4 x 64 current pcores
            // previous pcores in S sub-ranges so that the launch fills the machine (>= ~1024 workgroups)
            const int S = std::max(1, std::min((mp + CC_ASSOC_TQ - 1) / CC_ASSOC_TQ, (1024 + ctiles - 1) / ctiles));
            h->a_pdist.ensure((size_t)S * mc);
            h->a_pidx.ensure((size_t)S * mc);
            const dim3 grid(ctiles, S), block(256);
#define CC_ASSOC(DP)                                                                                                        \
    do {                                                                                                                    \
        if (unit) hipLaunchKernelGGL((k_assoc_tiled<DP, true>), grid, block, 0, h->stream, h->a_cur_cen.p, h->a_cur_pref.p,   \
                                     h->a_prev_cen.p, mc, mp, d, c_lo, c_hi, h->a_pdist.p, h->a_pidx.p);                      \
        else hipLaunchKernelGGL((k_assoc_tiled<DP, false>), grid, block, 0, h->stream, h->a_cur_cen.p, h->a_cur_pref.p,       \
                                h->a_prev_cen.p, mc, mp, d, c_lo, c_hi, h->a_pdist.p, h->a_pidx.p);                           \
    } while (0)
            if (d <= 4) CC_ASSOC(4);
            else if (d <= 8) CC_ASSOC(8);
            else if (d <= 16) CC_ASSOC(16);
            else if (d <= 24) CC_ASSOC(24);
            else if (d <= 40) CC_ASSOC(40);
            else if (d <= 64) CC_ASSOC(64);
            else CC_ASSOC(128);
#undef CC_ASSOC
            hipLaunchKernelGGL(k_assoc_merge, dim3((c_hi - c_lo + 255) / 256), dim3(256), 0, h->stream, h->a_pdist.p,
                               h->a_pidx.p, S, mc, c_lo, c_hi, h->a_idx.p, h->a_dist.p);
        } else if (c_hi > c_lo) {
            // no previous pcores: index -1, distance +inf (what the argmin kernel starts from)
            HIPCHK(hipMemsetAsync(h->a_idx.p + c_lo, 0xFF, (size_t)(c_hi - c_lo) * 4, h->stream));
            const std::vector<double> inf((size_t)(c_hi - c_lo), std::numeric_limits<double>::infinity());
            HIPCHK(hipMemcpyAsync(h->a_dist.p + c_lo, inf.data(), inf.size() * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));  // (`inf` is a local)
        }
        if (shard) {
            h->comm.all_gather(h->a_idx.p + (size_t)rank * share, h->a_idx.p, (size_t)share * 4, h->stream);
            h->comm.all_gather(h->a_dist.p + (size_t)rank * share, h->a_dist.p, (size_t)share * 8, h->stream);
        }
        HIPCHK(hipMemcpyAsync(out_idx, h->a_idx.p, (size_t)mc * 4, hipMemcpyDeviceToHost, h->stream));
        if (out_dist) HIPCHK(hipMemcpyAsync(out_dist, h->a_dist.p, (size_t)mc * 8, hipMemcpyDeviceToHost, h->stream));
        sync_stream(h, h->stream);
        HIPCHK(hipGetLastError());
        return (int)CC_OK;
    });
}

// ---- exact multi-GPU path: communicator set-up -------------------------------------------------------

int cc_comm_unique_id(void* out_id)
{
    if (!out_id) return CC_ERR_BAD_ARG;
    cc::RcclApi& api = cc::RcclApi::get();
    if (!api.ok()) return CC_ERR_COMM;
    ncclUniqueId id;
    if (api.GetUniqueId(&id) != ncclSuccess) return CC_ERR_COMM;
    static_assert(sizeof(id) == CC_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(out_id, &id, sizeof(id));
    return CC_OK;
}

int cc_comm_init_rccl(cc_handle* h, const void* id_bytes, int rank, int world)
{
    if (!h || !id_bytes || world < 1 || rank < 0 || rank >= world) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        if (h->comm.active()) return fail(h, CC_ERR_BAD_ARG, "the handle already belongs to a group");
        cc::RcclApi& api = cc::RcclApi::get();
        if (!api.ok()) return fail(h, CC_ERR_COMM, std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "missing symbol"));
        ncclUniqueId id;
        memcpy(&id, id_bytes, sizeof(id));
        ncclComm_t comm = nullptr;
        ncclResult_t r = api.CommInitRank(&comm, world, id, rank);  // (the handle's device is current)
        if (r != ncclSuccess) return fail(h, CC_ERR_COMM, std::string("ncclCommInitRank: ") + api.GetErrorString(r));
        h->comm.nccl[0] = comm;
        h->comm.rank = rank;
        h->comm.world = world;
        h->comm.broken = false;
        const char* to = getenv("CHRONOCLUST_HIP_COMM_TIMEOUT_S");
        if (to && atof(to) > 0.0) h->comm.timeout_s = atof(to);
        // ONE communicator serves both streams by default: RCCL then orders the lookahead scans' all-gathers (second
        // stream) with those of the validation stream, which costs some overlap but is the mode every RCCL user runs.
        // CHRONOCLUST_HIP_TWO_COMMS=1: a second communicator for the lookahead stream (its id is made by rank 0 and
        // travels through the first one), so that the two streams' collectives are independent - concurrent
        // communicators need both collective kernels co-resident on every rank and have never run on more than one
        // GPU in a build session: opt-in until a multi-GPU run has confirmed them
        const char* two = getenv("CHRONOCLUST_HIP_TWO_COMMS");
        if (two && two[0] == '1') {
            DevBuf<char> ids;
            ids.ensure((size_t)world * sizeof(ncclUniqueId) + sizeof(ncclUniqueId));
            ncclUniqueId id2;
            memset(&id2, 0, sizeof id2);
            if (rank == 0) {
                r = api.GetUniqueId(&id2);
                if (r != ncclSuccess) return fail(h, CC_ERR_COMM, std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
            }
            char* send = ids.p + (size_t)world * sizeof(ncclUniqueId);
            HIPCHK(hipMemcpyAsync(send, &id2, sizeof id2, hipMemcpyHostToDevice, h->stream));
            h->comm.all_gather(send, ids.p, sizeof id2, h->stream, 0);
            HIPCHK(hipMemcpyAsync(&id2, ids.p, sizeof id2, hipMemcpyDeviceToHost, h->stream));  // rank 0's block
            sync_stream(h, h->stream);
            ncclComm_t comm2 = nullptr;
            r = api.CommInitRank(&comm2, world, id2, rank);
            if (r != ncclSuccess) return fail(h, CC_ERR_COMM, std::string("ncclCommInitRank (second communicator): ") + api.GetErrorString(r));
            h->comm.nccl[1] = comm2;
        }
        // the split thresholds from a measurement of this group's own exchange (collective: every rank is here)
        const char* cal = getenv("CHRONOCLUST_HIP_CALIBRATE");
        if (!(cal && cal[0] == '0')) {
            const int rc = cc_comm_calibrate(h);
            if (rc != CC_OK) return rc;
        }
        return (int)CC_OK;
    });
}

int cc_comm_calibrate(cc_handle* h)
{
    if (!h) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        if (!h->comm.active()) return fail(h, CC_ERR_BAD_ARG, "cc_comm_calibrate: the handle belongs to no group");
        const int world = h->comm.world, rank = h->comm.rank;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        auto timed3 = [&](auto&& fn) {  // one untimed pass, then the minimum of three
            fn();
            sync_stream(h, h->stream);
            float best = 1e30f;
            for (int i = 0; i < 3; ++i) {
                HIPCHK(hipEventRecord(e0, h->stream));
                fn();
                HIPCHK(hipEventRecord(e1, h->stream));
                sync_stream(h, h->stream);
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            return (double)best * 1e3;  // us
        };
        // (1) the exchange of a split window: one full window's records from every rank
        const int win = std::min(h->tun.window, CC_MAX_WINDOW);
        const size_t rec = ((size_t)win * 4 + 4) * sizeof(Cand);
        DevBuf<char> sbuf, rbuf;
        sbuf.ensure(rec);
        rbuf.ensure(rec * (size_t)world);
        HIPCHK(hipMemsetAsync(sbuf.p, 0, rec, h->stream));
        const double ag_us = timed3([&]() { h->comm.all_gather(sbuf.p, rbuf.p, rec, h->stream, 0); });
        // (2) what a table row costs: the plain snapshot scan of a full window over 4 096 synthetic rows x 20 dimensions, on
        // scratch buffers and a control block of its own (the handle's state is not touched)
        constexpr int DPc = 20, Rc = 4096;
        constexpr int NWc = ScanShape<DPc, false>::NW;
        DevBuf<Ctl> cctl;
        DevBuf<double> cxt, ccen, cscl;
        DevBuf<int> ckind, ckey;
        DevBuf<Cand> cpart;
        const int tiles = (win + 63) / 64;
        const int Sc = std::max(1, std::min(16, (h->n_cus * scan_u_wgs_per_cu<DPc>()) / std::max(1, tiles)));
        cctl.ensure(1); cxt.ensure((size_t)win * DPc); ccen.ensure((size_t)Rc * DPc); cscl.ensure((size_t)Rc * DPc);
        ckind.ensure(Rc); ckey.ensure(Rc); cpart.ensure((size_t)2 * win * Sc * 4);
        {
            std::vector<double> x((size_t)win * DPc), cen((size_t)Rc * DPc), scl((size_t)Rc * DPc, 0.25);
            unsigned long long st = 0x9E3779B97F4A7C15ull;
            auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) * 0x1p-53; };
            for (auto& v : x) v = 0.1 + 0.8 * rnd();
            for (auto& v : cen) v = 0.1 + 0.8 * rnd();
            std::vector<int> kind(Rc, CC_KIND_PCORE), key(Rc);
            for (int i = 0; i < Rc; ++i) key[i] = i;
            Ctl c;
            memset(&c, 0, sizeof c);
            c.d = DPc; c.m_rows = Rc; c.win_b = win; c.win_cfg = win; c.n_points = win; c.xt_stride = win;
            c.k = 4.0; c.inv_k = 0.25; c.pow2 = 1;
            HIPCHK(hipMemcpyAsync(cctl.p, &c, sizeof c, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(cxt.p, x.data(), x.size() * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(ccen.p, cen.data(), cen.size() * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(cscl.p, scl.data(), scl.size() * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(ckind.p, kind.data(), (size_t)Rc * 4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(ckey.p, key.data(), (size_t)Rc * 4, hipMemcpyHostToDevice, h->stream));
            sync_stream(h, h->stream);  // (the host vectors go out of scope)
        }
        const double scan_us = timed3([&]() {
            hipLaunchKernelGGL((k_scan_u<DPc, NWc>), dim3(tiles, Sc), dim3(64 * NWc), 0, h->stream, (const Ctl*)cctl.p, (const double*)cxt.p,
                               (const double*)ccen.p, (const double*)cscl.p, (const int*)ckind.p, (const int*)ckey.p, cpart.p, 0, 0,
                               (size_t)win * Sc * 4, 0, 1);
        });
        HIPCHK(hipGetLastError());
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        // (3) every rank takes the group's maxima: the thresholds decide the sequence of collectives and must be the same
        // everywhere (an all-gather of two doubles per rank through the transport itself)
        double mine[2] = {ag_us, scan_us * 1e3 / ((double)Rc * DPc)};  // us, ns per (row, dim)
        DevBuf<double> dsend, drecv;
        dsend.ensure(2);
        drecv.ensure((size_t)2 * world);
        HIPCHK(hipMemcpyAsync(dsend.p, mine, sizeof mine, hipMemcpyHostToDevice, h->stream));
        h->comm.all_gather(dsend.p, drecv.p, sizeof mine, h->stream, 0);
        std::vector<double> all((size_t)2 * world);
        HIPCHK(hipMemcpyAsync(all.data(), drecv.p, all.size() * 8, hipMemcpyDeviceToHost, h->stream));
        sync_stream(h, h->stream);
        double ag = 0.0, sc = 0.0;
        for (int r = 0; r < world; ++r) { ag = std::max(ag, all[(size_t)2 * r]); sc = std::max(sc, all[(size_t)2 * r + 1]); }
        h->calib_ag_us = ag;
        h->calib_scan_ns = sc;
        if (world > 1 && sc > 0.0) {
            // time saved by the split = scan x (1 - 1 / world); it pays from scan >= exchange x world / (world - 1) on
            const double row_dims = ag * 1e3 * (double)world / (double)(world - 1) / sc;
            h->shard_min_row_dims = (long long)std::min(row_dims, 1e15);
            h->shard_min_row_dims_pruned = (long long)std::min(row_dims * 3.3, 1e15);
        }
        if (h->trace)
            fprintf(stderr, "[cc] rank %d of %d: all-gather of a %d-point window's records %.1f us, plain scan %.3f ns per (row, dim) "
                    "(group maxima) -> scans split from %lld (plain) / %lld (pruned) row-dims on\n", rank, world, win, ag, sc,
                    (long long)h->shard_min_row_dims, (long long)(h->shard_min_row_dims_pruned > 0 ? h->shard_min_row_dims_pruned : h->shard_min_row_dims));
        return (int)CC_OK;
    });
}

int cc_comm_init_local(cc_handle** handles, int world)
{
    if (!handles || world < 1) return CC_ERR_BAD_ARG;
    for (int r = 0; r < world; ++r)
        if (!handles[r] || handles[r]->comm.active()) return CC_ERR_BAD_ARG;
    auto grp = std::make_shared<cc::LocalGroup>(world);
    // every member's events first: a failure leaves no handle half inside a group
    for (int r = 0; r < world; ++r) {
        cc_handle* h = handles[r];
        int rc = guarded(h, [&]() {
            HIPCHK(hipEventCreateWithFlags(&h->comm.ev_ready, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->comm.ev_done, hipEventDisableTiming));
            return (int)CC_OK;
        });
        if (rc != CC_OK) {
            for (int q = 0; q <= r; ++q) handles[q]->comm.destroy();
            return rc;
        }
    }
    for (int r = 0; r < world; ++r) {
        handles[r]->comm.local = grp;
        handles[r]->comm.rank = r;
        handles[r]->comm.world = world;
    }
    return CC_OK;
}

int cc_comm_destroy(cc_handle* h)
{
    if (!h) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        try {
            sync_stream(h, h->stream);
            sync_stream(h, h->stream2);
        } catch (const cc::CommErr&) {  // (the group is already lost: nothing left to drain)
        }
        h->comm.destroy();
        return (int)CC_OK;
    });
}

int cc_comm_info(cc_handle* h, int32_t* rank, int32_t* world, int32_t* transport)
{
    if (!h) return CC_ERR_BAD_ARG;
    if (rank) *rank = h->comm.rank;
    if (world) *world = h->comm.world;
    if (transport) *transport = h->comm.rccl() ? 1 : (h->comm.local ? 2 : 0);
    return CC_OK;
}

int cc_comm_set_relaxed(cc_handle* h, int32_t minibatch_points)
{
    if (!h || minibatch_points < 0) return CC_ERR_BAD_ARG;
    if (minibatch_points > 0 && !h->comm.active()) return fail(h, CC_ERR_BAD_ARG, "the relaxed mode needs a group (cc_comm_init_*)");
    h->relaxed_minibatch = minibatch_points;
    return CC_OK;
}

int cc_get_relaxed_stats(cc_handle* h, cc_relaxed_stats* out)
{
    if (!h || !out) return CC_ERR_BAD_ARG;
    *out = h->rstats;
    return CC_OK;
}

int cc_set_shard_thresholds(cc_handle* h, int64_t min_row_dims, int32_t offline_min_rows)
{
    if (!h) return CC_ERR_BAD_ARG;
    if (min_row_dims >= 0) { h->shard_min_row_dims = min_row_dims; h->shard_min_row_dims_pruned = 0; }  // (one threshold for both kinds of scan)
    if (offline_min_rows >= 0) h->offline_shard_min_rows = offline_min_rows;
    return CC_OK;
}

// ---- per-point output: cluster index of every point (device), text of the per-point file (host) ----------

int cc_point_clusters(cc_handle* h, int32_t* out_idx)
{
    if (!h || !out_idx) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        const long long n = h->n_points;
        if (n == 0) return (int)CC_OK;
        // creation number -> cluster index, for the pcores the last cc_offline put into clusters (everything else,
        // outlier microclusters included: -1), built from the merge lists and uploaded as one dense table
        const long long n_uid = h->hc.outlier_last_id;
        std::vector<int32_t> map((size_t)std::max<long long>(n_uid, 1), -1);
        for (size_t c = 0; c < h->clusters.size(); ++c)
            for (int i = h->clusters.off[c]; i < h->clusters.off[c + 1]; ++i) {
                const long long u = h->pcore_uid_host[(size_t)h->clusters.mem[(size_t)i]];
                if (u >= 0 && u < n_uid) map[(size_t)u] = (int32_t)c;
            }
        h->pc_map.ensure(map.size());
        h->pc_out.ensure((size_t)n);
        HIPCHK(hipMemcpyAsync(h->pc_map.p, map.data(), map.size() * 4, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_point_clusters, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->lab_uid.p, n,
                           h->pc_map.p, n_uid, h->pc_out.p);
        HIPCHK(hipMemcpyAsync(out_idx, h->pc_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
        sync_stream(h, h->stream);  // (`map` is a local)
        HIPCHK(hipGetLastError());
        return (int)CC_OK;
    });
}

int cc_sync(cc_handle* h)
{
    if (!h) return CC_ERR_BAD_ARG;
    return guarded(h, [&]() {
        sync_stream(h, h->stream);
        sync_stream(h, h->stream2);
        HIPCHK(hipGetLastError());
        return (int)CC_OK;
    });
}

#include "cc_host_abi.inc"

int cc_get_stats(cc_handle* h, cc_stats* out)
{
    if (!h || !out) return CC_ERR_BAD_ARG;
    *out = h->stats;
    out->window = h->tun.window;
    out->calib_allgather_us = h->calib_ag_us;
    out->calib_scan_ns_per_row_dim = h->calib_scan_ns;
    out->split_threshold_row_dims = h->shard_min_row_dims;
    out->split_threshold_row_dims_pruned = h->shard_min_row_dims_pruned > 0 ? h->shard_min_row_dims_pruned : h->shard_min_row_dims;
    return CC_OK;
}

}  // extern "C"
