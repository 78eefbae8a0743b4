// cc_online.h — gfx950 kernels of the exact windowed online phase.
//
// Reference semantics (clustering/hddstream.py:220-237) are a strict per-point
// read-modify-write chain over the microcluster (MC) table.  The kernels below
// keep those semantics exactly while processing a window of B points at a time:
//
//   k_scan_u             every window point against every MC of the window-start
//   (k_scan<DIRTY=false>) snapshot: per point the two best pcore and the two best
//                        outlier candidates by (projected distance, list order).
//                        k_scan_u reads the MC rows as scalar operands (scalar loads,
//                        nothing through LDS) and serves k = 2^e, pi >= d and the
//                        compiled widths of d; k_scan stages rows in LDS and serves
//                        the rest (pdim filter, other k, padded d)
//   k_decide (round 0)   the decision each point would take if no earlier window
//                        point existed (exact for the first point of the window)
//   k_chain              replays the claimed decisions per MC in arrival order ->
//                        one "version row" per point = its MC's state right after it
//   k_scan<DIRTY=true>   every point against the version rows that are live when it
//                        arrives (earlier points' effects)
//   k_decide (round r)   re-derives every decision from snapshot candidates + live
//                        versions; the first index whose decision differs from the
//                        claim is the validation frontier; repeat until a fixed point
//   k_commit             writes the validated prefix back (ids, list-order keys and
//                        labels by prefix sums in point order) and opens the next window
//
// Everything before the frontier is exactly what the sequential loop would have
// produced; a window always commits at least its first point (the one exception:
// while the host does not launch the dirty scans - k_dseed ruled them out for every
// point of the last batch - k_decide refuses points that would need them, and a
// window whose first point is refused commits nothing and idles the batch).
//
// Lookahead: the snapshot scan of the next window may run on a second stream while
// this window is validated, against the table one commit earlier.  What it cannot
// see - the rows this window's commit changes - is kept as the "carry set" (struct
// Carry) and enters the next window's validation like version rows that precede its
// first point (k_dseed, k_scan<DIRTY=true, mode 1>, k_decide, k_chain, k_commit_b).
// Lookahead scans read one of two scan copies of the table (struct ScanCopy), kept
// up to date by cc_apply_carry and k_commit_b, so that a commit never waits for them.
//
// Beside the window pipeline:
//   k_chain_long      long chains on small tables (few MCs absorb every point): sequential CF additions per
//                     dimension, radius tests of a batch of steps in parallel, resume after the first rejected one
//   k_seq             no speculation: one wavefront walks the points in order on an LDS image of the table; the host
//                     switches to it while windows keep being cut short and it measures faster
//   k_merge_partials  exact multi-GPU path: a rank's partials per point -> the 64-byte record the ranks all-gather
//                     (k_scan then scans only the rank's share of the table rows, k_decide merges the records)
//   k_rel_*           relaxed multi-GPU mode (events sharded over the ranks): CF deltas, merge, promotions,
//                     set-aside points
//
// Arithmetic: IEEE double, no contraction by the compiler (-ffp-contract=off), sums
// over dimensions left to right as in utilities/mc_functions.py under numba; the one
// hand-written fusion (distance terms of the scans when k is a power of two) is
// exact, see CC_TINY.
#pragma once
#include <type_traits>

#include "cc_common.h"

#define CC_INF (__builtin_huge_val())
#define CC_GROUP_THREADS 256  // workgroup size of the one-32-lane-group-per-point kernels (k_decide, k_chain)

// Each wave stages its own LDS tile and is the only reader of it: DS operations of one wave execute in order,
// so a wavefront-scope fence (no workgroup barrier) is enough between filling a tile and reading it.
// The validation kernels are short dependent chains; beside a lookahead scan (four arithmetic-bound waves per SIMD)
// they would get every fifth issue slot.  They raise their wave priority so that the SIMD arbiter takes them first.
#define CC_LATENCY_KERNEL() __builtin_amdgcn_s_setprio(3)

#define CC_WAVE_SYNC()                                          \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// ---------------------------------------------------------------------------------
// per-MC arithmetic (utilities/mc_functions.py)
// ---------------------------------------------------------------------------------

// mc_functions.py:14-22
__device__ __forceinline__ double cc_sqvar(double cf1, double cf2, double w)
{
    double a = cf2 / w;
    double b = cf1 / w;
    b = b * b;
    return a - b;
}

// The parameters the per-MC arithmetic needs, read from the control block once per kernel into registers
// (reading them through the Ctl pointer inside loops would re-load them after every store).
struct Par {
    double delta_sq, k, inv_k, eps_sq, beta_mu;
    int pow2, pi, filter, d;
};

__device__ __forceinline__ Par cc_load_par(const Ctl* ctl)
{
    Par p;
    p.delta_sq = ctl->delta_sq; p.k = ctl->k; p.inv_k = ctl->inv_k; p.eps_sq = ctl->eps_sq; p.beta_mu = ctl->beta_mu;
    p.pow2 = ctl->pow2; p.pi = ctl->pi; p.filter = ctl->filter; p.d = ctl->d;
    return p;
}

// x / pref with pref in {1.0, k}; when k is a power of two x * (1/k) is the same double
__device__ __forceinline__ double cc_div_pref(double x, double pref, const Par& c)
{
    if (pref == 1.0) return x;
    return (c.pow2 && pref == c.k) ? x * c.inv_k : x / pref;
}

// microcluster.py:213-233 + mc_functions.py:45-56: projected radius^2 of (base + point) with the
// preferred dimensions of the enlarged MC.  base_cf1 == nullptr means an empty MC.
// Also returns count(pref' > 1) and count(pref' != 1) of the enlarged MC.
__device__ inline double cc_tentative_radius(const double* bcf1, const double* bcf2, double bw, const double* p,
                                             int d, const Par& c, int* cnt_gt1, int* cnt_ne1)
{
    const double w1 = bw + 1.0;
    double r2 = 0.0;
    int g = 0, n = 0;
    for (int i = 0; i < d; ++i) {
        double x = p[i];
        double c1 = (bcf1 ? bcf1[i] : 0.0) + x;
        double c2 = (bcf2 ? bcf2[i] : 0.0) + x * x;
        double var = cc_sqvar(c1, c2, w1);
        double pr = (var <= c.delta_sq) ? c.k : 1.0;  // microcluster.py:109-114 (NaN -> 1.0)
        g += (pr > 1.0);
        n += (pr != 1.0);
        r2 = r2 + cc_div_pref(var, pr, c);
    }
    if (cnt_gt1) *cnt_gt1 = g;
    if (cnt_ne1) *cnt_ne1 = n;
    return r2;
}

// insert x into the ordered pair (a, b); field-wise selects keep the structs in registers.  Bounds (CC_SLOT_BOUND,
// key -1) take part like candidates: the pair then reads "best, then either the exact second-best or a lower bound for
// everything else", whatever the order in which partial pairs are merged.
__device__ __forceinline__ void cc_top2_push(Cand& a, Cand& b, const Cand& x)
{
    const bool ok = x.slot != -1;
    const bool beats_a = ok && (a.slot == -1 || cand_less(x.dist, x.key, a.dist, a.key));
    const bool beats_b = ok && !beats_a && (b.slot == -1 || cand_less(x.dist, x.key, b.dist, b.key));
    b.dist = beats_a ? a.dist : (beats_b ? x.dist : b.dist);
    b.key = beats_a ? a.key : (beats_b ? x.key : b.key);
    b.slot = beats_a ? a.slot : (beats_b ? x.slot : b.slot);
    a.dist = beats_a ? x.dist : a.dist;
    a.key = beats_a ? x.key : a.key;
    a.slot = beats_a ? x.slot : a.slot;
}

// v_min_f64 / v_max_f64 as plain selections (the operands are never NaN here; the library fmin / fmax would
// add a canonicalising instruction per operand)
__device__ __forceinline__ double cc_vmin(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double cc_vmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int N, int I = 0, typename F>
__device__ __forceinline__ void cc_static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cc_static_for<N, I + 1>(f);
    }
}

// Fused last step of a distance term (mc_functions.py:39 + :41) when the divisor is a power of two.
// The reference computes acc + round(x2 * sc); x2 * sc is exact for sc = 2^e unless it lands in the subnormal
// range, so one rounding of x2 * sc + acc (v_fma_f64) is the same double.  The subnormal case is excluded by the
// caller: when every nonzero coordinate of the point and of the centroid has magnitude >= 2^-400, a nonzero
// difference is >= 2^-452 (both are multiples of 2^-452), its square >= 2^-905, and with 2^-64 <= sc <= 2^64 the
// product stays normal.  Tiles or waves that hold a smaller nonzero coordinate take the unfused path.
#define CC_TINY 0x1p-400
__device__ __forceinline__ bool cc_is_tiny(double v) { return v != 0.0 && __builtin_fabs(v) < CC_TINY; }

// Displacement maxima per 16 rows and kind (Versions::tile_dsq, Carry::tile_dsq): 0 = the tile holds no row of the kind,
// else the bits of the largest squared displacement (>= 0: the bits order like the values; +inf: no bound) plus one.
__device__ __forceinline__ unsigned long long cc_dsq_code(double dq) { return (unsigned long long)__double_as_longlong(dq) + 1ull; }
// can a tile / a window whose maximum is `code` be ruled out for a threshold tau?  (no row of the kind: yes)
__device__ __forceinline__ bool cc_dsq_below(unsigned long long code, double tau)
{
    if (code == 0ull) return true;
    const double dq = __longlong_as_double((long long)(code - 1ull));
    return dq < CC_INF && sqrt(dq) * (1.0 + 1e-9) < tau;
}

// wave-uniform operand of a dimension from bit BIT of its row mask: two scalar instructions (the compiler's own
// selection takes three, and the scalar unit issues one instruction per wave turn like the vector unit)
template <int BIT>
__device__ __forceinline__ double cc_sel_scale(unsigned mask, double scaled, double one)
{
    double r;
    asm("s_bitcmp1_b32 %1, %2\n\ts_cselect_b64 %0, %3, %4" : "=s"(r) : "s"(mask), "n"(BIT), "s"(scaled), "s"(one) : "scc");
    return r;
}

// ---------------------------------------------------------------------------------
// k_scan: points (one or PT per lane, in registers) x MC rows (wave-uniform, staged in LDS)
// ---------------------------------------------------------------------------------

#define CC_SCAN_TM 16     // MC rows per LDS tile
// window points per lane of the clean scan at DP == 20 / workgroups per CU it is compiled for (build-time knobs)
#ifndef CC_SCAN_PT_CLEAN
#define CC_SCAN_PT_CLEAN 1
#endif
#ifndef CC_SCAN_WGS_CLEAN
#define CC_SCAN_WGS_CLEAN 4
#endif
#ifndef CC_SCAN_WGS_CLEAN40
#define CC_SCAN_WGS_CLEAN40 3
#endif
#ifndef CC_SCAN_WGS_DIRTY32
#define CC_SCAN_WGS_DIRTY32 3
#endif
#ifndef CC_SCAN_NW_CLEAN
#define CC_SCAN_NW_CLEAN 4
#endif
template <int DP, bool DIRTY>
struct ScanShape {
    static constexpr int PT = (!DIRTY && DP == 20) ? CC_SCAN_PT_CLEAN : 1;
    // waves per workgroup: same points, disjoint MC sub-ranges, merged through LDS (8 halve the partials but measured
    // 5 % slower on C2)
    static constexpr int NW = (!DIRTY && DP == 20) ? CC_SCAN_NW_CLEAN : 4;
    static constexpr int WGS = (!DIRTY && DP == 20) ? CC_SCAN_WGS_CLEAN
                               : (DP <= 20 ? 4 : (DP <= 40 ? (DIRTY ? CC_SCAN_WGS_DIRTY32 : (DP == 40 ? CC_SCAN_WGS_CLEAN40 : 3)) : 2));
};
// waves per workgroup (template parameter NW): same points, disjoint MC sub-ranges, merged through LDS


// One workgroup = NW waves that hold the same 64*PT points in registers.  The MC rows of the launch
// are split into gridDim.y * NW sub-ranges; each wave streams its sub-range through its own LDS tile
// (centroid and 1/pref are wave-uniform broadcast reads), keeps the two best candidates per kind and point, and
// the workgroup writes ONE partial per point (merged in LDS), so the argmin partials in HBM stay small.
template <int DP, bool FILTER, bool POW2, bool DIRTY, int NW>
__global__ __launch_bounds__(64 * NW, (ScanShape<DP, DIRTY>::WGS)) void k_scan(const Ctl* __restrict__ ctl,
                                                             const double* __restrict__ X,
                                                             const double* __restrict__ Xt, Rows rows,
                                                             const Cand* __restrict__ clean,
                                                             Cand* __restrict__ part, int round, int mode,
                                                             size_t part_stride, int shard_rank, int shard_world)
{
    constexpr int PT = ScanShape<DP, DIRTY>::PT;  // window points per lane
    if (DIRTY) CC_LATENCY_KERNEL();
    // Which window, which rows:
    //   clean, mode 0: the current window against the table as it is (only if the window has no lookahead scan)
    //   clean, mode 1: lookahead - the window after the current one (parity `round` of its window_seq), while the
    //                  current one is being validated; its parameters sit in their own slot of the control block
    //   dirty, mode 0: the current window against its own version rows
    //   dirty, mode 1: the current window against the carry set of the previous window (lookahead windows only)
    int B, m_rows_scan;
    long long cursor;
    if (!DIRTY && mode == 1) {
        const int q = round & 1;
        B = ctl->la_b[q];
        m_rows_scan = ctl->la_rows[q];
        cursor = ctl->la_cursor[q];
        part += (size_t)q * part_stride;
    } else {
        B = ctl->win_b;
        m_rows_scan = ctl->m_rows;
        cursor = ctl->cursor;
        if (!DIRTY) {
            if (ctl->mode != 0) return;  // this window's snapshot scan ran ahead
            part += (size_t)(ctl->window_seq & 1ull) * part_stride;
        }
    }
    if (B == 0) return;
    if (DIRTY && ctl->fc[round - 1] >= B) return;  // already at a fixed point
    const bool carried = DIRTY && mode == 1;
    const int car_n = carried ? ((ctl->mode != 0) ? ctl->car_n : 0) : 0;
    if (carried && car_n == 0) return;
    const int bx = (int)blockIdx.x;
    {
    const int j0 = bx * (64 * PT);
    if (j0 >= B) return;
    if (DIRTY && rows.skip[bx] != 0) return;  // k_dseed: no row can matter to this tile; k_decide takes the seeds
    const int d = ctl->d;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, in an SGPR
    const int S = gridDim.y;  // partials per point
    const int nsub = S * NW;
    const int sub = blockIdx.y * NW + wv;
    // a version row i only matters to points j > i: the dirty scan of this tile covers rows [0, j0 + 64*PT - 1)
    // (a carried row matters to every point up to the first one that targets its MC)
    // Exact multi-GPU path: the table is replicated, rank r of `shard_world` scans the rows
    // [r * ceil(M / world), (r + 1) * ceil(M / world)) of the snapshot and the ranks exchange their per-point
    // candidates afterwards (k_merge_partials + all-gather); shard_world == 1: the whole table.
    int row_lo = 0, row_hi = DIRTY ? (carried ? car_n : min(B, j0 + 64 * PT - 1)) : m_rows_scan;
    if (!DIRTY && shard_world > 1) cc_shard_range(m_rows_scan, shard_world, shard_rank, 1, &row_lo, &row_hi);
    const int nrows = row_hi - row_lo;
    // dirty scan: sub-ranges are whole 16-row tiles so that the per-tile displacement maxima line up
    const int per = DIRTY ? (((nrows + nsub - 1) / nsub + CC_SCAN_TM - 1) / CC_SCAN_TM) * CC_SCAN_TM
                          : (nrows + nsub - 1) / nsub;
    const int r0 = row_lo + sub * per;
    const int r1 = min(row_hi, r0 + per);
    const int ntiles = (per + CC_SCAN_TM - 1) / CC_SCAN_TM;  // the same for every wave of the workgroup
    const size_t n_pts = (size_t)ctl->xt_stride;
    const Par par = cc_load_par(ctl);
    const double inv_k = par.inv_k;
    const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
    // FILTER = false: the host knows that the pdim filter of hddstream.py:317-321 is vacuous (pi >= d)
    const bool filter = FILTER && par.filter != 0;

    // LDS: per-wave tiles while scanning, then (same bytes) the candidate exchange of the final merge
    constexpr int TILE_DOUBLES = NW * CC_SCAN_TM * DP;
    constexpr int TILE_BYTES = TILE_DOUBLES * 16 + NW * CC_SCAN_TM * 12;
    constexpr int MERGE_BYTES = (NW - 1) * PT * 4 * 64 * (int)sizeof(Cand);
    __shared__ __attribute__((aligned(16))) unsigned char smem[TILE_BYTES > MERGE_BYTES ? TILE_BYTES : MERGE_BYTES];
    double* const s_c_base = reinterpret_cast<double*>(smem) + (size_t)wv * CC_SCAN_TM * DP;
    double* const s_s_base = reinterpret_cast<double*>(smem) + TILE_DOUBLES + (size_t)wv * CC_SCAN_TM * DP;
    int* const s_int = reinterpret_cast<int*>(smem + (size_t)TILE_DOUBLES * 16) + wv * CC_SCAN_TM * 3;
    int* const s_kind_w = s_int;
    int* const s_key_w = s_int + CC_SCAN_TM;
    int* const s_next_w = s_int + 2 * CC_SCAN_TM;

    double p[PT][DP];
    int jj[PT];
    bool valid[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        jj[t] = j0 + t * 64 + lane;
        valid[t] = jj[t] < B;
    }
    // dirty scan: no version whose displacement is below wave_tau can matter to any point of this wave; when that
    // rules out every tile of the sub-range the wave only hands its seeds on and never loads its points
    // (per kind: a version competes in the list of its kind, against that list's threshold)
    double wave_tau[2] = {-CC_INF, -CC_INF};
    bool any_tile = true;
    if (DIRTY) {
#pragma unroll
        for (int K = 0; K < 2; ++K) {
            double wt = CC_INF;
#pragma unroll
            for (int t = 0; t < PT; ++t) {
                const double tj = valid[t] ? rows.tau[(size_t)jj[t] * 2 + K] : CC_INF;
                wt = tj < wt ? tj : wt;
            }
            for (int off = 32; off >= 1; off >>= 1) {
                const double o = __shfl_xor(wt, off);
                wt = o < wt ? o : wt;
            }
            wave_tau[K] = wt;
        }
        any_tile = false;
        for (int rt = r0; rt < r1; rt += CC_SCAN_TM)
            if (!(cc_dsq_below(rows.tile_dsq[(size_t)(rt >> 4) * 2 + 0], wave_tau[0]) &&
                  cc_dsq_below(rows.tile_dsq[(size_t)(rt >> 4) * 2 + 1], wave_tau[1])))
                any_tile = true;
    }
#pragma unroll
    for (int t = 0; t < PT; ++t) {
#pragma unroll
        for (int i = 0; i < DP; ++i) p[t][i] = 0.0;
        if (!any_tile) continue;
        // Xt is the dimension-major copy of the points: consecutive lanes read consecutive doubles
        const double* xp = Xt + cursor + (valid[t] ? jj[t] : 0);
#pragma unroll
        for (int i = 0; i < DP; ++i) p[t][i] = (valid[t] && i < d) ? xp[(size_t)i * n_pts] : 0.0;
    }
    // fused distance terms (see cc_fma_term) are exact for this wave's points?
    bool fuse_wave = POW2 && par.k >= 0x1p-64 && par.k <= 0x1p64;
    if (POW2) {
        bool tn = false;
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int i = 0; i < DP; ++i) tn = tn || cc_is_tiny(p[t][i]);
        fuse_wave = fuse_wave && __builtin_amdgcn_ballot_w64(tn) == 0ull;
    }

    // running best-two per kind and point: [kind][pt][rank]
    double bd[2][PT][2];
    int bk[2][PT][2], bs[2][PT][2];
    double cap[2][PT];
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
#pragma unroll
        for (int t = 0; t < PT; ++t) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                bd[kd][t][r] = (DIRTY || valid[t]) ? CC_INF : -CC_INF;
                bk[kd][t][r] = CC_IDX_INF;
                bs[kd][t][r] = -1;
            }
            cap[kd][t] = CC_INF;
        }
    if (DIRTY) {
        // caps and first candidates prepared once per point by k_dseed (`clean` is the seed table here)
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            if (!valid[t]) continue;
#pragma unroll
            for (int kd = 0; kd < 2; ++kd) {
                const Cand sd = clean[(size_t)jj[t] * 4 + kd * 2];
                cap[kd][t] = clean[(size_t)jj[t] * 4 + kd * 2 + 1].dist;
                bd[kd][t][0] = sd.dist; bk[kd][t][0] = sd.key; bs[kd][t][0] = sd.slot;
            }
        }
    }

    for (int tt = 0; tt < ntiles; ++tt) {
        const int rt = r0 + tt * CC_SCAN_TM;
        const int tm = __builtin_amdgcn_readfirstlane(max(0, min(CC_SCAN_TM, r1 - rt)));
        if (tm == 0) break;
        if (DIRTY) {
            if (cc_dsq_below(rows.tile_dsq[(size_t)(rt >> 4) * 2 + 0], wave_tau[0]) &&
                cc_dsq_below(rows.tile_dsq[(size_t)(rt >> 4) * 2 + 1], wave_tau[1]))
                continue;  // nothing in this tile can matter
        }
        CC_WAVE_SYNC();
        bool fuse_tile = false;
        int rm_lo = 0, rm_hi = 0;  // lane m < CC_SCAN_TM: which dimensions of tile row m are scaled by 1/k
        {
            // the tile is one contiguous block of tm * d doubles per column: a straight copy, all loads of the
            // tile in flight before the first LDS store (LDS row stride = d; dimensions d..DP-1 are never read)
            constexpr int NL = (CC_SCAN_TM * DP + 63) / 64;
            const double* gc = rows.cen + (size_t)rt * d;
            const double* gs = rows.scl + (size_t)rt * d;
            double tc[NL], ts[NL];
            // LDS rows are DP doubles long (compile-time stride); when d < DP the padding holds (0, 1): zero terms
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const int e = lane + q * 64;  // index into the padded tile
                const int m = e / DP, i = e - m * DP;
                const bool in = (m < tm) && (i < d);
                const int ge = m * d + i;     // index into the contiguous global block (== e when d == DP)
                tc[q] = in ? gc[ge] : 0.0;
                ts[q] = in ? gs[ge] : 1.0;
            }
            bool tn = false;
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const int e = lane + q * 64;
                if (e < CC_SCAN_TM * DP) { s_c_base[e] = tc[q]; s_s_base[e] = ts[q]; }
                if (POW2) tn = tn || cc_is_tiny(tc[q]);
            }
            if (POW2) fuse_tile = fuse_wave && __builtin_amdgcn_ballot_w64(tn) == 0ull;
            if (POW2 && !DIRTY) {
                // When k is a power of two the distance operand of a dimension is 1 or 1/k: one bit.  The 64 * NL
                // staged operands give NL ballot words = the tile's CC_SCAN_TM * DP bits in row order; lane m keeps
                // the DP bits of row m, and the fused row loop builds its operands from them with scalar selects
                // instead of reading them from LDS (the loop is bound by wave-uniform LDS reads otherwise).
                unsigned long long rmask = 0ull;
                const int off = (lane & (CC_SCAN_TM - 1)) * DP;
#pragma unroll
                for (int q = 0; q < NL; ++q) {
                    const unsigned long long w = __builtin_amdgcn_ballot_w64(ts[q] != 1.0);
                    const int rel = off - 64 * q;
                    const unsigned long long a = (rel >= 0 && rel < 64) ? (w >> (rel & 63)) : 0ull;
                    const unsigned long long b = (rel < 0 && rel > -DP) ? (w << ((-rel) & 63)) : 0ull;
                    rmask |= a | b;
                }
                if (DP < 64) rmask &= (1ull << (DP & 63)) - 1ull;
                rm_lo = (int)(unsigned)(rmask & 0xFFFFFFFFull);
                rm_hi = (int)(unsigned)(rmask >> 32);
            }
        }
        // kinds of the tile's rows as two wave-uniform bit masks (clean scan) / LDS columns (dirty scan)
        unsigned pmask = 0, omask = 0;
        if (!DIRTY) {
            const int kd = (lane < tm) ? rows.kind[rt + lane] : CC_KIND_DEAD;
            pmask = (unsigned)__builtin_amdgcn_ballot_w64(kd == CC_KIND_PCORE);
            omask = (unsigned)__builtin_amdgcn_ballot_w64(kd == CC_KIND_OUTLIER);
        } else if (lane < tm) {
            s_kind_w[lane] = rows.kind[rt + lane];
            s_key_w[lane] = rows.key[rt + lane];
            if (!carried) s_next_w[lane] = rows.next[rt + lane];
            else if (rows.kind[rt + lane] == CC_KIND_DEAD) s_next_w[lane] = -1;  // not a carried row (no slot either)
            else {
                // a carried row is live up to and including the first point of this window that targets its MC
                const unsigned long long tc = rows.touch[(size_t)(round & 1) * rows.cap + (size_t)rows.slot[rt + lane]];
                s_next_w[lane] = ((tc >> 20) == stamp) ? (0xFFFFF - (int)(tc & 0xFFFFFull)) : CC_IDX_INF;
            }
        }
        // dirty scan: the rows of the tile that can matter to some point of the wave (the per-tile test above, per row;
        // while MCs are being created or promoted almost every tile holds a row without a bound, but few rows do)
        unsigned rowmask = 0xFFFFu;
        if (DIRTY) {
            const double rq = (lane < tm) ? rows.dsq[rt + lane] : 0.0;
            const int rk = (lane < tm) ? rows.kind[rt + lane] : CC_KIND_DEAD;
            const double wt = (rk == CC_KIND_PCORE) ? wave_tau[0] : wave_tau[1];
            rowmask = (unsigned)__builtin_amdgcn_ballot_w64(lane < tm && rk != CC_KIND_DEAD && !(rq < CC_INF && sqrt(rq) * (1.0 + 1e-9) < wt));
        }
        CC_WAVE_SYNC();

        if (!DIRTY) {
            // Clean scan: a lean row loop.  Rows run to the last dimension (the second-best bound is never tight
            // enough to drop a row early on 64 unrelated points: measured), so there are no exit checks.  The
            // running best-two hold (distance, row) only; while no distance of the wave equals a held one the
            // update is pure selection (min / max and three selects).  Exact ties - the only place where the
            // list-order keys decide (hddstream.py:326/373: strict `<`, first in list order wins) - and the pdim
            // filter take the general path, which fetches the keys it needs.
            // KSEL: 0 / 1 = every row of the tile is a pcore / outlier MC (no kind test per row, the running pair of that
            // kind stays in its registers), -1 = mixed tile
            auto clean_rows = [&](auto FUSEC, auto KSELC) {
            constexpr bool FUSE = decltype(FUSEC)::value;
            constexpr int KSEL = decltype(KSELC)::value;
            for (int m = 0; m < tm; ++m) {
                double acc[PT];
                // (rows of an even DP start on 16-byte boundaries: ds_read_b128)
                const double* rc = (DP % 2 == 0) ? (const double*)__builtin_assume_aligned(s_c_base + m * DP, 16) : s_c_base + m * DP;
                const double* rs = (DP % 2 == 0) ? (const double*)__builtin_assume_aligned(s_s_base + m * DP, 16) : s_s_base + m * DP;
                const unsigned mlo = FUSE ? (unsigned)__builtin_amdgcn_readlane(rm_lo, m) : 0u;
                const unsigned mhi = (FUSE && DP > 32) ? (unsigned)__builtin_amdgcn_readlane(rm_hi, m) : 0u;
                const double one = 1.0;
                // centroid of the row, two dimensions per LDS read
                typedef double cc_d2 __attribute__((ext_vector_type(2)));
                static_assert(DP % 2 == 0, "padded dimensionalities are even");
                const cc_d2* rc2 = reinterpret_cast<const cc_d2*>(rc);
                auto dim_step = [&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    const cc_d2 cp = rc2[i >> 1];
                    const double c = (i & 1) ? cp.y : cp.x;
                    // fused: the operand comes from the row's bit mask (wave-uniform, a scalar select)
                    double sc;
                    if constexpr (FUSE) sc = cc_sel_scale<(i & 31)>(i < 32 ? mlo : mhi, inv_k, one);
                    else sc = rs[i];
#pragma unroll
                    for (int t = 0; t < PT; ++t) {
                        double x = p[t][i] - c;       // mc_functions.py:37
                        x = x * x;                    // :38
                        // :39 + :41, left to right; the terms are >= +0, so 0.0 + x is x and the first one starts the sum
                        if (FUSE) {
                            acc[t] = (i == 0) ? x * sc : __builtin_fma(x, sc, acc[t]);  // see CC_TINY
                        } else {
                            x = POW2 ? x * sc : x / sc;
                            acc[t] = (i == 0) ? x : acc[t] + x;
                        }
                    }
                };
                cc_static_for<DP>(dim_step);
                const int rowg = rt + m;
                auto update = [&](auto KC) {
                    constexpr int K = decltype(KC)::value;
#pragma unroll
                    for (int t = 0; t < PT; ++t) {
                        const double a = acc[t];
                        double& d0 = bd[K][t][0];
                        double& d1 = bd[K][t][1];
                        int& s0 = bs[K][t][0];
                        int& s1 = bs[K][t][1];
                        // (lanes without a point hold -inf and never enter; no wave-level skip: with the few rows a
                        // wave sees, some lane enters on almost every row, and straight-line code updates in place)
                        bool ins = a < d1;    // enters the pair
                        bool first = a < d0;  // ... as its first element
                        const unsigned long long e1 = __builtin_amdgcn_ballot_w64(a == d1);
                        const unsigned long long e0 = __builtin_amdgcn_ballot_w64(a == d0);
                        if ((e1 | e0) != 0ull) {
                            if (a == d1 || a == d0) {
                                const int key = rows.key[rowg];
                                if (a == d1) ins = key < (s1 >= 0 ? rows.key[s1] : CC_IDX_INF);
                                if (a == d0) first = key < (s0 >= 0 ? rows.key[s0] : CC_IDX_INF);
                            }
                        }
                        if (K == 0 && filter) {
                            if (ins) {
                                // hddstream.py:317-321: pdim of the MC *with the point added* must be <= pi
                                int ne1 = 0;
                                cc_tentative_radius(rows.cf1 + (size_t)rowg * d, rows.cf2 + (size_t)rowg * d,
                                                    rows.w[rowg], X + (cursor + jj[t]) * d, d, par, nullptr, &ne1);
                                if (ne1 > par.pi) ins = false;
                            }
                            first = first && ins;
                            d1 = first ? d0 : (ins ? a : d1);
                            d0 = first ? a : d0;
                        } else {
                            // every row enters on distance alone: the distances of the pair are a plain selection
                            d1 = cc_vmin(d1, cc_vmax(d0, a));
                            d0 = cc_vmin(d0, a);
                        }
                        s1 = first ? s0 : (ins ? rowg : s1);
                        s0 = first ? rowg : s0;
                    }
                };
                if constexpr (KSEL == 0) update(std::integral_constant<int, 0>{});
                else if constexpr (KSEL == 1) update(std::integral_constant<int, 1>{});
                else {
                    if ((pmask >> m) & 1u) update(std::integral_constant<int, 0>{});
                    else if ((omask >> m) & 1u) update(std::integral_constant<int, 1>{});
                }
            }
            };
            const unsigned full = (tm >= 32) ? 0xFFFFFFFFu : ((1u << tm) - 1u);
            if (POW2 && fuse_tile) {
                if (pmask == full) clean_rows(std::true_type{}, std::integral_constant<int, 0>{});
                else if (omask == full) clean_rows(std::true_type{}, std::integral_constant<int, 1>{});
                else clean_rows(std::true_type{}, std::integral_constant<int, -1>{});
            } else clean_rows(std::false_type{}, std::integral_constant<int, -1>{});
            continue;
        }

        // The dirty scan (few waves, early exit after 4 dimensions) takes two MC rows per iteration: two
        // independent accumulation chains hide each other's latency.  The clean scan mostly runs rows to the end,
        // where pairing only adds work, and takes one.
        constexpr bool RB2 = DIRTY;
        auto dirty_rows = [&](auto FUSEC) {
        constexpr bool FUSE = decltype(FUSEC)::value;
        for (int m = 0; m < tm; m += (RB2 ? 2 : 1)) {
            const int kindA = ((rowmask >> m) & 1u) ? __builtin_amdgcn_readfirstlane(s_kind_w[m]) : CC_KIND_DEAD;
            const int kindB = (RB2 && m + 1 < tm && ((rowmask >> (m + 1)) & 1u))
                                  ? __builtin_amdgcn_readfirstlane(s_kind_w[m + 1]) : CC_KIND_DEAD;
            double boundA[PT], boundB[PT];
            auto row_bounds = [&](int mm, int kind, double (&bound)[PT]) -> bool {
                if (kind == CC_KIND_DEAD) {
#pragma unroll
                    for (int t = 0; t < PT; ++t) bound[t] = -1.0;
                    return false;
                }
                const int rowg = rt + mm;
                bool anyact = false;
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    bool a = valid[t];
                    if (DIRTY) {
                        const int nx = __builtin_amdgcn_readfirstlane(s_next_w[mm]);
                        a = a && (carried || rowg < jj[t]) && jj[t] <= nx;
                        const double b1 = (kind == 0) ? bd[0][t][0] : bd[1][t][0];
                        const double cp = (kind == 0) ? cap[0][t] : cap[1][t];
                        bound[t] = b1 < cp ? b1 : cp;
                    } else {
                        bound[t] = (kind == 0) ? bd[0][t][1] : bd[1][t][1];
                    }
                    if (!a) bound[t] = -1.0;
                    anyact = anyact || a;
                }
                return __builtin_amdgcn_ballot_w64(anyact) != 0ull;
            };
            const bool liveA = row_bounds(m, kindA, boundA);
            const bool liveB = RB2 ? row_bounds(m + 1, kindB, boundB) : false;
            if (!liveA && !liveB) continue;

            double accA[PT], accB[PT];
#pragma unroll
            for (int t = 0; t < PT; ++t) { accA[t] = 0.0; accB[t] = 0.0; }
            bool alive = true;
            const int mB = (m + 1 < CC_SCAN_TM) ? m + 1 : m;
#pragma unroll
            for (int i0 = 0; i0 < DP; i0 += 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = i0 + q;
                    if (i < DP) {
                        const double cA = s_c_base[m * DP + i], sA = s_s_base[m * DP + i];
                        const double cB = s_c_base[mB * DP + i], sB = s_s_base[mB * DP + i];
#pragma unroll
                        for (int t = 0; t < PT; ++t) {
                            double x = p[t][i] - cA;  // mc_functions.py:37
                            x = x * x;                // :38
                            if (FUSE) accA[t] = __builtin_fma(x, sA, accA[t]);  // :39 + :41 in one rounding, see CC_TINY
                            else {
                                x = POW2 ? x * sA : x / sA; // :39
                                accA[t] = accA[t] + x;    // :41, left to right
                            }
                            if (RB2) {
                                double y = p[t][i] - cB;
                                y = y * y;
                                if (FUSE) accB[t] = __builtin_fma(y, sB, accB[t]);
                                else {
                                    y = POW2 ? y * sB : y / sB;
                                    accB[t] = accB[t] + y;
                                }
                            }
                        }
                    }
                }
                if (i0 + 4 < DP) {
                    // terms are >= 0: once every point of the wave is past its bound for both rows, neither MC
                    // can enter any candidate list, whatever the remaining dimensions add
                    bool q = false;
#pragma unroll
                    for (int t = 0; t < PT; ++t) q = q || (accA[t] <= boundA[t]) || (RB2 && accB[t] <= boundB[t]);
                    if (__builtin_amdgcn_ballot_w64(q) == 0ull) {
                        alive = false;
                        break;
                    }
                }
            }
            if (!alive) continue;

            auto insert_row = [&](int mm, int kind, const double (&acc)[PT], const double (&bound)[PT]) {
                const int rowg = rt + mm;
                const int rowc = carried ? CC_CAR_BASE + rowg : rowg;  // what the candidate's slot says
                const int key = s_key_w[mm];
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    if (!(acc[t] <= bound[t])) continue;
                    auto consider = [&](auto KC) {
                        constexpr int K = decltype(KC)::value;
                        constexpr int R = DIRTY ? 0 : 1;  // rank that a newcomer has to beat
                        if (!cand_less(acc[t], key, bd[K][t][R], bk[K][t][R])) return;
                        if (K == 0 && filter) {
                            // hddstream.py:317-321: pdim of the MC *with the point added* must be <= pi
                            int ne1 = 0;
                            cc_tentative_radius(rows.cf1 + (size_t)rowg * d, rows.cf2 + (size_t)rowg * d,
                                                rows.w[rowg], X + (cursor + jj[t]) * d, d, par, nullptr, &ne1);
                            if (ne1 > par.pi) return;
                        }
                        if (cand_less(acc[t], key, bd[K][t][0], bk[K][t][0])) {
                            bd[K][t][1] = bd[K][t][0]; bk[K][t][1] = bk[K][t][0]; bs[K][t][1] = bs[K][t][0];
                            bd[K][t][0] = acc[t]; bk[K][t][0] = key; bs[K][t][0] = rowc;
                        } else {
                            bd[K][t][1] = acc[t]; bk[K][t][1] = key; bs[K][t][1] = rowc;
                        }
                    };
                    if (kind == 0) consider(std::integral_constant<int, 0>{});
                    else consider(std::integral_constant<int, 1>{});
                }
            };
            if (liveA) insert_row(m, kindA, accA, boundA);
            if (RB2 && liveB) insert_row(m + 1, kindB, accB, boundB);
        }
        };
        if (POW2 && fuse_tile) dirty_rows(std::true_type{});
        else dirty_rows(std::false_type{});
    }

    if (!DIRTY) {
        // the clean scan kept (distance, row) only: the list-order keys of the survivors
#pragma unroll
        for (int kd = 0; kd < 2; ++kd)
#pragma unroll
            for (int t = 0; t < PT; ++t)
#pragma unroll
                for (int r = 0; r < 2; ++r) bk[kd][t][r] = bs[kd][t][r] >= 0 ? rows.key[bs[kd][t][r]] : CC_IDX_INF;
    }
    // merge the waves' candidates through LDS; wave 0 writes the workgroup's partial
    Cand* const s_m = reinterpret_cast<Cand*>(smem);  // [NW - 1][PT][4][64], reuses the tile bytes
    auto s_m_at = [&](int w, int t, int c) -> Cand& { return s_m[((w * PT + t) * 4 + c) * 64 + lane]; };
    __syncthreads();  // every wave is done with its tile
    if (wv > 0) {
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            s_m_at(wv - 1, t, 0) = Cand{bd[0][t][0], bk[0][t][0], bs[0][t][0]};
            s_m_at(wv - 1, t, 1) = Cand{bd[0][t][1], bk[0][t][1], bs[0][t][1]};
            s_m_at(wv - 1, t, 2) = Cand{bd[1][t][0], bk[1][t][0], bs[1][t][0]};
            s_m_at(wv - 1, t, 3) = Cand{bd[1][t][1], bk[1][t][1], bs[1][t][1]};
        }
    }
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        if (!valid[t]) continue;
        Cand c0{bd[0][t][0], bk[0][t][0], bs[0][t][0]}, c1{bd[0][t][1], bk[0][t][1], bs[0][t][1]};
        Cand c2{bd[1][t][0], bk[1][t][0], bs[1][t][0]}, c3{bd[1][t][1], bk[1][t][1], bs[1][t][1]};
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) {
            cc_top2_push(c0, c1, s_m_at(w, t, 0));
            cc_top2_push(c0, c1, s_m_at(w, t, 1));
            cc_top2_push(c2, c3, s_m_at(w, t, 2));
            cc_top2_push(c2, c3, s_m_at(w, t, 3));
        }
        if (DIRTY) {
            Cand* o = part + ((size_t)jj[t] * S + blockIdx.y) * 2;
            o[0] = c0;
            o[1] = c2;
        } else {
            Cand* o = part + ((size_t)jj[t] * S + blockIdx.y) * 4;
            o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
        }
    }
    }
}

// ---------------------------------------------------------------------------------
// k_scan_u: the snapshot scan with the MC rows as scalar operands.  Same result as k_scan<DP, false, true, false>
// for d == DP (no pdim filter, k a power of two).  A row's centroid is the same for all 64 points of a wave: it is
// read with scalar loads (through the scalar cache, into SGPRs) and enters the FP64 instructions as their scalar
// operand; nothing of the row loop goes through LDS, whose data return rate k_scan's wave-uniform reads are bound by.
// The loads run one chunk (up to ten dimensions) ahead of the arithmetic: a chunk is requested right after the first
// use of the one before it, the first chunk of the next row after the first use of a row's last one (scalar loads
// return out of order, so a wait is always for everything outstanding: at each wait exactly one chunk is).
// Per tile of 16 rows the wave still reads the tile's 1/pref values (and the centroids, for the CC_TINY test) with
// coalesced vector loads: the ballots of `!= 1` give every row's bit mask, from which the scaled operands of two
// dimensions at a time are selected (scalar instructions), as in k_scan.
// ---------------------------------------------------------------------------------
// workgroups per CU the kernel is compiled for (register budget): the points of a lane alone are 2 * DP registers
#ifndef CC_SCANU_WGS40
#define CC_SCANU_WGS40 4  // (d = 40 at four per CU spills two registers and still measured 6 % faster than three per CU)
#endif
template <int DP>
struct ScanUShape {
    static constexpr int WGS = DP <= 32 ? 4 : (DP <= 40 ? CC_SCANU_WGS40 : 2);
};
// operands of dimensions BIT and BIT + 1 from a row's bit mask
template <int BIT>
__device__ __forceinline__ void cc_sel_scale2(unsigned mask, double scaled, double one, double& o0, double& o1)
{
    asm("s_bitcmp1_b32 %2, %3\n\ts_cselect_b64 %0, %5, %6\n\ts_bitcmp1_b32 %2, %4\n\ts_cselect_b64 %1, %5, %6"
        : "=&s"(o0), "=&s"(o1)
        : "s"(mask), "n"(BIT), "n"(BIT + 1), "s"(scaled), "s"(one)
        : "scc");
}

// `row`, usable only once `dep` has been computed: orders a scalar load after the first use of the previous one's data
__device__ __forceinline__ int cc_after(int row, double dep)
{
    asm("" : "+s"(row) : "v"(dep));
    return row;
}

template <int DP, int NW>
__global__ __launch_bounds__(64 * NW, ScanUShape<DP>::WGS) void k_scan_u(const Ctl* __restrict__ ctl, const double* __restrict__ Xt,
                                                                 const double* __restrict__ g_cen,
                                                                 const double* __restrict__ g_scl,
                                                                 const int* __restrict__ g_kind,
                                                                 const int* __restrict__ g_key, Cand* __restrict__ part,
                                                                 int round, int mode, size_t part_stride, int shard_rank,
                                                                 int shard_world)
{
    static_assert(DP % 2 == 0 && DP >= 4 && DP <= 64, "padded dimensionalities are even");
    // a row is read in NC chunks of whole pairs of dimensions (at most ten dimensions: 20 SGPRs), alternately into two
    // buffers; NC is even, so the first chunk of the next row follows the last one of a row in the other buffer
    constexpr int NP = DP / 2;
    constexpr int NC = 2 * ((NP + 9) / 10);
    constexpr int CH_MAX = 2 * ((NP + NC - 1) / NC);
    int B, m_rows_scan;
    long long cursor;
    if (mode == 1) {
        const int q = round & 1;
        B = ctl->la_b[q];
        m_rows_scan = ctl->la_rows[q];
        cursor = ctl->la_cursor[q];
        part += (size_t)q * part_stride;
    } else {
        B = ctl->win_b;
        m_rows_scan = ctl->m_rows;
        cursor = ctl->cursor;
        if (ctl->mode != 0) return;  // this window's snapshot scan ran ahead
        part += (size_t)(ctl->window_seq & 1ull) * part_stride;
    }
    if (B == 0) return;
    const int j0 = (int)blockIdx.x * 64;
    if (j0 >= B) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = gridDim.y;
    const int nsub = S * NW;
    const int sub = blockIdx.y * NW + wv;
    int row_lo = 0, row_hi = m_rows_scan;
    if (shard_world > 1) cc_shard_range(m_rows_scan, shard_world, shard_rank, 1, &row_lo, &row_hi);
    const int nrows = row_hi - row_lo;
    const int per = (nrows + nsub - 1) / nsub;
    const int r0 = row_lo + sub * per;
    const int r1 = min(row_hi, r0 + per);
    const size_t n_pts = (size_t)ctl->xt_stride;
    const double k = ctl->k;
    const double inv_k = ctl->inv_k;
    const int jj = j0 + lane;
    const bool valid = jj < B;

    double p[DP];
    {
        const double* xp = Xt + cursor + (valid ? jj : 0);
#pragma unroll
        for (int i = 0; i < DP; ++i) p[i] = valid ? xp[(size_t)i * n_pts] : 0.0;
    }
    bool fuse_wave = k >= 0x1p-64 && k <= 0x1p64;
    {
        bool tn = false;
#pragma unroll
        for (int i = 0; i < DP; ++i) tn = tn || cc_is_tiny(p[i]);
        fuse_wave = fuse_wave && __builtin_amdgcn_ballot_w64(tn) == 0ull;
    }
    // running best-two per kind: distances and rows ([kind][rank]); lanes without a point never enter
    double bd[2][2];
    int bs[2][2];
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            bd[kd][r] = valid ? CC_INF : -CC_INF;
            bs[kd][r] = -1;
        }

    for (int rt = r0; rt < r1; rt += CC_SCAN_TM) {
        const int tm = __builtin_amdgcn_readfirstlane(min(CC_SCAN_TM, r1 - rt));
        // first half of the tile's first row (in flight during the tile's vector loads; the row loop then keeps half a
        // row ahead)
        double buf[2][CH_MAX];
        {
            const double* __restrict__ c0 = g_cen + (size_t)rt * DP;
#pragma unroll
            for (int i = 0; i < 2 * (NP / NC); ++i) buf[0][i] = c0[i];
        }
        // the tile's 1/pref values and centroids, coalesced: bit masks of the rows, CC_TINY test
        constexpr int NL = (CC_SCAN_TM * DP + 63) / 64;
        int rm_lo = 0, rm_hi = 0;
        bool fuse_tile;
        {
            const double* gc = g_cen + (size_t)rt * DP;
            const double* gs = g_scl + (size_t)rt * DP;
            unsigned long long rmask = 0ull;
            bool tn = false;
            const int off = (lane & (CC_SCAN_TM - 1)) * DP;
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const int e = lane + q * 64;
                const bool in = e < tm * DP;
                const double tc = in ? gc[e] : 0.0;
                const double ts = in ? gs[e] : 1.0;
                tn = tn || cc_is_tiny(tc);
                const unsigned long long w = __builtin_amdgcn_ballot_w64(ts != 1.0);
                const int rel = off - 64 * q;
                const unsigned long long a = (rel >= 0 && rel < 64) ? (w >> (rel & 63)) : 0ull;
                const unsigned long long b = (rel < 0 && rel > -DP) ? (w << ((-rel) & 63)) : 0ull;
                rmask |= a | b;
            }
            if (DP < 64) rmask &= (1ull << (DP & 63)) - 1ull;
            rm_lo = (int)(unsigned)(rmask & 0xFFFFFFFFull);
            rm_hi = (int)(unsigned)(rmask >> 32);
            fuse_tile = fuse_wave && __builtin_amdgcn_ballot_w64(tn) == 0ull;
        }
        unsigned pmask, omask;
        {
            const int kd = (lane < tm) ? g_kind[rt + lane] : CC_KIND_DEAD;
            pmask = (unsigned)__builtin_amdgcn_ballot_w64(kd == CC_KIND_PCORE);
            omask = (unsigned)__builtin_amdgcn_ballot_w64(kd == CC_KIND_OUTLIER);
        }
        auto rows_of_tile = [&](auto FUSEC, auto KSELC) {
            constexpr bool FUSE = decltype(FUSEC)::value;
            constexpr int KSEL = decltype(KSELC)::value;
            for (int m = 0; m < tm; ++m) {
                const unsigned mlo = (unsigned)__builtin_amdgcn_readlane(rm_lo, m);
                const unsigned mhi = (DP > 32) ? (unsigned)__builtin_amdgcn_readlane(rm_hi, m) : 0u;
                const int rowg = rt + m;
                const int rown = min(rowg + 1, rt + tm - 1);  // (the last row of a tile requests its own first half again)
                const double one = 1.0;
                double acc = 0.0;
                // one pair of dimensions: mc_functions.py:37-41, left to right
                auto pair_step = [&](auto IC, double c0, double c1, double x0) {
                    constexpr int i = decltype(IC)::value;
                    double s0, s1;
                    cc_sel_scale2<(i & 31)>(i < 32 ? mlo : mhi, inv_k, one, s0, s1);
                    double x = x0;            // p[i] - c0, made by the caller
                    double y = p[i + 1] - c1;
                    x = x * x;
                    y = y * y;
                    if (FUSE) {
                        acc = (i == 0) ? x * s0 : __builtin_fma(x, s0, acc);  // :39 + :41 in one rounding, see CC_TINY
                        acc = __builtin_fma(y, s1, acc);
                    } else {
                        x = x * s0;  // :39 (the divisor is a power of two)
                        y = y * s1;
                        acc = (i == 0) ? x : acc + x;  // :41
                        acc = acc + y;
                    }
                };
                cc_static_for<NC>([&](auto CC) {
                    constexpr int c = decltype(CC)::value;
                    constexpr int lo = 2 * (c * NP / NC), hi = 2 * ((c + 1) * NP / NC);        // this chunk's dimensions
                    constexpr int cn = (c + 1) % NC;                                          // the chunk requested now
                    constexpr int nlo = 2 * (cn * NP / NC), nhi = 2 * ((cn + 1) * NP / NC);
                    // first use of the chunk requested one chunk ago: the wait is here, with nothing else outstanding
                    const double x0 = p[lo] - buf[c & 1][0];
                    // wave-uniform address: scalar loads; of this row, or of the next one after the last chunk
                    const double* __restrict__ nx = g_cen + (size_t)cc_after(c + 1 < NC ? rowg : rown, x0) * DP;
#pragma unroll
                    for (int i = 0; i < nhi - nlo; ++i) buf[cn & 1][i] = nx[nlo + i];
                    cc_static_for<(hi - lo) / 2>([&](auto QC) {
                        constexpr int i = 2 * decltype(QC)::value;
                        pair_step(std::integral_constant<int, lo + i>{}, buf[c & 1][i], buf[c & 1][i + 1],
                                  (i == 0) ? x0 : p[lo + i] - buf[c & 1][i]);
                    });
                });
                auto update = [&](auto KC) {
                    constexpr int K = decltype(KC)::value;
                    const double a = acc;
                    double& d0 = bd[K][0];
                    double& d1 = bd[K][1];
                    int& s0 = bs[K][0];
                    int& s1 = bs[K][1];
                    bool ins = a < d1;
                    bool first = a < d0;
                    // exact ties: list order decides (hddstream.py:326/373, strict `<`)
                    const unsigned long long e1 = __builtin_amdgcn_ballot_w64(a == d1);
                    const unsigned long long e0 = __builtin_amdgcn_ballot_w64(a == d0);
                    if ((e1 | e0) != 0ull) {
                        if (a == d1 || a == d0) {
                            const int key = g_key[rowg];
                            if (a == d1) ins = key < (s1 >= 0 ? g_key[s1] : CC_IDX_INF);
                            if (a == d0) first = key < (s0 >= 0 ? g_key[s0] : CC_IDX_INF);
                        }
                    }
                    d1 = cc_vmin(d1, cc_vmax(d0, a));
                    d0 = cc_vmin(d0, a);
                    s1 = first ? s0 : (ins ? rowg : s1);
                    s0 = first ? rowg : s0;
                };
                if constexpr (KSEL == 0) update(std::integral_constant<int, 0>{});
                else if constexpr (KSEL == 1) update(std::integral_constant<int, 1>{});
                else {
                    if ((pmask >> m) & 1u) update(std::integral_constant<int, 0>{});
                    else if ((omask >> m) & 1u) update(std::integral_constant<int, 1>{});
                }
            }
        };
        const unsigned full = (1u << tm) - 1u;
        if (fuse_tile) {
            if (pmask == full) rows_of_tile(std::true_type{}, std::integral_constant<int, 0>{});
            else if (omask == full) rows_of_tile(std::true_type{}, std::integral_constant<int, 1>{});
            else rows_of_tile(std::true_type{}, std::integral_constant<int, -1>{});
        } else rows_of_tile(std::false_type{}, std::integral_constant<int, -1>{});
    }

    // the list-order keys of the survivors, then the waves' candidates merged through LDS as in k_scan
    int bk[2][2];
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
#pragma unroll
        for (int r = 0; r < 2; ++r) bk[kd][r] = bs[kd][r] >= 0 ? g_key[bs[kd][r]] : CC_IDX_INF;
    __shared__ Cand s_m[(NW > 1 ? NW - 1 : 1) * 4 * 64];
    auto s_m_at = [&](int w, int c) -> Cand& { return s_m[(w * 4 + c) * 64 + lane]; };
    if (wv > 0) {
        s_m_at(wv - 1, 0) = Cand{bd[0][0], bk[0][0], bs[0][0]};
        s_m_at(wv - 1, 1) = Cand{bd[0][1], bk[0][1], bs[0][1]};
        s_m_at(wv - 1, 2) = Cand{bd[1][0], bk[1][0], bs[1][0]};
        s_m_at(wv - 1, 3) = Cand{bd[1][1], bk[1][1], bs[1][1]};
    }
    __syncthreads();
    if (wv != 0 || !valid) return;
    Cand c0{bd[0][0], bk[0][0], bs[0][0]}, c1{bd[0][1], bk[0][1], bs[0][1]};
    Cand c2{bd[1][0], bk[1][0], bs[1][0]}, c3{bd[1][1], bk[1][1], bs[1][1]};
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
        cc_top2_push(c0, c1, s_m_at(w, 0));
        cc_top2_push(c0, c1, s_m_at(w, 1));
        cc_top2_push(c2, c3, s_m_at(w, 2));
        cc_top2_push(c2, c3, s_m_at(w, 3));
    }
    Cand* o = part + ((size_t)jj * S + blockIdx.y) * 4;
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// ---------------------------------------------------------------------------------
// The PRUNED snapshot scan: k_seed -> k_seed_merge -> k_scan_p.  Same contract as k_scan_u (per point and kind the
// best candidates by (projected distance, list order)), for a fraction of its arithmetic.
//
// A distance is a sum of non-negative terms taken left to right (mc_functions.py:37-41), so its partial sums never
// decrease: a row whose partial sum already exceeds a threshold T cannot have a distance <= T.  With T well above the
// distance of the point's nearest microcluster, almost every row drops out after four or eight dimensions - for all 64
// points of a wave at once, because the other microclusters are far from every one of them (the test is wave-uniform:
// a row is abandoned when ALL lanes are over their thresholds; otherwise its distance is completed for all lanes).
//   k_seed        per point and kind the row with the smallest UNSCALED squared distance over the first eight
//                 dimensions, in single precision (a heuristic: nothing downstream relies on it being the nearest), per
//                 wave sub-range.  It has to be the point's nearest microcluster almost always, though: one lane with a
//                 far seed keeps its whole wave evaluating every row in full - hence eight dimensions, not four (in
//                 four, 1 % of the C2 points have another of the 5 000 microclusters closer than their own)
//   k_seed_merge  per point and kind: the three best of those, their exact distances, T = F x the smallest, and the
//                 single-precision threshold T32 that goes with it (see there)
//   k_scan_p      per tile of 16 rows: phase A abandons rows on an eight-dimension single-precision prefix sum, phase B
//                 completes the others in double precision with the abandon test every eight dimensions; per kind it keeps
//                 the two best EVALUATED rows and a lower bound (> = T) for every abandoned row's distance.  What leaves
//                 the kernel per kind is a pair (best, second) in which `second` may be a BOUND (CC_SLOT_BOUND): the
//                 best is exact whenever it is <= T (the seed row always is evaluated), the second is exact when it is
//                 smaller than every abandoned partial sum, else all that is known of the other rows is that none is
//                 closer than the bound.  Pairs merge like candidate pairs (cc_top2_push), in any order.
// k_decide treats a bound in second place like a second-best candidate that is dirty: the decision is exact iff a live
// version beats the bound, otherwise the point is undecidable in this window (CC_T_UNKNOWN) - with F = 16 that takes a
// microcluster whose live version is four times as far (in distance units) as the seed was.
// ---------------------------------------------------------------------------------

struct __attribute__((aligned(8))) SeedCand {
    float part;   // unscaled squared distance over the first eight dimensions (single precision)
    int row;      // -1: none
};

// the window a snapshot scan works on: mode 0 = the current window (in place), 1 = the lookahead window of parity round & 1
struct ScanWin {
    int B, rows, q;
    long long cursor;
};
__device__ __forceinline__ ScanWin cc_scan_window(const Ctl* __restrict__ ctl, int round, int mode)
{
    ScanWin w;
    if (mode == 1) {
        w.q = round & 1;
        w.B = ctl->la_b[w.q];
        w.rows = ctl->la_rows[w.q];
        w.cursor = ctl->la_cursor[w.q];
    } else {
        w.B = (ctl->mode != 0) ? 0 : ctl->win_b;  // (mode != 0: this window's snapshot scan ran ahead)
        w.rows = ctl->m_rows;
        w.cursor = ctl->cursor;
        w.q = (int)(ctl->window_seq & 1ull);
    }
    return w;
}

// single-precision pairs: the prefix arithmetic of k_seed and of k_scan_p's phase A runs on packed FP32 instructions
typedef float cc_f2 __attribute__((ext_vector_type(2)));
typedef float cc_f4 __attribute__((ext_vector_type(4)));
#define CC_PRE 8  // dimensions of the prefix (k_seed's score, phase A's bound): 8 floats = two 16-byte LDS reads per row

// the wave's tile of 16 row prefixes, converted to single precision and staged in LDS: tile[m * 8 + i]
// (lane + 64 q = 8 m + i); returns the largest |coordinate| this lane saw
template <int DP>
__device__ __forceinline__ void cc_load_prefix(const double* __restrict__ g_cen, const int* __restrict__ g_kind, int rt,
                                               int tm, int lane, double (&tc)[2], int& kd)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = lane + q * 64, m = e >> 3, i = e & 7;
        tc[q] = (m < tm) ? g_cen[(size_t)(rt + m) * DP + i] : 0.0;
    }
    kd = (lane < tm) ? g_kind[rt + lane] : CC_KIND_DEAD;
}

// sum over the prefix of (p - c)^2 in single precision, two dimensions per instruction
__device__ __forceinline__ float cc_prefix_score(const cc_f2 (&p2)[CC_PRE / 2], const cc_f4* __restrict__ row)
{
    const cc_f4 c01 = row[0], c23 = row[1];
    cc_f2 x0 = p2[0] - cc_f2{c01.x, c01.y};
    cc_f2 x1 = p2[1] - cc_f2{c01.z, c01.w};
    cc_f2 x2 = p2[2] - cc_f2{c23.x, c23.y};
    cc_f2 x3 = p2[3] - cc_f2{c23.z, c23.w};
    cc_f2 acc = x0 * x0;
    acc = __builtin_elementwise_fma(x1, x1, acc);
    acc = __builtin_elementwise_fma(x2, x2, acc);
    acc = __builtin_elementwise_fma(x3, x3, acc);
    return acc.x + acc.y;
}

// k_seed: per point, kind and sub-range of rows the row with the smallest squared distance over the first eight
// dimensions, in single precision.  Two points per lane (a workgroup covers 128 window points: a row's prefix is read
// from LDS once for both), and the score in its expanded form: with p' = p - o, c' = c - o (o = the prefix of table
// row 0: keeps the magnitudes at the size of the data's spread whatever its offset)
//     |p' - c'|^2 = |p'|^2 - 2 (p' . c' - |c'|^2 / 2),
// so the nearest row is the one with the LARGEST g = p' . c' - h, h = |c'|^2 / 2 staged with the tile: four packed
// multiply-adds, an add, a compare and two selects per row and point (the round-3 kernel's difference form - four
// packed subtractions more, one point per lane - took 61 us where this one takes 54, `profiles/r03_tool_seed.txt`).
// The cancellation costs a few units of 2^-24 |c'|^2: immaterial for a heuristic.
// cmax[q] (bits of a double): the largest |centroid coordinate| among the prefixes of the scanned rows, left by the
// workgroups of the window's first point tile (every row is in exactly one of their waves' sub-ranges)
template <int DP, int NW>
__global__ __launch_bounds__(64 * NW, 4) void k_seed(const Ctl* __restrict__ ctl, const double* __restrict__ Xt,
                                                   const double* __restrict__ g_cen, const int* __restrict__ g_kind,
                                                   SeedCand* __restrict__ spart, int round, int mode, size_t spart_stride,
                                                   unsigned long long* __restrict__ cmax)
{
    static_assert(CC_PRE == 8 && CC_PRE <= DP, "prefix dimensions");
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int B = win.B;
    if (B == 0) return;
    const int j0 = (int)blockIdx.x * 128;
    if (j0 >= B) return;
    spart += (size_t)win.q * spart_stride;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = gridDim.y;
    const int nsub = S * NW;
    const int sub = blockIdx.y * NW + wv;
    const int per = (win.rows + nsub - 1) / nsub;
    const int r0 = sub * per;
    const int r1 = min(win.rows, r0 + per);
    const size_t n_pts = (size_t)ctl->xt_stride;
    double org[CC_PRE];
#pragma unroll
    for (int i = 0; i < CC_PRE; ++i) org[i] = g_cen[i];  // (wave-uniform: scalar loads)
    int jj[2];
    bool valid[2];
    cc_f2 p2[2][CC_PRE / 2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        jj[u] = j0 + u * 64 + lane;
        valid[u] = jj[u] < B;
        const double* xp = Xt + win.cursor + (valid[u] ? jj[u] : 0);
#pragma unroll
        for (int i = 0; i < CC_PRE / 2; ++i)
            p2[u][i] = cc_f2{valid[u] ? (float)(xp[(size_t)(2 * i) * n_pts] - org[2 * i]) : 0.f,
                             valid[u] ? (float)(xp[(size_t)(2 * i + 1) * n_pts] - org[2 * i + 1]) : 0.f};
    }
    // per wave: the centred prefixes of a tile of 16 rows and their h in LDS, read back as wave-uniform broadcasts
    __shared__ __attribute__((aligned(16))) float s_pre[NW * CC_SCAN_TM * CC_PRE];
    __shared__ float s_h[NW * CC_SCAN_TM];
    float* const tile = s_pre + (size_t)wv * CC_SCAN_TM * CC_PRE;
    float* const th = s_h + (size_t)wv * CC_SCAN_TM;
    float best[2][2];
    int idx[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int K = 0; K < 2; ++K) { best[u][K] = -__builtin_inff(); idx[u][K] = -1; }
    double tc[2];
    int kdl = CC_KIND_DEAD;
    double cm = 0.0;
    if (r0 < r1) cc_load_prefix<DP>(g_cen, g_kind, r0, min(CC_SCAN_TM, r1 - r0), lane, tc, kdl);
    for (int rt = r0; rt < r1; rt += CC_SCAN_TM) {
        const int tm = __builtin_amdgcn_readfirstlane(min(CC_SCAN_TM, r1 - rt));
        CC_WAVE_SYNC();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            cm = __builtin_fmax(cm, __builtin_fabs(tc[q]));
            const float v = (float)(tc[q] - org[lane & 7]);
            tile[lane + q * 64] = v;
            // h of the row: the eight lanes that hold it add their squares (every lane ends up with the sum)
            float hsum = v * v;
            hsum += __shfl_xor(hsum, 1);
            hsum += __shfl_xor(hsum, 2);
            hsum += __shfl_xor(hsum, 4);
            if ((lane & 7) == 0) th[(lane >> 3) + q * 8] = 0.5f * hsum;
        }
        const unsigned pmask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_PCORE);
        const unsigned omask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_OUTLIER);
        CC_WAVE_SYNC();
        if (rt + CC_SCAN_TM < r1) cc_load_prefix<DP>(g_cen, g_kind, rt + CC_SCAN_TM, min(CC_SCAN_TM, r1 - rt - CC_SCAN_TM), lane, tc, kdl);
        const cc_f4* t4 = reinterpret_cast<const cc_f4*>(__builtin_assume_aligned(tile, 16));
        auto score2 = [&](int m, float (&g)[2]) {
            const cc_f4 c01 = t4[m * 2], c23 = t4[m * 2 + 1];
            const float h = th[m];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                cc_f2 acc = p2[u][0] * cc_f2{c01.x, c01.y};
                acc = __builtin_elementwise_fma(p2[u][1], cc_f2{c01.z, c01.w}, acc);
                acc = __builtin_elementwise_fma(p2[u][2], cc_f2{c23.x, c23.y}, acc);
                acc = __builtin_elementwise_fma(p2[u][3], cc_f2{c23.z, c23.w}, acc);
                g[u] = (acc.x + acc.y) - h;
            }
        };
        auto update = [&](auto KC, const float (&g)[2], int rowg) {
            constexpr int K = decltype(KC)::value;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool gt = g[u] > best[u][K];  // strict: the first row in scan order keeps a tie (deterministic)
                best[u][K] = gt ? g[u] : best[u][K];
                idx[u][K] = gt ? rowg : idx[u][K];
            }
        };
        auto rows_of_kind = [&](auto KC) {
            int m = 0;
            for (; m + 2 <= tm; m += 2) {
                float a[2][2];
#pragma unroll
                for (int v = 0; v < 2; ++v) score2(m + v, a[v]);
#pragma unroll
                for (int v = 0; v < 2; ++v) update(KC, a[v], rt + m + v);
            }
            for (; m < tm; ++m) {
                float a[2];
                score2(m, a);
                update(KC, a, rt + m);
            }
        };
        const unsigned full = (1u << tm) - 1u;
        if (pmask == full) rows_of_kind(std::integral_constant<int, 0>{});
        else if (omask == full) rows_of_kind(std::integral_constant<int, 1>{});
        else
            for (int m = 0; m < tm; ++m) {
                float a[2];
                score2(m, a);
                if ((pmask >> m) & 1u) update(std::integral_constant<int, 0>{}, a, rt + m);
                else if ((omask >> m) & 1u) update(std::integral_constant<int, 1>{}, a, rt + m);
            }
    }
    if (blockIdx.x == 0) {
        for (int off = 32; off >= 1; off >>= 1) cm = __builtin_fmax(cm, __shfl_xor(cm, off));
        if (lane == 0) atomicMax(cmax + win.q, (unsigned long long)__double_as_longlong(cm));  // (>= 0: bits order like values)
    }
    // back to squared prefix distances (what k_seed_merge ranks the sub-ranges' winners by): |p'|^2 - 2 g, never below 0
    float pp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        cc_f2 acc = p2[u][0] * p2[u][0];
#pragma unroll
        for (int i = 1; i < CC_PRE / 2; ++i) acc = __builtin_elementwise_fma(p2[u][i], p2[u][i], acc);
        pp[u] = acc.x + acc.y;
#pragma unroll
        for (int K = 0; K < 2; ++K) best[u][K] = idx[u][K] >= 0 ? __builtin_fmaxf(0.f, pp[u] - 2.f * best[u][K]) : __builtin_inff();
    }
    __shared__ float s_b[(NW > 1 ? NW - 1 : 1) * 4 * 64];
    __shared__ int s_i[(NW > 1 ? NW - 1 : 1) * 4 * 64];
    if (wv > 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int K = 0; K < 2; ++K) {
                s_b[((wv - 1) * 4 + u * 2 + K) * 64 + lane] = best[u][K];
                s_i[((wv - 1) * 4 + u * 2 + K) * 64 + lane] = idx[u][K];
            }
    }
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (!valid[u]) continue;
#pragma unroll
        for (int w = 0; w < NW - 1; ++w)
#pragma unroll
            for (int K = 0; K < 2; ++K) {
                const float b = s_b[(w * 4 + u * 2 + K) * 64 + lane];
                const int ix = s_i[(w * 4 + u * 2 + K) * 64 + lane];
                const bool lt = ix >= 0 && (idx[u][K] < 0 || b < best[u][K]);
                best[u][K] = lt ? b : best[u][K];
                idx[u][K] = lt ? ix : idx[u][K];
            }
        SeedCand* o = spart + ((size_t)jj[u] * S + blockIdx.y) * 2;
        o[0] = SeedCand{best[u][0], idx[u][0]};
        o[1] = SeedCand{best[u][1], idx[u][1]};
    }
}

// per point and kind (one thread each): the three best prefix scores of the sub-ranges -> their exact distances (the
// scans' own operations, in their order; the three sums advance together) -> T = F x the smallest; +inf when the kind
// has no row.  And T32, the threshold phase A's SINGLE-PRECISION prefix sum is compared with.  Phase A abandons a row when
//     Qf = smin * sum_{i < 8} fl32(fl32(p_i) - fl32(c_i))^2     exceeds T32,
// and that must imply that the row's exact partial sum P = sum_i s_i (p_i - c_i)^2 (s_i = 1 or 1/k) exceeds T.  With
// e = 2^-21 max(|p|, |c|) (twice the bound 2^-24 (|p_i| + |c_i| + |x_i|) on the error of a difference x_i),
//     P >= sum s_i (|x_i| - e)^2 >= Q - 2 e sum s_i |x_i| >= Q - a sqrt(Q),   a = 2 e sqrt(8 smax),  Q = sum s_i x_i^2 >= smin sum x_i^2
// (Cauchy-Schwarz), g(Q) = Q - a sqrt(Q) grows for sqrt(Q) > a / 2, so P > T follows from sqrt(Q) > u = (a + sqrt(a^2 + 4 T)) / 2.
// The nine roundings of Qf (relative 2^-24 each, all terms >= 0) are covered by the factor 1 + 2^-19; T32 is rounded up.
template <int DP>
__global__ __launch_bounds__(64) void k_seed_merge(const Ctl* __restrict__ ctl, const double* __restrict__ X,
                                                   const double* __restrict__ g_cen, const double* __restrict__ g_scl,
                                                   const SeedCand* __restrict__ spart, size_t spart_stride, int S,
                                                   double* __restrict__ thr, float* __restrict__ thr32, size_t thr_stride,
                                                   double F, int round, int mode, const unsigned long long* __restrict__ cmax)
{
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t >> 1, K = t & 1;
    if (j >= win.B) return;
    constexpr int d = DP;  // (the pruned scan runs for d == DP only: every loop below unrolls, its loads go out together)
    spart += (size_t)win.q * spart_stride;
    thr += (size_t)win.q * thr_stride;
    thr32 += (size_t)win.q * thr_stride;
    const double* p = X + (size_t)(win.cursor + j) * d;
    float b0 = __builtin_inff(), b1 = __builtin_inff(), b2 = __builtin_inff();
    int i0 = -1, i1 = -1, i2 = -1;
    for (int s = 0; s < S; ++s) {
        const SeedCand c = spart[((size_t)j * S + s) * 2 + K];
        if (c.row < 0) continue;
        if (i0 < 0 || c.part < b0) { b2 = b1; i2 = i1; b1 = b0; i1 = i0; b0 = c.part; i0 = c.row; }
        else if (i1 < 0 || c.part < b1) { b2 = b1; i2 = i1; b1 = c.part; i1 = c.row; }
        else if (i2 < 0 || c.part < b2) { b2 = c.part; i2 = c.row; }
    }
    double out = CC_INF;
    if (i0 >= 0) {
        const bool h1 = i1 >= 0, h2 = i2 >= 0;
        const size_t o0 = (size_t)i0 * d, o1 = (size_t)(h1 ? i1 : i0) * d, o2 = (size_t)(h2 ? i2 : i0) * d;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        // eight dimensions of the three rows per pass: 56 loads in flight, the sums left to right
        for (int i0 = 0; i0 < d; i0 += 8) {
            double pv[8], c0[8], c1[8], c2[8], s0[8], s1[8], s2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = (i0 + u < d) ? i0 + u : d - 1;
                pv[u] = p[i];
                c0[u] = g_cen[o0 + i]; c1[u] = g_cen[o1 + i]; c2[u] = g_cen[o2 + i];
                s0[u] = g_scl[o0 + i]; s1[u] = g_scl[o1 + i]; s2[u] = g_scl[o2 + i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u < d) {
                    double x0 = pv[u] - c0[u], x1 = pv[u] - c1[u], x2 = pv[u] - c2[u];
                    x0 = x0 * x0; x1 = x1 * x1; x2 = x2 * x2;
                    x0 = x0 * s0[u]; x1 = x1 * s1[u]; x2 = x2 * s2[u];
                    a0 = a0 + x0; a1 = a1 + x1; a2 = a2 + x2;
                }
            }
        }
        double dmin = a0;
        dmin = a1 < dmin ? a1 : dmin;
        dmin = a2 < dmin ? a2 : dmin;
        out = F * dmin;
    }
    thr[(size_t)j * 2 + K] = out;
    float t32 = __builtin_inff();
    if (out < CC_INF) {
        double pm = 0.0;
        for (int i = 0; i < CC_PRE; ++i) pm = __builtin_fmax(pm, __builtin_fabs(p[i]));
        const double cmx = __longlong_as_double((long long)cmax[win.q]);
        const double e = 0x1p-21 * __builtin_fmax(pm, cmx);
        const double inv_k = ctl->inv_k;
        const double smax = inv_k > 1.0 ? inv_k : 1.0, smin = inv_k < 1.0 ? inv_k : 1.0;
        const double a = 2.0 * e * sqrt(8.0 * smax);
        const double u = 0.5 * (a + sqrt(a * a + 4.0 * out)) * (1.0 + 0x1p-40);
        // the kernel compares sum x^2 (without smin) with T32 = u^2 (1 + 2^-19) / smin
        const double t64 = u * u * (1.0 + 0x1p-19) / smin * (1.0 + 0x1p-40);
        t32 = (float)t64;
        if ((double)t32 < t64) t32 = __uint_as_float(__float_as_uint(t32) + 1u);  // (t32 >= 0 and finite here: the next float up)
    }
    thr32[(size_t)j * 2 + K] = t32;
}

// Per wave and tile of 16 rows two phases:
//   A  every row, straight-line, in SINGLE precision: the sum over the first eight dimensions of (p - c)^2 (packed FP32
//      instructions: 4 subtractions, 4 multiply-adds and an add per row) against the lane's threshold T32 (k_seed_merge:
//      exceeding it implies that the row's exact partial sum exceeds T, whatever the row's preferred dimensions are),
//      the wave-uniform test "some lane within its threshold", one bit per row; a row that no lane keeps leaves T in the
//      kind's bound.  Only the first eight dimensions of the tile's rows are fetched (16 x 8 doubles, coalesced,
//      converted and staged in the wave's LDS tile, read back as wave-uniform broadcasts; the next tile's loads are in
//      flight during the row loop of the current one) - whole rows, as k_scan stages them, would be five times the bytes
//      at d = 20 for 2 % of the rows, and the same lines are wanted by every point tile's workgroup at the same moment.
//   B  the rows phase A kept (few): the whole distance from its first dimension with the reference's four operations
//      per term in double precision (sub, square, scale, add - no fusion, so no CC_TINY condition to check), centroid
//      and operand as scalar loads of eight dimensions at a time, the abandon test (now exact: partial sum against T)
//      every eight dimensions, then the best-two update of k_scan_u.
#ifndef CC_SCANP_WGS20
#define CC_SCANP_WGS20 4  // workgroups per CU k_scan_p is compiled for at d <= 20
#endif
template <int DP, int NW>
__global__ __launch_bounds__(64 * NW, (DP <= 20 ? CC_SCANP_WGS20 : (DP <= 40 ? 3 : 2))) void k_scan_p(
    Ctl* __restrict__ ctl, const double* __restrict__ Xt, const double* __restrict__ g_cen, const double* __restrict__ g_scl,
    const int* __restrict__ g_kind, const int* __restrict__ g_key, const double* __restrict__ thr,
    const float* __restrict__ thr32, size_t thr_stride, Cand* __restrict__ part, int round, int mode, size_t part_stride)
{
    static_assert(DP % 2 == 0 && DP > CC_PRE && DP <= 64, "k_scan_p shapes");
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int B = win.B;
    if (B == 0) return;
    const int j0 = (int)blockIdx.x * 64;
    if (j0 >= B) return;
    part += (size_t)win.q * part_stride;
    thr += (size_t)win.q * thr_stride;
    thr32 += (size_t)win.q * thr_stride;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = gridDim.y;
    const int nsub = S * NW;
    const int sub = blockIdx.y * NW + wv;
    const int per = (win.rows + nsub - 1) / nsub;
    const int r0 = sub * per;
    const int r1 = min(win.rows, r0 + per);
    const size_t n_pts = (size_t)ctl->xt_stride;
    const int jj = j0 + lane;
    const bool valid = jj < B;

    constexpr int TILE_BYTES = NW * CC_SCAN_TM * CC_PRE * 4;
    constexpr int MERGE_BYTES = (NW - 1) * 4 * 64 * (int)sizeof(Cand);
    __shared__ __attribute__((aligned(16))) unsigned char smem[TILE_BYTES > MERGE_BYTES ? TILE_BYTES : MERGE_BYTES];
    float* const tile = reinterpret_cast<float*>(smem) + (size_t)wv * CC_SCAN_TM * CC_PRE;

    double p[DP];
    {
        const double* xp = Xt + win.cursor + (valid ? jj : 0);
#pragma unroll
        for (int i = 0; i < DP; ++i) p[i] = valid ? xp[(size_t)i * n_pts] : 0.0;
    }
    cc_f2 p2[CC_PRE / 2];
#pragma unroll
    for (int i = 0; i < CC_PRE / 2; ++i) p2[i] = cc_f2{(float)p[2 * i], (float)p[2 * i + 1]};
    // thresholds (exact: th, single-precision prefix: th32) and the bound of what was abandoned, per kind; lanes without
    // a point keep no row alive
    double th[2], lb[2] = {CC_INF, CC_INF};
    float th32[2];
#pragma unroll
    for (int K = 0; K < 2; ++K) {
        th[K] = valid ? thr[(size_t)jj * 2 + K] : -CC_INF;
        th32[K] = valid ? thr32[(size_t)jj * 2 + K] : -__builtin_inff();
    }
    bool dropped[2] = {false, false};  // (wave-uniform) phase A abandoned a row of the kind: every lane's bound is its T
    double bd[2][2];
    int bs[2][2];
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            bd[kd][r] = valid ? CC_INF : -CC_INF;
            bs[kd][r] = -1;
        }
    int n_rows = 0, n_full = 0;  // statistics (wave-uniform)

    double tc[2];
    int kdl = CC_KIND_DEAD;
    if (r0 < r1) cc_load_prefix<DP>(g_cen, g_kind, r0, min(CC_SCAN_TM, r1 - r0), lane, tc, kdl);
    for (int rt = r0; rt < r1; rt += CC_SCAN_TM) {
        const int tm = __builtin_amdgcn_readfirstlane(min(CC_SCAN_TM, r1 - rt));
        CC_WAVE_SYNC();
#pragma unroll
        for (int q = 0; q < 2; ++q) tile[lane + q * 64] = (float)tc[q];
        const unsigned pmask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_PCORE);
        const unsigned omask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_OUTLIER);
        CC_WAVE_SYNC();
        if (rt + CC_SCAN_TM < r1) cc_load_prefix<DP>(g_cen, g_kind, rt + CC_SCAN_TM, min(CC_SCAN_TM, r1 - rt - CC_SCAN_TM), lane, tc, kdl);
        n_rows += tm;
        const cc_f4* t4 = reinterpret_cast<const cc_f4*>(__builtin_assume_aligned(tile, 16));
        const unsigned full = (1u << tm) - 1u;

        // ---- phase A ----
        unsigned surv = 0u;
        // straight-line: one compare per row, the wave's verdict as one bit (which kinds lost rows follows from the
        // bits at the end of the tile)
        auto verdict = [&](auto KSELC, int m, float q) {
            constexpr int KSEL = decltype(KSELC)::value;
            float t;
            if constexpr (KSEL == 0) t = th32[0];
            else if constexpr (KSEL == 1) t = th32[1];
            else t = ((pmask >> m) & 1u) ? th32[0] : th32[1];
            surv |= (__builtin_amdgcn_ballot_w64(q <= t) != 0ull) ? (1u << m) : 0u;
        };
        auto phase_a = [&](auto KSELC) {
            int m = 0;
            for (; m + 4 <= tm; m += 4) {
                float a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = cc_prefix_score(p2, t4 + (m + u) * 2);
#pragma unroll
                for (int u = 0; u < 4; ++u) verdict(KSELC, m + u, a[u]);
            }
            for (; m < tm; ++m) verdict(KSELC, m, cc_prefix_score(p2, t4 + m * 2));
        };
        if (pmask == full) phase_a(std::integral_constant<int, 0>{});
        else if (omask == full) phase_a(std::integral_constant<int, 1>{});
        else phase_a(std::integral_constant<int, -1>{});
        if ((~surv & pmask & full) != 0u) dropped[0] = true;
        if ((~surv & omask & full) != 0u) dropped[1] = true;

        // ---- phase B: the rows that stayed ----
        while (surv != 0u) {
            const int m = __builtin_ctz(surv);
            surv &= surv - 1u;
            const int rowg = rt + m;
            const bool is_p = ((pmask >> m) & 1u) != 0u;
            if (!is_p && ((omask >> m) & 1u) == 0u) continue;  // (neither list)
            const double* __restrict__ rc = g_cen + (size_t)rowg * DP;  // wave-uniform addresses: scalar loads
            const double* __restrict__ rs = g_scl + (size_t)rowg * DP;
            double acc = 0.0;
            bool gone = false;
            cc_static_for<(DP + 7) / 8>([&](auto CC) {
                constexpr int lo = 8 * decltype(CC)::value, hi = (lo + 8) < DP ? (lo + 8) : DP;
                if (gone) return;
                double c[hi - lo], sc[hi - lo];
#pragma unroll
                for (int i = 0; i < hi - lo; ++i) {
                    c[i] = rc[lo + i];
                    sc[i] = rs[lo + i];
                }
#pragma unroll
                for (int i = 0; i < hi - lo; ++i) {
                    double x = p[lo + i] - c[i];          // mc_functions.py:37
                    x = x * x;                             // :38
                    x = x * sc[i];                         // :39 (the divisor is a power of two)
                    acc = (lo + i == 0) ? x : acc + x;     // :41
                }
                if constexpr (hi < DP) {
                    // all lanes over their thresholds: the row is abandoned; its partial sum bounds its distance from below
                    if (is_p) {
                        if (__builtin_amdgcn_ballot_w64(acc <= th[0]) == 0ull) { lb[0] = cc_vmin(lb[0], acc); gone = true; }
                    } else {
                        if (__builtin_amdgcn_ballot_w64(acc <= th[1]) == 0ull) { lb[1] = cc_vmin(lb[1], acc); gone = true; }
                    }
                }
            });
            if (gone) continue;
            ++n_full;
            auto update = [&](auto KC) {
                constexpr int K = decltype(KC)::value;
                const double a = acc;
                double& d0 = bd[K][0];
                double& d1 = bd[K][1];
                int& s0 = bs[K][0];
                int& s1 = bs[K][1];
                bool ins = a < d1;
                bool first = a < d0;
                // exact ties: list order decides (hddstream.py:326/373, strict `<`)
                const unsigned long long e1 = __builtin_amdgcn_ballot_w64(a == d1);
                const unsigned long long e0 = __builtin_amdgcn_ballot_w64(a == d0);
                if ((e1 | e0) != 0ull) {
                    if (a == d1 || a == d0) {
                        const int key = g_key[rowg];
                        if (a == d1) ins = key < (s1 >= 0 ? g_key[s1] : CC_IDX_INF);
                        if (a == d0) first = key < (s0 >= 0 ? g_key[s0] : CC_IDX_INF);
                    }
                }
                d1 = cc_vmin(d1, cc_vmax(d0, a));
                d0 = cc_vmin(d0, a);
                s1 = first ? s0 : (ins ? rowg : s1);
                s0 = first ? rowg : s0;
            };
            if (is_p) update(std::integral_constant<int, 0>{});
            else update(std::integral_constant<int, 1>{});
        }
    }
    // rows abandoned in phase A: their exact partial sums exceed every lane's T (k_seed_merge), which is all that is
    // recorded of them
    if (dropped[0]) lb[0] = cc_vmin(lb[0], th[0]);
    if (dropped[1]) lb[1] = cc_vmin(lb[1], th[1]);
    // statistics for the host's policy: a sample - the waves of the window's first point tile (atomics of every wave on
    // one address serialise: 30 000 of them cost more than the scan)
    if (lane == 0 && blockIdx.x == 0 && n_rows > 0) {
        atomicAdd(&ctl->stat_prune_rows, (unsigned long long)n_rows);
        atomicAdd(&ctl->stat_prune_full, (unsigned long long)n_full);
    }

    // the survivors' list-order keys; every kind's pair then takes in the bound of what the wave abandoned; the waves'
    // pairs are merged through LDS as in k_scan_u
    int bk[2][2];
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
#pragma unroll
        for (int r = 0; r < 2; ++r) bk[kd][r] = bs[kd][r] >= 0 ? g_key[bs[kd][r]] : CC_IDX_INF;
    Cand c0{bd[0][0], bk[0][0], bs[0][0]}, c1{bd[0][1], bk[0][1], bs[0][1]};
    Cand c2{bd[1][0], bk[1][0], bs[1][0]}, c3{bd[1][1], bk[1][1], bs[1][1]};
    cc_top2_push(c0, c1, Cand{lb[0], -1, (valid && lb[0] < CC_INF) ? CC_SLOT_BOUND : -1});
    cc_top2_push(c2, c3, Cand{lb[1], -1, (valid && lb[1] < CC_INF) ? CC_SLOT_BOUND : -1});
    Cand* s_m = reinterpret_cast<Cand*>(smem);
    auto s_m_at = [&](int w, int c) -> Cand& { return s_m[(w * 4 + c) * 64 + lane]; };
    __syncthreads();  // every wave is done with its tile: the same bytes now carry the candidate exchange
    if (wv > 0) {
        s_m_at(wv - 1, 0) = c0;
        s_m_at(wv - 1, 1) = c1;
        s_m_at(wv - 1, 2) = c2;
        s_m_at(wv - 1, 3) = c3;
    }
    __syncthreads();
    if (wv != 0 || !valid) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
        cc_top2_push(c0, c1, s_m_at(w, 0));
        cc_top2_push(c0, c1, s_m_at(w, 1));
        cc_top2_push(c2, c3, s_m_at(w, 2));
        cc_top2_push(c2, c3, s_m_at(w, 3));
    }
    Cand* o = part + ((size_t)jj * S + blockIdx.y) * 4;
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// ---------------------------------------------------------------------------------
// k_merge_partials (exact multi-GPU path): the S partials a rank's snapshot scan left per window point -> ONE record
// of four candidates per point, the unit the ranks all-gather (64 B per point instead of S x 64 B).  Candidates are
// totally ordered by (distance, list-order key), so the best two of a union do not depend on the merge order and
// every rank derives the same lists from the gathered records.  One thread per point; `round` / `mode` select the
// window exactly as in k_scan.  Always recomputed from the scan's partials (idempotent), also when the in-place scan
// it follows found that the window had been scanned ahead.
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_merge_partials(const Ctl* __restrict__ ctl, const Cand* __restrict__ part,
                                                        size_t part_stride, int S, Cand* __restrict__ out,
                                                        size_t out_stride, int round, int mode)
{
    int B, q;
    if (mode == 1) {
        q = round & 1;
        B = ctl->la_b[q];
    } else {
        q = (int)(ctl->window_seq & 1ull);
        B = ctl->win_b;
    }
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B) return;
    part += (size_t)q * part_stride;
    out += (size_t)q * out_stride;
    const Cand none = Cand{CC_INF, CC_IDX_INF, -1};
    Cand p1 = none, p2 = none, o1 = none, o2 = none;
    for (int s = 0; s < S; ++s) {
        const Cand* c = part + ((size_t)j * S + s) * 4;
        cc_top2_push(p1, p2, c[0]);
        cc_top2_push(p1, p2, c[1]);
        cc_top2_push(o1, o2, c[2]);
        cc_top2_push(o1, o2, c[3]);
    }
    Cand* o = out + (size_t)j * 4;
    o[0] = p1; o[1] = p2; o[2] = o1; o[3] = o2;
}

// ---------------------------------------------------------------------------------
// 32-lane groups: one group per window point in k_decide / k_chain.  Lane l owns dimensions l and l + 32
// (d <= 64); sums over dimensions stay strictly left to right through an ordered shuffle loop.
// ---------------------------------------------------------------------------------

__device__ __forceinline__ unsigned cc_group_ballot(bool p)
{
    const unsigned long long b = __builtin_amdgcn_ballot_w64(p);
    return (unsigned)(b >> (threadIdx.x & 32));
}

// Sum over the 32 lanes of a group whose order does not matter (it feeds conservative bounds only), result valid in
// lanes 0..15 of the group: four DPP steps inside the rows of 16 lanes, then the other row's total (the shuffle
// butterfly goes through the LDS crossbar five times, one latency each)
template <int CTRL>
__device__ __forceinline__ double cc_dpp_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cc_group_sum_any_order(double x)
{
    x += cc_dpp_f64<0xB1>(x);   // quad_perm [1,0,3,2]
    x += cc_dpp_f64<0x4E>(x);   // quad_perm [2,3,0,1]
    x += cc_dpp_f64<0x141>(x);  // row_half_mirror
    x += cc_dpp_f64<0x140>(x);  // row_mirror: every lane of a row of 16 holds the row's sum
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const double r16 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r48 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return x + ((threadIdx.x & 32) ? r48 : r16);
}

struct GroupAdd {
    double c1[2], c2[2], pr[2];  // this lane's two dimensions of (base + point): CF1, CF2, preferred-dimension entry
    double cen[2];               // ... and CF1 / W, the centroid (mc_functions.py:31-33; the same quotient the variance uses)
    double r2;                   // projected radius^2 of the enlarged MC (all lanes)
    int gt1, ne1;                // count(pref' > 1), count(pref' != 1)
};

// microcluster.py:213-233 + mc_functions.py:45-56, computed by the 32 lanes of a group together.
// Every lane of the group must call it with the same bw / d; b1, b2, px are this lane's two dimensions of the
// base CF1, CF2 and of the point.
__device__ inline GroupAdd cc_group_add_regs(const double (&b1)[2], const double (&b2)[2], double bw,
                                             const double (&px)[2], int d, const Par& c)
{
    const int gl = threadIdx.x & 31;
    GroupAdd g;
    const double w1 = bw + 1.0;
    double term[2];
    bool gt[2], ne[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = gl + 32 * h;
        g.c1[h] = 0.0; g.c2[h] = 0.0; g.pr[h] = 1.0; g.cen[h] = 0.0;
        term[h] = 0.0; gt[h] = false; ne[h] = false;
        if (i < d) {
            const double x = px[h];
            g.c1[h] = b1[h] + x;
            g.c2[h] = b2[h] + x * x;
            // mc_functions.py:14-22 (cc_sqvar), keeping the quotient CF1 / W
            const double qa = g.c2[h] / w1;
            const double qb = g.c1[h] / w1;
            g.cen[h] = qb;
            const double var = qa - qb * qb;
            const double pr = (var <= c.delta_sq) ? c.k : 1.0;
            g.pr[h] = pr;
            term[h] = cc_div_pref(var, pr, c);
            gt[h] = pr > 1.0;
            ne[h] = pr != 1.0;
        }
    }
    g.gt1 = __builtin_popcount(cc_group_ballot(gt[0])) + __builtin_popcount(cc_group_ballot(gt[1]));
    g.ne1 = __builtin_popcount(cc_group_ballot(ne[0])) + __builtin_popcount(cc_group_ballot(ne[1]));
    // ordered sum over dimensions: the terms go through LDS (one 64-double row per group) and every lane adds
    // them left to right from broadcast reads; the wave owns its rows, so a wavefront fence is enough
    __shared__ double s_term[(CC_GROUP_THREADS / 32)][64];
    double* const row = s_term[(threadIdx.x >> 5) % (CC_GROUP_THREADS / 32)];
    CC_WAVE_SYNC();
    row[gl] = term[0];
    row[gl + 32] = term[1];
    CC_WAVE_SYNC();
    double r2 = 0.0;
    // mc_functions.py:54, left to right; entries d..63 of the row hold +0.0 (x + 0.0 == x), so the loop runs over
    // whole groups of eight (four 16-byte LDS reads in flight) without a one-by-one remainder
    const int d8 = (d + 7) & ~7;
    for (int i = 0; i < d8; ++i) r2 = r2 + row[i];
    g.r2 = r2;
    return g;
}

// the same from memory: bcf1 == nullptr means an empty base
__device__ inline GroupAdd cc_group_add(const double* bcf1, const double* bcf2, double bw, const double* p, int d,
                                        const Par& c)
{
    const int gl = threadIdx.x & 31;
    double b1[2] = {0.0, 0.0}, b2[2] = {0.0, 0.0}, px[2] = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = gl + 32 * h;
        if (i < d) {
            px[h] = p[i];
            if (bcf1) { b1[h] = bcf1[i]; b2[h] = bcf2[i]; }
        }
    }
    return cc_group_add_regs(b1, b2, bw, px, d, c);
}

// The same candidate from another lane of this lane's row of 16 (DPP: no trip through the LDS crossbar, which the
// co-running snapshot scan keeps busy)
template <int CTRL>
__device__ __forceinline__ Cand cc_dpp_cand(const Cand& c)
{
    Cand o;
    o.dist = cc_dpp_f64<CTRL>(c.dist);
    o.key = __builtin_amdgcn_update_dpp(0, c.key, CTRL, 0xF, 0xF, false);
    o.slot = __builtin_amdgcn_update_dpp(0, c.slot, CTRL, 0xF, 0xF, false);
    return o;
}
// All-to-all merge inside every row of 16 lanes in four exchanges with disjoint holdings: neighbours, pairs of a
// quad, the two quads of a half row (half mirror), the two half rows (mirror).  `f(ctrl_constant)` does one exchange.
template <typename F>
__device__ __forceinline__ void cc_row16_exchanges(F&& f)
{
    f(std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    f(std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    f(std::integral_constant<int, 0x141>{});  // row_half_mirror
    f(std::integral_constant<int, 0x140>{});  // row_mirror
}

// ---------------------------------------------------------------------------------
// k_dseed: per window point and kind, the cap and the first candidate of the dirty scan (one thread per point).
// A live version only matters to point j if it beats what j already has.  If j's best snapshot candidate c1 is
// still untouched when j arrives, that is c1 itself (cap = d1).  If c1 was touched, the live version of c1's MC
// is itself a candidate: find it (member list, then walk or backward read of the claims), take its exact distance
// as the first candidate;
// everything else has to beat that.  Loose fallback: the snapshot's second-best distance d2.
// seed[j*4 + kd*2] = first candidate (slot -1: none), seed[j*4 + kd*2 + 1].dist = cap.
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_dseed(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                              Versions ver, Carry car, const Cand* __restrict__ clean,
                                              Cand* __restrict__ seed, const int* __restrict__ T, int round)
{
    CC_LATENCY_KERNEL();
    const int B = ctl->win_b;
    if (B == 0) return;
    if (ctl->fc[round - 1] >= B) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= B) return;
    const bool la_mode = ctl->mode != 0;
    // largest displacement of any version row / carried row (the workgroup is one wave)
    // (per kind, in cc_dsq_code form: 0 = the window / the carry set holds no row of the kind)
    unsigned long long maxd[2] = {0ull, 0ull}, maxd_car[2] = {0ull, 0ull};
    {
        for (int i = threadIdx.x; i < 2 * ((B + 15) / 16); i += 64) {
            const unsigned long long v = ver.tile_dsq[i];
            if (i & 1) maxd[1] = v > maxd[1] ? v : maxd[1];
            else maxd[0] = v > maxd[0] ? v : maxd[0];
        }
        if (la_mode)
            for (int i = threadIdx.x; i < 2 * ((ctl->car_n + 15) / 16); i += 64) {
                const unsigned long long v = car.tile_dsq[i];
                if (i & 1) maxd_car[1] = v > maxd_car[1] ? v : maxd_car[1];
                else maxd_car[0] = v > maxd_car[0] ? v : maxd_car[0];
            }
#pragma unroll
        for (int K = 0; K < 2; ++K)
            for (int off = 32; off >= 1; off >>= 1) {
                const unsigned long long o = __shfl_xor(maxd[K], off), oc = __shfl_xor(maxd_car[K], off);
                maxd[K] = o > maxd[K] ? o : maxd[K];
                maxd_car[K] = oc > maxd_car[K] ? oc : maxd_car[K];
            }
    }
    double tau_out[2] = {CC_INF, CC_INF};  // lanes past the window do not constrain the tile
    bool flag_unprov = false, flag_unsafe = false;
    if (j < B) {
    const Par par = cc_load_par(ctl);
    const int d = par.d;
    const bool filter = par.filter != 0;
    const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
    const double* p = X + (ctl->cursor + j) * d;
    Cand first0 = Cand{CC_INF, CC_IDX_INF, -1}, first1 = Cand{CC_INF, CC_IDX_INF, -1};
    double cap[2] = {CC_INF, CC_INF};
    bool provable = par.k > 0.0;  // false: some live version of a list MC could not be located -> no pruning
    // Ratio of a dimension's weight before / after, for the rows the threshold below is applied to: 1.  A version whose
    // preferred dimensions differ from its MC's at window start carries no bound (k_chain, k_chain_long, k_commit_b give
    // it dsq = +inf like a new or promoted MC), so every bounded row has the window-start metric itself and the bound is
    // the plain triangle inequality.  (With the worst-case ratio k instead, a list whose MCs are all far from the point -
    // second-best < k x best: every noise point, every point of a stream with a few stale outlier MCs - could never be
    // pruned, and one such point keeps its whole tile's dirty scans running.)
    const double K = 1.0;

    const unsigned long long wseq = ctl->window_seq;
    // Four lookups per point - best and second-best snapshot candidate of either kind -, each a chain of dependent
    // loads (validation stamp -> member list -> version row -> its centroid).  One thread per point: the four chains
    // advance in lock step, so that every step's loads are in flight together.
    Cand cq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cq[q] = clean[(size_t)j * 4 + q];
    double d2v[2] = {CC_INF, CC_INF};
    bool have1[2];
    bool look[4];
#pragma unroll
    for (int kd = 0; kd < 2; ++kd) {
        // (a bound in second place - pruned snapshot scan - serves as d2 like an exact second-best distance: what is
        // needed of d2 below is that no MC outside the list was closer than it at window start)
        if (cq[kd * 2 + 1].slot != -1) d2v[kd] = cq[kd * 2 + 1].dist;
        have1[kd] = cq[kd * 2].slot >= 0;  // no snapshot candidate of this kind: cap stays +inf
        if (cq[kd * 2].slot == CC_SLOT_BOUND) provable = false;  // (never left in first place; k_decide refuses the point)
        look[kd * 2] = have1[kd];
        look[kd * 2 + 1] = have1[kd] && cq[kd * 2 + 1].slot >= 0;
    }
    // step 1: validation stamp, member count, carry mark of the four MCs
    unsigned long long tcq[4], cwq[4], coq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t sl = (size_t)(look[q] ? cq[q].slot : 0);
        tcq[q] = look[q] ? tab.touch[(size_t)(round & 1) * tab.cap + sl] : 0ull;
        cwq[q] = look[q] ? tab.cnt[sl] : 0ull;
        coq[q] = (look[q] && la_mode) ? tab.carry_of[sl] : 0ull;
    }
    // the live version of the MC when point j arrives (-1: untouched so far, -2: not found,
    // >= CC_CAR_BASE: the carried row - the previous window changed the MC after this window's snapshot scan)
    int lv[4], n_memb[4];
    bool walk[4];
    int max_list = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int head = 0xFFFFF - (int)(tcq[q] & 0xFFFFFull);
        lv[q] = -1;
        walk[q] = false;
        n_memb[q] = 0;
        if (!look[q]) continue;
        if ((tcq[q] >> 20) != stamp || head >= j) {
            if (la_mode && (coq[q] >> 20) == wseq) lv[q] = CC_CAR_BASE + (int)(coq[q] & 0xFFFFFull);
        } else {
            // the latest claimant before j: the largest listed member of the MC's chain below j (k_decide listed up
            // to CC_CHAIN_MEMB of them), then along the chain for the members the list does not hold
            lv[q] = head;
            walk[q] = true;
            n_memb[q] = ((cwq[q] >> 24) == stamp) ? (int)(cwq[q] & 0xFFFFFFull) : 0;
            const int n_list = n_memb[q] < CC_CHAIN_MEMB ? n_memb[q] : CC_CHAIN_MEMB;
            max_list = n_list > max_list ? n_list : max_list;
        }
    }
    // step 2: the member lists, four entries of each list per pass
    for (int pos = 0; pos * 4 < max_list; ++pos) {
        int4 mm[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool on = walk[q] && pos * 4 < n_memb[q];
            mm[q] = on ? reinterpret_cast<const int4*>(tab.memb + (size_t)cq[q].slot * CC_CHAIN_MEMB)[pos] : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n_list = n_memb[q] < CC_CHAIN_MEMB ? n_memb[q] : CC_CHAIN_MEMB;
            const int e[4] = {mm[q].x, mm[q].y, mm[q].z, mm[q].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int m = e[c];
                if (walk[q] && pos * 4 + c < n_list && m < j && m > lv[q]) lv[q] = m;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (walk[q] && n_memb[q] > CC_CHAIN_MEMB) {
            // More members than the list holds (the counter stops there; k_chain left the chain's length in clen).
            // Members far apart: along the chain from the latest listed member before j, one dependent load per member.  Close
            // together (a long chain): backwards through the claims (T, the ones k_chain replayed) from j - 1, 32
            // independent loads per pass; neighbouring threads read overlapping ranges.
            const int clen = tab.clen[cq[q].slot];
            const int n = clen > CC_CHAIN_MEMB ? clen : CC_CHAIN_MEMB + 1;
            const int gap = B / n;  // members of this chain lie about this many claims apart
            // The listed members are mostly the chain's earliest (k_decide's workgroups start in point order), so
            // the walk from the latest listed one takes up to n - 32 dependent steps, a backward read about gap / 32
            // passes: whichever is expected to be shorter goes first, the other one is the fallback.
            const bool scan_first = (n - CC_CHAIN_MEMB) > gap / 8;  // (a pass of 32 loads costs about four steps)
            int v = lv[q];
            int res = -2;
            for (int attempt = 0; attempt < 2 && res == -2; ++attempt) {
                if ((attempt == 0) == scan_first) {
                    const int want = cq[q].slot;
                    // (members are spread like arrivals: a distance of 16 gaps is exceeded once in 10^7 lookups)
                    const int budget = min(16384, 16 * gap + 64);
                    for (int hi = j - 1, scanned = 0; res == -2 && scanned < budget; hi -= 32, scanned += 32) {
                        int tv[32];
#pragma unroll
                        for (int c = 0; c < 32; ++c) tv[c] = (hi - c > v) ? T[hi - c] : CC_T_UNKNOWN;
                        int hit = -1;
#pragma unroll
                        for (int c = 31; c >= 0; --c)
                            if (tv[c] == want) hit = hi - c;  // (ends on the smallest c = the largest index)
                        if (hit < 0 && hi - 32 <= v) hit = v;  // nothing between v and j: v is the latest
                        if (hit >= 0) res = hit;
                    }
                } else {
                    int w = v;
                    for (int steps = 0; steps < 64; ++steps) {
                        const int nxv = ver.next[w];
                        if (nxv >= j) { res = w; break; }
                        w = nxv;
                    }
                }
            }
            lv[q] = res;
        }
    // what the lookups mean for the caps (hddstream.py:326/373 via the candidate lists)
#pragma unroll
    for (int kd = 0; kd < 2; ++kd) {
        if (!have1[kd]) continue;
        const int v1 = lv[kd * 2];
        if (v1 == -1) cap[kd] = cq[kd * 2].dist;  // c1 is clean at j: a live version has to beat c1 itself
        else {
            cap[kd] = d2v[kd];
            if (v1 < 0) provable = false;
        }
        if (look[kd * 2 + 1] && lv[kd * 2 + 1] == -2) provable = false;
    }
    // step 3: kind and key of the (up to four) version rows; step 4: their exact distances to point j, four
    // dimensions of all rows per pass, every sum left to right
    bool sd[4];
    int kvq[4], keyq[4];
    const double* vcen[4];
    const double* vpref[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sd[q] = look[q] && lv[q] >= 0;
        const bool cr = sd[q] && lv[q] >= CC_CAR_BASE;
        const size_t r = sd[q] ? (size_t)(cr ? lv[q] - CC_CAR_BASE : lv[q]) : 0;
        kvq[q] = sd[q] ? (cr ? car.kind[r] : ver.kind[r]) : CC_KIND_DEAD;
        keyq[q] = sd[q] ? (cr ? car.key[r] : ver.key[r]) : 0;
        vcen[q] = (cr ? car.cen : ver.cen) + r * d;
        vpref[q] = (cr ? car.pref : ver.pref) + r * d;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sd[q] = sd[q] && kvq[q] != CC_KIND_DEAD;
    double accq[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = 0; i0 < d; i0 += 4) {
        double pv[4], cv[4][4], fv[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = (i0 + c < d) ? i0 + c : d - 1;
            pv[c] = p[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cv[q][c] = sd[q] ? vcen[q][i] : 0.0;
                fv[q][c] = sd[q] ? vpref[q][i] : 1.0;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (i0 + c < d) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double x = pv[c] - cv[q][c];
                    x = x * x;
                    accq[q] = accq[q] + cc_div_pref(x, fv[q][c], par);
                }
            }
        }
    }
    // the seeds enter the first-candidate slot of their version's kind, in lookup order
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!sd[q]) continue;
        if (kvq[q] == 0 && filter) {
            const bool cr = lv[q] >= CC_CAR_BASE;
            const size_t r = (size_t)(cr ? lv[q] - CC_CAR_BASE : lv[q]);
            int ne1 = 0;
            cc_tentative_radius((cr ? car.cf1 : ver.cf1) + r * d, (cr ? car.cf2 : ver.cf2) + r * d,
                                cr ? car.w[r] : ver.w[r], p, d, par, nullptr, &ne1);
            if (ne1 > par.pi) continue;
        }
        if (kvq[q] == 0) {
            if (cand_less(accq[q], keyq[q], first0.dist, first0.key)) first0 = Cand{accq[q], keyq[q], lv[q]};
        } else {
            if (cand_less(accq[q], keyq[q], first1.dist, first1.key)) first1 = Cand{accq[q], keyq[q], lv[q]};
        }
    }
    seed[(size_t)j * 4 + 0] = first0;
    seed[(size_t)j * 4 + 1] = Cand{cap[0], 0, 0};
    seed[(size_t)j * 4 + 2] = first1;
    seed[(size_t)j * 4 + 3] = Cand{cap[1], 0, 0};

    // Pruning threshold.  Let v be a live version of a MC s that is in neither list of its kind for this point and
    // was of that kind at window start: its window-start distance is >= d2.  Weighted norms obey the triangle
    // inequality and a dimension's weight changes by at most the factor K, so
    //     dist_v >= (sqrt(d2) - sqrt(dsq_v))^2 / K,
    // and v cannot beat `cap` (or the seeded candidate, whichever is smaller) when
    //     sqrt(dsq_v) < sqrt(d2) - sqrt(K * cap).
    // Live versions of the list MCs themselves are seeded above.  With the pdim filter on, pcore MCs outside the
    // list may be closer than d2 (they were filtered out), so nothing is pruned for that kind.
    // One threshold per kind: a version competes in the list of its (current) kind.  A kind without any version row in
    // the window constrains nothing - stale outlier MCs that no point touches must not cost anything.
    bool ok_v = true, ok_c = true;
    for (int kd = 0; kd < 2; ++kd) {
        double t;
        if (kd == 0 && filter) t = -CC_INF;
        else if (!have1[kd]) t = CC_INF;  // no MC of this kind at window start: its versions have dsq = +inf
        else {
            const double fb = (kd == 0) ? first0.dist : first1.dist;
            const double ce = fb < cap[kd] ? fb : cap[kd];
            t = (d2v[kd] == CC_INF) ? CC_INF : (sqrt(d2v[kd]) - sqrt(K * ce));
        }
        if (!provable) t = -CC_INF;
        t = (t == CC_INF) ? CC_INF : t * (1.0 - 1e-9) - 1e-290;  // margin for the rounding of all of the above
        tau_out[kd] = t;
        ver.tau[(size_t)j * 2 + kd] = t;
        // the same test as for the tile below, for this point alone
        ok_v = ok_v && cc_dsq_below(maxd[kd], t);
        ok_c = ok_c && (!la_mode || cc_dsq_below(maxd_car[kd], t));
    }
    ver.unsafe[j] = (ok_v && ok_c) ? 0 : 1;
    flag_unprov = !provable;
    flag_unsafe = !(ok_v && ok_c);
    }
    {
        // statistics for the host's trace line (one atomic per wave and only when something is flagged)
        const unsigned long long b1 = __builtin_amdgcn_ballot_w64(flag_unprov), b2 = __builtin_amdgcn_ballot_w64(flag_unsafe);
        if (threadIdx.x == 0 && b1) atomicAdd((unsigned long long*)&ctl->stat_unprovable, (unsigned long long)__builtin_popcountll(b1));
        if (threadIdx.x == 0 && b2) atomicAdd((unsigned long long*)&ctl->stat_unsafe, (unsigned long long)__builtin_popcountll(b2));
    }
    // the tile as a whole: when even the largest displacement stays below every point's threshold, no row can matter
    // to any point of the tile and its dirty scan is not run at all (the same test k_scan makes per 16 rows)
    double tile_tau[2] = {tau_out[0], tau_out[1]};
#pragma unroll
    for (int K = 0; K < 2; ++K)
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(tile_tau[K], off);
            tile_tau[K] = o < tile_tau[K] ? o : tile_tau[K];
        }
    if (threadIdx.x == 0) {
        const int sk = (cc_dsq_below(maxd[0], tile_tau[0]) && cc_dsq_below(maxd[1], tile_tau[1])) ? 1 : 0;
        ver.skip[blockIdx.x] = sk;  // (k_commit_a counts the tiles of the last round for the host's window policy)
        ver.skip_car[blockIdx.x] = (!la_mode || (cc_dsq_below(maxd_car[0], tile_tau[0]) && cc_dsq_below(maxd_car[1], tile_tau[1]))) ? 1 : 0;
    }
}

// Scan copies of the table (lookahead).  Lookahead scans do not read the table but one of two copies of the columns a
// scan needs, so that a commit never waits for a scan that is still reading.  The copy with the parity of window W
// is read by the snapshot scan of W (which saw the table two commits earlier... one commit before W - 1's) and is
// brought up to date during W's validation: first the rows the previous commit changed (its carry set,
// cc_apply_carry), then the rows W's own commit changes (k_commit_b).  It is next read by the scan of W + 2.
struct ScanCopy {
    double* cen;
    double* scl;
    double* cf1;  // cf1, cf2, w: read by the pdim filter only
    double* cf2;
    double* w;
    int* kind;
    int* key;
};

struct CommitRec {
    int n;        // validated prefix length
    int M0;       // table rows at window start
    int pk0, ok0; // list-order key bases
    long long pid0, oid0;
    const int* T; // the claims the prefix was validated against
    long long cursor;  // first point of the window in the call's input
    int carry;    // 1: the next window is a lookahead window -> k_commit_b also writes the carry set
    unsigned long long next_seq;  // its window_seq
};

// The rows the previous commit changed (its carry set) into the scan copy of this window's parity.  Runs as extra
// workgroups of k_decide's round-0 launch (`part` of `parts`): nothing it writes is read by k_decide, and a launch of its
// own would cost the validation chain one more kernel slot per window.
__device__ __forceinline__ void cc_apply_carry(const CommitRec* __restrict__ rec, const Carry& car, const ScanCopy& sc, int d,
                                               int filter, int part, int parts)
{
    if (rec->carry == 0) return;
    const int n = rec->n;
    const int gl = threadIdx.x & 31;
    const int groups = (parts * (int)blockDim.x) >> 5;
    for (int j = (part * (int)blockDim.x + (int)threadIdx.x) >> 5; j < n; j += groups) {
        const int kind = car.kind[j];
        if (kind == CC_KIND_DEAD) continue;  // a later point holds the last version of this MC
        const size_t row = (size_t)car.slot[j];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = gl + 32 * h;
            if (i >= d) continue;
            const size_t e = row * d + i, v = (size_t)j * d + i;
            sc.cen[e] = car.cen[v];
            sc.scl[e] = car.scl[v];
            if (filter) { sc.cf1[e] = car.cf1[v]; sc.cf2[e] = car.cf2[v]; }
        }
        if (gl == 0) {
            sc.kind[row] = kind;
            sc.key[row] = car.key[j];
            if (filter) sc.w[row] = car.w[j];
        }
    }
}

// ---------------------------------------------------------------------------------
// k_decide: one 32-lane group per window point.  Segment partials are merged inside each row of 16 lanes with DPP
// exchanges (per-point argmin over the MC range), then the reference's decision procedure runs group-uniformly.
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_decide(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                                Versions ver, Carry car, const Cand* __restrict__ part,
                                                size_t part_stride, Cand* __restrict__ clean,
                                                const Cand* __restrict__ dpart, const Cand* __restrict__ dpart2,
                                                const Cand* __restrict__ dseed,
                                                const int* __restrict__ Told, int* __restrict__ Tnew,
                                                int8_t* __restrict__ dpath, int S, int Sd, int round, int nodirty,
                                                int scan_rows, int part_inner, size_t part_outer,
                                                const CommitRec* __restrict__ ac_rec, ScanCopy ac_sc, int ac_blocks,
                                                int* __restrict__ long_list)
{
    CC_LATENCY_KERNEL();
    // the last ac_blocks workgroups of a round-0 launch before a lookahead window's validation: cc_apply_carry
    if (ac_blocks > 0 && (int)blockIdx.x >= (int)gridDim.x - ac_blocks) {
        cc_apply_carry(ac_rec, car, ac_sc, ctl->d, ctl->filter, (int)blockIdx.x - ((int)gridDim.x - ac_blocks), ac_blocks);
        return;
    }
    const int B = ctl->win_b;
    if (B == 0) return;
    if (round > 0 && ctl->fc[round - 1] >= B) return;
    const int gl = threadIdx.x & 31;
    const int j = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (j >= B) return;
    const Par par = cc_load_par(ctl);
    const int d = par.d;
    const Cand none = Cand{CC_INF, CC_IDX_INF, -1};

    // candidates are kept in named scalars (no runtime-indexed arrays: those would live in scratch memory)
    Cand p1 = none, p2 = none, o1 = none, o2 = none;  // best two pcore / outlier snapshot candidates
    const unsigned long long wseq = ctl->window_seq;
    const bool la_mode = ctl->mode != 0;
    if (round == 0) {
        part += (size_t)(wseq & 1ull) * part_stride;  // the snapshot scan of this window wrote the copy of its parity
        // partial s of point j: one launch wrote S partials per point (part_inner = S, part_outer unused); on the
        // exact multi-GPU path every rank contributed one merged record (part_inner = 1, part_outer = the distance
        // between the ranks' blocks in the gathered buffer)
        // both rows of 16 lanes of the group merge all S partials (lane l of a row takes l, l + 16, ..), so the two
        // rows end with the same result and nothing crosses between them
        for (int s = gl & 15; s < S; s += 16) {
            const Cand* q = part + (size_t)(s / part_inner) * part_outer + ((size_t)j * part_inner + (s % part_inner)) * 4;
            cc_top2_push(p1, p2, q[0]);
            cc_top2_push(p1, p2, q[1]);
            cc_top2_push(o1, o2, q[2]);
            cc_top2_push(o1, o2, q[3]);
        }
        cc_row16_exchanges([&](auto CT) {
            constexpr int C = decltype(CT)::value;
            const Cand a0 = cc_dpp_cand<C>(p1), a1 = cc_dpp_cand<C>(p2);
            const Cand a2 = cc_dpp_cand<C>(o1), a3 = cc_dpp_cand<C>(o2);
            cc_top2_push(p1, p2, a0);
            cc_top2_push(p1, p2, a1);
            cc_top2_push(o1, o2, a2);
            cc_top2_push(o1, o2, a3);
        });
        if (gl == 0) {
            Cand* out = clean + (size_t)j * 4;
            out[0] = p1; out[1] = p2; out[2] = o1; out[3] = o2;
        }
    } else {
        const Cand* in = clean + (size_t)j * 4;
        p1 = in[0]; p2 = in[1]; o1 = in[2]; o2 = in[3];
    }
    Cand dvp = none, dvo = none;  // best live version per kind
    if (round > 0) {
        Cand dummy = none;
        // a dirty scan that k_dseed ruled out for this point's tile was not run: the seeds are its whole result
        // (nodirty: the host did not launch the dirty scans at all; points that would have needed them are refused below)
        const bool ran = nodirty == 0 && ver.skip[j >> 6] == 0;
        const bool ran_car = nodirty == 0 && la_mode && ver.skip_car[j >> 6] == 0;
        if (!ran && !ran_car) {
            // no dirty scan ran for this point's tile (the steady state): the seeds are the whole result, every lane
            // reads them itself and nothing has to be merged
            dvp = dseed[(size_t)j * 4 + 0];
            dvo = dseed[(size_t)j * 4 + 2];
        } else {
        if ((gl & 15) == 0 && !(ran && (ran_car || !la_mode))) {
            dvp = dseed[(size_t)j * 4 + 0];
            dvo = dseed[(size_t)j * 4 + 2];
        }
        for (int s = gl & 15; s < Sd; s += 16) {  // (per row of 16 lanes, as in round 0)
            if (ran) {
                const Cand* q = dpart + ((size_t)j * Sd + s) * 2;
                cc_top2_push(dvp, dummy, q[0]);
                cc_top2_push(dvo, dummy, q[1]);
            }
            if (ran_car) {  // the carry set is scanned separately
                const Cand* q2 = dpart2 + ((size_t)j * Sd + s) * 2;
                cc_top2_push(dvp, dummy, q2[0]);
                cc_top2_push(dvo, dummy, q2[1]);
            }
        }
        cc_row16_exchanges([&](auto CT) {
            constexpr int C = decltype(CT)::value;
            const Cand b0 = cc_dpp_cand<C>(dvp), b1 = cc_dpp_cand<C>(dvo);
            cc_top2_push(dvp, dummy, b0);
            cc_top2_push(dvo, dummy, b1);
        });
        }
    }

    const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
    // the snapshot distance of `slot` no longer describes the MC: a point before j targets it, or (lookahead) the
    // previous window changed it after the snapshot was scanned
    auto dirty = [&](int slot) -> bool {
        if (round == 0) return false;
        const unsigned long long t = tab.touch[(size_t)(round & 1) * tab.cap + slot];
        if ((t >> 20) == stamp && (0xFFFFF - (int)(t & 0xFFFFFull)) < j) return true;
        return la_mode && (tab.carry_of[slot] >> 20) == wseq;
    };

    const int M0 = ctl->m_rows;
    const double* p = X + (ctl->cursor + j) * d;
    int T = -1;
    int path = 2;
    // one stage of the reference's procedure: stage 0 = _add_to_pcore (hddstream.py:288-343), 1 = _add_to_outlier
    auto run_stage = [&](const Cand& c1, const Cand& c2, const Cand& dd, int stage) {
        int state;  // 0: no clean candidate, 1: cb is the exact clean best, 2: cb only bounds the clean best from below
        Cand cb = none;
        if (c1.slot == -1) state = 0;
        else if (c1.slot == CC_SLOT_BOUND) { T = CC_T_UNKNOWN; return; }  // (a pruned scan never leaves this)
        else if (!dirty(c1.slot)) { state = 1; cb = c1; }
        else if (c2.slot == -1) state = 0;
        else if (c2.slot == CC_SLOT_BOUND) { state = 2; cb = c2; }  // the clean rows are only known to be >= c2.dist
        else if (!dirty(c2.slot)) { state = 1; cb = c2; }
        else { state = 2; cb = c2; }

        int wkind = 0;  // 0 none, 1 table row, 2 version row
        int wrow = -1;
        if (state == 0) {
            if (dd.slot >= 0) { wkind = 2; wrow = dd.slot; }
        } else if (state == 1) {
            if (dd.slot >= 0 && cand_less(dd.dist, dd.key, cb.dist, cb.key)) { wkind = 2; wrow = dd.slot; }
            else { wkind = 1; wrow = cb.slot; }
        } else {
            if (dd.slot >= 0 && cand_less(dd.dist, dd.key, cb.dist, cb.key)) { wkind = 2; wrow = dd.slot; }
            else { T = CC_T_UNKNOWN; return; }
        }
        if (wkind == 0) return;
        const double *bcf1, *bcf2;
        double bw;
        int target;
        if (wkind == 1) {
            bcf1 = tab.cf1 + (size_t)wrow * d; bcf2 = tab.cf2 + (size_t)wrow * d; bw = tab.w[wrow];
            target = wrow;
        } else if (wrow >= CC_CAR_BASE) {
            const size_t r = (size_t)(wrow - CC_CAR_BASE);
            bcf1 = car.cf1 + r * d; bcf2 = car.cf2 + r * d; bw = car.w[r];
            target = car.slot[r];
        } else {
            bcf1 = ver.cf1 + (size_t)wrow * d; bcf2 = ver.cf2 + (size_t)wrow * d; bw = ver.w[wrow];
            target = ver.tgt[wrow];
        }
        const GroupAdd g = cc_group_add(bcf1, bcf2, bw, p, d, par);  // hddstream.py:334-337
        if (g.r2 <= par.eps_sq) {
            T = target;
            path = stage;
        }
    };
    run_stage(p1, p2, dvp, 0);
    if (T == -1) run_stage(o1, o2, dvo, 1);
    if (T == -1) {
        if (ctl->no_create != 0) {
            T = CC_T_NONE;  // relaxed multi-GPU mode: set aside for the replicated second half of the super-step
            path = 8;
        } else {  // hddstream.py:434-462: new outlier MC, provisional id = rows-at-window-start + j
            T = M0 + j;
            path = 2;
            if (gl == 0) ctl->any_new[round] = 1;
        }
    }
    if (round > 0 && nodirty != 0 && ver.unsafe[j] != 0) T = CC_T_UNKNOWN;  // the seeds are not this point's whole story
    if (gl == 0) {
        Tnew[j] = T;
        dpath[j] = (int8_t)path;
        if (round > 0 && (T == CC_T_UNKNOWN || T != Told[j])) atomicMin(&ctl->fc[round], j);
        if ((j & 15) == 0) {  // the next k_chain takes maxima into them
            ver.tile_dsq[(size_t)(j >> 4) * 2] = 0ull;
            ver.tile_dsq[(size_t)(j >> 4) * 2 + 1] = 0ull;
        }
        // (claims on the first scan_rows table rows are gathered by k_claims instead, without atomics)
        if (T >= 0 && !(T < M0 && T < scan_rows)) {
            // first / last point of this window that targets T, for the round that replays these claims
            // (provisional ids of new MCs index the free rows behind the table)
            const unsigned long long sn = (stamp + 1ull) << 20;
            const size_t wr = (size_t)((round + 1) & 1) * tab.cap + (size_t)T;  // the copy the next round reads
            atomicMax(&tab.touch[wr], sn | (unsigned long long)(0xFFFFF - j));
            atomicMax(&tab.last[wr], sn | (unsigned long long)j);
            // ... and the list of all of them (the counter restarts whenever its stamp is an old one)
            const unsigned long long sc = (stamp + 1ull) << 24;
            unsigned long long* cw = tab.cnt + T;
            const unsigned long long old = __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int pos = CC_CHAIN_MEMB + 1;
            if ((old & ~0xFFFFFFull) == sc) {
                // live counter; one that already says "more than the list holds" needs no further count
                if ((int)(old & 0xFFFFFFull) <= CC_CHAIN_MEMB) pos = (int)(atomicAdd(cw, 1ull) & 0xFFFFFFull);
            } else if (atomicCAS(cw, old, sc | 1ull) == old) {
                pos = 0;  // restarted the counter
            } else {
                pos = (int)(atomicAdd(cw, 1ull) & 0xFFFFFFull);  // somebody else of this launch restarted it meanwhile
            }
            if (pos < CC_CHAIN_MEMB) tab.memb[(size_t)T * CC_CHAIN_MEMB + pos] = j;
            else if (pos == CC_CHAIN_MEMB && T < M0) {
                // the one claimant that finds the list full: a long chain of an existing MC
                atomicAdd((unsigned long long*)&ctl->stat_long, 1ull);
                if (long_list != nullptr) {
                    const int idx = atomicAdd(&ctl->n_long[round + 1], 1);
                    if (idx < CC_LONG_CAP) {
                        long_list[(size_t)((round + 1) & 1) * CC_LONG_CAP + idx] = T;
                        atomicOr(cw, CC_LONG_LISTED);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// k_claims: first / last claimant, number of claimants and (up to CC_CHAIN_MEMB) members of the chains of the first
// scan_rows table rows, one workgroup per MC reading the claims once.  With few MCs the per-point atomics of
// k_decide pile up on a handful of addresses (68 us per call at 50 MCs, measured); the host launches this kernel
// instead while the table is small.  Same stamps and formats as k_decide writes.
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_claims(const Ctl* __restrict__ ctl, Table tab, const int* __restrict__ T,
                                                int round, int scan_rows)
{
    CC_LATENCY_KERNEL();
    const int B = ctl->win_b;
    if (B == 0) return;
    if (round > 0 && ctl->fc[round - 1] >= B) return;  // k_decide of this round did not run either
    const int M0 = ctl->m_rows;
    const int m = blockIdx.x;
    if (m >= M0 || m >= scan_rows) return;
    __shared__ int s_pos, s_first, s_last;
    if (threadIdx.x == 0) { s_pos = 0; s_first = CC_IDX_INF; s_last = -1; }
    __syncthreads();
    int lmin = CC_IDX_INF, lmax = -1;
    const int4* T4 = reinterpret_cast<const int4*>(T);  // (the buffer is padded to whole 128-entry blocks)
    for (int base = (int)threadIdx.x * 4; base < B; base += 256 * 4) {
        const int4 v = T4[base >> 2];
        const int e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = base + c;
            if (j < B && e[c] == m) {
                lmin = j < lmin ? j : lmin;
                lmax = j > lmax ? j : lmax;
                const int pos = atomicAdd(&s_pos, 1);
                if (pos < CC_CHAIN_MEMB) tab.memb[(size_t)m * CC_CHAIN_MEMB + pos] = j;
            }
        }
    }
    if (lmax >= 0) { atomicMin(&s_first, lmin); atomicMax(&s_last, lmax); }
    __syncthreads();
    if (threadIdx.x == 0 && s_pos > 0) {
        const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
        const unsigned long long sn = (stamp + 1ull) << 20;
        const size_t wr = (size_t)((round + 1) & 1) * tab.cap + (size_t)m;  // the copy the next round reads
        tab.touch[wr] = sn | (unsigned long long)(0xFFFFF - s_first);
        tab.last[wr] = sn | (unsigned long long)s_last;
        tab.cnt[m] = ((stamp + 1ull) << 24) | (unsigned long long)s_pos;
    }
}

// ---------------------------------------------------------------------------------
// k_chain: replay the claimed decisions per MC in arrival order.  One 32-lane group per window point; the
// group of the first point that targets a MC walks that MC's chain, every step dimension-parallel.
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_chain(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                               Versions ver, Carry car, const int* __restrict__ T, int round,
                                               int long_rows)
{
    CC_LATENCY_KERNEL();
    const int B = ctl->win_b;
    if (B == 0) return;
    if (ctl->fc[round - 1] >= B) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->last_round = round;
    const int gl = threadIdx.x & 31;
    const int j = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (j >= B) return;
    const int t = T[j];
    if (t < 0) {  // undecided (CC_T_UNKNOWN) or set aside (CC_T_NONE): no MC is touched
        if (gl == 0) {
            ver.kind[j] = CC_KIND_DEAD; ver.next[j] = j; ver.tgt[j] = t; ver.acc[j] = 0; ver.upg[j] = -1;
        }
        return;
    }
    const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
    // k_decide recorded the first and the last window point that target t: the first one heads the chain and
    // walks it; everybody else is walked over
    const size_t rd = (size_t)(round & 1) * tab.cap + (size_t)t;
    const unsigned long long ft = tab.touch[rd], lt = tab.last[rd];
    if ((ft >> 20) != stamp || 0xFFFFF - (int)(ft & 0xFFFFFull) != j) return;
    const int last_j = ((lt >> 20) == stamp) ? (int)(lt & 0xFFFFFull) : j;
    // the members of the chain: up to CC_CHAIN_MEMB of them were listed by k_decide (unordered) - lane l keeps the
    // l-th smallest; a longer chain is found by scanning the claims (16-byte loads; the buffer is padded)
    const unsigned long long cw = tab.cnt[t];
    const int n_memb = ((cw >> 24) == stamp) ? (int)(cw & 0xFFFFFFull) : 0;
    const bool listed = n_memb <= CC_CHAIN_MEMB;
    // a long chain on one of the first long_rows table rows is replayed by k_chain_long (launched right after)
    if (!listed && ((t < long_rows && t < ctl->m_rows) || (cw & CC_LONG_LISTED) != 0ull)) return;
    int sorted_memb = CC_IDX_INF;
    if (listed) {
        const int mine = (gl < n_memb) ? tab.memb[(size_t)t * CC_CHAIN_MEMB + gl] : CC_IDX_INF;
        int rank = 0;
        for (int q = 0; q < n_memb; ++q) rank += (__shfl(mine, q, 32) < mine) ? 1 : 0;
        // lane l takes the member whose rank is l (ranks are distinct: the members are)
        for (int q = 0; q < n_memb; ++q) {
            const int v = __shfl(mine, q, 32), r = __shfl(rank, q, 32);
            if (r == gl) sorted_memb = v;
        }
    }
    const int4* T4 = reinterpret_cast<const int4*>(T);
    int step = 0;
    int blk = -1;  // block of claims held in vb (chains that are not listed)
    bool have_nb = false;
    int4 vb = make_int4(0, 0, 0, 0), vn = make_int4(0, 0, 0, 0);

    const Par par = cc_load_par(ctl);
    const int d = par.d;
    const int M0 = ctl->m_rows;
    const long long cursor = ctl->cursor;
    const int pk_base = ctl->n_pkeys;
    const bool isnew = t >= M0;
    const bool valid_chain = !isnew || (t == M0 + j);  // a claim on a MC nobody creates any more is void

    // this lane's two dimensions of the chain's running state stay in registers from step to step
    double bc1[2] = {0.0, 0.0}, bc2[2] = {0.0, 0.0}, bce[2] = {0.0, 0.0}, bpr[2] = {1.0, 1.0};
    double bw = 0.0;
    int bkind = CC_KIND_OUTLIER, bkey = ctl->n_okeys + j, bupg = -1;
    if (!isnew) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = gl + 32 * h;
            if (i < d) {
                bc1[h] = tab.cf1[(size_t)t * d + i]; bc2[h] = tab.cf2[(size_t)t * d + i];
                bce[h] = tab.cen[(size_t)t * d + i]; bpr[h] = tab.pref[(size_t)t * d + i];
            }
        }
        bw = tab.w[t]; bkind = tab.kind[t]; bkey = tab.key[t];
    }
    // centroid, metric and kind of this MC in the snapshot the window was scanned against, for the displacement of
    // its versions: the table row, unless (lookahead) the previous window changed it after that scan
    double c0[2] = {bce[0], bce[1]};
    double w0[2] = {1.0 / bpr[0], 1.0 / bpr[1]};
    int kind0 = bkind;
    if (!isnew && ctl->mode != 0) {
        const unsigned long long co = tab.carry_of[t];
        if ((co >> 20) == ctl->window_seq) {
            const size_t r = (size_t)(co & 0xFFFFFull);
            kind0 = car.kind0[r];  // CC_KIND_DEAD (never a live kind): not in the snapshot -> no bound
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = gl + 32 * h;
                if (i < d) { c0[h] = car.c0[r * d + i]; w0[h] = car.w0[r * d + i]; }
            }
        }
    }
    // The chain is walked in batches of CC_CHAIN_AHEAD members: the members of a batch are located first and their
    // points requested together (the rows of one MC's points are scattered over the window: with one point in
    // flight per step a long chain ran at the memory latency, 2.5 us per step); the steps themselves stay strictly
    // sequential.  mem[q] = q-th member of the batch, mem[CC_CHAIN_AHEAD] = first member of the next one.
    constexpr int NB = CC_CHAIN_AHEAD;
    // the member after `after` of a chain that is not listed: scan of the claims, 128 per block, in registers
    auto find_next = [&](const int after) -> int {
        int nx = CC_IDX_INF;
        for (int base = (after + 1) & ~127; base <= last_j && after < last_j; base += 128) {
            const int i = base + gl * 4;
            // the 128 claims of a block stay in registers while the chain moves inside it; the following block is
            // requested as soon as a block is entered, so its latency hides behind the chain steps
            if (base != blk) {
                vb = (have_nb && base == blk + 128) ? vn : T4[(base >> 2) + gl];
                blk = base;
                have_nb = base + 128 <= last_j;
                if (have_nb) vn = T4[((base + 128) >> 2) + gl];
            }
            const int4 v = vb;
            const unsigned mm = ((i > after && i < B && v.x == t) ? 1u : 0u) | ((i + 1 > after && i + 1 < B && v.y == t) ? 2u : 0u) |
                                ((i + 2 > after && i + 2 < B && v.z == t) ? 4u : 0u) | ((i + 3 > after && i + 3 < B && v.w == t) ? 8u : 0u);
            const unsigned b = cc_group_ballot(mm != 0u);
            if (b) {
                const int l = __builtin_ctz(b);
                const unsigned ml = __shfl(mm, l, 32);
                nx = base + l * 4 + __builtin_ctz(ml);
                break;
            }
        }
        return nx;
    };
    int first = j;  // first member of the next batch
    int walked = 0;
    while (first != CC_IDX_INF) {
        int mem[NB + 1];
        mem[0] = first;
#pragma unroll
        for (int q = 1; q <= NB; ++q) {
            int m = CC_IDX_INF;
            if (mem[q - 1] != CC_IDX_INF) {
                if (listed) m = (step + q < n_memb) ? __shfl(sorted_memb, (step + q) & 31, 32) : CC_IDX_INF;
                else m = find_next(mem[q - 1]);
            }
            mem[q] = m;
        }
        step += NB;
        double pxb[NB][2];  // this lane's two dimensions of the batch's points
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = gl + 32 * h;
                pxb[q][h] = (valid_chain && mem[q] != CC_IDX_INF && i < d) ? X[(cursor + mem[q]) * d + i] : 0.0;
            }
        // All of the batch's points have to be in before its first step anyway.  Waiting here, once, keeps the steps
        // free of vector-memory waits: gfx9 counts loads and stores in one in-order counter, so a wait for a point
        // inside the loop is also a wait for every version row stored before it (2 us per step, measured).
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (mem[q] == CC_IDX_INF) break;
            ++walked;
            const int cur = mem[q];
            const int nx = mem[q + 1];
            const double px[2] = {pxb[q][0], pxb[q][1]};
            if (!valid_chain) {
                if (gl == 0) {
                    ver.tgt[cur] = t; ver.kind[cur] = CC_KIND_DEAD; ver.next[cur] = cur; ver.acc[cur] = 0; ver.upg[cur] = -1;
                }
            } else {
                const double w1 = bw + 1.0;  // microcluster.py:147
                const GroupAdd g = cc_group_add_regs(bc1, bc2, bw, px, d, par);
                const bool creates = isnew && cur == j;
                const bool ok = creates || (g.r2 <= par.eps_sq);
                if (ok) {
                    // hddstream.py:416-430: promotion is only examined after an add to an existing outlier MC
                    if (bkind == CC_KIND_OUTLIER && !creates && w1 >= par.beta_mu && g.gt1 <= par.pi) {
                        bkind = CC_KIND_PCORE; bkey = pk_base + cur; bupg = cur;
                        if (gl == 0) ctl->any_up[round] = 1;
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        bc1[h] = g.c1[h]; bc2[h] = g.c2[h];
                        bce[h] = g.cen[h];  // mc_functions.py:31-33: CF1 / W, the quotient the variance was formed from
                        bpr[h] = g.pr[h];
                    }
                    bw = w1;
                }
                // the version row of `cur` = the MC's state right after `cur` (unchanged if the radius test failed)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = gl + 32 * h;
                    if (i < d) {
                        ver.cf1[(size_t)cur * d + i] = bc1[h]; ver.cf2[(size_t)cur * d + i] = bc2[h];
                        ver.cen[(size_t)cur * d + i] = bce[h]; ver.pref[(size_t)cur * d + i] = bpr[h];
                        ver.scl[(size_t)cur * d + i] = par.pow2 ? (bpr[h] == 1.0 ? 1.0 : par.inv_k) : bpr[h];
                    }
                }
                // squared displacement from the window-start centroid in the window-start metric (any summation order:
                // it only feeds a conservative bound); +inf when no bound exists (new MC, promoted inside the window)
                double dq = 0.0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double df = bce[h] - c0[h];
                    dq += df * df * w0[h];
                }
                dq = cc_group_sum_any_order(dq);  // (valid in the lane that stores it)
                // (a change of the preferred dimensions since the snapshot takes the bound away as well: k_dseed's
                // threshold assumes the window-start metric)
                bool mv = false;
#pragma unroll
                for (int h = 0; h < 2; ++h) mv = mv || ((1.0 / bpr[h]) != w0[h]);
                if (isnew || bkind != kind0 || !(dq >= 0.0) || cc_group_ballot(mv) != 0u) dq = CC_INF;
                if (gl == 0) {
                    ver.w[cur] = bw;
                    ver.tgt[cur] = t; ver.kind[cur] = bkind; ver.key[cur] = bkey; ver.upg[cur] = bupg;
                    ver.acc[cur] = ok ? 1 : 0; ver.next[cur] = nx;
                    ver.dsq[cur] = dq;
                    atomicMax(&ver.tile_dsq[(size_t)(cur >> 4) * 2 + (bkind == CC_KIND_PCORE ? 0 : 1)], cc_dsq_code(dq));
                }
            }
        }
        first = mem[NB];
    }
    if (gl == 0 && !isnew) tab.clen[t] = walked;  // k_dseed chooses its way of finding live versions by it
}

// ---------------------------------------------------------------------------------
// k_chain_long: the chains k_chain leaves alone - more than CC_CHAIN_MEMB claimants on one of the first scan_rows
// table rows (few microclusters: every MC absorbs hundreds of window points).  k_chain replays such a chain one
// point after the other (~1.6 us per step: locate the member, fetch its point, two IEEE divisions per dimension,
// ordered radius sum, 14 stores).  Only the CF sums are sequential by nature (microcluster.py:147, mc_functions.py:
// 24-29: CF1 += p, CF2 += p * p, W += 1); the radius test of step k (mc_functions.py:45-56) is a function of the sums
// after k alone.  One workgroup per MC therefore works in batches of K members:
//   1. the members are collected in order from the claims (ordered compaction of 1 024 claims per pass),
//   2. their points are staged in LDS, thread i < d runs the two additions per step of dimension i over the batch
//      (the same additions in the same order as k_chain), one more thread the additions of W,
//   3. thread k evaluates step k - variances, preferred dimensions, ordered radius sum, promotion test - assuming
//      that every earlier step of the batch was accepted,
//   4. up to the first rejected step f that assumption holds, so the version rows 0 .. f - 1 (and the unchanged state
//      as the row of f) are exactly what the sequential replay produces; the chain resumes after f from the state
//      of f - 1.
// Rows and stamps are written in k_chain's formats.  New MCs (created inside the window) stay with k_chain.
// ---------------------------------------------------------------------------------

#define CC_LONG_XY_DOUBLES 6144  // staged CF1 / CF2 prefixes of a batch: 2 * K * d doubles (48 KB)
#define CC_LONG_QUEUE 2048       // pending chain members (ring buffer)

__global__ __launch_bounds__(256) void k_chain_long(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                                    Versions ver, Carry car, const int* __restrict__ T, int round,
                                                    int scan_rows, const int* __restrict__ long_list)
{
    CC_LATENCY_KERNEL();
    const int B = ctl->win_b;
    if (B == 0) return;
    if (ctl->fc[round - 1] >= B) return;
    const int M0 = ctl->m_rows;
    // small tables (k_claims): one workgroup per table row; otherwise one per entry of the round's list of long chains
    int t;
    if (long_list == nullptr) {
        t = blockIdx.x;
        if (t >= scan_rows) return;
    } else {
        const int n_listed = min(ctl->n_long[round], CC_LONG_CAP);
        if ((int)blockIdx.x >= n_listed) return;
        t = long_list[(size_t)(round & 1) * CC_LONG_CAP + blockIdx.x];
    }
    if (t >= M0) return;
    const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
    const size_t rd = (size_t)(round & 1) * tab.cap + (size_t)t;
    const unsigned long long ft = tab.touch[rd], lt = tab.last[rd];
    if ((ft >> 20) != stamp) return;  // nobody targets this MC in this round
    const unsigned long long cw = tab.cnt[t];
    const int n_memb = ((cw >> 24) == stamp) ? (int)(cw & 0xFFFFFFull) : 0;
    if (n_memb <= CC_CHAIN_MEMB) return;  // a listed chain: k_chain walks it
    const int head = 0xFFFFF - (int)(ft & 0xFFFFFull);
    const int last_j = ((lt >> 20) == stamp) ? (int)(lt & 0xFFFFFull) : head;

    const Par par = cc_load_par(ctl);
    const int d = par.d;
    const int tid = threadIdx.x;
    const long long cursor = ctl->cursor;
    const int pk_base = ctl->n_pkeys;
    // steps per batch: the CF1 / CF2 prefixes of a batch have to fit the staging area
    const int K = (d <= 24) ? 128 : ((d <= 48) ? 64 : 32);

    __shared__ __attribute__((aligned(16))) double s_xy[CC_LONG_XY_DOUBLES];
    __shared__ double s_w[128], s_dq[128];
    __shared__ unsigned long long s_mask[128];  // bit i: dimension i is a preferred one after the step (var <= delta^2)
    __shared__ int s_flag[128];                 // bit 0: radius test passed, bit 1: promotion condition holds
    __shared__ int s_queue[CC_LONG_QUEUE];
    __shared__ double s_b1[64], s_b2[64], s_bcen[64], s_bpref[64], s_c0[64], s_w0[64];  // running state / snapshot metric
    __shared__ double s_bw, s_bdq;
    __shared__ unsigned long long s_m0, s_bmask;  // preferred dimensions in the snapshot / of the running state (bit i)
    __shared__ int s_wsum[4];
    __shared__ int s_first_fail, s_first_up;
    double* const xs = s_xy;
    double* const ys = s_xy + (size_t)K * d;

    // running state of the chain (same meaning as k_chain's registers)
    int bkind = tab.kind[t], bkey = tab.key[t], bupg = -1;
    if (tid < d) {
        s_b1[tid] = tab.cf1[(size_t)t * d + tid]; s_b2[tid] = tab.cf2[(size_t)t * d + tid];
        s_bcen[tid] = tab.cen[(size_t)t * d + tid]; s_bpref[tid] = tab.pref[(size_t)t * d + tid];
    }
    if (tid == 0) s_bw = tab.w[t];
    __syncthreads();
    // centroid, metric and kind in the snapshot the window was scanned against (k_chain: c0, w0, kind0)
    int kind0 = bkind;
    {
        bool from_carry = false;
        size_t r = 0;
        if (ctl->mode != 0) {
            const unsigned long long co = tab.carry_of[t];
            if ((co >> 20) == ctl->window_seq) {
                from_carry = true;
                r = (size_t)(co & 0xFFFFFull);
                kind0 = car.kind0[r];
            }
        }
        if (tid < d) {
            s_c0[tid] = from_carry ? car.c0[r * d + tid] : s_bcen[tid];
            s_w0[tid] = from_carry ? car.w0[r * d + tid] : 1.0 / s_bpref[tid];
        }
    }
    __syncthreads();
    if (tid == 0) {
        double dq = 0.0;
        for (int i = 0; i < d; ++i) {
            const double df = s_bcen[i] - s_c0[i];
            dq += df * df * s_w0[i];
        }
        s_bdq = dq;
        unsigned long long m0 = 0ull, bm = 0ull;
        for (int i = 0; i < d; ++i) {
            m0 |= (s_w0[i] != 1.0) ? (1ull << i) : 0ull;
            bm |= (s_bpref[i] != 1.0) ? (1ull << i) : 0ull;
        }
        s_m0 = m0;
        s_bmask = bm;
    }

    const int4* T4 = reinterpret_cast<const int4*>(T);  // (the claims buffer is padded to whole 128-entry blocks)
    int qhead = 0, qcount = 0;     // ring buffer of pending members (the same in every thread)
    int scan_pos = head & ~1023;   // next block of 1 024 claims to look at
    bool scan_done = false;
    int walked = 0;
    bool promoted_any = false;

    for (;;) {
        // ---- 1. members in order: ordered compaction of the next claims into the queue ----
        while (!scan_done && qcount < K + 1 && qcount + 1024 <= CC_LONG_QUEUE) {
            const int i0 = scan_pos + tid * 4;
            int4 v = make_int4(-1, -1, -1, -1);
            if (i0 <= last_j) v = T4[i0 >> 2];
            const int e[4] = {v.x, v.y, v.z, v.w};
            int f[4], cnt = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = i0 + c;
                f[c] = (j >= head && j <= last_j && j < B && e[c] == t) ? 1 : 0;
                cnt += f[c];
            }
            int incl = cnt;
            const int lane = tid & 63, wv = tid >> 6;
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off);
                if (lane >= off) incl += o;
            }
            __syncthreads();  // (s_wsum of the previous pass has been read)
            if (lane == 63) s_wsum[wv] = incl;
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w < wv) base += s_wsum[w];
                total += s_wsum[w];
            }
            int pos = qcount + base + incl - cnt;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (f[c]) { s_queue[(qhead + pos) & (CC_LONG_QUEUE - 1)] = i0 + c; ++pos; }
            qcount += total;
            scan_pos += 1024;
            if (scan_pos > last_j) scan_done = true;
        }
        __syncthreads();
        if (qcount == 0) break;
        const int n = qcount < K ? qcount : K;  // steps of this batch (the member after it is known, or the chain ends)

        // ---- 2. stage the points, then the sequential additions per dimension ----
        for (int e = tid; e < n * d; e += 256) {
            const int k = e / d, i = e - k * d;
            const int m = s_queue[(qhead + k) & (CC_LONG_QUEUE - 1)];
            xs[e] = X[(cursor + m) * d + i];
        }
        if (tid == 0) { s_first_fail = n; s_first_up = n; }
        __syncthreads();
        if (tid < d) {
            double c1 = s_b1[tid], c2 = s_b2[tid];
            for (int k = 0; k < n; ++k) {
                const double x = xs[k * d + tid];
                c1 = c1 + x;          // mc_functions.py:24-29, the additions k_chain makes, in its order
                c2 = c2 + x * x;
                xs[k * d + tid] = c1;
                ys[k * d + tid] = c2;
            }
        } else if (tid == 64) {
            double w = s_bw;
            for (int k = 0; k < n; ++k) {
                w = w + 1.0;  // microcluster.py:147
                s_w[k] = w;
            }
        }
        __syncthreads();

        // ---- 3. every step evaluated on its own prefix ----
        if (tid < n) {
            const int k = tid;
            const double w1 = s_w[k];
            double r2 = 0.0, dq = 0.0;
            int gt1 = 0;
            unsigned long long mask = 0ull;
            for (int i = 0; i < d; ++i) {
                const double qa = ys[k * d + i] / w1;  // mc_functions.py:14-22 (cc_sqvar), keeping CF1 / W
                const double qb = xs[k * d + i] / w1;
                const double var = qa - qb * qb;
                const bool prefd = var <= par.delta_sq;  // microcluster.py:109-114 (NaN -> 1.0)
                const double pr = prefd ? par.k : 1.0;
                r2 = r2 + cc_div_pref(var, pr, par);     // mc_functions.py:54, left to right
                gt1 += (pr > 1.0) ? 1 : 0;
                mask |= prefd ? (1ull << i) : 0ull;
                const double df = qb - s_c0[i];
                dq += df * df * s_w0[i];
            }
            const bool ok = r2 <= par.eps_sq;                          // hddstream.py:334-337
            const bool up = w1 >= par.beta_mu && gt1 <= par.pi;        // hddstream.py:416-417
            s_flag[k] = (ok ? 1 : 0) | (up ? 2 : 0);
            s_mask[k] = mask;
            s_dq[k] = dq;
            if (!ok) atomicMin(&s_first_fail, k);
            if (up) atomicMin(&s_first_up, k);
        }
        __syncthreads();
        const int f = s_first_fail;                 // first rejected step (n: none)
        const int n_ok = f < n ? f : n;             // accepted steps 0 .. n_ok - 1
        const int n_rows = f < n ? f + 1 : n;       // members consumed by this batch (the rejected one included)
        // hddstream.py:416-430: the first accepted add to an outlier MC that fulfils the condition promotes it
        int u = -1;
        if (bkind == CC_KIND_OUTLIER && s_first_up < n_ok) u = s_first_up;
        const int up_point = (u >= 0) ? s_queue[(qhead + u) & (CC_LONG_QUEUE - 1)] : -1;

        // ---- 4. version rows: vectors by (row, dimension), the rest by row ----
        for (int e = tid; e < n_rows * d; e += 256) {
            const int k = e / d, i = e - k * d;
            const int m = s_queue[(qhead + k) & (CC_LONG_QUEUE - 1)];
            const int src = (k < n_ok) ? k : k - 1;  // a rejected step leaves the state of the step before it
            double c1, c2, ce, pr;
            if (src >= 0) {
                c1 = xs[src * d + i]; c2 = ys[src * d + i];
                ce = c1 / s_w[src];  // mc_functions.py:31-33: the quotient the variance was formed from
                pr = ((s_mask[src] >> i) & 1ull) ? par.k : 1.0;
            } else {
                c1 = s_b1[i]; c2 = s_b2[i]; ce = s_bcen[i]; pr = s_bpref[i];
            }
            const size_t o = (size_t)m * d + i;
            ver.cf1[o] = c1; ver.cf2[o] = c2; ver.cen[o] = ce; ver.pref[o] = pr;
            ver.scl[o] = par.pow2 ? (pr == 1.0 ? 1.0 : par.inv_k) : pr;
        }
        if (tid < n_rows) {
            const int k = tid;
            const int m = s_queue[(qhead + k) & (CC_LONG_QUEUE - 1)];
            const int src = (k < n_ok) ? k : k - 1;
            const bool promoted = u >= 0 && k >= u;
            const int kind = promoted ? CC_KIND_PCORE : bkind;
            double dq = (src >= 0) ? s_dq[src] : s_bdq;
            // (no bound either when the preferred dimensions differ from the snapshot's, see k_chain)
            const unsigned long long vmask = (src >= 0) ? s_mask[src] : s_bmask;
            if (kind != kind0 || !(dq >= 0.0) || vmask != s_m0) dq = CC_INF;
            const int nx = (k + 1 < qcount) ? s_queue[(qhead + k + 1) & (CC_LONG_QUEUE - 1)] : CC_IDX_INF;
            ver.w[m] = (src >= 0) ? s_w[src] : s_bw;
            ver.tgt[m] = t;
            ver.kind[m] = kind;
            ver.key[m] = promoted ? pk_base + up_point : bkey;
            ver.upg[m] = promoted ? up_point : bupg;
            ver.acc[m] = (k < n_ok) ? 1 : 0;
            ver.next[m] = nx;
            ver.dsq[m] = dq;
            atomicMax(&ver.tile_dsq[(size_t)(m >> 4) * 2 + (kind == CC_KIND_PCORE ? 0 : 1)], cc_dsq_code(dq));
        }
        __syncthreads();  // every read of the running state and of the queue slots is done

        // ---- 5. the running state moves on to the last accepted step ----
        if (n_ok > 0) {
            const int l = n_ok - 1;
            if (tid < d) {
                s_b1[tid] = xs[l * d + tid]; s_b2[tid] = ys[l * d + tid];
                s_bcen[tid] = xs[l * d + tid] / s_w[l];
                s_bpref[tid] = ((s_mask[l] >> tid) & 1ull) ? par.k : 1.0;
            }
            if (tid == 64) { s_bw = s_w[l]; s_bdq = s_dq[l]; s_bmask = s_mask[l]; }
        }
        if (u >= 0) {
            bkind = CC_KIND_PCORE; bkey = pk_base + up_point; bupg = up_point;
            promoted_any = true;
        }
        qhead = (qhead + n_rows) & (CC_LONG_QUEUE - 1);
        qcount -= n_rows;
        walked += n_rows;
        __syncthreads();
    }
    if (tid == 0) {
        tab.clen[t] = walked;  // k_dseed chooses its way of finding live versions by it
        if (promoted_any) ctl->any_up[round] = 1;
    }
}

// ---------------------------------------------------------------------------------
// commit: k_commit_a (one workgroup) ranks the new MCs / promotions of the validated prefix in point order and
// opens the next window; k_commit_b (many workgroups) writes the labels and copies the last version of every
// touched MC back into the table.
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(1024) void k_commit_a(Ctl* __restrict__ ctl, Table tab, Versions ver, Carry car,
                                                   const int* __restrict__ Tbuf0, const int* __restrict__ Tbuf1,
                                                   int* __restrict__ rk, CommitRec* __restrict__ rec)
{
    CC_LATENCY_KERNEL();
    const int B = ctl->win_b;
    if (B == 0) {
        if (threadIdx.x == 0) rec->n = 0;
        return;
    }
    __shared__ unsigned wsum[16];
    __shared__ unsigned tot;
    __shared__ int dirty_tiles;
    const int r = ctl->last_round;
    const bool la_win = ctl->mode != 0;  // this window's snapshot scan ran ahead (read before thread 0 moves on)
    if (threadIdx.x == 0) dirty_tiles = 0;
    __syncthreads();
    // point tiles whose dirty scan ran in the last validation round (the host keeps windows short while most do)
    const int n_tiles = (r >= 1) ? (B + 63) / 64 : 0;
    for (int i = threadIdx.x; i < n_tiles; i += 1024)
        if (ver.skip[i] == 0 || (la_win && ver.skip_car[i] == 0)) atomicAdd(&dirty_tiles, 1);
    const int* T = ((r - 1) & 1) ? Tbuf1 : Tbuf0;
    const int fcv = ctl->fc[r];
    const int n = fcv < B ? fcv : B;
    const int M0 = ctl->m_rows;
    const int tid = threadIdx.x;
    const long long cursor = ctl->cursor;
    const long long oid0 = ctl->outlier_last_id, pid0 = ctl->pcore_last_id;
    const int pk0 = ctl->n_pkeys, ok0 = ctl->n_okeys;
    const long long n_points = ctl->n_points;
    const int win_cfg = ctl->win_cfg;

    // Per point: bit 0 "creates a MC", bit 1 "its add promoted the MC"; read with coalesced loads into LDS, then
    // every thread ranks a contiguous run of points (packed counts, unsigned: low 16 bits creations, high 16 bits
    // promotions; B <= CC_MAX_WINDOW < 2^16, so creations never carry into the promotions and promotions fit the upper
    // 16 bits; readers take the upper half as unsigned).
    __shared__ unsigned char sflag[CC_MAX_WINDOW];
    // (steady state: nothing was created or promoted in this window - nothing to rank, k_commit_b never reads rk)
    const bool events = ctl->any_new[r - 1] != 0 || ctl->any_up[r] != 0;
    if (threadIdx.x == 0) tot = 0u;
    if (events) {
    for (int j = tid; j < B; j += 1024)
        sflag[j] = (j < n) ? (unsigned char)(((T[j] == M0 + j) ? 1 : 0) | ((ver.upg[j] == j) ? 2 : 0)) : (unsigned char)0;
    __syncthreads();
    const int per = (B + 1023) >> 10;  // points per thread
    unsigned mine = 0u;
    for (int q = 0; q < per; ++q) {
        const int j = tid * per + q;
        const unsigned f = (j < B) ? (unsigned)sflag[j] : 0u;
        mine += (f & 1u) | ((f & 2u) << 15);
    }
    // inclusive wave scan of the per-thread sums
    unsigned v = mine;
    const int lane = tid & 63, wid = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    if (lane == 63) wsum[wid] = v;
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0u;
        for (int i = 0; i < 16; ++i) { const unsigned x = wsum[i]; wsum[i] = run; run += x; }
        tot = run;
    }
    __syncthreads();
    unsigned run = v - mine + wsum[wid];
    for (int q = 0; q < per; ++q) {
        const int j = tid * per + q;
        if (j < B) {
            rk[j] = (int)run;  // exclusive prefix (two unsigned 16-bit counts)
            const unsigned f = (unsigned)sflag[j];
            run += (f & 1u) | ((f & 2u) << 15);
        }
    }
    }
    __syncthreads();
    const int tot_new = (int)(tot & 0xFFFFu), tot_up = (int)(tot >> 16);
    // Lookahead: the snapshot scan of the next window is already under way (or done) if the host enqueues such
    // scans; it is usable when this window committed in full, so that the next one starts where that scan assumed.
    const unsigned long long seq = ctl->window_seq;
    const long long next_cursor = cursor + n;
    const long long left = n_points - next_cursor;
    const int next_b = (int)(left < (long long)win_cfg ? left : (long long)win_cfg);
    const int qn = (int)((seq + 1ull) & 1ull);
    const bool la_ok = ctl->la_on != 0 && n == B && next_b > 0 && ctl->la_b[qn] == next_b &&
                       ctl->la_cursor[qn] == next_cursor;
    if (la_ok)
        for (int i = tid; i < 2 * ((B + 15) / 16 + 1); i += 1024) car.tile_dsq[i] = 0ull;  // k_commit_b takes maxima into them
    if (tid == 0) {
        rec->n = n; rec->M0 = M0; rec->pk0 = pk0; rec->ok0 = ok0; rec->pid0 = pid0; rec->oid0 = oid0; rec->T = T;
        rec->carry = la_ok ? 1 : 0;
        rec->cursor = cursor;
        rec->next_seq = seq + 1ull;
        ctl->mode = la_ok ? 1 : 0;
        ctl->car_n = la_ok ? B : 0;
        ctl->stat_lookahead += la_ok ? 1 : 0;
        // what the lookahead scan launched after this commit covers: the window after the next one, assuming the
        // next one commits in full
        const int q2 = (int)(seq & 1ull);  // parity of seq + 2
        const long long c2 = next_cursor + next_b;
        const long long left2 = n_points - c2;
        ctl->la_cursor[q2] = c2;
        ctl->la_b[q2] = (ctl->la_on != 0 && left2 > 0) ? (int)(left2 < (long long)win_cfg ? left2 : (long long)win_cfg) : 0;
        ctl->la_rows[q2] = M0 + tot_new;
        ctl->m_rows = M0 + tot_new;
        ctl->n_okeys = ok0 + tot_new;
        ctl->outlier_last_id = oid0 + tot_new;
        ctl->n_pkeys = pk0 + tot_up;
        ctl->pcore_last_id = pid0 + tot_up;
        ctl->cursor = cursor + n;
        ctl->stat_windows += 1;
        ctl->stat_rounds += r;
        ctl->round_hist[r] += 1;
        ctl->stat_truncated += (n < B) ? 1 : 0;
        ctl->stat_tiles += n_tiles;
        ctl->stat_dirty_tiles += dirty_tiles;
        ctl->stat_trunc_unknown += (n < B && T[n] == CC_T_UNKNOWN) ? 1 : 0;
        ctl->stat_table_rows += M0;
        ctl->stat_pair_rows += (double)B * (double)M0;
        // next window
        ctl->window_seq = seq + 1ull;
        if ((ctl->la_on != 0 && !la_ok && next_b > 0) || n == 0) {
            // (n == 0: the first point could not be decided without the dirty scans the host had stopped launching)
            // no usable lookahead scan and, in a lookahead batch, no in-place scan either: wait for the host
            ctl->stall_b = next_b;
            ctl->win_b = 0;
            ctl->la_b[0] = 0;
            ctl->la_b[1] = 0;
        } else {
            ctl->win_b = next_b;
        }
        ctl->last_round = 0;
        ctl->fc[0] = 0;
        for (int i = 1; i < CC_MAX_ROUNDS + 2; ++i) ctl->fc[i] = CC_IDX_INF;
        for (int i = 0; i < CC_MAX_ROUNDS + 2; ++i) { ctl->any_new[i] = 0; ctl->any_up[i] = 0; ctl->n_long[i] = 0; }
    }
}

// One 32-lane group per point of the validated prefix; the group whose point holds the last version of a MC copies
// it into the table.  When the next window is a lookahead window (rec->carry) the same rows, together with what
// the table row held before, become the carry set (see Carry).
__global__ __launch_bounds__(256) void k_commit_b(const CommitRec* __restrict__ rec, Table tab, Versions ver,
                                                  Carry car, const int* __restrict__ rk,
                                                  const int8_t* __restrict__ dpath, long long* __restrict__ lab_uid,
                                                  int8_t* __restrict__ lab_path, int d, ScanCopy sc, int filter)
{
    CC_LATENCY_KERNEL();
    const int n = rec->n;
    if (n == 0) return;
    const int M0 = rec->M0;
    const int* T = rec->T;
    const bool carry = rec->carry != 0;
    const int gl = threadIdx.x & 31;
    const int groups = (gridDim.x * blockDim.x) >> 5;
    for (int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 5; j < n; j += groups) {
        if (gl == 0) {
            // the label of point j: creation number of the MC that holds it (microcluster.py:149); rows below M0 keep
            // their uid in this commit
            const int tj = T[j];
            lab_uid[rec->cursor + j] = (tj < 0) ? -1ll : ((tj < M0) ? tab.uid[tj] : rec->oid0 + (rk[tj - M0] & 0xFFFF));
            lab_path[rec->cursor + j] = (int8_t)(dpath[j] | ((ver.upg[j] == j) ? 4 : 0));
        }
        if (ver.next[j] < n) {  // a later point of the prefix holds the MC's last version
            if (carry && gl == 0) { car.kind[j] = CC_KIND_DEAD; car.slot[j] = 0; }
            continue;
        }
        const int t = T[j];
        const int c = t - M0;
        const size_t row = (t < M0) ? (size_t)t : (size_t)(M0 + (rk[c] & 0xFFFF));
        const int u = ver.upg[j];
        const int kind = ver.kind[j];
        int key;
        if (u >= 0) key = rec->pk0 + (int)((unsigned)rk[u] >> 16);
        else if (t >= M0) key = rec->ok0 + (rk[c] & 0xFFFF);
        else key = tab.key[row];
        const int kind0 = (t < M0) ? tab.kind[row] : CC_KIND_DEAD;
        double dq = 0.0;
        bool metric_moved = false;  // the preferred dimensions differ from what the snapshot held
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = gl + 32 * h;
            if (i >= d) continue;
            const size_t e = row * d + i, v = (size_t)j * d + i;
            const double ncen = ver.cen[v], npref = ver.pref[v], nscl = ver.scl[v], n1 = ver.cf1[v], n2 = ver.cf2[v];
            if (carry) {
                const double oc = (t < M0) ? tab.cen[e] : 0.0;
                const double ow = (t < M0) ? 1.0 / tab.pref[e] : 0.0;
                metric_moved = metric_moved || (t < M0 && npref != tab.pref[e]);
                car.c0[v] = oc; car.w0[v] = ow;
                car.cf1[v] = n1; car.cf2[v] = n2; car.cen[v] = ncen; car.pref[v] = npref; car.scl[v] = nscl;
                const double df = ncen - oc;
                dq += df * df * ow;
            }
            tab.cf1[e] = n1; tab.cf2[e] = n2; tab.cen[e] = ncen; tab.pref[e] = npref; tab.scl[e] = nscl;
            if (sc.cen) {
                sc.cen[e] = ncen; sc.scl[e] = nscl;
                if (filter) { sc.cf1[e] = n1; sc.cf2[e] = n2; }
            }
        }
        if (carry) {
            for (int off = 16; off >= 1; off >>= 1) dq += __shfl_xor(dq, off, 32);
            if (kind0 == CC_KIND_DEAD || kind != kind0 || !(dq >= 0.0) || cc_group_ballot(metric_moved) != 0u) dq = CC_INF;
        }
        if (gl == 0) {
            tab.w[row] = ver.w[j];
            tab.kind[row] = kind;
            if (u >= 0) {
                tab.key[row] = key;
                tab.id[row] = rec->pid0 + (long long)((unsigned)rk[u] >> 16);
            } else if (t >= M0) {
                tab.key[row] = key;
                tab.id[row] = rec->oid0 + (rk[c] & 0xFFFF);
            }
            if (t >= M0) tab.uid[row] = rec->oid0 + (rk[c] & 0xFFFF);
            if (sc.cen) {
                sc.kind[row] = kind;
                sc.key[row] = key;
                if (filter) sc.w[row] = ver.w[j];
            }
            if (carry) {
                car.w[j] = ver.w[j];
                car.kind[j] = kind;
                car.key[j] = key;
                car.slot[j] = (int)row;
                car.kind0[j] = kind0;
                car.dsq[j] = dq;
                atomicMax(&car.tile_dsq[(size_t)(j >> 4) * 2 + (kind == CC_KIND_PCORE ? 0 : 1)], cc_dsq_code(dq));
                tab.carry_of[row] = (rec->next_seq << 20) | (unsigned long long)j;
            }
        }
    }
}


// ---------------------------------------------------------------------------------
// k_seq: the reference's loop taken literally (hddstream.py:220-237), for streams on which speculation does not pay:
// a handful of microclusters absorb every point (the bundled d0-d4 data: 2-15 pcore MCs), so the chains of a window
// are hundreds of points long, decisions keep moving and windows commit a few hundred points per validation pass.
// One wavefront walks the points in order with the whole table in LDS (structure of arrays, row = lane-strided):
//   per point: lanes take the rows r = lane, lane + 64, ...: projected distance to the pcore rows (with the
//   tentative-add pdim filter when pi < d), wave argmin by (distance, list-order key) through DPP row operations,
//   tentative add of the winner with lane = dimension (two IEEE divisions per dimension side by side), ordered
//   radius sum, commit into LDS; only if that fails the same over the outlier rows (+ promotion), else a new row.
// No speculation, nothing to validate: ~0.2-0.4 us per point whatever the data.  The host uses it while the table
// fits the LDS image (seq_cap_rows) and the windows of the speculative path keep being cut short; both paths are
// exact, so switching between them between windows never changes a result.
// ---------------------------------------------------------------------------------

#define CC_SEQ_DOUBLES 6600  // LDS image of the table: (4 d + 5) doubles per row
#define CC_SEQ_CHUNK_DOUBLES 512  // points staged ahead: 512 / d of them (at most 64), eight doubles per lane in flight

__host__ __device__ inline int cc_seq_cap_rows(int d) { return CC_SEQ_DOUBLES / (4 * d + 5); }

__device__ __forceinline__ double cc_readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// minimum over the wavefront (no NaN among the operands), the same value in every lane
__device__ __forceinline__ double cc_wave_min_f64(double x)
{
    x = cc_vmin(x, cc_dpp_f64<0xB1>(x));   // quad_perm [1,0,3,2]
    x = cc_vmin(x, cc_dpp_f64<0x4E>(x));   // quad_perm [2,3,0,1]
    x = cc_vmin(x, cc_dpp_f64<0x141>(x));  // row_half_mirror
    x = cc_vmin(x, cc_dpp_f64<0x140>(x));  // row_mirror: every lane of a row of 16 holds the row's minimum
    const double a = cc_readlane_f64(x, 0), b = cc_readlane_f64(x, 16), c = cc_readlane_f64(x, 32), e = cc_readlane_f64(x, 48);
    const double ab = a < b ? a : b, ce = c < e ? c : e;
    return ab < ce ? ab : ce;
}

// FILTER: pi < d (the tentative-add pdim filter of the pcore stage is not vacuous); POW2: k is a power of two
template <bool FILTER, bool POW2>
__global__ __launch_bounds__(64) void k_seq(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                            long long* __restrict__ lab_uid, int8_t* __restrict__ lab_path, int n_max)
{
    const long long clk0 = clock64(), wall0 = wall_clock64();
    const Par par = cc_load_par(ctl);
    const int d = par.d;
    const int lane = threadIdx.x;
    const int cap = cc_seq_cap_rows(d);
    int M = ctl->m_rows;
    if (M > cap) return;  // (the host checks the same bound)
    const long long cursor0 = ctl->cursor;
    const long long left = ctl->n_points - cursor0;
    const int n = (int)(left < (long long)n_max ? left : (long long)n_max);
    if (n <= 0) return;
    int n_pkeys = ctl->n_pkeys, n_okeys = ctl->n_okeys;
    long long pcore_last_id = ctl->pcore_last_id, outlier_last_id = ctl->outlier_last_id;
    constexpr bool filter = FILTER;
    constexpr bool pow2 = POW2;

    // table image, structure of arrays with the row as the fast index.  `op` is the distance operand of a dimension:
    // 1 or 1/k when k is a power of two (x / k == x * (1/k) bit for bit), else the preferred-dimension entry itself
    __shared__ __attribute__((aligned(16))) double s_tab[CC_SEQ_DOUBLES];
    __shared__ __attribute__((aligned(16))) double s_pts[CC_SEQ_CHUNK_DOUBLES];
    __shared__ long long s_luid[64];
    __shared__ int s_lpath[64];
    double* const Lcf1 = s_tab;
    double* const Lcf2 = Lcf1 + (size_t)d * cap;
    double* const Lcen = Lcf2 + (size_t)d * cap;
    double* const Lop = Lcen + (size_t)d * cap;
    double* const Lw = Lop + (size_t)d * cap;
    int* const Lkind = reinterpret_cast<int*>(Lw + cap);
    int* const Lkey = Lkind + cap;
    long long* const Lid = reinterpret_cast<long long*>(Lw + 2 * (size_t)cap);
    long long* const Luid = Lid + cap;
    int* const Lplist = reinterpret_cast<int*>(Lw + 4 * (size_t)cap);  // rows of the pcore MCs / of the outlier MCs, any order
    int* const Lolist = Lplist + cap;
    auto op_of = [&](double pr) { return pow2 ? (pr == 1.0 ? 1.0 : par.inv_k) : pr; };
    auto pref_of = [&](double op) { return pow2 ? (op == 1.0 ? 1.0 : par.k) : op; };
    // x / pref through the operand (mc_functions.py:39)
    auto scaled = [&](double x, double op) { return pow2 ? x * op : (op == 1.0 ? x : x / op); };

    for (int r = lane; r < M; r += 64) {
        for (int i = 0; i < d; ++i) {
            Lcf1[i * cap + r] = tab.cf1[(size_t)r * d + i]; Lcf2[i * cap + r] = tab.cf2[(size_t)r * d + i];
            Lcen[i * cap + r] = tab.cen[(size_t)r * d + i]; Lop[i * cap + r] = op_of(tab.pref[(size_t)r * d + i]);
        }
        Lw[r] = tab.w[r]; Lkind[r] = tab.kind[r]; Lkey[r] = tab.key[r]; Lid[r] = tab.id[r]; Luid[r] = tab.uid[r];
    }
    int n_p = 0, n_o = 0;
    CC_WAVE_SYNC();
    if (lane == 0)
        for (int r = 0; r < M; ++r) {
            if (Lkind[r] == CC_KIND_PCORE) Lplist[n_p++] = r;
            else Lolist[n_o++] = r;
        }
    n_p = __builtin_amdgcn_readfirstlane(n_p);
    n_o = __builtin_amdgcn_readfirstlane(n_o);

    // points are fetched a chunk ahead: element e of a chunk (row-major, C points x d) by lane e % 64
    const int C = (CC_SEQ_CHUNK_DOUBLES / d) < 64 ? (CC_SEQ_CHUNK_DOUBLES / d) : 64;
    double pf[8];
    auto fetch_chunk = [&](int first) {
        const int cnt = (n - first) < C ? (n - first) : C;  // >= 1 at every call
        const int last = cnt * d - 1;
        const double* src = X + (cursor0 + first) * d;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = q * 64 + lane;
            pf[q] = src[e < last ? e : last];  // (clamped: eight loads in flight, no branch around any of them)
        }
    };
    fetch_chunk(0);
    CC_WAVE_SYNC();

    int done = 0;
    bool full = false;
    for (int c0 = 0; c0 < n && !full; c0 += C) {
        const int cnt = (n - c0) < C ? (n - c0) : C;
        CC_WAVE_SYNC();  // (the previous chunk's points and labels have been consumed)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = q * 64 + lane;
            if (e < CC_SEQ_CHUNK_DOUBLES) s_pts[e] = pf[q];
        }
        if (c0 + C < n) fetch_chunk(c0 + C);  // in flight while this chunk is processed
        CC_WAVE_SYNC();
        int cdone = 0;
        for (int jj = 0; jj < cnt; ++jj) {
            const double* sp = s_pts + jj * d;
            const double myp = (lane < d) ? sp[lane] : 0.0;  // this lane's dimension of the point
            int target = -1, path = 2;
            bool promoted = false;
            // stage 0: _add_to_pcore (hddstream.py:288-343), stage 1: _add_to_outlier (:345-395)
            for (int stage = 0; stage < 2 && target < 0; ++stage) {
                const int* list = stage == 0 ? Lplist : Lolist;
                const int n_list = stage == 0 ? n_p : n_o;
                double bd = CC_INF;
                int bk = CC_IDX_INF, br = -1;
#pragma nounroll
                for (int q = lane; q < n_list; q += 64) {
                    const int r = list[q];
                    if (stage == 0 && filter) {
                        // hddstream.py:317-321: pdim of the MC with the point added must be <= pi
                        const double w1 = Lw[r] + 1.0;
                        int ne1 = 0;
#pragma nounroll
                        for (int i = 0; i < d; ++i) {
                            const double x = sp[i];
                            const double c1 = Lcf1[i * cap + r] + x, c2 = Lcf2[i * cap + r] + x * x;
                            const double var = cc_sqvar(c1, c2, w1);
                            ne1 += (((var <= par.delta_sq) ? par.k : 1.0) != 1.0) ? 1 : 0;
                        }
                        if (ne1 > par.pi) continue;
                    }
                    double acc = 0.0;
                    int i = 0;
#pragma nounroll
                    for (; i + 4 <= d; i += 4) {  // loads of four dimensions together, sums left to right
                        const double p0 = sp[i], p1 = sp[i + 1], p2 = sp[i + 2], p3 = sp[i + 3];
                        const double e0 = Lcen[i * cap + r], e1 = Lcen[(i + 1) * cap + r], e2 = Lcen[(i + 2) * cap + r], e3 = Lcen[(i + 3) * cap + r];
                        const double o0 = Lop[i * cap + r], o1 = Lop[(i + 1) * cap + r], o2 = Lop[(i + 2) * cap + r], o3 = Lop[(i + 3) * cap + r];
                        double x0 = p0 - e0, x1 = p1 - e1, x2 = p2 - e2, x3 = p3 - e3;  // mc_functions.py:37
                        x0 = x0 * x0; x1 = x1 * x1; x2 = x2 * x2; x3 = x3 * x3;          // :38
                        acc = acc + scaled(x0, o0);                                      // :39 + :41
                        acc = acc + scaled(x1, o1);
                        acc = acc + scaled(x2, o2);
                        acc = acc + scaled(x3, o3);
                    }
#pragma nounroll
                    for (; i < d; ++i) {
                        double x = sp[i] - Lcen[i * cap + r];
                        x = x * x;
                        acc = acc + scaled(x, Lop[i * cap + r]);
                    }
                    const int key = Lkey[r];
                    if (cand_less(acc, key, bd, bk)) { bd = acc; bk = key; br = r; }  // strict <, first in list order wins (:326/:373)
                }
                // wave argmin by (distance, key): the minimum distance, then the smallest key among the lanes that hold it
                const double D = cc_wave_min_f64(bd);
                const unsigned long long tied = __builtin_amdgcn_ballot_w64(br >= 0 && bd == D);
                if (tied == 0ull) continue;  // no (admissible) MC of this kind
                int wl = __builtin_ctzll(tied);
                if (tied & (tied - 1ull)) {
                    int best_key = CC_IDX_INF;
                    for (unsigned long long m = tied; m; m &= m - 1ull) {
                        const int l = __builtin_ctzll(m);
                        const int k2 = __builtin_amdgcn_readlane(bk, l);
                        if (k2 < best_key) { best_key = k2; wl = l; }
                    }
                }
                const int R = __builtin_amdgcn_readlane(br, wl);
                // tentative add (microcluster.py:213-233) with lane = dimension, then the radius test (:334-337 / :378-381)
                const double w1 = Lw[R] + 1.0;
                double c1 = 0.0, c2 = 0.0, qb = 0.0, pr = 1.0, term = 0.0;
                if (lane < d) {
                    c1 = Lcf1[lane * cap + R] + myp;
                    c2 = Lcf2[lane * cap + R] + myp * myp;
                    const double qa = c2 / w1;
                    qb = c1 / w1;
                    const double var = qa - qb * qb;
                    pr = (var <= par.delta_sq) ? par.k : 1.0;
                    term = scaled(var, op_of(pr));  // mc_functions.py:52: var / pref'
                }
                double r2 = 0.0;
#pragma nounroll
                for (int i = 0; i < d; ++i) r2 = r2 + cc_readlane_f64(term, i);  // mc_functions.py:54, left to right
                if (!(r2 <= par.eps_sq)) continue;
                if (lane < d) {
                    Lcf1[lane * cap + R] = c1; Lcf2[lane * cap + R] = c2; Lcen[lane * cap + R] = qb; Lop[lane * cap + R] = op_of(pr);
                }
                if (lane == 0) Lw[R] = w1;
                target = R;
                path = stage;
                if (stage == 1) {
                    // hddstream.py:416-430
                    const int gt1 = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < d && pr > 1.0));
                    if (w1 >= par.beta_mu && gt1 <= par.pi) {
                        promoted = true;
                        // out of the outlier list (the last entry takes its place), onto the pcore list
                        if (lane == 0) {
                            Lkind[R] = CC_KIND_PCORE; Lkey[R] = n_pkeys; Lid[R] = pcore_last_id;
                            int at = 0;
                            while (Lolist[at] != R) ++at;
                            Lolist[at] = Lolist[n_o - 1];
                            Lplist[n_p] = R;
                        }
                        n_o -= 1;
                        n_p += 1;
                        n_pkeys += 1;
                        pcore_last_id += 1;
                    }
                }
                CC_WAVE_SYNC();  // the row as committed is what the next point sees
            }
            if (target < 0) {
                // hddstream.py:434-462: a new outlier MC holding this point (an add to an empty MC)
                if (M >= cap) { full = true; break; }  // the LDS image is full: the host continues with the windowed path
                const int R = M;
                if (lane < d) {
                    const double c1 = 0.0 + myp, c2 = 0.0 + myp * myp;
                    const double qa = c2 / 1.0, qb = c1 / 1.0;
                    const double var = qa - qb * qb;
                    Lcf1[lane * cap + R] = c1; Lcf2[lane * cap + R] = c2; Lcen[lane * cap + R] = qb;
                    Lop[lane * cap + R] = op_of((var <= par.delta_sq) ? par.k : 1.0);
                }
                if (lane == 0) {
                    Lw[R] = 0.0 + 1.0; Lkind[R] = CC_KIND_OUTLIER; Lkey[R] = n_okeys; Lid[R] = outlier_last_id; Luid[R] = outlier_last_id;
                    Lolist[n_o] = R;
                }
                n_o += 1;
                n_okeys += 1;
                outlier_last_id += 1;
                M += 1;
                target = R;
                path = 2;
                CC_WAVE_SYNC();
            }
            if (lane == 0) {
                s_luid[jj] = Luid[target];
                s_lpath[jj] = path | (promoted ? 4 : 0);
            }
            cdone = jj + 1;
        }
        CC_WAVE_SYNC();
        if (lane < cdone) {
            lab_uid[cursor0 + c0 + lane] = s_luid[lane];
            lab_path[cursor0 + c0 + lane] = (int8_t)s_lpath[lane];
        }
        done = c0 + cdone;
    }
    CC_WAVE_SYNC();
    // the table image back to HBM (every column the windowed path reads, scl included)
    for (int r = lane; r < M; r += 64) {
        for (int i = 0; i < d; ++i) {
            const double op = Lop[i * cap + r];
            tab.cf1[(size_t)r * d + i] = Lcf1[i * cap + r]; tab.cf2[(size_t)r * d + i] = Lcf2[i * cap + r];
            tab.cen[(size_t)r * d + i] = Lcen[i * cap + r]; tab.pref[(size_t)r * d + i] = pref_of(op);
            tab.scl[(size_t)r * d + i] = op;
        }
        tab.w[r] = Lw[r]; tab.kind[r] = Lkind[r]; tab.key[r] = Lkey[r]; tab.id[r] = Lid[r]; tab.uid[r] = Luid[r];
    }
    if (lane == 0) {
        ctl->cursor = cursor0 + done;
        ctl->m_rows = M;
        ctl->n_pkeys = n_pkeys; ctl->n_okeys = n_okeys;
        ctl->pcore_last_id = pcore_last_id; ctl->outlier_last_id = outlier_last_id;
        ctl->window_seq += 1ull;  // stamps and carry marks of earlier windows are history
        ctl->mode = 0; ctl->car_n = 0;
        ctl->stat_seq_points += done;
        ctl->stat_seq_clk += clock64() - clk0;
        ctl->stat_seq_wall += wall_clock64() - wall0;
    }
}

// ---------------------------------------------------------------------------------
// Relaxed multi-GPU mode (events of a timepoint sharded over the ranks).  A super-step:
//   A  every rank runs the exact windowed path over its next mini-batch with no_create set: points join existing MCs
//      (of the table all ranks share at the start of the super-step), points nobody absorbs are set aside;
//   M  the changes of the existing rows are merged: delta = local - snapshot per row (k_rel_delta), summed over the
//      ranks (RCCL all-reduce), snapshot + sum written back with centroid / preferred dimensions recomputed
//      (k_rel_merge), promotions decided on the merged rows in row order (k_rel_promote);
//   B  the set-aside points of all ranks, in rank order, go through the exact path on every rank redundantly
//      (k_rel_collect, k_rel_gather_points, k_rel_scatter_labels), so new MCs are created once and identically.
// Not the reference's semantics: within a super-step a rank does not see the other ranks' adds.
// ---------------------------------------------------------------------------------

// delta[r][0..d) = CF1 change, [d..2d) = CF2 change, [2d] = weight change of row r during phase A
__global__ void k_rel_delta(Table tab, const double* __restrict__ s_cf1, const double* __restrict__ s_cf2,
                            const double* __restrict__ s_w, int m, int d, double* __restrict__ delta)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * d) return;
    const int r = e / d, i = e - r * d;
    double* o = delta + (size_t)r * (2 * d + 1);
    o[i] = tab.cf1[e] - s_cf1[e];
    o[d + i] = tab.cf2[e] - s_cf2[e];
    if (i == 0) o[2 * d] = tab.w[r] - s_w[r];
}

// rows that absorbed points on some rank: CF = snapshot + summed delta, then what add_new_point leaves behind
// (microcluster.py:117-165: centroid = CF1 / W, preferred dimensions from the variances); kind / key / id of every row
// back to the snapshot (promotions are decided on the merged rows by k_rel_promote)
__global__ void k_rel_merge(Table tab, const double* __restrict__ s_cf1, const double* __restrict__ s_cf2,
                            const double* __restrict__ s_w, const int* __restrict__ s_kind,
                            const int* __restrict__ s_key, const long long* __restrict__ s_id, int m, int d,
                            const double* __restrict__ delta, double delta_sq, double k, int pow2, double inv_k)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * d) return;
    const int r = e / d, i = e - r * d;
    const double* o = delta + (size_t)r * (2 * d + 1);
    const double dw = o[2 * d];
    if (i == 0) { tab.kind[r] = s_kind[r]; tab.key[r] = s_key[r]; tab.id[r] = s_id[r]; }
    if (dw == 0.0) {
        // untouched everywhere: the row as it was (stored centroid and preferred dimensions included)
        tab.cf1[e] = s_cf1[e]; tab.cf2[e] = s_cf2[e];
        if (i == 0) tab.w[r] = s_w[r];
        return;
    }
    const double w = s_w[r] + dw;
    const double c1 = s_cf1[e] + o[i], c2 = s_cf2[e] + o[d + i];
    const double qa = c2 / w, qb = c1 / w;
    const double var = qa - qb * qb;
    const double pr = (var <= delta_sq) ? k : 1.0;
    tab.cf1[e] = c1; tab.cf2[e] = c2; tab.cen[e] = qb; tab.pref[e] = pr;
    tab.scl[e] = pow2 ? (pr == 1.0 ? 1.0 : inv_k) : pr;
    if (i == 0) tab.w[r] = w;
}

// hddstream.py:416-430 on the merged rows, in row order: an outlier MC that absorbed a point in this super-step and
// now fulfils W >= beta * mu and count(pref > 1) <= pi becomes a pcore MC (next list position, next pcore id).
// One workgroup; the counters of the control block continue from the values at the start of the super-step.
__global__ __launch_bounds__(1024) void k_rel_promote(Ctl* __restrict__ ctl, Table tab, int m, int d,
                                                      const double* __restrict__ delta, double beta_mu, int pi,
                                                      int n_pkeys0, long long pcore_last_id0)
{
    __shared__ int s_cnt[1024];
    const int tid = threadIdx.x;
    const int per = (m + 1023) / 1024;
    const int r0 = tid * per, r1 = min(m, r0 + per);
    int mine = 0;
    for (int r = r0; r < r1; ++r) {
        bool up = false;
        if (tab.kind[r] == CC_KIND_OUTLIER && delta[(size_t)r * (2 * d + 1) + 2 * d] != 0.0 && tab.w[r] >= beta_mu) {
            int gt1 = 0;
            for (int i = 0; i < d; ++i) gt1 += (tab.pref[(size_t)r * d + i] > 1.0) ? 1 : 0;
            up = gt1 <= pi;
        }
        mine += up ? 1 : 0;
    }
    s_cnt[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 1024; ++t) { const int x = s_cnt[t]; s_cnt[t] = run; run += x; }
        ctl->n_pkeys = n_pkeys0 + run;
        ctl->pcore_last_id = pcore_last_id0 + run;
    }
    __syncthreads();
    int rank = s_cnt[tid];
    for (int r = r0; r < r1; ++r) {
        bool up = false;
        if (tab.kind[r] == CC_KIND_OUTLIER && delta[(size_t)r * (2 * d + 1) + 2 * d] != 0.0 && tab.w[r] >= beta_mu) {
            int gt1 = 0;
            for (int i = 0; i < d; ++i) gt1 += (tab.pref[(size_t)r * d + i] > 1.0) ? 1 : 0;
            up = gt1 <= pi;
        }
        if (up) {
            tab.kind[r] = CC_KIND_PCORE;
            tab.key[r] = n_pkeys0 + rank;
            tab.id[r] = pcore_last_id0 + rank;
            ++rank;
        }
    }
}

// the set-aside points (label -1) of [a, e) in ascending order: out[0] = their number, out[1 ..] = their indices.
// One workgroup, ordered compaction 1 024 points per pass.
__global__ __launch_bounds__(1024) void k_rel_collect(const long long* __restrict__ lab_uid, long long a, long long e,
                                                      int* __restrict__ out)
{
    __shared__ int s_wsum[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (long long p0 = a; p0 < e; p0 += 1024) {
        const long long j = p0 + tid;
        const int f = (j < e && lab_uid[j] == -1ll) ? 1 : 0;
        int incl = f;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        int base = s_base;
        for (int w = 0; w < wv; ++w) base += s_wsum[w];
        if (f) out[1 + base + incl - 1] = (int)j;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += s_wsum[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) out[0] = s_base;
}

__global__ void k_rel_gather_points(const double* __restrict__ x, const int* __restrict__ idx, int n, int d,
                                    double* __restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * d) return;
    const int q = e / d, i = e - q * d;
    out[e] = x[(size_t)idx[q] * d + i];
}

__global__ void k_rel_scatter_labels(const long long* __restrict__ g_uid, const int8_t* __restrict__ g_path,
                                     const int* __restrict__ idx, int n, long long* __restrict__ lab_uid,
                                     int8_t* __restrict__ lab_path)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    lab_uid[idx[q]] = g_uid[q];
    lab_path[idx[q]] = g_path[q];
}

// out[i] = sum over the ranks, in rank order, of in[r * count + i]  (the in-process transport's all-reduce)
__global__ void k_sum_ranks(const double* __restrict__ in, int world, size_t count, double* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double acc = in[i];
    for (int r = 1; r < world; ++r) acc = acc + in[(size_t)r * count + i];
    out[i] = acc;
}

// dimension-major copy of the points for the scan's coalesced loads: xt[i * n + r] = x[r * d + i]
__global__ void k_transpose_points(const double* __restrict__ x, double* __restrict__ xt, long long n, int d)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * d) return;
    const long long r = e / d;
    const int i = (int)(e - r * d);
    xt[(size_t)i * n + r] = x[e];
}

// NaN / Inf check of the uploaded points (cc_points_upload)
__global__ void k_check_finite(const double* __restrict__ x, long long n, int* __restrict__ bad)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    int b = 0;
    for (; i < n; i += stride) {
        const double v = x[i];
        b |= !(v - v == 0.0);
    }
    if (b) atomicOr(bad, 1);
}

// ---------------------------------------------------------------------------------
// MinMax scaling on the device (scaling/scaler.py:27-47 = scikit-learn's MinMaxScaler restated, see
// chronoclust_amd/scaling/scaler.py): HBM-bound elementwise passes and one column reduction.
// ---------------------------------------------------------------------------------

// per-column minimum / maximum ignoring NaN (np.nanmin / np.nanmax): one workgroup per (column, row chunk),
// coalescing is across the columns of a row (consecutive threads read consecutive doubles), partials in part[2][chunks][d]
__global__ __launch_bounds__(256) void k_col_minmax(const double* __restrict__ x, long long n, int d,
                                                    double* __restrict__ part, int chunks)
{
    // thread t handles column t % d of rows t / d, t / d + rows_per_pass, ...
    const int rows_per_pass = 256 / d > 0 ? 256 / d : 1;
    const int col = (d <= 256) ? (int)(threadIdx.x % d) : 0;
    const int rsub = (int)(threadIdx.x / d);
    const long long per = (n + chunks - 1) / chunks;
    const long long r0 = (long long)blockIdx.x * per, r1 = (r0 + per < n) ? r0 + per : n;
    double mn = CC_INF, mx = -CC_INF;
    if (rsub < rows_per_pass && d <= 256)
        for (long long r = r0 + rsub; r < r1; r += rows_per_pass) {
            const double v = x[r * d + col];
            mn = __builtin_fmin(mn, v);  // fmin / fmax return the non-NaN operand
            mx = __builtin_fmax(mx, v);
        }
    __shared__ double smn[256], smx[256];
    smn[threadIdx.x] = mn;
    smx[threadIdx.x] = mx;
    __syncthreads();
    if ((int)threadIdx.x < d && d <= 256) {
        for (int q = 1; q < rows_per_pass; ++q) {
            mn = __builtin_fmin(mn, smn[q * d + threadIdx.x]);
            mx = __builtin_fmax(mx, smx[q * d + threadIdx.x]);
        }
        part[(size_t)blockIdx.x * d + threadIdx.x] = mn;
        part[(size_t)(chunks + blockIdx.x) * d + threadIdx.x] = mx;
    }
}

// MinMaxScaler.transform: X * scale_ + min_ (two roundings, in place); inverse: (X - min_) / scale_
__global__ void k_scale_points(double* __restrict__ x, long long tot, int d, const double* __restrict__ scale,
                               const double* __restrict__ mn)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tot) return;
    const int c = (int)(e % d);
    double v = x[e];
    v = v * scale[c];
    v = v + mn[c];
    x[e] = v;
}

__global__ void k_unscale_points(const double* __restrict__ x, double* __restrict__ out, long long tot, int d,
                                 const double* __restrict__ scale, const double* __restrict__ mn)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tot) return;
    const int c = (int)(e % d);
    double v = x[e];
    v = v - mn[c];
    v = v / scale[c];
    out[e] = v;
}
