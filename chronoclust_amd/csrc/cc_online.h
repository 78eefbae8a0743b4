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
//                     <.., PREP>: ahead of k_chain, the running sums of a pcore MC's long chain alone; k_chain's group
//                     of every member then evaluates that member's step (few long chains: one CU is not enough)
//   k_seq             no speculation: one wavefront walks the points in order on an LDS image of the table; the host
//                     switches to it while windows keep being cut short and it measures faster
//   k_seq_r           the same with the table in registers (d <= 4, a few hundred rows: the reference's own data)
//   k_seq_g           the same on the table in HBM, one workgroup of 1 024 threads: tables beyond k_seq's LDS image
//   k_claims_heavy    the claims of a microcluster that takes a large share of a window, gathered by one workgroup
//                     instead of three same-address atomics per claimant in k_decide (Table::heavy)
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

// class of a version / carried row (CC_DSQ_STRIDE) from its current kind and its stored dsq (negated: promoted since the
// snapshot), and the stored form of a displacement for a row whose snapshot kind was kind0
__device__ __forceinline__ int cc_dsq_class(int kind, double dsq_stored)
{
    return (kind == CC_KIND_PCORE) ? (__builtin_signbit(dsq_stored) ? 2 : 0) : 1;
}
// dq >= 0 or +inf (no bound) -> what Versions::dsq / Carry::dsq hold and the class whose maximum it enters.
// A row without a bound stays in the class of its kind (+inf is never below a threshold).
__device__ __forceinline__ double cc_dsq_store(double dq, int kind, int kind0, int* cls)
{
    const bool promoted = kind == CC_KIND_PCORE && kind0 == CC_KIND_OUTLIER && dq < CC_INF;
    *cls = (kind == CC_KIND_PCORE) ? (promoted ? 2 : 0) : 1;
    return promoted ? -dq : dq;
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

// The kernels of the online phase, by stage (all of them see the helpers above):
#include "cc_scan.h"      // snapshot scans and dirty scans
#include "cc_scan16.h"    // the pruned scan's prefix test on the matrix cores (k_prefix16, k_scan_p3)
#include "cc_validate.h"  // k_dseed, k_decide, k_claims, k_chain, k_chain_long, k_commit_a / b
#include "cc_link.h"      // round 0 that knows the window's own creators (k_link_scan, k_link_apply)
#include "cc_seq.h"       // the sequential kernel (table in LDS)
#include "cc_seq_r.h"     // ... and with the table in registers, for d <= 4 and a few hundred rows
#include "cc_seq_g.h"     // ... and on the table in HBM, one workgroup: tables beyond the LDS image
#include "cc_relaxed.h"   // relaxed multi-GPU mode
#include "cc_points.h"    // transposed copy, finiteness check, scaler
