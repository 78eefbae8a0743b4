// cc_online.h — gfx950 kernels of the exact windowed online phase.
//
// Reference semantics (clustering/hddstream.py:220-237) are a strict per-point
// read-modify-write chain over the microcluster (MC) table.  The kernels below
// keep those semantics exactly while processing a window of B points at a time:
//
//   k_scan<DIRTY=false>  every window point against every MC of the window-start
//                        snapshot: per point the two best pcore and the two best
//                        outlier candidates by (projected distance, list order)
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
// produced; a window always commits at least its first point.
//
// Arithmetic: IEEE double, no contraction (-ffp-contract=off), sums over dimensions
// left to right as in utilities/mc_functions.py under numba.
#pragma once
#include <type_traits>

#include "cc_common.h"

#define CC_INF (__builtin_huge_val())

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

// x / pref with pref in {1.0, k}; when k is a power of two x * (1/k) is the same double
__device__ __forceinline__ double cc_div_pref(double x, double pref, const Ctl* c)
{
    if (pref == 1.0) return x;
    return (c->pow2 && pref == c->k) ? x * c->inv_k : x / pref;
}

// microcluster.py:213-233 + mc_functions.py:45-56: projected radius^2 of (base + point) with the
// preferred dimensions of the enlarged MC.  base_cf1 == nullptr means an empty MC.
// Also returns count(pref' > 1) and count(pref' != 1) of the enlarged MC.
__device__ inline double cc_tentative_radius(const double* bcf1, const double* bcf2, double bw, const double* p,
                                             int d, const Ctl* c, int* cnt_gt1, int* cnt_ne1)
{
    const double w1 = bw + 1.0;
    double r2 = 0.0;
    int g = 0, n = 0;
    for (int i = 0; i < d; ++i) {
        double x = p[i];
        double c1 = (bcf1 ? bcf1[i] : 0.0) + x;
        double c2 = (bcf2 ? bcf2[i] : 0.0) + x * x;
        double var = cc_sqvar(c1, c2, w1);
        double pr = (var <= c->delta_sq) ? c->k : 1.0;  // microcluster.py:109-114 (NaN -> 1.0)
        g += (pr > 1.0);
        n += (pr != 1.0);
        r2 = r2 + cc_div_pref(var, pr, c);
    }
    if (cnt_gt1) *cnt_gt1 = g;
    if (cnt_ne1) *cnt_ne1 = n;
    return r2;
}

// ---------------------------------------------------------------------------------
// k_scan: points (one or PT per lane, in registers) x MC rows (wave-uniform, staged in LDS)
// ---------------------------------------------------------------------------------

#define CC_SCAN_TM 16  // MC rows per LDS tile

template <int DP, int PT, bool POW2, bool DIRTY>
__global__ __launch_bounds__(64) void k_scan(const Ctl* __restrict__ ctl, const double* __restrict__ X, Rows rows,
                                             const Cand* __restrict__ clean, Cand* __restrict__ part, int S,
                                             int round)
{
    const int B = ctl->win_b;
    if (B == 0) return;
    if (DIRTY && ctl->fc[round - 1] >= B) return;  // already at a fixed point
    const int j0 = blockIdx.x * (64 * PT);
    if (j0 >= B) return;
    const int d = ctl->d;
    const int lane = threadIdx.x;
    const int seg = blockIdx.y;
    const int nrows = DIRTY ? B : ctl->m_rows;
    const int per = (nrows + S - 1) / S;
    const int r0 = seg * per;
    int r1 = min(nrows, r0 + per);
    if (DIRTY) r1 = min(r1, j0 + 64 * PT - 1);  // a version row i only matters to points j > i
    const long long cursor = ctl->cursor;
    const double inv_k = ctl->inv_k;
    const bool filter = ctl->filter != 0;

    __shared__ double s_c[CC_SCAN_TM][DP];
    __shared__ double s_s[CC_SCAN_TM][DP];
    __shared__ int s_kind[CC_SCAN_TM], s_key[CC_SCAN_TM], s_next[CC_SCAN_TM];

    double p[PT][DP];
    int jj[PT];
    bool valid[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        jj[t] = j0 + t * 64 + lane;
        valid[t] = jj[t] < B;
        const double* xp = X + (cursor + (valid[t] ? jj[t] : 0)) * d;
#pragma unroll
        for (int i = 0; i < DP; ++i) p[t][i] = (valid[t] && i < d) ? xp[i] : 0.0;
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
                bd[kd][t][r] = CC_INF;
                bk[kd][t][r] = CC_IDX_INF;
                bs[kd][t][r] = -1;
            }
            cap[kd][t] = CC_INF;
            if (DIRTY && valid[t]) cap[kd][t] = clean[(size_t)jj[t] * 4 + kd * 2 + 1].dist;
        }

    for (int rt = r0; rt < r1; rt += CC_SCAN_TM) {
        const int tm = min(CC_SCAN_TM, r1 - rt);
        __syncthreads();
        for (int e = lane; e < tm * DP; e += 64) {
            const int m = e / DP, i = e - m * DP;
            double c = 0.0, s = 1.0;
            if (i < d) {
                const size_t g = (size_t)(rt + m) * d + i;
                c = rows.cen[g];
                const double pr = rows.pref[g];
                s = POW2 ? (pr == 1.0 ? 1.0 : inv_k) : pr;
            }
            s_c[m][i] = c;
            s_s[m][i] = s;
        }
        if (lane < tm) {
            s_kind[lane] = rows.kind[rt + lane];
            s_key[lane] = rows.key[rt + lane];
            s_next[lane] = DIRTY ? rows.next[rt + lane] : 0;
        }
        __syncthreads();

        for (int m = 0; m < tm; ++m) {
            const int kind = __builtin_amdgcn_readfirstlane(s_kind[m]);
            if (kind == CC_KIND_DEAD) continue;
            const int rowg = rt + m;
            bool act[PT];
            double bound[PT];
            bool anyact = false;
#pragma unroll
            for (int t = 0; t < PT; ++t) {
                act[t] = valid[t];
                if (DIRTY) {
                    const int nx = __builtin_amdgcn_readfirstlane(s_next[m]);
                    act[t] = act[t] && rowg < jj[t] && jj[t] <= nx;
                    const double b1 = (kind == 0) ? bd[0][t][0] : bd[1][t][0];
                    const double cp = (kind == 0) ? cap[0][t] : cap[1][t];
                    bound[t] = b1 < cp ? b1 : cp;
                } else {
                    bound[t] = (kind == 0) ? bd[0][t][1] : bd[1][t][1];
                }
                if (!act[t]) bound[t] = -1.0;
                anyact = anyact || act[t];
            }
            if (__builtin_amdgcn_ballot_w64(anyact) == 0ull) continue;

            double acc[PT];
#pragma unroll
            for (int t = 0; t < PT; ++t) acc[t] = 0.0;
            bool alive = true;
#pragma unroll
            for (int i0 = 0; i0 < DP; i0 += 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = i0 + q;
                    if (i < DP) {
                        const double c = s_c[m][i];
                        const double s = s_s[m][i];
#pragma unroll
                        for (int t = 0; t < PT; ++t) {
                            double x = p[t][i] - c;   // mc_functions.py:37
                            x = x * x;                // :38
                            x = POW2 ? x * s : x / s; // :39
                            acc[t] = acc[t] + x;      // :41, left to right
                        }
                    }
                }
                if (i0 + 4 < DP) {
                    // terms are >= 0: once every point of the wave is past its bound this MC cannot
                    // enter any candidate list, whatever the remaining dimensions add
                    bool q = false;
#pragma unroll
                    for (int t = 0; t < PT; ++t) q = q || (acc[t] <= bound[t]);
                    if (__builtin_amdgcn_ballot_w64(q) == 0ull) {
                        alive = false;
                        break;
                    }
                }
            }
            if (!alive) continue;

            const int key = s_key[m];
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
                        cc_tentative_radius(rows.cf1 + (size_t)rowg * d, rows.cf2 + (size_t)rowg * d, rows.w[rowg],
                                            X + (cursor + jj[t]) * d, d, ctl, nullptr, &ne1);
                        if (ne1 > ctl->pi) return;
                    }
                    if (cand_less(acc[t], key, bd[K][t][0], bk[K][t][0])) {
                        bd[K][t][1] = bd[K][t][0]; bk[K][t][1] = bk[K][t][0]; bs[K][t][1] = bs[K][t][0];
                        bd[K][t][0] = acc[t]; bk[K][t][0] = key; bs[K][t][0] = rowg;
                    } else {
                        bd[K][t][1] = acc[t]; bk[K][t][1] = key; bs[K][t][1] = rowg;
                    }
                };
                if (kind == 0) consider(std::integral_constant<int, 0>{});
                else consider(std::integral_constant<int, 1>{});
            }
        }
    }

#pragma unroll
    for (int t = 0; t < PT; ++t) {
        if (!valid[t]) continue;
        if (DIRTY) {
            Cand* o = part + ((size_t)jj[t] * S + seg) * 2;
            o[0] = Cand{bd[0][t][0], bk[0][t][0], bs[0][t][0]};
            o[1] = Cand{bd[1][t][0], bk[1][t][0], bs[1][t][0]};
        } else {
            Cand* o = part + ((size_t)jj[t] * S + seg) * 4;
            o[0] = Cand{bd[0][t][0], bk[0][t][0], bs[0][t][0]};
            o[1] = Cand{bd[0][t][1], bk[0][t][1], bs[0][t][1]};
            o[2] = Cand{bd[1][t][0], bk[1][t][0], bs[1][t][0]};
            o[3] = Cand{bd[1][t][1], bk[1][t][1], bs[1][t][1]};
        }
    }
}

// ---------------------------------------------------------------------------------
// k_decide: one thread per window point
// ---------------------------------------------------------------------------------

__device__ inline void cc_top2_push(Cand& a, Cand& b, const Cand& x)
{
    if (x.slot < 0) return;
    if (a.slot < 0 || cand_less(x.dist, x.key, a.dist, a.key)) {
        b = a;
        a = x;
    } else if (b.slot < 0 || cand_less(x.dist, x.key, b.dist, b.key)) {
        b = x;
    }
}

__global__ __launch_bounds__(64) void k_decide(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                               Versions ver, const Cand* __restrict__ part, Cand* __restrict__ clean,
                                               const Cand* __restrict__ dpart, const int* __restrict__ Told,
                                               int* __restrict__ Tnew, int8_t* __restrict__ dpath, int S, int round)
{
    const int B = ctl->win_b;
    if (B == 0) return;
    if (round > 0 && ctl->fc[round - 1] >= B) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B) return;
    const int d = ctl->d;
    const Cand none = Cand{CC_INF, CC_IDX_INF, -1};

    Cand c[4];
    if (round == 0) {
        c[0] = c[1] = c[2] = c[3] = none;
        for (int s = 0; s < S; ++s) {
            const Cand* q = part + ((size_t)j * S + s) * 4;
            cc_top2_push(c[0], c[1], q[0]);
            cc_top2_push(c[0], c[1], q[1]);
            cc_top2_push(c[2], c[3], q[2]);
            cc_top2_push(c[2], c[3], q[3]);
        }
        for (int i = 0; i < 4; ++i) clean[(size_t)j * 4 + i] = c[i];
    } else {
        for (int i = 0; i < 4; ++i) c[i] = clean[(size_t)j * 4 + i];
    }
    Cand dv[2] = {none, none};  // best live version per kind
    if (round > 0) {
        Cand dummy = none;
        for (int s = 0; s < S; ++s) {
            const Cand* q = dpart + ((size_t)j * S + s) * 2;
            cc_top2_push(dv[0], dummy, q[0]);
            dummy = none;
            cc_top2_push(dv[1], dummy, q[1]);
            dummy = none;
        }
    }

    const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
    auto dirty = [&](int slot) -> bool {
        if (round == 0) return false;
        const unsigned long long t = tab.touch[slot];
        return (t >> 20) == stamp && (int)(t & 0xFFFFFull) < j;
    };

    const int M0 = ctl->m_rows;
    const double* p = X + (ctl->cursor + j) * d;
    int T = -1;
    int path = 2;
    for (int stage = 0; stage < 2 && T == -1; ++stage) {
        const Cand c1 = c[stage * 2], c2 = c[stage * 2 + 1], dd = dv[stage];
        int state;  // 0: no clean candidate, 1: cb is the exact clean best, 2: cb only bounds the clean best from below
        Cand cb = none;
        if (c1.slot < 0) state = 0;
        else if (!dirty(c1.slot)) { state = 1; cb = c1; }
        else if (c2.slot < 0) state = 0;
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
            else { T = CC_T_UNKNOWN; break; }
        }
        if (wkind == 0) continue;
        const double *bcf1, *bcf2;
        double bw;
        int target;
        if (wkind == 1) {
            bcf1 = tab.cf1 + (size_t)wrow * d; bcf2 = tab.cf2 + (size_t)wrow * d; bw = tab.w[wrow];
            target = wrow;
        } else {
            bcf1 = ver.cf1 + (size_t)wrow * d; bcf2 = ver.cf2 + (size_t)wrow * d; bw = ver.w[wrow];
            target = ver.tgt[wrow];
        }
        const double r2 = cc_tentative_radius(bcf1, bcf2, bw, p, d, ctl, nullptr, nullptr);  // hddstream.py:334-337
        if (r2 <= ctl->eps_sq) {
            T = target;
            path = stage;
        }
    }
    if (T == -1) {  // hddstream.py:434-462: new outlier MC, provisional id = rows-at-window-start + j
        T = M0 + j;
        path = 2;
    }
    Tnew[j] = T;
    dpath[j] = (int8_t)path;
    if (round > 0 && (T == CC_T_UNKNOWN || T != Told[j])) atomicMin(&ctl->fc[round], j);
}

// ---------------------------------------------------------------------------------
// k_chain: replay the claimed decisions per MC in arrival order
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_chain(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                               Versions ver, const int* __restrict__ T, int round)
{
    const int B = ctl->win_b;
    if (B == 0) return;
    if (ctl->fc[round - 1] >= B) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->last_round = round;
    extern __shared__ int sT[];
    for (int i = threadIdx.x; i < B; i += blockDim.x) sT[i] = T[i];
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B) return;
    const int t = sT[j];
    if (t == CC_T_UNKNOWN) {
        ver.kind[j] = CC_KIND_DEAD; ver.next[j] = j; ver.tgt[j] = t; ver.acc[j] = 0; ver.upg[j] = -1;
        return;
    }
    for (int i = j - 1; i >= 0; --i)
        if (sT[i] == t) return;  // an earlier point heads this chain and walks over j

    const int d = ctl->d;
    const int M0 = ctl->m_rows;
    const bool isnew = t >= M0;
    const bool valid_chain = !isnew || (t == M0 + j);  // a claim on a MC nobody creates any more is void
    const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
    if (!isnew) tab.touch[t] = (stamp << 20) | (unsigned long long)j;

    const double *bcf1 = nullptr, *bcf2 = nullptr, *bcen = nullptr, *bpref = nullptr;
    double bw = 0.0;
    int bkind = CC_KIND_OUTLIER, bkey = ctl->n_okeys + j, bupg = -1;
    if (!isnew) {
        bcf1 = tab.cf1 + (size_t)t * d; bcf2 = tab.cf2 + (size_t)t * d;
        bcen = tab.cen + (size_t)t * d; bpref = tab.pref + (size_t)t * d;
        bw = tab.w[t]; bkind = tab.kind[t]; bkey = tab.key[t];
    }
    int cur = j;
    while (true) {
        int nx = CC_IDX_INF;
        for (int i = cur + 1; i < B; ++i)
            if (sT[i] == t) { nx = i; break; }
        double* vcf1 = ver.cf1 + (size_t)cur * d; double* vcf2 = ver.cf2 + (size_t)cur * d;
        double* vcen = ver.cen + (size_t)cur * d; double* vpref = ver.pref + (size_t)cur * d;
        ver.tgt[cur] = t;
        if (!valid_chain) {
            ver.kind[cur] = CC_KIND_DEAD; ver.next[cur] = cur; ver.acc[cur] = 0; ver.upg[cur] = -1;
        } else {
            const double* p = X + (ctl->cursor + cur) * d;
            const double w1 = bw + 1.0;  // microcluster.py:147
            double r2 = 0.0;
            int gt1 = 0;
            for (int i = 0; i < d; ++i) {
                const double x = p[i];
                const double c1 = (bcf1 ? bcf1[i] : 0.0) + x;      // mc_functions.py:26
                const double c2 = (bcf2 ? bcf2[i] : 0.0) + x * x;  // :27
                const double var = cc_sqvar(c1, c2, w1);
                const double pr = (var <= ctl->delta_sq) ? ctl->k : 1.0;
                vcf1[i] = c1; vcf2[i] = c2;
                vcen[i] = c1 / w1;  // mc_functions.py:31-33
                vpref[i] = pr;
                gt1 += (pr > 1.0);
                r2 = r2 + cc_div_pref(var, pr, ctl);
            }
            const bool creates = isnew && cur == j;
            const bool ok = creates || (r2 <= ctl->eps_sq);
            if (ok) {
                ver.w[cur] = w1;
                // hddstream.py:416-430: promotion is only examined after an add to an existing outlier MC
                if (bkind == CC_KIND_OUTLIER && !creates && w1 >= ctl->beta_mu && gt1 <= ctl->pi) {
                    bkind = CC_KIND_PCORE; bkey = ctl->n_pkeys + cur; bupg = cur;
                }
            } else {
                for (int i = 0; i < d; ++i) {
                    vcf1[i] = bcf1[i]; vcf2[i] = bcf2[i]; vcen[i] = bcen[i]; vpref[i] = bpref[i];
                }
                ver.w[cur] = bw;
            }
            ver.kind[cur] = bkind; ver.key[cur] = bkey; ver.upg[cur] = bupg; ver.acc[cur] = ok ? 1 : 0;
            ver.next[cur] = nx;
            bcf1 = vcf1; bcf2 = vcf2; bcen = vcen; bpref = vpref;
            bw = ver.w[cur];
        }
        if (nx == CC_IDX_INF) break;
        cur = nx;
    }
}

// ---------------------------------------------------------------------------------
// k_commit: write the validated prefix back and open the next window (one workgroup)
// ---------------------------------------------------------------------------------

__device__ inline void cc_block_exclusive_scan(int* a, int n, int* total, int* scratch)
{
    // a[0..n) -> exclusive prefix sums in place; blockDim.x threads, scratch[blockDim.x]
    const int tid = threadIdx.x, nt = blockDim.x;
    const int chunk = (n + nt - 1) / nt;
    const int lo = min(n, tid * chunk), hi = min(n, lo + chunk);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    scratch[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < nt; ++i) { int v = scratch[i]; scratch[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    int run = scratch[tid];
    for (int i = lo; i < hi; ++i) { int v = a[i]; a[i] = run; run += v; }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_commit(Ctl* __restrict__ ctl, Table tab, Versions ver,
                                                 const int* __restrict__ Tbuf0, const int* __restrict__ Tbuf1,
                                                 const int8_t* __restrict__ dpath, long long* __restrict__ lab_uid,
                                                 int8_t* __restrict__ lab_path)
{
    const int B = ctl->win_b;
    if (B == 0) return;
    extern __shared__ int sm[];
    int* rnew = sm;          // [B]
    int* rup = sm + B;       // [B]
    int* scratch = sm + 2 * B;  // [blockDim.x]
    __shared__ int tot_new, tot_up;
    const int r = ctl->last_round;
    const int* T = ((r - 1) & 1) ? Tbuf1 : Tbuf0;
    const int fcv = ctl->fc[r];
    const int n = fcv < B ? fcv : B;
    const int d = ctl->d;
    const int M0 = ctl->m_rows;
    const int tid = threadIdx.x, nt = blockDim.x;

    for (int j = tid; j < B; j += nt) {
        rnew[j] = (j < n && T[j] == M0 + j) ? 1 : 0;
        rup[j] = (j < n && ver.upg[j] == j) ? 1 : 0;
    }
    __syncthreads();
    cc_block_exclusive_scan(rnew, B, &tot_new, scratch);
    cc_block_exclusive_scan(rup, B, &tot_up, scratch);

    const long long cursor = ctl->cursor;
    const long long oid0 = ctl->outlier_last_id, pid0 = ctl->pcore_last_id;
    const int pk0 = ctl->n_pkeys, ok0 = ctl->n_okeys;
    for (int j = tid; j < n; j += nt) {
        const int t = T[j];
        lab_uid[cursor + j] = (t < M0) ? tab.uid[t] : oid0 + rnew[t - M0];
        lab_path[cursor + j] = (int8_t)(dpath[j] | ((ver.upg[j] == j) ? 4 : 0));
    }
    __syncthreads();  // tab.uid reads above precede the row writes below
    // scalar columns of the last version of every touched MC
    for (int j = tid; j < n; j += nt) {
        if (ver.next[j] < n) continue;
        const int t = T[j];
        const int row = (t < M0) ? t : M0 + rnew[t - M0];
        tab.w[row] = ver.w[j];
        tab.kind[row] = ver.kind[j];
        const int u = ver.upg[j];
        if (u >= 0) {
            tab.key[row] = pk0 + rup[u];
            tab.id[row] = pid0 + rup[u];
        } else if (t >= M0) {
            tab.key[row] = ok0 + rnew[t - M0];
            tab.id[row] = oid0 + rnew[t - M0];
        }
        if (t >= M0) tab.uid[row] = oid0 + rnew[t - M0];
    }
    // vector columns
    for (int e = tid; e < n * d; e += nt) {
        const int j = e / d, i = e - j * d;
        if (ver.next[j] < n) continue;
        const int t = T[j];
        const size_t row = (t < M0) ? (size_t)t : (size_t)(M0 + rnew[t - M0]);
        tab.cf1[row * d + i] = ver.cf1[(size_t)j * d + i];
        tab.cf2[row * d + i] = ver.cf2[(size_t)j * d + i];
        tab.cen[row * d + i] = ver.cen[(size_t)j * d + i];
        tab.pref[row * d + i] = ver.pref[(size_t)j * d + i];
    }
    __syncthreads();
    if (tid == 0) {
        ctl->m_rows = M0 + tot_new;
        ctl->n_okeys = ok0 + tot_new;
        ctl->outlier_last_id = oid0 + tot_new;
        ctl->n_pkeys = pk0 + tot_up;
        ctl->pcore_last_id = pid0 + tot_up;
        ctl->cursor = cursor + n;
        ctl->stat_windows += 1;
        ctl->stat_rounds += r;
        ctl->stat_truncated += (n < B) ? 1 : 0;
        ctl->stat_table_rows += M0;
        ctl->stat_pair_rows += (double)B * (double)M0;
        // next window
        ctl->window_seq += 1;
        long long left = ctl->n_points - (cursor + n);
        ctl->win_b = (int)(left < (long long)ctl->win_cfg ? left : (long long)ctl->win_cfg);
        ctl->last_round = 0;
        ctl->fc[0] = 0;
        for (int i = 1; i < CC_MAX_ROUNDS + 2; ++i) ctl->fc[i] = CC_IDX_INF;
    }
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
