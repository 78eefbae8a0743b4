// cc_offline.h — gfx950 kernels of the timestep boundary (decay / downgrade), the offline PreDeCon phase
// and the association tracker.  All of them are embarrassingly parallel given fixed inputs; the only ordered
// parts (Python list mutation while iterating, the BFS expansion) run on the host over small integer arrays.
#pragma once
#include "cc_common.h"

// ---------------------------------------------------------------------------------
// K4: decay (hddstream.py:283-286) and the downgrade / delete predicates (:529-532, :547)
// ---------------------------------------------------------------------------------

__global__ void k_decay(Table tab, int m_rows, int d, double f)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < m_rows * d) {
        tab.cf1[e] = tab.cf1[e] * f;
        tab.cf2[e] = tab.cf2[e] * f;
    }
    if (e < m_rows) tab.w[e] = tab.w[e] * f;
}

// flags[r] bit 0: would be downgraded as a pcore (w < beta*mu or count(pref > 1) > pi)
//          bit 1: would be deleted as an outlier (w <= omicron)
__global__ void k_downgrade_flags(Table tab, int m_rows, int d, double beta_mu, int pi, double omicron, int* flags)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m_rows) return;
    int cnt = 0;
    for (int i = 0; i < d; ++i) cnt += (tab.pref[(size_t)r * d + i] > 1.0);
    const double w = tab.w[r];
    flags[r] = ((w < beta_mu || cnt > pi) ? 1 : 0) | ((w <= omicron) ? 2 : 0);
}

// scl column from pref (run before every online phase: covers injected rows and parameter changes)
// ... and Ctl::cen_absmax, the largest |centroid coordinate| of the table as it is now (the host zeroes the word first)
__global__ __launch_bounds__(256) void k_rebuild_scl(Ctl* __restrict__ ctl, Table tab, int m_rows, int d, int pow2, double inv_k)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double a = 0.0;
    if (e < m_rows * d) {
        const double pr = tab.pref[e];
        tab.scl[e] = pow2 ? (pr == 1.0 ? 1.0 : inv_k) : pr;
        a = __builtin_fabs(tab.cen[e]);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(a, off);
        a = o > a ? o : a;
    }
    if ((threadIdx.x & 63) == 0 && a > 0.0) atomicMax(&ctl->cen_absmax, (unsigned long long)__double_as_longlong(a));
}

// dst row i <- src row perm[i]; kind / key / id are rewritten from the host-computed lists
__global__ void k_gather_rows(Table src, Table dst, const int* perm, const int* nkind, const int* nkey,
                              const long long* nid, int n, int d)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * d) {
        const int i = e / d, c = e - i * d;
        const size_t s = (size_t)perm[i] * d + c;
        dst.cf1[e] = src.cf1[s];
        dst.cf2[e] = src.cf2[s];
        dst.cen[e] = src.cen[s];
        dst.pref[e] = src.pref[s];
        dst.scl[e] = src.scl[s];
    }
    if (e < n) {
        const int s = perm[e];
        dst.w[e] = src.w[s];
        dst.uid[e] = src.uid[s];
        dst.kind[e] = nkind[e];
        dst.key[e] = nkey[e];
        dst.id[e] = nid[e];
        dst.touch[e] = 0ull; dst.touch[dst.cap + e] = 0ull;
        dst.last[e] = 0ull; dst.last[dst.cap + e] = 0ull;
    }
}

// ---------------------------------------------------------------------------------
// offline phase on the pcore list (dense copies in list order)
// ---------------------------------------------------------------------------------

struct PcoreView {
    double* cf1;
    double* cf2;
    double* cen;
    double* pref;
    double* w;
    long long* id;
};

__global__ void k_gather_pcores(Table tab, PcoreView pv, const int* rows, int mp, int d)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < mp * d) {
        const int i = e / d, c = e - i * d;
        const size_t s = (size_t)rows[i] * d + c;
        pv.cf1[e] = tab.cf1[s];
        pv.cf2[e] = tab.cf2[s];
        pv.cen[e] = tab.cen[s];
        pv.pref[e] = tab.pref[s];
    }
    if (e < mp) {
        pv.w[e] = tab.w[rows[e]];
        pv.id[e] = tab.id[rows[e]];
    }
}

// K5: mc_functions.py:64-77 with the thresholds of hddstream.py:489
__global__ void k_core_flags(PcoreView pv, int mp, int d, double eps_sq, double mu, int pi, double k, double inv_k,
                             int pow2, int8_t* core)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= mp) return;
    const double w = pv.w[r];
    double r2 = 0.0;
    int cnt = 0;
    for (int i = 0; i < d; ++i) {
        const size_t g = (size_t)r * d + i;
        double a = pv.cf2[g] / w;
        double b = pv.cf1[g] / w;
        b = b * b;
        double v = a - b;
        const double pr = pv.pref[g];
        if (pr != 1.0) v = (pow2 && pr == k) ? v * inv_k : v / pr;
        r2 = r2 + v;
        cnt += (pr > 1.0);
    }
    core[r] = (r2 <= eps_sq && w >= mu && cnt <= pi) ? 1 : 0;
}

// K6: predecon.py:161-188.  Lane = q with its centroid in registers; a workgroup is four waves = four words of 64 q's and
// walks a chunk of CC_EPS_PCH p rows, staged 32 at a time in LDS with coalesced loads and read back as broadcasts (the
// shape of k_scan and k_assoc_tiled: per (p, q, dim) a subtraction, a square and an addition, one LDS broadcast per two
// dimensions and wave).  ballot -> one word of the adjacency bitmask per (p, 64 q).
// Euclidean distance = sqrt of the left-to-right sum of squares (the reference's np.linalg.norm is
// platform-defined in the last ulp: nrm2 under numba, sqrt(dot) under numpy).
// (p_base / p_end: the p rows of this launch - on the multi-GPU path a rank takes a block of p rows)
#define CC_EPS_PCH 256  // p rows per workgroup on large tables; small ones take fewer (`pch`), so that the launch fills the machine
#define CC_EPS_TP 32

template <int DP>
__global__ __launch_bounds__(256) void k_eps_neighbours(const double* __restrict__ cen, int mp, int d, double eps,
                                                        unsigned long long* __restrict__ adj, int words, int p_base,
                                                        int p_end, int pch)
{
    __shared__ __attribute__((aligned(16))) double s_p[CC_EPS_TP * DP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qword = blockIdx.x * 4 + wv;
    const int q = qword * 64 + lane;
    const bool qvalid = qword < words && q < mp;
    double cq[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) cq[i] = (qvalid && i < d) ? cen[(size_t)q * d + i] : 0.0;
    const int p0 = p_base + blockIdx.y * pch;
    const int p1 = min(p_end, p0 + pch);
    for (int pt = p0; pt < p1; pt += CC_EPS_TP) {
        const int tp = min(CC_EPS_TP, p1 - pt);
        __syncthreads();
        for (int e = threadIdx.x; e < CC_EPS_TP * DP; e += 256) {
            const int m = e / DP, i = e - m * DP;
            s_p[e] = (m < tp && i < d) ? cen[(size_t)(pt + m) * d + i] : 0.0;
        }
        __syncthreads();
        if (qword >= words) continue;  // (a wave without a word of its own still helps staging)
        for (int m = 0; m < tp; ++m) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < DP; ++i) {
                double t = cq[i] - s_p[m * DP + i];  // predeconmc_functions.py:4-17 (padded dimensions: 0 - 0)
                t = t * t;
                acc = acc + t;
            }
            const bool in = qvalid && sqrt(acc) <= eps;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
            if (lane == 0) adj[(size_t)(pt + m) * words + qword] = mask;
        }
    }
}

// K7: predecon.py:190-217 + predeconmc_functions.py:19-42.  One thread per (p, dim): mean squared deviation of
// the neighbours' centroids, summed in dict (= list) order.  Note `<= delta`, not delta^2 (predecon.py:213).
__global__ void k_subspace_pref(const double* __restrict__ cen, const unsigned long long* __restrict__ adj,
                                int words, int mp, int d, double delta, double k, double* __restrict__ wvec,
                                int* __restrict__ nn, int p_base, int p_end)
{
    const int e = p_base * d + blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p_end * d) return;
    const int p = e / d, c = e - p * d;
    const double cp = cen[e];
    double acc = 0.0;
    int n = 0;
    for (int wd = 0; wd < words; ++wd) {
        unsigned long long m = adj[(size_t)p * words + wd];
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const int q = wd * 64 + b;
            double t = cp - cen[(size_t)q * d + c];
            t = t * t;
            acc = acc + t;
            ++n;
        }
    }
    const double var = acc / (double)n;
    wvec[e] = (var <= delta) ? k : 1.0;
    if (c == 0) nn[p] = n;
}

__global__ void k_pdim(const double* __restrict__ wvec, int mp, int d, int* __restrict__ pdim)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= mp) return;
    int cnt = 0;
    for (int i = 0; i < d; ++i) cnt += (wvec[(size_t)p * d + i] > 1.0);  // predecon_mc.py:81
    pdim[p] = cnt;
}

// K8: predecon.py:155-159, 219-239 + predeconmc_functions.py:44-62 on the eps-neighbour pairs.  One wavefront per p:
// the words of p's eps-neighbour row are read 64 at a time (coalesced) and only the non-empty ones - a handful per row
// when the MCs are well separated - are evaluated, lane = q; every word of the output row is written (zeros included).
__global__ __launch_bounds__(64) void k_weighted_reach(const double* __restrict__ cen, const double* __restrict__ wvec,
                                                       const unsigned long long* __restrict__ adj,
                                                       unsigned long long* __restrict__ adjw, int words, int mp,
                                                       int d, double eps_sq, int p_base, int p_end)
{
    const int p = p_base + blockIdx.x;
    if (p >= p_end) return;
    const int lane = threadIdx.x;
    for (int w0 = 0; w0 < words; w0 += 64) {
        const int wd = w0 + lane;
        const unsigned long long nb = (wd < words) ? adj[(size_t)p * words + wd] : 0ull;
        unsigned long long res = 0ull;
        for (unsigned long long todo = __builtin_amdgcn_ballot_w64(nb != 0ull); todo; todo &= todo - 1ull) {
            const int l = __builtin_ctzll(todo);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(nb & 0xFFFFFFFFull), l);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(nb >> 32), l);
            const unsigned long long bits = ((unsigned long long)hi << 32) | lo;
            const int q = (w0 + l) * 64 + lane;
            bool in = false;
            if (q < mp && ((bits >> lane) & 1ull)) {
                double dpq = 0.0, dqp = 0.0;
                for (int i = 0; i < d; ++i) {
                    const double a = cen[(size_t)p * d + i], b = cen[(size_t)q * d + i];
                    double t = a - b;
                    t = t * t;
                    t = wvec[(size_t)p * d + i] * t;
                    dpq = dpq + t;
                    double u = b - a;
                    u = u * u;
                    u = wvec[(size_t)q * d + i] * u;
                    dqp = dqp + u;
                }
                const double dist = dpq > dqp ? dpq : dqp;  // Python max(a, b): b only if b > a
                in = dist <= eps_sq;
            }
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
            if (lane == l) res = mask;
        }
        if (wd < words) adjw[(size_t)p * words + wd] = res;
    }
}

// The weighted-reachability rows as neighbour lists (what the ordered expansion on the host walks): per p the number
// of set bits, then - offsets being the host's prefix sums - the set bit positions in ascending order.  One wavefront
// per p in both kernels.
__global__ __launch_bounds__(64) void k_adj_counts(const unsigned long long* __restrict__ adjw, int words, int mp,
                                                   int* __restrict__ cnt)
{
    const int p = blockIdx.x;
    if (p >= mp) return;
    int c = 0;
    for (int wd = threadIdx.x; wd < words; wd += 64) c += __builtin_popcountll(adjw[(size_t)p * words + wd]);
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    if (threadIdx.x == 0) cnt[p] = c;
}

__global__ __launch_bounds__(64) void k_adj_fill(const unsigned long long* __restrict__ adjw, int words, int mp,
                                                 const long long* __restrict__ off, int* __restrict__ nbr)
{
    const int p = blockIdx.x;
    if (p >= mp) return;
    const int lane = threadIdx.x;
    long long base = off[p];
    for (int w0 = 0; w0 < words; w0 += 64) {
        const int wd = w0 + lane;
        unsigned long long m = (wd < words) ? adjw[(size_t)p * words + wd] : 0ull;
        const int mine = __builtin_popcountll(m);
        int incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        long long pos = base + incl - mine;
        while (m) {
            nbr[pos++] = wd * 64 + __builtin_ctzll(m);
            m &= m - 1ull;
        }
        base += __shfl(incl, 63);
    }
}

// predecon_mc.py:50-68 merge_mc in merge order + predecon.py:80 update_preferred_dimensions.
// One thread per (cluster, dim); members[off[c] .. off[c+1]) are pcore list positions in merge order.
__global__ void k_cluster_merge(PcoreView pv, const int* __restrict__ members, const int* __restrict__ off, int nc,
                                int d, double delta_sq, double k, double* __restrict__ ccf1, double* __restrict__ ccf2,
                                double* __restrict__ ccen, double* __restrict__ cpref, double* __restrict__ cw)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nc * d) return;
    const int c = e / d, i = e - c * d;
    double s1 = 0.0, s2 = 0.0, w = 0.0;
    for (int t = off[c]; t < off[c + 1]; ++t) {
        const int m = members[t];
        s1 = s1 + pv.cf1[(size_t)m * d + i];
        s2 = s2 + pv.cf2[(size_t)m * d + i];
        w = w + pv.w[m];
    }
    ccf1[e] = s1;
    ccf2[e] = s2;
    ccen[e] = s1 / w;
    double a = s2 / w;
    double b = s1 / w;
    b = b * b;
    const double var = a - b;
    cpref[e] = (var <= delta_sq) ? k : 1.0;
    if (i == 0) cw[c] = w;
}

// ---------------------------------------------------------------------------------
// K9: association tracker, cluster_tracker.py:127-141: for every current pcore the previous pcore with the smallest
// sum_d (prev - cur)^2 / cur_pref, strict <, first minimum wins.
//
// Lane = one current pcore, its centroid and distance operands in registers; the previous centroids are wave-uniform:
// a workgroup (4 waves = 4 tiles of 64 current pcores) stages 32 of them at a time in LDS with coalesced loads and
// every lane walks them in ascending order (broadcast reads), so the running minimum needs no cross-lane step and
// "first minimum wins" is the loop order.  The previous pcores are split into gridDim.y sub-ranges for parallelism;
// k_assoc_merge folds the partial minima in sub-range order.  UNIT: every preference entry is 1 or k with k a power
// of two (x / k == x * (1/k) bit for bit, the operand is 1 or 1/k); otherwise the operand is the entry itself and
// the term is divided.
// ---------------------------------------------------------------------------------

#define CC_ASSOC_TQ 32

template <int DP, bool UNIT>
__global__ __launch_bounds__(256) void k_assoc_tiled(const double* __restrict__ cur_cen, const double* __restrict__ cur_op,
                                                     const double* __restrict__ prev_cen, int mc, int mp, int d, int c_lo,
                                                     int c_hi, double* __restrict__ part_dist, int* __restrict__ part_idx)
{
    __shared__ __attribute__((aligned(16))) double s_q[CC_ASSOC_TQ * DP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = c_lo + (blockIdx.x * 4 + wv) * 64 + lane;
    const bool valid = c < c_hi;
    double cc[DP], op[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        cc[i] = (valid && i < d) ? cur_cen[(size_t)c * d + i] : 0.0;
        op[i] = (valid && i < d) ? cur_op[(size_t)c * d + i] : 1.0;
    }
    const int S = gridDim.y;
    const int per = (mp + S - 1) / S;
    const int q0 = blockIdx.y * per, q1 = min(mp, q0 + per);
    double best = __builtin_huge_val();
    int bidx = -1;
    for (int qt = q0; qt < q1; qt += CC_ASSOC_TQ) {
        const int tq = min(CC_ASSOC_TQ, q1 - qt);
        __syncthreads();
        for (int e = threadIdx.x; e < CC_ASSOC_TQ * DP; e += 256) {
            const int m = e / DP, i = e - m * DP;
            s_q[e] = (m < tq && i < d) ? prev_cen[(size_t)(qt + m) * d + i] : 0.0;
        }
        __syncthreads();
        for (int m = 0; m < tq; ++m) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < DP; ++i) {
                double t = s_q[m * DP + i] - cc[i];  // padded dimensions: 0 - 0, operand 1: the terms add +0.0
                t = t * t;
                t = UNIT ? t * op[i] : ((op[i] != 1.0) ? t / op[i] : t);
                acc = acc + t;
            }
            if (bidx < 0 || acc < best) {  // ascending q: strict < keeps the first minimum
                best = acc;
                bidx = qt + m;
            }
        }
    }
    if (valid) {
        part_dist[(size_t)blockIdx.y * mc + c] = best;
        part_idx[(size_t)blockIdx.y * mc + c] = bidx;
    }
}

__global__ void k_assoc_merge(const double* __restrict__ part_dist, const int* __restrict__ part_idx, int S, int mc,
                              int c_lo, int c_hi, int* __restrict__ out_idx, double* __restrict__ out_dist)
{
    const int c = c_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= c_hi) return;
    double best = __builtin_huge_val();
    int bidx = -1;
    for (int s = 0; s < S; ++s) {  // sub-ranges in ascending order of q: strict < keeps the first minimum
        const int i = part_idx[(size_t)s * mc + c];
        const double v = part_dist[(size_t)s * mc + c];
        if (i >= 0 && (bidx < 0 || v < best)) {
            best = v;
            bidx = i;
        }
    }
    out_idx[c] = bidx;
    if (out_dist) out_dist[c] = best;
}

// cc_point_clusters: out[i] = cluster index of the microcluster whose creation number labels point i, or -1
// (app.py:303-332 walks the per-microcluster `points` dicts of every cluster's pcores for this join)
__global__ void k_point_clusters(const long long* __restrict__ lab_uid, long long n, const int* __restrict__ map,
                                 long long n_uid, int* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long u = lab_uid[i];
    out[i] = (u >= 0 && u < n_uid) ? map[u] : -1;
}
