// chronoclust_amd/csrc: kernels of the relaxed multi-GPU mode (k_rel_*, k_sum_ranks).  (included by cc_online.h; one translation unit, cc_api.hip)
#pragma once

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
