// chronoclust_amd/csrc: round 0 that knows the window's own creators - k_link_scan, k_link_apply.  (included by cc_online.h; one translation unit, cc_api.hip)
//
// While microclusters are being created (the start of a stream, a new population) round 0 decides "create" for EVERY
// point of a population the snapshot does not know (hddstream.py:434-462 reached through :288-395): the first of them is
// right, the others would join the microcluster the first one creates.  Left to the validation rounds that takes three of
// them: round 1 sees the creators' version rows and retargets the others - to the NEAREST of their population's
// would-be creators, not the first -, round 2 retargets those whose target no longer creates anything, round 3 confirms
// (profiles/r06_tool_round_debug.txt: 214 such retargets in the first 17 000 points of C2 - every window took all three
// rounds, each with its chain replay, seeds, dirty scans and decisions).
// Here the points that decided "create" ("orphans") are linked among themselves right behind round 0: k_link_scan finds,
// per orphan j, the EARLIEST orphan i < j whose one-point microcluster would absorb j (the radius test of the pair,
// mc_functions.py:45-56 on CF1 = p_i + p_j, CF2 = p_i^2 + p_j^2, W = 2: variance (p_i - p_j)^2 / 4 per dimension,
// preferred where that is <= delta^2); k_link_apply follows those links to their root - the orphan that stays a creator -,
// rewrites the claim of j to "join the microcluster the root creates" and registers the claims of all orphans (k_decide
// leaves that to it).  A PREDICTION, like every claim of round 0: the validation rounds re-derive every decision from the
// live versions as before, so a wrong link costs a round, never a result.  On well-separated populations the links are
// right and a creation-phase window converges in its first validation round.
#pragma once

#define CC_LINK_SUB 64   // earlier window points per wave of k_link_scan (one pass of the staging area at d <= 20)

template <int DP>
struct LinkShape {
    static constexpr int ROWS = DP <= 20 ? 64 : (DP <= 40 ? 32 : 16);  // earlier points staged per pass (<= 10 KB per wave)
};

// grid (point tiles of 64, ceil(window / (4 * CC_LINK_SUB))), 256 threads: wave w of workgroup (x, y) compares the orphans of
// tile x with the orphans among the earlier points [(4 y + w) * CC_LINK_SUB, ...) - staged in LDS block by block, read back
// as broadcasts - and takes the first hit of each lane into near[j] by atomic minimum (the earliest over all waves).
template <int DP>
__global__ __launch_bounds__(256) void k_link_scan(const Ctl* __restrict__ ctl, const double* __restrict__ X,
                                                   const double* __restrict__ Xt, const int* __restrict__ T0,
                                                   int* __restrict__ near)
{
    CC_LATENCY_KERNEL();
    constexpr int ROWS = LinkShape<DP>::ROWS;
    const int B = ctl->win_b;
    if (B == 0 || ctl->no_create != 0) return;
    const int j0 = (int)blockIdx.x * 64;
    if (j0 >= B) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int M0 = ctl->m_rows;
    const int jj = j0 + lane;
    const bool orphan = jj < B && T0[jj] == M0 + jj;
    if (__builtin_amdgcn_ballot_w64(orphan) == 0ull) return;  // (the steady state: nobody creates anything)
    const int i_lo = ((int)blockIdx.y * 4 + wv) * CC_LINK_SUB;
    const int i_hi = min(i_lo + CC_LINK_SUB, min(B, j0 + 63));  // i < j <= j0 + 63
    if (i_lo >= i_hi) return;
    const int d = ctl->d;
    const long long cursor = ctl->cursor;
    const size_t n_pts = (size_t)ctl->xt_stride;
    double p[DP];
    {
        const double* xp = Xt + cursor + (jj < B ? jj : 0);
#pragma unroll
        for (int c = 0; c < DP; ++c) p[c] = (orphan && c < d) ? xp[(size_t)c * n_pts] : 0.0;
    }
    const double inv_k = 1.0 / ctl->k;
    const double thr_d = 4.0 * ctl->delta_sq, thr_e = 4.0 * ctl->eps_sq;
    __shared__ __attribute__((aligned(16))) double s_all[4 * ROWS * DP];
    double* const s = s_all + (size_t)wv * ROWS * DP;
    if (d < DP)  // (padded dimensions: zero terms)
        for (int q = lane; q < ROWS * DP; q += 64) s[q] = 0.0;
    int hit = CC_IDX_INF;
    for (int ib = i_lo; ib < i_hi; ib += ROWS) {
        const int e = ib + lane;
        const bool flag = lane < ROWS && e < i_hi && T0[e] == M0 + e;
        unsigned long long mask = __builtin_amdgcn_ballot_w64(flag);
        if (mask == 0ull) continue;
        if (__builtin_amdgcn_ballot_w64(orphan && hit == CC_IDX_INF && jj > ib) == 0ull) break;  // everybody has its link
        CC_WAVE_SYNC();
        {
            // the block's points are one contiguous run of the row-major copy
            const int nrow = min(ROWS, i_hi - ib);
            const double* src = X + (size_t)(cursor + ib) * d;
            for (int q = lane; q < nrow * d; q += 64) {
                const int m = q / d, c = q - m * d;
                s[m * DP + c] = src[q];
            }
        }
        CC_WAVE_SYNC();
        while (mask != 0ull) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1ull;
            const int i = ib + b;
            const bool act = orphan && hit == CC_IDX_INF && i < jj;
            if (__builtin_amdgcn_ballot_w64(act) == 0ull) continue;
            const double* r = s + b * DP;
            // four dimensions at a time; the terms are >= 0, so a pair whose partial sum is past the threshold is decided -
            // and two populations are apart in (nearly) every dimension: the lanes of a wave agree after the first four
            double acc = 0.0;
            bool open = true;
#pragma unroll
            for (int c0 = 0; c0 < DP; c0 += 4) {
                if (open) {
#pragma unroll
                    for (int c = c0; c < c0 + 4 && c < DP; ++c) {
                        double x = p[c] - r[c];
                        x = x * x;
                        acc += (x > thr_d) ? x : x * inv_k;
                    }
                    if (c0 + 4 < DP) open = __builtin_amdgcn_ballot_w64(act && acc <= thr_e) != 0ull;
                }
            }
            if (act && open && acc <= thr_e) hit = i;
        }
    }
    if (hit != CC_IDX_INF) atomicMin(&near[jj], hit);
}

// One thread per window point.  An orphan follows its links to the root (an orphan without a link: it stays a creator),
// claims the microcluster the root creates (provisional row M0 + root, path 1: an outlier microcluster absorbs it) and
// registers the claim for the round that replays it, in k_decide's formats (the creators' own claims as well: k_decide
// left the orphans to this kernel).
__global__ __launch_bounds__(256) void k_link_apply(Ctl* __restrict__ ctl, Table tab, int* __restrict__ T0,
                                                    const int* __restrict__ near, int8_t* __restrict__ dpath)
{
    CC_LATENCY_KERNEL();
    const int B = ctl->win_b;
    if (B == 0 || ctl->no_create != 0) return;
    const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= B) return;
    const int M0 = ctl->m_rows;
    if (T0[j] != M0 + j) return;
    int r = near[j];
    int t = M0 + j;
    if (r != CC_IDX_INF) {
        for (int steps = 0; steps < B; ++steps) {  // (links point backwards: the walk ends)
            const int up = near[r];
            if (up == CC_IDX_INF) break;
            r = up;
        }
        t = M0 + r;
        T0[j] = t;
        dpath[j] = (int8_t)1;
    }
    const unsigned long long stamp = ctl->window_seq * 16ull;  // round 0
    const unsigned long long sn = (stamp + 1ull) << 20;
    const size_t wr = tab.cap + (size_t)t;  // the copy round 1 reads
    atomicMax(&tab.touch[wr], sn | (unsigned long long)(0xFFFFF - j));
    atomicMax(&tab.last[wr], sn | (unsigned long long)j);
    const unsigned long long sc = (stamp + 1ull) << 24;
    unsigned long long* cw = tab.cnt + t;
    const unsigned long long old = __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int pos = CC_CHAIN_MEMB + 1;
    if ((old & ~0xFFFFFFull) == sc) {
        if ((int)(old & 0xFFFFFFull) <= CC_CHAIN_MEMB) pos = (int)(atomicAdd(cw, 1ull) & 0xFFFFFFull);
    } else if (atomicCAS(cw, old, sc | 1ull) == old) {
        pos = 0;
    } else {
        pos = (int)(atomicAdd(cw, 1ull) & 0xFFFFFFull);
    }
    if (pos < CC_CHAIN_MEMB) tab.memb[(size_t)t * CC_CHAIN_MEMB + pos] = j;
}
