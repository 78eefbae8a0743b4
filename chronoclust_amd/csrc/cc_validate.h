// chronoclust_amd/csrc: validation of a window - 32-lane group helpers, k_dseed, k_decide, k_claims, k_chain, k_chain_long, k_commit_a / k_commit_b.  (included by cc_online.h; one translation unit, cc_api.hip)
#pragma once

// ---------------------------------------------------------------------------------
// 32-lane groups: one group per window point in k_decide / k_chain.  Lane l owns dimensions l and l + 32
// (d <= 64); sums over dimensions stay strictly left to right through an ordered shuffle loop.
// ---------------------------------------------------------------------------------

#define CC_LSTAT_ROWS 1024  // words of `lstat` per round parity: one per workgroup of a k_chain_long launch (table rows <= 1 024, or list entries); two counters behind them

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
                                              Cand* __restrict__ seed, const int* __restrict__ T, int round,
                                              const int8_t* __restrict__ dpath, int* __restrict__ sparse_list, int sparse_cap)
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
    // (per class, see CC_DSQ_STRIDE)
    unsigned long long maxd[3] = {0ull, 0ull, 0ull}, maxd_car[3] = {0ull, 0ull, 0ull};
    {
        for (int i = threadIdx.x; i < (B + 15) / 16; i += 64) {
            const unsigned long long* w = ver.tile_dsq + (size_t)i * CC_DSQ_STRIDE;
            const unsigned long long v0 = w[0], v1 = w[1], v2 = w[2];
            maxd[0] = v0 > maxd[0] ? v0 : maxd[0];
            maxd[1] = v1 > maxd[1] ? v1 : maxd[1];
            maxd[2] = v2 > maxd[2] ? v2 : maxd[2];
        }
        if (la_mode)
            for (int i = threadIdx.x; i < (ctl->car_n + 15) / 16; i += 64) {
                const unsigned long long* w = car.tile_dsq + (size_t)i * CC_DSQ_STRIDE;
                const unsigned long long v0 = w[0], v1 = w[1], v2 = w[2];
                maxd_car[0] = v0 > maxd_car[0] ? v0 : maxd_car[0];
                maxd_car[1] = v1 > maxd_car[1] ? v1 : maxd_car[1];
                maxd_car[2] = v2 > maxd_car[2] ? v2 : maxd_car[2];
            }
#pragma unroll
        for (int K = 0; K < 3; ++K)
            for (int off = 32; off >= 1; off >>= 1) {
                const unsigned long long o = __shfl_xor(maxd[K], off), oc = __shfl_xor(maxd_car[K], off);
                maxd[K] = o > maxd[K] ? o : maxd[K];
                maxd_car[K] = oc > maxd_car[K] ? oc : maxd_car[K];
            }
    }
    bool flag_unprov = false, flag_unsafe = false, flag_n1skip = false;
    bool repeats = false;  // k_decide of this round would repeat this point's claim (see the end of the per-point block)
    int flag_word = 0;
    double tau_w[3] = {CC_INF, CC_INF, CC_INF};
    bool need_ver = false, need_car = false;  // this point's stages need rows only a dirty scan of the versions / the carry set covers
    if (j < B) {
    const Par par = cc_load_par(ctl);
    const int d = par.d;
    const bool filter = par.filter != 0;
    const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
    const double* p = X + (ctl->cursor + j) * d;
    Cand first0 = Cand{CC_INF, CC_IDX_INF, -1}, first1 = Cand{CC_INF, CC_IDX_INF, -1};
    double cap[2] = {CC_INF, CC_INF};
    // per candidate list (0 pcore, 1 outlier): lk_ok - the live versions of its entries were located; lbound - its first
    // place is a bound (a pruned scan whose threshold lay below every row of the kind: nothing is known of the list but
    // that bound, k_decide refuses the point if it has to evaluate that stage)
    bool lk_ok[2] = {par.k > 0.0, par.k > 0.0}, lbound[2] = {false, false};
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
        // a bound in FIRST place (guessed thresholds, k_scan_p): every row of the kind is at least that far
        lbound[kd] = cq[kd * 2].slot == CC_SLOT_BOUND;
        if (lbound[kd]) d2v[kd] = cq[kd * 2].dist;
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
            if (v1 < 0) lk_ok[kd] = false;
        }
        if (look[kd * 2 + 1] && lv[kd * 2 + 1] == -2) lk_ok[kd] = false;
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
    // Class 2 - a MC promoted since the snapshot (hddstream.py:416-430): it competes in the pcore list, but the snapshot
    // scanned it as an OUTLIER MC, so its snapshot distance is >= the second-best of the point's outlier list unless
    // it is one of that list's two entries (whose live versions are seeded above whatever their kind has become):
    //     sqrt(dsq_v) < sqrt(d2_outlier) - sqrt(K * ce_pcore)    rules it out.
    double tau[3];
    for (int kd = 0; kd < 3; ++kd) {
        double t;
        const int list = (kd == 1) ? 1 : 0;        // the list the row competes in
        const int from = (kd == 0) ? 0 : 1;        // the list its snapshot distance is bounded by
        if (list == 0 && filter) t = -CC_INF;
        else if (kd != 2 && !have1[kd]) t = CC_INF;  // no MC of this kind at window start: its versions have dsq = +inf
        else {
            const double fb = (list == 0) ? first0.dist : first1.dist;
            const double ce = fb < cap[list] ? fb : cap[list];
            // (d2 = +inf: the snapshot held no other MC of that kind; cap = +inf: nothing to beat yet - every row matters)
            t = (d2v[from] == CC_INF) ? CC_INF : (sqrt(d2v[from]) - sqrt(K * ce));
        }
        // no threshold for a list that is only a bound or whose entries' live versions were not all found; the promoted
        // rows' bound also rests on the outlier list's entries being seeded
        if (!lk_ok[list] || lbound[list] || (kd == 2 && !lk_ok[1])) t = -CC_INF;
        t = (t == CC_INF) ? CC_INF : t * (1.0 - 1e-9) - 1e-290;  // margin for the rounding of all of the above
        tau[kd] = t;
    }
    // Stage 1 (hddstream.py:345-395) is only reached when stage 0 fails.  A point whose previous decision was to join a
    // pcore MC is expected to do so again: the outlier-kind rows are not held against its tile, and k_decide refuses
    // the point (CC_T_UNKNOWN) if stage 0 fails after all and those rows were not covered.
    const bool need0 = true;
    const bool need1 = !(T[j] >= 0 && dpath[j] == 0);
    int flags = 0;
    if (!(cc_dsq_below(maxd[0], tau[0]) && cc_dsq_below(maxd[2], tau[2]))) flags |= CC_FLAG_U0;
    if (!cc_dsq_below(maxd[1], tau[1])) flags |= CC_FLAG_U1;
    if (la_mode && !(cc_dsq_below(maxd_car[0], tau[0]) && cc_dsq_below(maxd_car[2], tau[2]))) flags |= CC_FLAG_C0;
    if (la_mode && !cc_dsq_below(maxd_car[1], tau[1])) flags |= CC_FLAG_C1;
    need_ver = (need0 && (flags & CC_FLAG_U0)) || (need1 && (flags & CC_FLAG_U1));
    need_car = (need0 && (flags & CC_FLAG_C0)) || (need1 && (flags & CC_FLAG_C1));
    flag_unprov = !(lk_ok[0] && lk_ok[1]);
    flag_unsafe = need_ver || need_car;
    // (a point that goes on the sparse list keeps its real thresholds: its scans cover both kinds)
    flag_n1skip = !need1;
    flag_word = flags;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) tau_w[kd] = tau[kd];
    // Would k_decide of this round repeat the claim?  In the steady state it does for every point of the window, and the
    // launch (one 32-lane group per point re-reading what this thread holds) only confirms it.  The test below is
    // k_decide's own evaluation of stage 0 for a point whose tile's dirty scans do not run (its live versions are then
    // the seeds written above), restricted to the two ways it ends on the claimed MC with the radius test's verdict
    // already known from the chain (ver.acc, see run_stage); anything else - another stage, a flag, a bound in the
    // list, a verdict that has to be computed - leaves `repeats` false, and one such point makes k_decide run.
    {
        const int t = T[j];
        const Cand c1 = cq[0], c2 = cq[1];
        if (flags == 0 && t >= 0 && dpath[j] == 0 && c1.slot >= 0 && ver.tgt[j] == t && ver.acc[j] != 0) {
            const int head0 = 0xFFFFF - (int)(tcq[0] & 0xFFFFFull);
            const bool dirty0 = ((tcq[0] >> 20) == stamp && head0 < j) || (la_mode && (coq[0] >> 20) == wseq);
            if (!dirty0) {
                // the clean best stands for its MC: it wins unless the best live version beats it; it must be the claim
                repeats = c1.slot == t && !(first0.slot >= 0 && cand_less(first0.dist, first0.key, c1.dist, c1.key));
            } else if (first0.slot >= 0 && (c2.slot == -1 || cand_less(first0.dist, first0.key, c2.dist, c2.key))) {
                // the clean best was changed before j: the best live version has to beat whatever the second place
                // holds (an exact candidate, clean or not, or a bound), belong to the claimed MC and be the state
                // the chain added j to (its predecessor's version row, or the carried row)
                if (first0.slot >= CC_CAR_BASE) repeats = car.slot[first0.slot - CC_CAR_BASE] == t;
                else repeats = ver.tgt[first0.slot] == t && ver.next[first0.slot] == j;
            }
        }
    }
    }
    // Sparse dirty scans (sparse_cap > 0: the host launches them instead of the tiles' scans): the points that need rows
    // only a dirty scan covers go on the round's list, one atomic per wave that holds any; a point that finds the list
    // full keeps its flags and is refused by k_decide (the window then commits up to it)
    if (sparse_cap > 0) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(flag_unsafe);
        if (m != 0ull) {
            int base = 0;
            if (threadIdx.x == (unsigned)__builtin_ctzll(m)) base = atomicAdd(&ctl->n_sparse, __builtin_popcountll(m));
            base = __shfl(base, __builtin_ctzll(m));
            const int idx = base + __builtin_popcountll(m & ((1ull << threadIdx.x) - 1ull));
            if (flag_unsafe && idx < sparse_cap) {
                sparse_list[idx] = j;
                flag_word |= CC_FLAG_SPARSE;
                flag_n1skip = false;
            }
        }
    }
    // the tile as a whole: when no point of it needs a row that only a dirty scan covers - even the largest displacement
    // of every class stays below the point's threshold, for the stages the point is expected to evaluate -, the tile's
    // dirty scan is not run at all
    const unsigned long long any_v = __builtin_amdgcn_ballot_w64(need_ver), any_c = __builtin_amdgcn_ballot_w64(need_car);
    if (j < B) {
        // (the outlier-kind threshold is only withheld where that spares the tile its scans: in a tile whose scans run
        // anyway every point gets both kinds covered, and a point whose stage 0 fails against expectation is decided
        // all the same - on overlapping data that happens all the time while the table is young)
        if (flag_n1skip && any_v == 0ull && any_c == 0ull) { tau_w[1] = CC_INF; flag_word |= CC_FLAG_N1SKIP; }
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) ver.tau[(size_t)j * CC_TAU_STRIDE + kd] = tau_w[kd];
        ver.unsafe[j] = flag_word;
    }
    // (a tile whose dirty scans run gives k_decide more than the seeds to merge: not covered by the test above)
    if (__builtin_amdgcn_ballot_w64(j < B && !(repeats && any_v == 0ull && any_c == 0ull)) != 0ull && threadIdx.x == 0)
        ctl->rdiff[round] = 1;
    {
        // statistics for the host's trace line (one atomic per wave and only when something is flagged)
        const unsigned long long b1 = __builtin_amdgcn_ballot_w64(flag_unprov), b2 = __builtin_amdgcn_ballot_w64(flag_unsafe);
        if (threadIdx.x == 0 && b1) atomicAdd((unsigned long long*)&ctl->stat_unprovable, (unsigned long long)__builtin_popcountll(b1));
        if (threadIdx.x == 0 && b2) atomicAdd((unsigned long long*)&ctl->stat_unsafe, (unsigned long long)__builtin_popcountll(b2));
    }
    if (threadIdx.x == 0) {
        ver.skip[blockIdx.x] = any_v == 0ull ? 1 : 0;  // (k_commit_a counts the tiles of the last round for the host's window policy)
        ver.skip_car[blockIdx.x] = (!la_mode || any_c == 0ull) ? 1 : 0;
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
                                                int* __restrict__ long_list, int long_cap, int stat_tail, int last_round,
                                                int heavy_on, int quiet_ok, int* __restrict__ link_near)
{
    CC_LATENCY_KERNEL();
    // the last ac_blocks workgroups of a round-0 launch before a lookahead window's validation: cc_apply_carry
    if (ac_blocks > 0 && (int)blockIdx.x >= (int)gridDim.x - ac_blocks) {
        cc_apply_carry(ac_rec, car, ac_sc, ctl->d, ctl->filter, (int)blockIdx.x - ((int)gridDim.x - ac_blocks), ac_blocks);
        return;
    }
    const int B = ctl->win_b;
    if (B == 0) return;
    // exact multi-GPU path: the ranks' samples of their split pruned scans (the record behind each rank's candidates,
    // k_merge_partials) add up to the counters the host policy reads - the same sum on every rank
    // ... and a single GPU's own sample, and the count of points a guessed threshold missed: left per window parity by
    // kernels that may have run on the lookahead stream (Ctl::pstat, Ctl::n_missed_all), taken over here
    if (round == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        const int q = (int)(ctl->window_seq & 1ull);
        unsigned long long rows = 0ull, full = 0ull;
        if (stat_tail >= 0) {
            const Cand* g = part + (size_t)q * part_stride + (size_t)stat_tail;
            for (int r = 0; r < S; ++r) {
                rows += (unsigned long long)(unsigned)g[(size_t)r * part_outer].key;
                full += (unsigned long long)(unsigned)g[(size_t)r * part_outer].slot;
            }
        } else {
            rows = ctl->pstat[q][0];
            full = ctl->pstat[q][1];
        }
        ctl->pstat[q][0] = 0ull;
        ctl->pstat[q][1] = 0ull;
        ctl->stat_prune_rows += rows;
        ctl->stat_prune_full += full;
        // (atomic: the groups of this launch that refuse a point whose first place is a bound add to the same counter)
        if (ctl->n_missed_all[q] != 0) atomicAdd((unsigned long long*)&ctl->stat_missed, (unsigned long long)ctl->n_missed_all[q]);
        ctl->n_missed_all[q] = 0;
    }
    if (round > 0 && ctl->fc[round - 1] >= B) return;
    // a quiet round: k_dseed established that every point's decision repeats its claim (Ctl::rdiff).  Nothing to write:
    // the claims stand, the frontier stays at the window's end (fc[round] was reset when the window was opened), later
    // rounds and the commit see a converged window.
    if (round > 0 && quiet_ok != 0 && ctl->rdiff[round] == 0) return;
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
    // what k_dseed found out about this point (CC_FLAG_*), and which dirty scans covered its tile
    const int flags = (round > 0) ? ver.unsafe[j] : 0;
    // a dirty scan that k_dseed ruled out for this point's tile was not run: the seeds are its whole result
    // (nodirty: the host did not launch the dirty scans at all; points that would have needed them are refused below)
    // (nodirty == 2: the sparse dirty scans ran instead - for the points on the round's list, both kinds covered)
    const bool on_list = nodirty == 2 && (flags & CC_FLAG_SPARSE) != 0;
    const bool ran = round > 0 && ((nodirty == 0 && ver.skip[j >> 6] == 0) || on_list);
    const bool ran_car = round > 0 && la_mode && ((nodirty == 0 && ver.skip_car[j >> 6] == 0) || on_list);
    if (round > 0) {
        Cand dummy = none;
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
        else if (c1.slot == CC_SLOT_BOUND) {
            // a guessed threshold missed the point's own MC and nobody rescanned it (the list of missed points was full,
            // or the scan ran lean - without that list, see cc_policy.h): refused; counted, so that the policy hears of it
            // (stage 1 as well: the pcore list was resolved by the guess, the pcore stage failed and every outlier MC lies
            // beyond the guess - the seeded chain has to look at the point, see Ctl::seed_at)
            if (gl == 0 && round == 0) atomicAdd((unsigned long long*)&ctl->stat_missed, 1ull);
            T = CC_T_UNKNOWN;
            return;
        }
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
        // hddstream.py:334-337.  In a validation round the very same add has usually been made already: k_chain (or
        // k_chain_long) of this round replayed the claims of the previous one, and if j claimed `target` there and
        // the state it is added to now is the one its step started from - the version row of its predecessor in the
        // chain, or the row itself when nobody before j claims the MC (j heads the chain) - the radius test's verdict
        // is in ver.acc[j]: the same operands in the same order, so the same bits.
        bool known = false, fits = false;
        if (round > 0 && Told[j] == target && ver.tgt[j] == target) {
            known = (wkind == 2 && wrow < CC_CAR_BASE) ? (ver.next[wrow] == j) : true;
            fits = ver.acc[j] != 0;
        }
        if (!known) {
            const GroupAdd g = cc_group_add(bcf1, bcf2, bw, p, d, par);
            fits = g.r2 <= par.eps_sq;
        }
        if (fits) {
            T = target;
            path = stage;
        }
    };
    // A stage is evaluated only when the rows it needs were covered: by the seeds alone (no flag), or by the dirty scans
    // of the point's tile.
    {
        const bool missing = ((flags & CC_FLAG_U0) != 0 && !ran) || ((flags & CC_FLAG_C0) != 0 && la_mode && !ran_car);
        if (missing) T = CC_T_UNKNOWN;
        else run_stage(p1, p2, dvp, 0);
    }
    if (T == -1) {
        const bool covers1 = (flags & CC_FLAG_N1SKIP) == 0;  // (the outlier-kind threshold was not withheld from the scans)
        const bool missing = ((flags & CC_FLAG_U1) != 0 && !(ran && covers1)) ||
                             ((flags & CC_FLAG_C1) != 0 && la_mode && !(ran_car && covers1));
        if (missing) T = CC_T_UNKNOWN;
        else run_stage(o1, o2, dvo, 1);
    }
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
    if (gl == 0) {
        Tnew[j] = T;
        dpath[j] = (int8_t)path;
        // (round 0 with the creators' links, cc_link.h: no link yet; the claims of the points that decided "create" are
        // registered by k_link_apply, for the microcluster they end up claiming)
        if (link_near != nullptr) link_near[j] = CC_IDX_INF;
        if (round > 0 && (T == CC_T_UNKNOWN || T != Told[j])) atomicMin(&ctl->fc[round], j);
#ifdef CC_ROUND_DEBUG
        if (round > 0) {
            const int to = Told[j];
            atomicAdd(&ctl->dbg_round[round][5], 1ull);
            if (T == CC_T_UNKNOWN) atomicAdd(&ctl->dbg_round[round][0], 1ull);
            else if (T != to) {
                const int c = (to == M0 + j) ? (T >= M0 ? 1 : 2) : (T == M0 + j ? 3 : 4);
                atomicAdd(&ctl->dbg_round[round][c], 1ull);
            }
        }
#endif
        if ((j & 15) == 0) {  // the next k_chain takes maxima into them
            ver.tile_dsq[(size_t)(j >> 4) * CC_DSQ_STRIDE] = 0ull;
            ver.tile_dsq[(size_t)(j >> 4) * CC_DSQ_STRIDE + 1] = 0ull;
            ver.tile_dsq[(size_t)(j >> 4) * CC_DSQ_STRIDE + 2] = 0ull;
        }
        // (claims on the first scan_rows table rows are gathered by k_claims instead, without atomics)
        // Nobody replays the claims of the last round the host enqueued for this window - no k_chain follows it, the
        // commit reads the claims themselves -, so they are not registered: three atomics per point saved, and the ones
        // that serialise when a population takes a large share of the events (one address per MC)
        // (heavy_on: k_claims_heavy follows this launch and gathers the claims of the heavy rows - Table::heavy)
        if (last_round == 0 && T >= 0 && !(T < M0 && T < scan_rows) && !(heavy_on != 0 && T < M0 && tab.heavy[T] != 0) &&
            !(link_near != nullptr && T == M0 + j)) {
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
                if (tab.heavy[T] == 0) {  // nominated as a heavy row (k_commit_a takes it from there)
                    const int hn = atomicAdd(&ctl->n_heavy_new, 1);
                    if (hn < CC_HEAVY_NEW) ctl->heavy_new[hn] = T;
                }
                if (long_list != nullptr) {
                    const int idx = atomicAdd(&ctl->n_long[round + 1], 1);
                    if (idx < long_cap) {  // (= the workgroups of the k_chain_long launch that follows)
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
                                                int round, int scan_rows, int quiet_ok)
{
    CC_LATENCY_KERNEL();
    const int B = ctl->win_b;
    if (B == 0) return;
    if (round > 0 && ctl->fc[round - 1] >= B) return;  // k_decide of this round did not run either
    // (... nor in a quiet round: it left T unwritten - the claims of the round before stand -, nothing to gather)
    if (round > 0 && quiet_ok != 0 && ctl->rdiff[round] == 0) return;
    const int M0 = ctl->m_rows;
    const int m = blockIdx.x;
    if (m >= M0 || m >= scan_rows) return;
    // (as in k_claims_heavy below: every thread counts its own claimants, only the first CC_CHAIN_MEMB of them go through
    // the LDS counter - with a dozen MCs a workgroup finds thousands -, and eight 16-byte loads are in flight per thread)
    __shared__ int s_pos, s_cnt, s_first, s_last;
    if (threadIdx.x == 0) { s_pos = 0; s_cnt = 0; s_first = CC_IDX_INF; s_last = -1; }
    __syncthreads();
    int lmin = CC_IDX_INF, lmax = -1, mine = 0;
    volatile int* const pos_now = &s_pos;
    const int4* T4 = reinterpret_cast<const int4*>(T);  // (the buffer is padded to whole 128-entry blocks)
    constexpr int AHEAD = 8;
    for (int base0 = (int)threadIdx.x * 4; base0 < B; base0 += 256 * 4 * AHEAD) {
        int4 v[AHEAD];
#pragma unroll
        for (int q = 0; q < AHEAD; ++q) {
            const int base = base0 + q * 256 * 4;
            v[q] = T4[(base < B ? base : 0) >> 2];  // (no branch around the load: claims beyond B are ignored by index below)
        }
#pragma unroll
        for (int q = 0; q < AHEAD; ++q) {
            const int base = base0 + q * 256 * 4;
            const int e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = base + c;
                if (j < B && e[c] == m) {
                    lmin = j < lmin ? j : lmin;
                    lmax = j > lmax ? j : lmax;
                    ++mine;
                    if (*pos_now < CC_CHAIN_MEMB) {
                        const int pos = atomicAdd(&s_pos, 1);
                        if (pos < CC_CHAIN_MEMB) tab.memb[(size_t)m * CC_CHAIN_MEMB + pos] = j;
                    }
                }
            }
        }
    }
    if (lmax >= 0) { atomicMin(&s_first, lmin); atomicMax(&s_last, lmax); atomicAdd(&s_cnt, mine); }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt > 0) {
        const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
        const unsigned long long sn = (stamp + 1ull) << 20;
        const size_t wr = (size_t)((round + 1) & 1) * tab.cap + (size_t)m;  // the copy the next round reads
        tab.touch[wr] = sn | (unsigned long long)(0xFFFFF - s_first);
        tab.last[wr] = sn | (unsigned long long)s_last;
        tab.cnt[m] = ((stamp + 1ull) << 24) | (unsigned long long)s_cnt;
    }
}

// ---------------------------------------------------------------------------------
// k_claims_heavy: k_claims for the heavy rows of a large table (Table::heavy, Ctl::heavy_list): one workgroup per
// heavy row reads the claims once and leaves first / last claimant, count and members as k_decide's atomics would have
// - without the thousands of same-address atomics a MC that takes a tenth of the window's points costs there.  A chain
// of more than CC_CHAIN_MEMB claimants goes on the round's list of long chains exactly as in k_decide.  Launched behind
// every k_decide whose claims are replayed, while the previous batch ended with heavy rows (k_decide: heavy_on).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_claims_heavy(Ctl* __restrict__ ctl, Table tab, const int* __restrict__ T, int round,
                                                      int* __restrict__ long_list, int long_cap, int quiet_ok)
{
    CC_LATENCY_KERNEL();
    const int B = ctl->win_b;
    if (B == 0) return;
    if (round > 0 && ctl->fc[round - 1] >= B) return;  // k_decide of this round did not run either
    if (round > 0 && quiet_ok != 0 && ctl->rdiff[round] == 0) return;  // (a quiet round: see k_claims)
    if ((int)blockIdx.x >= ctl->n_heavy) return;
    const int m = ctl->heavy_list[blockIdx.x];
    if (m < 0 || m >= ctl->m_rows) return;
    // (thousands of claimants: each thread counts its own, and only the first CC_CHAIN_MEMB of them - any of them, the
    // list is unordered - take a slot of the member list through the LDS counter)
    __shared__ int s_pos, s_cnt, s_first, s_last;
    if (threadIdx.x == 0) { s_pos = 0; s_cnt = 0; s_first = CC_IDX_INF; s_last = -1; }
    __syncthreads();
    int lmin = CC_IDX_INF, lmax = -1, mine = 0;
    volatile int* const pos_now = &s_pos;
    const int4* T4 = reinterpret_cast<const int4*>(T);  // (the buffer is padded to whole 128-entry blocks)
    // (eight 16-byte loads in flight per thread: with one load per pass of the loop - the matches' LDS traffic keeps the
    // compiler from moving the next load up - a window of 32 768 claims was 32 dependent round trips, 29 us per launch)
    constexpr int AHEAD = 8;
    for (int base0 = (int)threadIdx.x * 4; base0 < B; base0 += 256 * 4 * AHEAD) {
        int4 v[AHEAD];
#pragma unroll
        for (int q = 0; q < AHEAD; ++q) {
            const int base = base0 + q * 256 * 4;
            v[q] = T4[(base < B ? base : 0) >> 2];  // (no branch around the load: claims beyond B are ignored by index below)
        }
#pragma unroll
        for (int q = 0; q < AHEAD; ++q) {
            const int base = base0 + q * 256 * 4;
            const int e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = base + c;
                if (j < B && e[c] == m) {
                    lmin = j < lmin ? j : lmin;
                    lmax = j > lmax ? j : lmax;
                    ++mine;
                    if (*pos_now < CC_CHAIN_MEMB) {
                        const int pos = atomicAdd(&s_pos, 1);
                        if (pos < CC_CHAIN_MEMB) tab.memb[(size_t)m * CC_CHAIN_MEMB + pos] = j;
                    }
                }
            }
        }
    }
    if (lmax >= 0) { atomicMin(&s_first, lmin); atomicMax(&s_last, lmax); atomicAdd(&s_cnt, mine); }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int n_claims = s_cnt;
    if (n_claims > 0) {
        const unsigned long long stamp = ctl->window_seq * 16ull + (unsigned long long)round;
        const unsigned long long sn = (stamp + 1ull) << 20;
        const size_t wr = (size_t)((round + 1) & 1) * tab.cap + (size_t)m;  // the copy the next round reads
        tab.touch[wr] = sn | (unsigned long long)(0xFFFFF - s_first);
        tab.last[wr] = sn | (unsigned long long)s_last;
        unsigned long long cw = ((stamp + 1ull) << 24) | (unsigned long long)n_claims;
        if (n_claims > CC_CHAIN_MEMB) {
            atomicAdd((unsigned long long*)&ctl->stat_long, 1ull);
            if (long_list != nullptr) {
                const int idx = atomicAdd(&ctl->n_long[round + 1], 1);
                if (idx < long_cap) {
                    long_list[(size_t)((round + 1) & 1) * CC_LONG_CAP + idx] = m;
                    cw |= CC_LONG_LISTED;
                }
            }
        }
        tab.cnt[m] = cw;
    }
    // the population has thinned out: back to k_decide's atomics from the next window on
    if (round == 0 && n_claims <= CC_CHAIN_MEMB / 2) tab.heavy[m] = 2;
}

// ---------------------------------------------------------------------------------
// k_chain: replay the claimed decisions per MC in arrival order.  One 32-lane group per window point; the
// group of the first point that targets a MC walks that MC's chain, every step dimension-parallel.
// ---------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_chain(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                               Versions ver, Carry car, const int* __restrict__ T, int round,
                                               int long_rows, const unsigned long long* __restrict__ lprev,
                                               unsigned long long* __restrict__ lstat)
{
    CC_LATENCY_KERNEL();
    const int B = ctl->win_b;
    if (B == 0) return;
    if (ctl->fc[round - 1] >= B) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->last_round = round;
        ctl->n_sparse = 0;  // (k_dseed of this round fills the list of points for the sparse dirty scans)
        ctl->rdiff[round] = 0;  // (... and says whether any decision of this round may differ from its claim)
    }
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
    // a member of a long chain whose running sums k_chain_long<.., true> has laid out (see there): this group evaluates
    // the member's own step - the body of the walk below for one step, from the sums before the member
    if (lprev != nullptr) {
        const unsigned long long lp = lprev[j];
        if ((lp >> 27) == stamp) {
            const int pv = (int)(lp & 0x1FFFFull) - 1;
            const size_t lslot = (size_t)(round & 1) * CC_LSTAT_ROWS + (size_t)((lp >> 17) & 0x3FFull);
            const Par par = cc_load_par(ctl);
            const int d = par.d;
            double bc1[2] = {0.0, 0.0}, bc2[2] = {0.0, 0.0}, px[2] = {0.0, 0.0}, c0[2] = {0.0, 0.0}, w0[2] = {1.0, 1.0};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = gl + 32 * h;
                if (i < d) {
                    bc1[h] = (pv < 0) ? tab.cf1[(size_t)t * d + i] : ver.cf1[(size_t)pv * d + i];
                    bc2[h] = (pv < 0) ? tab.cf2[(size_t)t * d + i] : ver.cf2[(size_t)pv * d + i];
                    px[h] = X[(ctl->cursor + j) * d + i];
                    c0[h] = tab.cen[(size_t)t * d + i];
                    w0[h] = 1.0 / tab.pref[(size_t)t * d + i];
                }
            }
            const double bw = (pv < 0) ? tab.w[t] : ver.w[pv];
            const int bkind = tab.kind[t];  // (a pcore MC: nothing to promote)
            int kind0 = bkind;
            if (ctl->mode != 0) {
                const unsigned long long co = tab.carry_of[t];
                if ((co >> 20) == ctl->window_seq) {
                    const size_t r = (size_t)(co & 0xFFFFFull);
                    kind0 = car.kind0[r];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int i = gl + 32 * h;
                        if (i < d) { c0[h] = car.c0[r * d + i]; w0[h] = car.w0[r * d + i]; }
                    }
                }
            }
            const GroupAdd g = cc_group_add_regs(bc1, bc2, bw, px, d, par);
            if (!(g.r2 <= par.eps_sq)) {
                if (gl == 0) atomicMin(&lstat[lslot], (stamp << 20) | (unsigned long long)j);  // (the FIRST rejected step is what counts)
                return;
            }
            double dq = 0.0;
            bool mv = false;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = gl + 32 * h;
                if (i < d) {
                    ver.cen[(size_t)j * d + i] = g.cen[h]; ver.pref[(size_t)j * d + i] = g.pr[h];
                    ver.scl[(size_t)j * d + i] = par.pow2 ? (g.pr[h] == 1.0 ? 1.0 : par.inv_k) : g.pr[h];
                }
                const double df = g.cen[h] - c0[h];
                dq += df * df * w0[h];
                mv = mv || (1.0 / g.pr[h] != w0[h]);
            }
            dq = cc_group_sum_any_order(dq);
            const bool promoted = bkind == CC_KIND_PCORE && kind0 == CC_KIND_OUTLIER;
            if ((bkind != kind0 && !promoted) || !(dq >= 0.0) || cc_group_ballot(mv) != 0u) dq = CC_INF;
            if (gl == 0) {
                ver.tgt[j] = t; ver.kind[j] = bkind; ver.key[j] = tab.key[t]; ver.upg[j] = -1; ver.acc[j] = 1;
                int cls;
                ver.dsq[j] = cc_dsq_store(dq, bkind, kind0, &cls);
                atomicMax(&ver.tile_dsq[(size_t)(j >> 4) * CC_DSQ_STRIDE + cls], cc_dsq_code(dq));
            }
            return;
        }
    }
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
    double binv[2] = {w0[0], w0[1]};  // 1 / bpr of the running state: divided anew only when a preferred dimension changes
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
                        if (g.pr[h] != bpr[h]) binv[h] = 1.0 / g.pr[h];
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
                for (int h = 0; h < 2; ++h) mv = mv || (binv[h] != w0[h]);
                // (a MC promoted since the snapshot keeps its bound: it was scanned as an outlier MC, and k_dseed bounds it
                // through the point's outlier list - class 2)
                const bool promoted = bkind == CC_KIND_PCORE && kind0 == CC_KIND_OUTLIER;
                if (isnew || (bkind != kind0 && !promoted) || !(dq >= 0.0) || cc_group_ballot(mv) != 0u) dq = CC_INF;
                if (gl == 0) {
                    ver.w[cur] = bw;
                    ver.tgt[cur] = t; ver.kind[cur] = bkind; ver.key[cur] = bkey; ver.upg[cur] = bupg;
                    ver.acc[cur] = ok ? 1 : 0; ver.next[cur] = nx;
                    int cls;
                    ver.dsq[cur] = cc_dsq_store(dq, bkind, isnew ? CC_KIND_DEAD : kind0, &cls);
                    atomicMax(&ver.tile_dsq[(size_t)(cur >> 4) * CC_DSQ_STRIDE + cls], cc_dsq_code(dq));
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
#define CC_LONG_QUEUE 2048       // pending chain members (ring buffer); four times as many in the SPLIT workgroups
#define CC_LONG_THREADS 1024      // threads of a k_chain_long workgroup

// SPLIT: workgroups of CC_LONG_THREADS threads (four waves per SIMD) with the per-(step, dimension) work of the radius
// tests spread over all of them - twice the LDS, so one workgroup per CU: for the few long chains of few-microcluster
// streams and skewed populations.  !SPLIT: 256 threads, two workgroups per CU: tables of hundreds of rows, where every
// row's chain is long-ish and the workgroups are many.
//
// PREP (round 5): the sequential part alone.  One workgroup's vector-memory pipeline and its per-(step, dimension) work
// were most of a long chain's time (a chain of 3 277 members: 220 us, of which the running sums are 60) - and everything
// but the running sums is a function of one step's sums.  k_chain_long<true, true>, launched BEFORE k_chain, walks the
// chain of a pcore MC through phases 1 and 2 only and leaves, per member m: the sums after m (ver.cf1 / cf2 / w), the
// member after it (ver.next) and `lprev[m]` = stamp, the chain's slot in `lstat`, the member before it.  k_chain's
// 32-lane group of m then evaluates that one step from the sums before it - the same function of the same operands as a
// step of its own walk, on all CUs instead of one - and if the radius test rejects the step takes the member into the
// chain's word of `lstat` (atomic minimum: the first rejected step).  The sums behind a rejected step are not the chain's:
// the launch of k_chain_long<.., false> that follows replays such a chain FROM that member on, starting from the version
// row of the member before it (every step up to there was accepted and evaluated) - and, from the start, the chains that
// were never prepared: outlier MCs, whose promotion is a sequential matter.
#define CC_LPREV(stamp, slot, prev) (((unsigned long long)(stamp) << 27) | ((unsigned long long)(slot) << 17) | (unsigned long long)((prev) + 1))
template <bool SPLIT, bool PREP>
__global__ __launch_bounds__(SPLIT ? CC_LONG_THREADS : 256) void k_chain_long(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                                    Versions ver, Carry car, const int* __restrict__ T, int round,
                                                    int scan_rows, const int* __restrict__ long_list,
                                                    unsigned long long* __restrict__ lstat, unsigned long long* __restrict__ lprev)
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
        const int n_listed = min(ctl->n_long[round], (int)gridDim.x);  // (k_decide listed at most as many as this launch has workgroups)
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
    int head = 0xFFFFF - (int)(ft & 0xFFFFFull);
    const int last_j = ((lt >> 20) == stamp) ? (int)(lt & 0xFFFFFull) : head;
    // the chain's word in lstat (per round parity; table row or list entry): stamp << 20 | first rejected member of a
    // prepared chain (0xFFFFF: none)
    const size_t lslot = (size_t)(round & 1) * CC_LSTAT_ROWS + (size_t)blockIdx.x;
    int resume_prev = -1;   // the member before the one the replay starts at (-1: the table row is the state before it)
    bool resumed = false;
    if constexpr (PREP) {
        if (tab.kind[t] != CC_KIND_PCORE) return;
        if (threadIdx.x == 0) {
            lstat[lslot] = (stamp << 20) | 0xFFFFFull;
            atomicAdd(&lstat[2 * CC_LSTAT_ROWS], 1ull);
        }
    } else {
        if (lstat != nullptr) {
            const unsigned long long ls = lstat[lslot];
            if ((ls >> 20) == stamp) {
                const int f = (int)(ls & 0xFFFFFull);
                if (f == 0xFFFFF) return;  // prepared, and k_chain accepted every step
                if (threadIdx.x == 0) atomicAdd(&lstat[2 * CC_LSTAT_ROWS + 1], 1ull);
                resumed = true;
                head = f;
                resume_prev = (int)(lprev[f] & 0x1FFFFull) - 1;
            }
        }
    }

    const Par par = cc_load_par(ctl);
    const int d = par.d;
    const int tid = threadIdx.x;
    const long long cursor = ctl->cursor;
    const int pk_base = ctl->n_pkeys;
    // steps per batch: the CF1 / CF2 prefixes of a batch have to fit the staging area
    const int K = min(256, (CC_LONG_XY_DOUBLES / (2 * d)) & ~7);  // (256 at d <= 12, 216 at 14, 152 at 20, 72 at 40, 48 at 64)

    __shared__ __attribute__((aligned(16))) double s_xy[CC_LONG_XY_DOUBLES + 2 * CC_MAX_DIM];  // (+ one odd-length pad per dimension)
    __shared__ double s_w[256], s_dq[256];
    // (PREP: a second staging area - the running sums of one batch run beside the staging of the next, see the loop)
    __shared__ __attribute__((aligned(16))) double s_xy2[PREP && SPLIT ? CC_LONG_XY_DOUBLES + 2 * CC_MAX_DIM : 1];
    __shared__ double s_w2[PREP && SPLIT ? 256 : 1];
    __shared__ unsigned long long s_mask[256];  // bit i: dimension i is a preferred one after the step (var <= delta^2)
    __shared__ int s_flag[256];                 // bit 0: radius test passed, bit 1: promotion condition holds
    // claims looked at per pass of the member collection: four per thread; the ring buffer holds two passes' worth
    constexpr int PASS = (SPLIT ? CC_LONG_THREADS : 256) * 4;
    constexpr int QUEUE = SPLIT ? 4 * CC_LONG_QUEUE : CC_LONG_QUEUE;
    constexpr int SCAN_WAVES = PASS / 256;
    __shared__ int s_queue[QUEUE];
    __shared__ double s_b1[64], s_b2[64], s_bcen[64], s_bpref[64], s_c0[64], s_w0[64];  // running state / snapshot metric
    __shared__ double s_bw, s_bdq;
    __shared__ unsigned long long s_m0, s_bmask;  // preferred dimensions in the snapshot / of the running state (bit i)
    __shared__ int s_wsum[SCAN_WAVES];
    __shared__ int s_first_fail, s_first_up;
    // per (step, dimension) of a batch: the term of the radius sum, the displacement term, "preferred after the step"
    __shared__ double s_term[SPLIT && !PREP ? CC_LONG_XY_DOUBLES / 2 : 1], s_dqt[SPLIT && !PREP ? CC_LONG_XY_DOUBLES / 2 : 1];
    __shared__ unsigned char s_pf[SPLIT && !PREP ? CC_LONG_XY_DOUBLES / 2 : 1];
    const int NT = (int)blockDim.x;  // CC_LONG_THREADS: four waves per SIMD - these phases are instruction chains, not bandwidth
    // staged coordinates / CF prefixes of a batch, dimension-major: entry (step k, dimension i) at i * Kp + k.  The chains
    // walk a dimension's steps - contiguous, so their LDS operands have immediate offsets: a lone wave issues one
    // instruction every ~8 cycles, and the address arithmetic of a step-major layout was two thirds of a step's
    // instructions (54 -> 24 cycles per step, timed).  Kp is odd: threads that walk a step's dimensions (lanes = consecutive
    // dimensions) then hit 32 different LDS banks.
    const int Kp = K | 1;
    const bool one_wave = d <= 31;  // the three running sums in one wave (see the chains)
    double* const xs = s_xy;
    double* const ys = s_xy + (size_t)Kp * d;
    auto at = [&](int k, int i) { return i * Kp + k; };

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
    // a replay that starts inside the chain: the running state is the version row of the member before
    if (resume_prev >= 0) {
        if (tid < d) {
            const size_t o = (size_t)resume_prev * d + tid;
            s_b1[tid] = ver.cf1[o]; s_b2[tid] = ver.cf2[o]; s_bcen[tid] = ver.cen[o]; s_bpref[tid] = ver.pref[o];
        }
        if (tid == 0) s_bw = ver.w[resume_prev];
        __syncthreads();
    }
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
    int scan_pos = head & ~(PASS - 1);  // next block of PASS claims to look at
    bool scan_done = false;
    int walked = 0;
    int prev_last = -1;  // (PREP) the member before the batch at hand: none before the head
    bool promoted_any = false;
    // (every thread scans four claims per pass.  The claims of a pass are loaded one pass ahead: the latency of the load
    // runs beside the batch in between, not in front of the pass.)
    const bool scanner = true;
    auto load_claims = [&](int pos) {
        const int i0 = pos + (scanner ? tid : 0) * 4;
        return T4[((scanner && i0 <= last_j) ? i0 : 0) >> 2];  // (no branch around the load - it would wait for it -: claims beyond last_j are dropped by index)
    };
    int4 v_next = load_claims(scan_pos);
    // coordinates requested one batch ahead (see phase 2): element e = tid + q NT of the next batch's staging area
    constexpr int PRE = (CC_LONG_XY_DOUBLES / 2 + (SPLIT ? CC_LONG_THREADS : 256) - 1) / (SPLIT ? CC_LONG_THREADS : 256);
    double pre[PRE];
    int pre_elems = 0, pre_next = 0;  // elements of `pre` that are valid for the batch at hand / requested for the one after
#ifdef CC_LONG_TIMERS
    // build variant (-DCC_LONG_TIMERS): shader cycles per phase, workgroup 0 of every launch, printed at the end of a call
    long long tk_prev = clock64();
    unsigned long long tk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define CC_TICK(i) do { const long long now_ = clock64(); tk_acc[i] += (unsigned long long)(now_ - tk_prev); tk_prev = now_; } while (0)
#else
#define CC_TICK(i) do { } while (0)
#endif

    // members in order: ordered compaction of the next claims into the queue, until it holds `want` of them (or the claims
    // are through).  K slots behind the head stay untouched: the batch before the one at hand is still read there (PREP).
    auto collect = [&](const int want) {
        while (!scan_done && qcount < want && qcount + PASS + K <= QUEUE) {
            const int i0 = scan_pos + (scanner ? tid : 0) * 4;
            const int4 v = v_next;
            v_next = load_claims(scan_pos + PASS);
            const int e[4] = {v.x, v.y, v.z, v.w};
            int f[4], cnt = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = i0 + c;
                f[c] = (scanner && j >= head && j <= last_j && j < B && e[c] == t) ? 1 : 0;
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
            for (int w = 0; w < SCAN_WAVES; ++w) {
                if (w < wv) base += s_wsum[w];
                total += s_wsum[w];
            }
            int pos = qcount + base + incl - cnt;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (f[c]) { s_queue[(qhead + pos) & (QUEUE - 1)] = i0 + c; ++pos; }
            qcount += total;
            scan_pos += PASS;
            if (scan_pos > last_j) scan_done = true;
        }
    };

    // the running sums of a batch of n steps over one staged row (see phase 2 of the loop).  Sixteen steps at a time: the
    // sixteen LDS reads go out together, then the sixteen additions in order, then the sixteen writes.  (What a step costs is
    // the LDS's time per instruction, not the round trip: a dependent v_add_f64 takes 7 cycles, a step 28 with its half a
    // ds_read2_b64 and half a ds_write2_b64, whatever the number of active lanes; issuing the next sixteen reads ahead of
    // these additions gave 27, storing the sums straight to HBM instead of the LDS 35-40 - tools/micro/dep_add.hip.)
#ifndef CC_LONG_UNROLL
#define CC_LONG_UNROLL 16
#endif
    auto prefix_chain = [&](double* const row, double c, const int n) {
        constexpr int U = CC_LONG_UNROLL;
        int k = 0;
        for (; k + U <= n; k += U) {
            double v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = row[k + u];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c = c + v[u];
                v[u] = c;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) row[k + u] = v[u];
        }
        for (; k < n; ++k) {
            c = c + row[k];
            row[k] = c;
        }
        return c;
    };

    if constexpr (PREP && SPLIT) {
        if (one_wave) {
            // The pipelined form (d <= 31: the running sums fit one wave).  The first wave adds up batch b in one staging area
            // while the other fifteen write out batch b - 1 from the second one - sums, neighbours, stamps - and stage batch
            // b + 1 into it: every thread of theirs stores the sums of a slot and then overwrites that same slot, so the
            // two steps need no barrier between them, and the running sums themselves stay in the first wave's registers
            // from batch to batch.  One barrier per batch; the member collection (all waves) runs between batches and keeps
            // two batches' worth of members queued, so that the batch after the one at hand is known in full.
            auto bxq = [&](const int q) -> double* { return q ? s_xy2 : s_xy; };
            auto bwq = [&](const int q) -> double* { return q ? s_w2 : s_w; };
            const int st = tid - 64, NS = NT - 64;  // the staging threads
            const size_t ys_off = (size_t)Kp * d;
            // (a thread's coordinates - up to four - are requested together, no branch around a load: one round trip per
            // batch, not one per coordinate; the staging waves are the longer side of a batch otherwise)
            constexpr int SE = (CC_LONG_XY_DOUBLES / 2 + (CC_LONG_THREADS - 64) - 1) / (CC_LONG_THREADS - 64);
            auto stage = [&](double* const xb, double* const wb, const int first, const int n_) {
                double x[SE];
                int slot[SE];
#pragma unroll
                for (int q = 0; q < SE; ++q) {
                    const int e = st + q * NS;
                    const bool valid = e < n_ * d;
                    const int k = valid ? e / d : 0, i = valid ? e - k * d : 0;
                    slot[q] = valid ? at(k, i) : -1;
                    x[q] = X[(cursor + s_queue[(first + k) & (QUEUE - 1)]) * d + i];
                }
#pragma unroll
                for (int q = 0; q < SE; ++q)
                    if (slot[q] >= 0) {
                        xb[slot[q]] = x[q];
                        xb[ys_off + slot[q]] = x[q] * x[q];
                    }
                for (int k = st; k < n_; k += NS) wb[k] = 1.0;  // (what the W chain adds per step)
            };
            // (n_after: members the queue holds behind this batch's first one; before: the member before it)
            auto write_out = [&](const double* const xb, const double* const wb, const int first, const int n_, const int n_after,
                                 const int before) {
                for (int e = st; e < n_ * d; e += NS) {
                    const int k = e / d, i = e - k * d;
                    const size_t o = (size_t)s_queue[(first + k) & (QUEUE - 1)] * d + i;
                    ver.cf1[o] = xb[at(k, i)]; ver.cf2[o] = xb[ys_off + at(k, i)];
                }
                for (int k = st; k < n_; k += NS) {
                    const int m = s_queue[(first + k) & (QUEUE - 1)];
                    const int pv = (k > 0) ? s_queue[(first + k - 1) & (QUEUE - 1)] : before;
                    ver.w[m] = wb[k];
                    ver.next[m] = (k + 1 < n_after) ? s_queue[(first + k + 1) & (QUEUE - 1)] : CC_IDX_INF;
                    lprev[m] = CC_LPREV(stamp, blockIdx.x, pv);
                }
            };
            collect(2 * K + 1);
            __syncthreads();
            CC_TICK(0);
            int n = qcount < K ? qcount : K;
            if (tid >= 64) stage(bxq(0), bwq(0), qhead, n);
            __syncthreads();
            CC_TICK(1);
            const bool c1 = tid < d, c2 = tid >= 32 && tid < 32 + d, cw = tid == 63;
            double run = c1 ? s_b1[tid] : c2 ? s_b2[tid - 32] : s_bw;  // this lane's running sum (first wave)
            int cur = 0, n_prev = 0, first_prev = 0, before_prev = -1;
            while (n > 0) {
                const int n_next = min(K, qcount - n);  // (exact: either the claims are through or the queue holds 2 K + 1)
                if (tid < 64) {
                    if (c1 || c2 || cw)
                        run = prefix_chain(c1 ? bxq(cur) + (size_t)tid * Kp : c2 ? bxq(cur) + ys_off + (size_t)(tid - 32) * Kp : bwq(cur), run, n);
                } else {
                    if (n_prev > 0) write_out(bxq(cur ^ 1), bwq(cur ^ 1), first_prev, n_prev, n_prev + 1, before_prev);
                    if (n_next > 0) stage(bxq(cur ^ 1), bwq(cur ^ 1), qhead + n, n_next);
                }
                __syncthreads();
                CC_TICK(2);
                before_prev = (n_prev > 0) ? s_queue[(first_prev + n_prev - 1) & (QUEUE - 1)] : -1;
                first_prev = qhead; n_prev = n;
                qhead = (qhead + n) & (QUEUE - 1);
                qcount -= n;
                walked += n;
                cur ^= 1;
                n = n_next;
                if (n == 0) break;
                collect(2 * K + 1);
                __syncthreads();
                CC_TICK(0);
#ifdef CC_LONG_TIMERS
                tk_acc[7] += 1ull;
#endif
            }
            // the last batch's sums (nothing behind it: the chain ends there)
            if (tid >= 64 && n_prev > 0) write_out(bxq(cur ^ 1), bwq(cur ^ 1), first_prev, n_prev, n_prev, before_prev);
            CC_TICK(5);
#ifdef CC_LONG_TIMERS
            tk_acc[7] += 1ull;
            if (tid == 0 && blockIdx.x == 0)
                for (int i = 0; i < 8; ++i) atomicAdd(&ctl->dbg_long[i], tk_acc[i]);
#endif
            if (tid == 0) tab.clen[t] = walked;
            return;
        }
    }
    for (;;) {
        // ---- 1. members in order ----
        collect(K + 1);
        __syncthreads();
        CC_TICK(0);  // member collection
        if (qcount == 0) break;
        const int n = qcount < K ? qcount : K;  // steps of this batch (the member after it is known, or the chain ends)

        // ---- 2. stage the points, then the sequential additions per dimension ----
        // (the coordinates of the members the queue already held behind the previous batch were requested while that batch
        // was evaluated - `pre`, valid if that batch consumed exactly its n members: a memory round trip per batch less)
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
            const int e = tid + q * NT;
            if (e < n * d) {
                const int k = e / d, i = e - k * d;
                double x;
                if (e < pre_elems) x = pre[q];
                else x = X[(cursor + s_queue[(qhead + k) & (QUEUE - 1)]) * d + i];
                xs[at(k, i)] = x;
                ys[at(k, i)] = x * x;  // (the square every step adds to CF2: formed here, by all threads, instead of inside the chain)
            }
        }
        // the next batch's coordinates, as far as its members are known: in flight during this batch's phases
        {
            const int n_ahead = min(K, qcount - n);
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const int e = tid + q * NT;
                if (e < n_ahead * d) {
                    const int k = e / d, i = e - k * d;
                    pre[q] = X[(cursor + s_queue[(qhead + n + k) & (QUEUE - 1)]) * d + i];
                }
            }
            pre_next = n_ahead * d;
        }
        if (tid == 0) { s_first_fail = n; s_first_up = n; }
        if (one_wave)
            for (int k = tid; k < n; k += NT) s_w[k] = 1.0;  // (what the W chain adds per step)
        __syncthreads();
        CC_TICK(1);  // staging
        // The three running sums are chains of dependent additions, a dozen instructions per step for ONE wave that has
        // its SIMD to itself: CF1 in the first wave, CF2 in the second, W in the third, side by side
        // (mc_functions.py:24-29 / microcluster.py:147: the additions k_chain makes, in its order).
        // (sixteen steps at a time: the sixteen LDS reads go out together, then the sixteen additions in order, then the
        // sixteen writes - read, add, write per step would wait out one LDS round trip per step)
        if (one_wave) {
            // (round 5) d <= 31: all three in ONE wave - CF1 in lanes 0 .., CF2 in lanes 32 .., W (a row of ones, see the
            // staging) in lane 63.  The LDS serves a read or write of a wave in the same time whatever the number of
            // active lanes (16 cycles per 16-byte-per-lane instruction, tools/micro/dep_add.hip), and three waves' requests
            // queue up behind one another: 43 cycles per step side by side, 29 in one wave.
            if (tid < 64) {
                const bool c1 = tid < d, c2 = tid >= 32 && tid < 32 + d, cw = tid == 63;
                if (c1 || c2 || cw)
                    prefix_chain(c1 ? xs + (size_t)tid * Kp : c2 ? ys + (size_t)(tid - 32) * Kp : s_w,
                                 c1 ? s_b1[tid] : c2 ? s_b2[tid - 32] : s_bw, n);
            }
        } else if (tid < d) {
            prefix_chain(xs + (size_t)tid * Kp, s_b1[tid], n);
        } else if (tid >= 64 && tid < 64 + d) {
            prefix_chain(ys + (size_t)(tid - 64) * Kp, s_b2[tid - 64], n);
        } else if (tid == 128) {
            // (unrolled like the two above: with a loop per step, this chain - one addition per step - was the slowest of
            // the three once theirs had lost their address arithmetic)
            double w = s_bw;
            int k = 0;
            for (; k + 16 <= n; k += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    w = w + 1.0;
                    v[u] = w;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) s_w[k + u] = v[u];
            }
            for (; k < n; ++k) {
                w = w + 1.0;
                s_w[k] = w;
            }
        }
        __syncthreads();
        CC_TICK(2);  // chains

        if constexpr (PREP) {
            // the sums after every member, its neighbours in the chain: what k_chain's group of the member needs for its step
            for (int e = tid; e < n * d; e += NT) {
                const int k = e / d, i = e - k * d;
                const size_t o = (size_t)s_queue[(qhead + k) & (QUEUE - 1)] * d + i;
                ver.cf1[o] = xs[at(k, i)]; ver.cf2[o] = ys[at(k, i)];
            }
            if (tid < n) {
                const int k = tid;
                const int m = s_queue[(qhead + k) & (QUEUE - 1)];
                const int pv = (k > 0) ? s_queue[(qhead + k - 1) & (QUEUE - 1)] : prev_last;
                ver.w[m] = s_w[k];
                ver.next[m] = (k + 1 < qcount) ? s_queue[(qhead + k + 1) & (QUEUE - 1)] : CC_IDX_INF;
                lprev[m] = CC_LPREV(stamp, blockIdx.x, pv);
            }
            prev_last = s_queue[(qhead + n - 1) & (QUEUE - 1)];
            __syncthreads();
            CC_TICK(5);  // (version rows: the sums)
            const int l = n - 1;
            if (tid < d) { s_b1[tid] = xs[at(l, tid)]; s_b2[tid] = ys[at(l, tid)]; }
            if (tid == 64) s_bw = s_w[l];
            qhead = (qhead + n) & (QUEUE - 1);
            qcount -= n;
            walked += n;
            pre_elems = pre_next;
            __syncthreads();
            CC_TICK(6);
#ifdef CC_LONG_TIMERS
            tk_acc[7] += 1ull;
#endif
            continue;
        }

        // ---- 3. every step evaluated on its own prefix ----
        // (a) per (step, dimension), all threads: the two quotients, the variance, the term of the radius sum - the
        // divisions are most of a step's instructions and independent of one another
        if constexpr (SPLIT) {
        for (int e = tid; e < n * d; e += NT) {
            const int k = e / d, i = e - k * d;
            const double w1 = s_w[k];
            const double qa = ys[at(k, i)] / w1;  // mc_functions.py:14-22 (cc_sqvar), keeping CF1 / W
            const double qb = xs[at(k, i)] / w1;
            const double var = qa - qb * qb;
            const bool prefd = var <= par.delta_sq;  // microcluster.py:109-114 (NaN -> 1.0)
            const double pr = prefd ? par.k : 1.0;
            s_term[e] = cc_div_pref(var, pr, par);
            s_pf[e] = prefd ? (unsigned char)1 : (unsigned char)0;
            const double df = qb - s_c0[i];
            s_dqt[e] = df * df * s_w0[i];
        }
        __syncthreads();
        CC_TICK(3);  // per (step, dimension)
        }
        // (b) per step, one thread: the ordered sums over the dimensions (!SPLIT: the terms themselves as well)
        // (dealing the steps round all sixteen waves instead of packing them into four was timed: 7.0 -> 10.5 M cycles)
        if (tid < n) {
            const int k = tid;
            const double w1 = s_w[k];
            double r2 = 0.0, dq = 0.0;
            int gt1 = 0;
            unsigned long long mask = 0ull;
            for (int i = 0; i < d; ++i) {
                if constexpr (SPLIT) {
                    r2 = r2 + s_term[k * d + i];             // mc_functions.py:54, left to right
                    dq += s_dqt[k * d + i];
                    const bool prefd = s_pf[k * d + i] != 0;
                    gt1 += (prefd && par.k > 1.0) ? 1 : 0;    // count(pref' > 1): pref' = k where preferred
                    mask |= prefd ? (1ull << i) : 0ull;
                } else {
                    const double qa = ys[at(k, i)] / w1;  // mc_functions.py:14-22 (cc_sqvar), keeping CF1 / W
                    const double qb = xs[at(k, i)] / w1;
                    const double var = qa - qb * qb;
                    const bool prefd = var <= par.delta_sq;  // microcluster.py:109-114 (NaN -> 1.0)
                    const double pr = prefd ? par.k : 1.0;
                    r2 = r2 + cc_div_pref(var, pr, par);     // mc_functions.py:54, left to right
                    gt1 += (pr > 1.0) ? 1 : 0;
                    mask |= prefd ? (1ull << i) : 0ull;
                    const double df = qb - s_c0[i];
                    dq += df * df * s_w0[i];
                }
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
        CC_TICK(4);  // per step
        const int f = s_first_fail;                 // first rejected step (n: none)
        const int n_ok = f < n ? f : n;             // accepted steps 0 .. n_ok - 1
        const int n_rows = f < n ? f + 1 : n;       // members consumed by this batch (the rejected one included)
        // hddstream.py:416-430: the first accepted add to an outlier MC that fulfils the condition promotes it
        int u = -1;
        if (bkind == CC_KIND_OUTLIER && s_first_up < n_ok) u = s_first_up;
        const int up_point = (u >= 0) ? s_queue[(qhead + u) & (QUEUE - 1)] : -1;

        // ---- 4. version rows: vectors by (row, dimension), the rest by row ----
        for (int e = tid; e < n_rows * d; e += NT) {
            const int k = e / d, i = e - k * d;
            const int m = s_queue[(qhead + k) & (QUEUE - 1)];
            const int src = (k < n_ok) ? k : k - 1;  // a rejected step leaves the state of the step before it
            double c1, c2, ce, pr;
            if (src >= 0) {
                c1 = xs[at(src, i)]; c2 = ys[at(src, i)];
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
            const int m = s_queue[(qhead + k) & (QUEUE - 1)];
            const int src = (k < n_ok) ? k : k - 1;
            const bool promoted = u >= 0 && k >= u;
            const int kind = promoted ? CC_KIND_PCORE : bkind;
            double dq = (src >= 0) ? s_dq[src] : s_bdq;
            // (no bound either when the preferred dimensions differ from the snapshot's, see k_chain)
            const unsigned long long vmask = (src >= 0) ? s_mask[src] : s_bmask;
            const bool was_outlier = kind == CC_KIND_PCORE && kind0 == CC_KIND_OUTLIER;  // promoted since the snapshot: class 2
            if ((kind != kind0 && !was_outlier) || !(dq >= 0.0) || vmask != s_m0) dq = CC_INF;
            const int nx = (k + 1 < qcount) ? s_queue[(qhead + k + 1) & (QUEUE - 1)] : CC_IDX_INF;
            ver.w[m] = (src >= 0) ? s_w[src] : s_bw;
            ver.tgt[m] = t;
            ver.kind[m] = kind;
            ver.key[m] = promoted ? pk_base + up_point : bkey;
            ver.upg[m] = promoted ? up_point : bupg;
            ver.acc[m] = (k < n_ok) ? 1 : 0;
            ver.next[m] = nx;
            int cls;
            ver.dsq[m] = cc_dsq_store(dq, kind, kind0, &cls);
            atomicMax(&ver.tile_dsq[(size_t)(m >> 4) * CC_DSQ_STRIDE + cls], cc_dsq_code(dq));
        }
        __syncthreads();  // every read of the running state and of the queue slots is done
        CC_TICK(5);  // version rows

        // ---- 5. the running state moves on to the last accepted step ----
        if (n_ok > 0) {
            const int l = n_ok - 1;
            if (tid < d) {
                s_b1[tid] = xs[at(l, tid)]; s_b2[tid] = ys[at(l, tid)];
                s_bcen[tid] = xs[at(l, tid)] / s_w[l];
                s_bpref[tid] = ((s_mask[l] >> tid) & 1ull) ? par.k : 1.0;
            }
            if (tid == 64) { s_bw = s_w[l]; s_bdq = s_dq[l]; s_bmask = s_mask[l]; }
        }
        if (u >= 0) {
            bkind = CC_KIND_PCORE; bkey = pk_base + up_point; bupg = up_point;
            promoted_any = true;
        }
        qhead = (qhead + n_rows) & (QUEUE - 1);
        qcount -= n_rows;
        walked += n_rows;
        pre_elems = (n_rows == n) ? pre_next : 0;  // (a rejected step ends the batch early: the next one starts elsewhere)
        __syncthreads();
        CC_TICK(6);  // state moves on
#ifdef CC_LONG_TIMERS
        tk_acc[7] += 1ull;
#endif
    }
#ifdef CC_LONG_TIMERS
    if (tid == 0 && blockIdx.x == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&ctl->dbg_long[i], tk_acc[i]);
#endif
    if (tid == 0) {
        if (!resumed) tab.clen[t] = walked;  // k_dseed chooses its way of finding live versions by it (a resumed replay: the whole chain's count is there already)
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
                                                   int* __restrict__ rk, CommitRec* __restrict__ rec,
                                                   const Cand* __restrict__ clean, const int8_t* __restrict__ dpath)
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
    // heavy rows (Table::heavy): rows marked 2 leave the list, this window's nominations join it (the last wave; the
    // list and the marks change nowhere else, so k_decide and k_claims_heavy of one window always agree on them)
    if (threadIdx.x >= 960 && (ctl->n_heavy > 0 || ctl->n_heavy_new > 0)) {
        const int lane = (int)threadIdx.x - 960;
        const unsigned long long below = (1ull << lane) - 1ull;
        const int n_h = min(ctl->n_heavy, CC_HEAVY_CAP);
        const int row = lane < n_h ? ctl->heavy_list[lane] : -1;
        const bool keep = row >= 0 && tab.heavy[row] == 1;
        if (row >= 0 && !keep) tab.heavy[row] = 0;
        const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
        const int n_keep = __builtin_popcountll(km);
        const int n_new = min(ctl->n_heavy_new, CC_HEAVY_NEW);
        const int nr = lane < n_new ? ctl->heavy_new[lane] : -1;
        bool add = nr >= 0 && atomicExch(&tab.heavy[nr], 1) == 0;  // (a row nominated twice is taken once)
        const unsigned long long am = __builtin_amdgcn_ballot_w64(add);
        const int apos = n_keep + __builtin_popcountll(am & below);
        if (add && apos >= CC_HEAVY_CAP) {
            tab.heavy[nr] = 0;  // no room
            add = false;
        }
        if (keep) ctl->heavy_list[__builtin_popcountll(km & below)] = row;
        if (add) ctl->heavy_list[apos] = nr;
        if (lane == 0) {
            ctl->n_heavy = min(n_keep + __builtin_popcountll(am), CC_HEAVY_CAP);
            ctl->n_heavy_new = 0;
        }
    }
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
        for (int i = tid; i < CC_DSQ_STRIDE * ((B + 15) / 16 + 1); i += 1024) car.tile_dsq[i] = 0ull;  // k_commit_b takes maxima into them
    // Ctl::tg - the mean snapshot distance at which the committed points joined a MC of either kind: what a later
    // window's pruned scan takes its guessed thresholds from (k_scan_p).  Slot = parity of this window; a window without
    // such points passes on what the previous one left.
    __shared__ double s_sum[2][16];
    __shared__ int s_cnt[2][16];
    {
        double sm[2] = {0.0, 0.0};
        int cn[2] = {0, 0};
        // (every eighth point: a mean is all that is wanted, and three dependent loads per point are 15 us of a kernel
        // the whole window waits for when every point is read)
        for (int j = tid * 8; j < n; j += 8192) {
            const int pth = dpath[j];
            if (T[j] >= 0 && (pth == 0 || pth == 1)) {
                const double dist = clean[(size_t)j * 4 + pth * 2].dist;
                if (dist < CC_INF) { sm[pth] += dist; cn[pth] += 1; }
            }
        }
#pragma unroll
        for (int K = 0; K < 2; ++K) {
            for (int off = 32; off >= 1; off >>= 1) {
                sm[K] += __shfl_xor(sm[K], off);
                cn[K] += __shfl_xor(cn[K], off);
            }
            if ((tid & 63) == 0) { s_sum[K][tid >> 6] = sm[K]; s_cnt[K][tid >> 6] = cn[K]; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int slot = (int)(seq & 1ull);
        for (int K = 0; K < 2; ++K) {
            double sm = 0.0;
            int cn = 0;
            for (int i = 0; i < 16; ++i) { sm += s_sum[K][i]; cn += s_cnt[K][i]; }
            if (cn > 0) { ctl->tg[slot][K] = sm / (double)cn; ctl->tg_ok[slot][K] = 1; }
            else { ctl->tg[slot][K] = ctl->tg[slot ^ 1][K]; ctl->tg_ok[slot][K] = ctl->tg_ok[slot ^ 1][K]; }
        }
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
        ctl->seed_at = (n < B && T[n] == CC_T_UNKNOWN) ? cursor + n : -1ll;
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
            const bool promoted = kind == CC_KIND_PCORE && kind0 == CC_KIND_OUTLIER;  // class 2, see k_chain
            if (kind0 == CC_KIND_DEAD || (kind != kind0 && !promoted) || !(dq >= 0.0) || cc_group_ballot(metric_moved) != 0u) dq = CC_INF;
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
                int cls;
                car.dsq[j] = cc_dsq_store(dq, kind, kind0, &cls);
                atomicMax(&car.tile_dsq[(size_t)(j >> 4) * CC_DSQ_STRIDE + cls], cc_dsq_code(dq));
                tab.carry_of[row] = (rec->next_seq << 20) | (unsigned long long)j;
            }
        }
    }
}
