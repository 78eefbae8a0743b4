// chronoclust_amd/csrc: k_seq_g, the sequential kernel on the table in HBM.  (included by cc_online.h; one translation unit, cc_api.hip)
#pragma once

// ---------------------------------------------------------------------------------
// k_seq_g: the reference's loop taken literally (hddstream.py:220-237) like k_seq, for the streams k_seq is made for -
// overlapping microclusters, decisions that keep moving, windows cut short after a handful of points - once the table has
// outgrown k_seq's LDS image (77 rows at d = 20, 25 at d = 64).  There the windowed path commits ~10 points per
// millisecond whatever is done to its kernels; this kernel takes 8-14 us per point at 150-450 rows
// (cycles per point with the -DCC_SEQG_TIMERS build at d = 20 / 434 rows: scan 12 k, workgroup minimum 2 k, the first wave's
// tentative add 6 k; at d = 128 / 535 rows with the pdim filter 50 k / 10 k / 15 k - tools/overlap_stream.py).
// One workgroup of 1 024 threads works on the table where it lies:
//   per point and stage (pcore rows, then outlier rows: hddstream.py:288-343 / 345-395) every thread takes the rows
//   q = tid, tid + 1 024, ... of the stage's list (row indices in HBM scratch) - projected distance over the dimensions
//   left to right (mc_functions.py:35-43), with the tentative-add pdim filter when pi < d -, the workgroup's minimum by
//   (distance, list-order key) goes through DPP row operations and one LDS exchange, the first wave makes the tentative
//   add of the winner with lane = dimension (microcluster.py:213-233), the ordered radius sum and the test, and commits
//   the row (and a promotion, hddstream.py:416-430) in place; a point nobody absorbs opens a new row (:434-462).
// Two barriers per stage: the waves of one workgroup share their CU's vector L1, a barrier orders the first wave's stores
// before everybody's loads of the next point.  Same arithmetic, same order of operations as k_seq.
// ---------------------------------------------------------------------------------

#define CC_SEQG_THREADS 1024
#define CC_SEQG_CHUNK_DOUBLES 512  // points staged per chunk: 512 / d of them (four at d = 128), at most 64

template <bool FILTER, bool POW2>
__global__ __launch_bounds__(CC_SEQG_THREADS) void k_seq_g(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                                            long long* __restrict__ lab_uid, int8_t* __restrict__ lab_path,
                                                            int n_max, int* __restrict__ lists, int list_cap,
                                                            double* __restrict__ img)
{
    const long long clk0 = clock64(), wall0 = wall_clock64();
    const Par par = cc_load_par(ctl);
    const int d = par.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = CC_SEQG_THREADS, NW = CC_SEQG_THREADS / 64;
    int M = ctl->m_rows;
    const long long cursor0 = ctl->cursor;
    const long long left = ctl->n_points - cursor0;
    const int n = (int)(left < (long long)n_max ? left : (long long)n_max);
    if (n <= 0) return;
    int n_pkeys = ctl->n_pkeys, n_okeys = ctl->n_okeys;
    long long pcore_last_id = ctl->pcore_last_id, outlier_last_id = ctl->outlier_last_id;
    constexpr bool pow2 = POW2;
    auto op_of = [&](double pr) { return pow2 ? (pr == 1.0 ? 1.0 : par.inv_k) : pr; };
    // x / pref through the operand (mc_functions.py:39)
    auto scaled = [&](double x, double op) { return pow2 ? x * op : (op == 1.0 ? x : x / op); };

    __shared__ __attribute__((aligned(16))) double s_pts[CC_SEQG_CHUNK_DOUBLES];
    __shared__ double s_cd[NW];          // per wave: best distance, key, row, list position
    __shared__ int s_ck[NW], s_cr[NW], s_cq[NW];
    __shared__ int s_np, s_no;
    __shared__ int s_verdict;            // the first wave's word on the winner: 0 rejected, 1 accepted, 3 accepted and promoted
    __shared__ int s_tgt[64], s_lpath[64];
    int* const plist = lists;            // rows of the pcore MCs / of the outlier MCs, any order
    int* const olist = lists + list_cap;
    // What the threads scan is a dimension-major copy of the rows' centroids and distance operands (and, for the pdim
    // filter, of CF1 / CF2): entry (dimension i, row r) at i * list_cap + r, so that the loads of a wave - consecutive rows,
    // one dimension - coalesce.  (Walking the table's own row-major rows, every load of a wave touched 64 cache lines: 18 us
    // per point at d = 64 and 159 rows, 14 this way; at d = 20 it makes no difference.)  The first wave keeps it in line with every row it commits.
    const size_t plane = (size_t)d * (size_t)list_cap;
    double* const icen = img;
    double* const iscl = img + plane;
    double* const ic1 = img + 2 * plane;
    double* const ic2 = img + 3 * plane;
    for (size_t e = tid; e < (size_t)M * d; e += NT) {
        const size_t r = e / d, i = e - r * d;
        icen[i * list_cap + r] = tab.cen[e]; iscl[i * list_cap + r] = tab.scl[e];
        if (FILTER) { ic1[i * list_cap + r] = tab.cf1[e]; ic2[i * list_cap + r] = tab.cf2[e]; }
    }

    if (tid == 0) { s_np = 0; s_no = 0; }
    __syncthreads();
    for (int r = tid; r < M; r += NT) {
        if (tab.kind[r] == CC_KIND_PCORE) plist[atomicAdd(&s_np, 1)] = r;
        else olist[atomicAdd(&s_no, 1)] = r;
    }
    __syncthreads();
    int n_p = s_np, n_o = s_no;
#ifdef CC_SEQG_TIMERS
    // build variant (-DCC_SEQG_TIMERS -DCC_LONG_TIMERS: the latter declares Ctl::dbg_long): shader cycles per phase, thread 0,
    // printed at the end of a call
    long long tq_prev = clock64();
    unsigned long long tq_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define CC_TQ(i) do { const long long now_ = clock64(); tq_acc[i] += (unsigned long long)(now_ - tq_prev); tq_prev = now_; } while (0)
#else
#define CC_TQ(i) do { } while (0)
#endif

    const int C = (CC_SEQG_CHUNK_DOUBLES / d) < 64 ? (CC_SEQG_CHUNK_DOUBLES / d) : 64;
    int done = 0;
    bool full = false;
    for (int c0 = 0; c0 < n && !full; c0 += C) {
        const int cnt = (n - c0) < C ? (n - c0) : C;
        __syncthreads();  // (the previous chunk's points and targets have been consumed)
        for (int e = tid; e < cnt * d; e += NT) s_pts[e] = X[(cursor0 + c0) * d + e];
        __syncthreads();
        CC_TQ(5);  // chunk of points
        int cdone = 0;
        for (int jj = 0; jj < cnt; ++jj) {
            const double* sp = s_pts + jj * d;
            // (first wave) this lane's dimensions of the point: lane and lane + 64 (d <= 128 = CC_MAX_DIM)
            const double myp[2] = {(tid < d) ? sp[tid] : 0.0, (tid < 64 && tid + 64 < d) ? sp[tid + 64] : 0.0};
            int target = -1, path = 2;
            bool promoted = false;
            // stage 0: _add_to_pcore (hddstream.py:288-343), stage 1: _add_to_outlier (:345-395)
            for (int stage = 0; stage < 2 && target < 0; ++stage) {
                const int* list = stage == 0 ? plist : olist;
                const int n_list = stage == 0 ? n_p : n_o;
                if (n_list == 0) continue;
                double bd = CC_INF;
                int bk = CC_IDX_INF, br = -1, bq = -1;
#pragma nounroll
                for (int q = tid; q < n_list; q += NT) {
                    const int r = list[q];
                    const double* const rcen = icen + r;
                    const double* const rscl = iscl + r;
                    double acc = 0.0;
                    // (a pass over the dimensions is a chain of round trips to L2, ~1 700 cycles each, whatever is computed in
                    // between - the filter's divisions included: as many loads per trip as the registers take)
                    if (stage == 0 && FILTER) {
                        // hddstream.py:317-321: pdim of the MC with the point added must be <= pi - evaluated beside the
                        // distance, whose loads share the trip (a row the filter drops has its distance computed in vain)
                        // Only the verdict "var <= delta^2" per dimension is needed, and the reference's var = fl(fl(c2 / w) -
                        // fl(fl(c1 / w)^2)) costs two IEEE divisions - ~70 instructions, 280 cycles of a wave's issue - per
                        // dimension and row: three quarters of the scan at d = 128.  With one reciprocal per row,
                        // v = c2 rw - (c1 rw)^2 differs from it by at most 10 u (|A| + B^2), u = 2^-53 (the reference's expression:
                        // within 2 u |A| + 4 u B^2 of the exact A - B^2; v: within 3 u |A| + 6 u B^2): whenever v is further than
                        // 4e-15 (|a| + b^2) from delta^2 the verdict is certain; a wave in which some lane is not certain (or
                        // sees something that is not finite) evaluates the reference's own expression for those dimensions.
                        const double w1 = tab.w[r] + 1.0;
                        const double rw = 1.0 / w1;
                        const double* const rc1 = ic1 + r;
                        const double* const rc2 = ic2 + r;
                        int ne1 = 0;
                        int i = 0;
#pragma nounroll
                        for (; i + 8 <= d; i += 8) {
                            double a1[8], a2[8], e[8], o[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const size_t at = (size_t)(i + u) * list_cap;
                                a1[u] = rc1[at]; a2[u] = rc2[at]; e[u] = rcen[at]; o[u] = rscl[at];
                            }
                            unsigned unsure = 0u;
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const double x = sp[i + u];
                                a1[u] = a1[u] + x; a2[u] = a2[u] + x * x;  // (the sums the reference divides)
                                const double a = a2[u] * rw, b = a1[u] * rw;
                                const double bb = b * b;
                                const double v = a - bb;
                                const double slack = 4e-15 * (fabs(a) + bb);
                                const bool yes = par.delta_sq - v > slack, no = v - par.delta_sq > slack;
                                ne1 += (yes && par.k != 1.0) ? 1 : 0;
                                unsure |= (!yes && !no) ? (1u << u) : 0u;
                                double y = x - e[u];               // mc_functions.py:37
                                y = y * y;                          // :38
                                acc = acc + scaled(y, o[u]);        // :39 + :41
                            }
                            if (__builtin_amdgcn_ballot_w64(unsure != 0u) != 0ull) {
#pragma unroll
                                for (int u = 0; u < 8; ++u)
                                    if ((unsure >> u) & 1u) {
                                        const double var = cc_sqvar(a1[u], a2[u], w1);
                                        ne1 += (((var <= par.delta_sq) ? par.k : 1.0) != 1.0) ? 1 : 0;
                                    }
                            }
                        }
#pragma nounroll
                        for (; i < d; ++i) {
                            const size_t at = (size_t)i * list_cap;
                            const double x = sp[i];
                            const double var = cc_sqvar(rc1[at] + x, rc2[at] + x * x, w1);
                            ne1 += (((var <= par.delta_sq) ? par.k : 1.0) != 1.0) ? 1 : 0;
                            double y = x - rcen[at];
                            y = y * y;
                            acc = acc + scaled(y, rscl[at]);
                        }
                        if (ne1 > par.pi) continue;
                    } else {
                        int i = 0;
#pragma nounroll
                        for (; i + 16 <= d; i += 16) {  // the loads of sixteen dimensions together, sums left to right
                            double e[16], o[16];
#pragma unroll
                            for (int u = 0; u < 16; ++u) { e[u] = rcen[(size_t)(i + u) * list_cap]; o[u] = rscl[(size_t)(i + u) * list_cap]; }
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                double x = sp[i + u] - e[u];       // mc_functions.py:37
                                x = x * x;                          // :38
                                acc = acc + scaled(x, o[u]);        // :39 + :41
                            }
                        }
#pragma nounroll
                        for (; i + 4 <= d; i += 4) {
                            double e[4], o[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) { e[u] = rcen[(size_t)(i + u) * list_cap]; o[u] = rscl[(size_t)(i + u) * list_cap]; }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                double x = sp[i + u] - e[u];
                                x = x * x;
                                acc = acc + scaled(x, o[u]);
                            }
                        }
#pragma nounroll
                        for (; i < d; ++i) {
                            double x = sp[i] - rcen[(size_t)i * list_cap];
                            x = x * x;
                            acc = acc + scaled(x, rscl[(size_t)i * list_cap]);
                        }
                    }
                    const int key = tab.key[r];
                    if (cand_less(acc, key, bd, bk)) { bd = acc; bk = key; br = r; bq = q; }  // strict <, first in list order wins (:326/:373)
                }
                CC_TQ(0);  // scan
                // the wave's minimum by (distance, key) ...
                {
                    const double D = cc_wave_min_f64(bd);
                    const unsigned long long tied = __builtin_amdgcn_ballot_w64(br >= 0 && bd == D);
                    int wl = 0;
                    if (tied != 0ull) {
                        wl = __builtin_ctzll(tied);
                        if (tied & (tied - 1ull)) {
                            int best_key = CC_IDX_INF;
                            for (unsigned long long m = tied; m; m &= m - 1ull) {
                                const int l = __builtin_ctzll(m);
                                const int k2 = __builtin_amdgcn_readlane(bk, l);
                                if (k2 < best_key) { best_key = k2; wl = l; }
                            }
                        }
                    }
                    const int wr = __builtin_amdgcn_readlane(br, wl), wk = __builtin_amdgcn_readlane(bk, wl);
                    const int wq = __builtin_amdgcn_readlane(bq, wl);
                    if (lane == 0) {
                        s_cd[wave] = (tied != 0ull) ? D : CC_INF;
                        s_ck[wave] = (tied != 0ull) ? wk : CC_IDX_INF;
                        s_cr[wave] = (tied != 0ull) ? wr : -1;
                        s_cq[wave] = wq;
                    }
                }
                __syncthreads();
                CC_TQ(1);  // wave minimum + barrier
                // ... and the workgroup's (every thread looks at the sixteen entries: the same winner everywhere)
                double D = CC_INF;
                int K = CC_IDX_INF, R = -1, Q = -1;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const double wd = s_cd[w];
                    const int wk = s_ck[w], wr = s_cr[w];
                    if (wr >= 0 && cand_less(wd, wk, D, K)) { D = wd; K = wk; R = wr; Q = s_cq[w]; }
                }
                if (R < 0) {  // no (admissible) MC of this kind
                    __syncthreads();  // (the entries are rewritten by the next stage)
                    continue;
                }
                // tentative add (microcluster.py:213-233) with lane = dimension, then the radius test (:334-337 / :378-381)
                if (wave == 0) {
                    const double w1 = tab.w[R] + 1.0;
                    double c1[2] = {0.0, 0.0}, c2[2] = {0.0, 0.0}, qb[2] = {0.0, 0.0}, pr[2] = {1.0, 1.0}, term[2] = {0.0, 0.0};
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int i = lane + 64 * hh;
                        if (i < d) {
                            c1[hh] = tab.cf1[(size_t)R * d + i] + myp[hh];
                            c2[hh] = tab.cf2[(size_t)R * d + i] + myp[hh] * myp[hh];
                            const double qa = c2[hh] / w1;
                            qb[hh] = c1[hh] / w1;
                            const double var = qa - qb[hh] * qb[hh];
                            pr[hh] = (var <= par.delta_sq) ? par.k : 1.0;
                            term[hh] = scaled(var, op_of(pr[hh]));  // mc_functions.py:52: var / pref'
                        }
                    }
                    double r2 = 0.0;
                    {
                        const int d0 = d < 64 ? d : 64;
#pragma nounroll
                        for (int i = 0; i < d0; ++i) r2 = r2 + cc_readlane_f64(term[0], i);  // mc_functions.py:54, left to right
#pragma nounroll
                        for (int i = 64; i < d; ++i) r2 = r2 + cc_readlane_f64(term[1], i - 64);
                    }
                    int verdict = 0;
                    if (r2 <= par.eps_sq) {
                        verdict = 1;
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int i = lane + 64 * hh;
                            if (i < d) {
                                const size_t o = (size_t)R * d + i;
                                tab.cf1[o] = c1[hh]; tab.cf2[o] = c2[hh]; tab.cen[o] = qb[hh]; tab.pref[o] = pr[hh]; tab.scl[o] = op_of(pr[hh]);
                                const size_t io = (size_t)i * list_cap + R;
                                icen[io] = qb[hh]; iscl[io] = op_of(pr[hh]);
                                if (FILTER) { ic1[io] = c1[hh]; ic2[io] = c2[hh]; }
                            }
                        }
                        if (lane == 0) tab.w[R] = w1;
                        if (stage == 1) {
                            // hddstream.py:416-430
                            const int gt1 = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < d && pr[0] > 1.0)) +
                                            __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane + 64 < d && pr[1] > 1.0));
                            if (w1 >= par.beta_mu && gt1 <= par.pi) {
                                verdict = 3;
                                // out of the outlier list (the last entry takes its place), onto the pcore list
                                if (lane == 0) {
                                    tab.kind[R] = CC_KIND_PCORE; tab.key[R] = n_pkeys; tab.id[R] = pcore_last_id;
                                    olist[Q] = olist[n_o - 1];
                                    plist[n_p] = R;
                                }
                            }
                        }
                    }
                    if (lane == 0) s_verdict = verdict;
                }
                CC_TQ(2);  // selection + tentative add (first wave)
                __syncthreads();  // the row as committed is what the next point sees
                CC_TQ(3);  // barrier
                const int verdict = s_verdict;
                if (verdict == 0) continue;
                target = R;
                path = stage;
                if (verdict == 3) {
                    promoted = true;
                    n_o -= 1;
                    n_p += 1;
                    n_pkeys += 1;
                    pcore_last_id += 1;
                }
            }
            if (target < 0) {
                // hddstream.py:434-462: a new outlier MC holding this point (an add to an empty MC)
                if (M >= (int)tab.cap || n_o >= list_cap) { full = true; break; }  // (the host makes room and comes back)
                const int R = M;
                if (wave == 0) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int i = lane + 64 * hh;
                        if (i < d) {
                            const double c1 = 0.0 + myp[hh], c2 = 0.0 + myp[hh] * myp[hh];
                            const double qa = c2 / 1.0, qb = c1 / 1.0;
                            const double var = qa - qb * qb;
                            const double pr = (var <= par.delta_sq) ? par.k : 1.0;
                            const size_t o = (size_t)R * d + i;
                            tab.cf1[o] = c1; tab.cf2[o] = c2; tab.cen[o] = qb; tab.pref[o] = pr; tab.scl[o] = op_of(pr);
                            const size_t io = (size_t)i * list_cap + R;
                            icen[io] = qb; iscl[io] = op_of(pr);
                            if (FILTER) { ic1[io] = c1; ic2[io] = c2; }
                        }
                    }
                    if (lane == 0) {
                        tab.w[R] = 0.0 + 1.0; tab.kind[R] = CC_KIND_OUTLIER; tab.key[R] = n_okeys; tab.id[R] = outlier_last_id;
                        tab.uid[R] = outlier_last_id;
                        olist[n_o] = R;
                    }
                }
                n_o += 1;
                n_okeys += 1;
                outlier_last_id += 1;
                M += 1;
                target = R;
                path = 2;
                __syncthreads();
            }
            if (tid == 0) {
                s_tgt[jj] = target;
                s_lpath[jj] = path | (promoted ? 4 : 0);
            }
            cdone = jj + 1;
            CC_TQ(4);  // new row, bookkeeping
        }
        __syncthreads();
        if (tid < cdone) {
            lab_uid[cursor0 + c0 + tid] = tab.uid[s_tgt[tid]];  // (a row's creation number never changes)
            lab_path[cursor0 + c0 + tid] = (int8_t)s_lpath[tid];
        }
        done = c0 + cdone;
    }
    __syncthreads();
#ifdef CC_SEQG_TIMERS
    if (tid == 0) {
        tq_acc[7] = (unsigned long long)done;
        for (int i = 0; i < 8; ++i) atomicAdd(&ctl->dbg_long[i], tq_acc[i]);
    }
#endif
    if (tid == 0) {
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
