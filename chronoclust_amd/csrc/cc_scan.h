// chronoclust_amd/csrc: the snapshot scans - k_scan (LDS-staged rows; also the dirty scans), k_scan_u (rows as scalar operands), the pruned chain k_seed / k_seed_merge / k_scan_p, k_merge_partials.  (included by cc_online.h; one translation unit, cc_api.hip)
#pragma once

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
    // Sparse dirty scans: the tile is not 64 consecutive window points but 64 entries of the round's list of points
    // that need rows only a dirty scan covers (k_dseed compacted them: a few per cent of the window, in any order).
    const bool sparse = DIRTY && rows.plist != nullptr;
    const int n_list = sparse ? ctl->n_sparse : 0;
    if (sparse ? (j0 >= n_list) : (j0 >= B)) return;
    if (DIRTY && !sparse && rows.skip[bx] != 0) return;  // k_dseed: no row can matter to this tile; k_decide takes the seeds
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
    int jj[PT];
    bool valid[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const int e = j0 + t * 64 + lane;
        if (sparse) {
            valid[t] = e < n_list;
            jj[t] = valid[t] ? rows.plist[e] : 0;
        } else {
            jj[t] = e;
            valid[t] = e < B;
        }
    }
    // (version rows at or beyond the tile's last point cannot matter to it)
    int rows_upto = min(B, j0 + 64 * PT - 1);
    if (sparse) {
        int mx = 0;
#pragma unroll
        for (int t = 0; t < PT; ++t) mx = max(mx, valid[t] ? jj[t] : 0);
        for (int off = 32; off >= 1; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
        rows_upto = min(B, __builtin_amdgcn_readfirstlane(mx));
    }
    int row_lo = 0, row_hi = DIRTY ? (carried ? car_n : rows_upto) : m_rows_scan;
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
    // dirty scan: no version whose displacement is below wave_tau can matter to any point of this wave; when that
    // rules out every tile of the sub-range the wave only hands its seeds on and never loads its points
    // (per kind: a version competes in the list of its kind, against that list's threshold)
    // (per class of row, see CC_DSQ_STRIDE: 0 pcore, 1 outlier, 2 promoted since the snapshot)
    double wave_tau[3] = {-CC_INF, -CC_INF, -CC_INF};
    bool any_tile = true;
    auto tile_below = [&](int rt) -> bool {
        const unsigned long long* w = rows.tile_dsq + (size_t)(rt >> 4) * CC_DSQ_STRIDE;
        return cc_dsq_below(w[0], wave_tau[0]) && cc_dsq_below(w[1], wave_tau[1]) && cc_dsq_below(w[2], wave_tau[2]);
    };
    if (DIRTY) {
#pragma unroll
        for (int K = 0; K < 3; ++K) {
            double wt = CC_INF;
#pragma unroll
            for (int t = 0; t < PT; ++t) {
                const double tj = valid[t] ? rows.tau[(size_t)jj[t] * CC_TAU_STRIDE + K] : CC_INF;
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
            if (!tile_below(rt)) any_tile = true;
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
            if (tile_below(rt)) continue;  // nothing in this tile can matter
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
            const double rqs = (lane < tm) ? rows.dsq[rt + lane] : 0.0;  // (negated: class 2)
            const int rk = (lane < tm) ? rows.kind[rt + lane] : CC_KIND_DEAD;
            const int cls = cc_dsq_class(rk, rqs);
            const double rq = __builtin_fabs(rqs);
            const double wt = (cls == 0) ? wave_tau[0] : (cls == 1 ? wave_tau[1] : wave_tau[2]);
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
                                                                 int shard_world, const int* __restrict__ plist = nullptr)
{
    static_assert(DP % 2 == 0 && DP >= 4 && DP <= 64, "padded dimensionalities are even");
    // a row is read in NC chunks of whole pairs of dimensions (at most ten dimensions: 20 SGPRs), alternately into two
    // buffers; NC is even, so the first chunk of the next row follows the last one of a row in the other buffer
    constexpr int NP = DP / 2;
    constexpr int NC = 2 * ((NP + 9) / 10);
    constexpr int CH_MAX = 2 * ((NP + NC - 1) / NC);
    int B, m_rows_scan, q;
    long long cursor;
    if (mode == 1) {
        q = round & 1;
        B = ctl->la_b[q];
        m_rows_scan = ctl->la_rows[q];
        cursor = ctl->la_cursor[q];
    } else {
        B = ctl->win_b;
        m_rows_scan = ctl->m_rows;
        cursor = ctl->cursor;
        if (ctl->mode != 0) return;  // this window's snapshot scan ran ahead
        q = (int)(ctl->window_seq & 1ull);
    }
    part += (size_t)q * part_stride;
    if (B == 0) return;
    // plist (round 6): not the window's tiles but the points a guessed-threshold scan missed (k_missed's / k_missed_g's list of
    // the window's parity).  Such a point has no microcluster within the guess: a seeded threshold of its own is as wide as
    // its nearest microcluster is far, the pruned chain completes most rows for it - one by one, each a chain of dependent
    // round trips (187 us per launch for ten points of the C2 stream) -, and the plain scan is the faster kernel by ten.
    if (plist) plist += (size_t)q * CC_MISSED_CAP;
    const int n_here = plist ? ctl->n_missed[q] : B;
    const int j0 = (int)blockIdx.x * 64;
    if (j0 >= n_here) return;
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
    const bool valid = j0 + lane < n_here;
    const int jj = plist ? (valid ? plist[j0 + lane] : 0) : j0 + lane;

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
                                                   unsigned long long* __restrict__ cmax, const int* __restrict__ plist)
{
    static_assert(CC_PRE == 8 && CC_PRE <= DP, "prefix dimensions");
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int B = win.B;
    if (B == 0) return;
    const int j0 = (int)blockIdx.x * 128;
    // (plist: not the window's points but the ones a guessed threshold missed, k_missed's list of the window's parity)
    if (plist) plist += (size_t)win.q * CC_MISSED_CAP;
    const int n_pts_here = plist ? ctl->n_missed[win.q] : B;
    if (j0 >= n_pts_here) return;
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
        const int e = j0 + u * 64 + lane;
        valid[u] = e < n_pts_here;
        jj[u] = plist ? (valid[u] ? plist[e] : 0) : e;
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
// T32 of a threshold T for a point whose prefix coordinates are at most pm in magnitude, the scanned rows' at most cmx
// (derivation above k_seed_merge: exceeding T32 in single precision implies exceeding T exactly)
__device__ __forceinline__ float cc_thr32(double T, double pm, double cmx, double inv_k)
{
    if (!(T < CC_INF)) return __builtin_inff();
    const double e = 0x1p-21 * __builtin_fmax(pm, cmx);
    const double smax = inv_k > 1.0 ? inv_k : 1.0, smin = inv_k < 1.0 ? inv_k : 1.0;
    const double a = 2.0 * e * sqrt(8.0 * smax);
    const double u = 0.5 * (a + sqrt(a * a + 4.0 * T)) * (1.0 + 0x1p-40);
    // the kernel compares sum x^2 (without smin) with T32 = u^2 (1 + 2^-19) / smin
    const double t64 = u * u * (1.0 + 0x1p-19) / smin * (1.0 + 0x1p-40);
    float t32 = (float)t64;
    if ((double)t32 < t64) t32 = __uint_as_float(__float_as_uint(t32) + 1u);  // (t32 >= 0 and finite here: the next float up)
    return t32;
}

template <int DP>
__global__ __launch_bounds__(64) void k_seed_merge(const Ctl* __restrict__ ctl, const double* __restrict__ X,
                                                   const double* __restrict__ g_cen, const double* __restrict__ g_scl,
                                                   const SeedCand* __restrict__ spart, size_t spart_stride, int S,
                                                   double* __restrict__ thr, float* __restrict__ thr32, size_t thr_stride,
                                                   double F, int round, int mode, const unsigned long long* __restrict__ cmax,
                                                   unsigned long long* __restrict__ pstat, const int* __restrict__ plist)
{
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int K = t & 1;
    int j = t >> 1;
    if (plist) {
        if (win.B == 0 || j >= ctl->n_missed[win.q]) return;
        j = plist[(size_t)win.q * CC_MISSED_CAP + j];
    } else {
        if (j >= win.B) return;
        // the sample counters of this window's k_scan_p (split scans only: see there) start at zero
        if (t < 2) pstat[win.q * 2 + t] = 0ull;
    }
    constexpr int d = DP;  // (the pruned scan runs for d == DP only: every loop below unrolls, its loads go out together)
    const bool pow2 = ctl->pow2 != 0;
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
                    // (k not a power of two - round 6, k_scan_p3<GENERAL> -: the operand column holds the preference entries)
                    if (pow2) { x0 = x0 * s0[u]; x1 = x1 * s1[u]; x2 = x2 * s2[u]; }
                    else { x0 = x0 / s0[u]; x1 = x1 / s1[u]; x2 = x2 / s2[u]; }
                    a0 = a0 + x0; a1 = a1 + x1; a2 = a2 + x2;
                }
            }
        }
        double dmin = a0;
        dmin = a1 < dmin ? a1 : dmin;
        dmin = a2 < dmin ? a2 : dmin;
        out = F * dmin;
        if (F <= 0.0) {
            // TIGHT (round 6, behind k_seed16's seeds): the SECOND smallest of the exact distances - at least two rows of the kind
            // lie within it, so the scan that abandons what lies beyond returns the exact two best, which is all a plain scan
            // returns.  One seed only (one sub-range, or one row of the kind): its distance; the second place is then a bound.
            const double lo = a0 < a1 ? a0 : a1, hi = a0 < a1 ? a1 : a0;
            const double mid = a2 < hi ? a2 : hi;  // min(hi, a2)
            out = !h1 ? a0 : (!h2 ? hi : (lo < mid ? mid : lo));
            // (strictly above it: the bound the scan leaves for what it abandoned - T - must not tie with the exact second
            // candidate, a bound sorts first on a tie)
            out = out * (1.0 + 0x1p-20) + 0x1p-1000;
        }
    }
    thr[(size_t)j * 2 + K] = out;
    double pm = 0.0;
    for (int i = 0; i < CC_PRE; ++i) pm = __builtin_fmax(pm, __builtin_fabs(p[i]));
    thr32[(size_t)j * 2 + K] = cc_thr32(out, pm, __longlong_as_double((long long)cmax[win.q]), ctl->inv_k);
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

// ---- phase A as a kernel of its own (round 5): k_scan_a -> k_scan_p<.., MASKED = true> ----------------------------------
// k_scan_p's phase A is bound by the round trip of its wave-uniform LDS reads (two 16-byte broadcasts per row and wave of 64
// points; neither the VALU nor the LDS array is busy half the time, profiles/r05_tool_scanp_variants.txt).  k_scan_a runs the
// prefix test the way k_seed scores prefixes: TWO points per lane - a row's prefix is read from LDS once for 128 points -
// and the EXPANDED form, everything centred on the prefix o of table row 0:
//     |p' - c'|^2 = |p'|^2 - 2 (p'.c' - |c'|^2 / 2),   p' = p - o,  c' = c - o.
// With the lane's constant folded into the accumulator's start the test "the row's exact partial sum over the first eight
// dimensions exceeds the point's threshold" is
//     t' = p^.c^ - tau  <  h^        (4 packed multiply-adds, 1 add, 1 compare per row and point),
// p^ = fl32(p'), c^ = fl32(c'), h^ = |c^|^2 / 2 as the wave computes it.  It needs no coordinates beyond the prefix, so the
// 2 x d doubles of the points stay out of the registers; what it leaves per (tile of 128 points, row sub-range, tile of 16
// rows) is one word: the rows some lane of either 64-point half keeps (bits 0-15 / 16-31).  k_scan_p<MASKED> then runs
// phase B alone over those rows (1.3 % at C2, 0.13 % at the C5 shape) - same thresholds, same exact abandon test, same
// candidates and bounds: a row phase A keeps is evaluated exactly, so only "abandoned implies beyond T" matters.
//
// cc_tau32: tau for a point.  T = its exact threshold of the kind, s = |p^|^2 (double), pm = max |p^_i|, cm >= max |c^_i|.
// Reals, u = 2^-24:  U = sum_{i<8} (p_i - c_i)^2; phase B's partial sum P >= smin U (smin = min(1, 1/k)).  The conversions
// perturb every difference by at most u' (|p^_i| + |c^_i|), so sqrt(U) >= sqrt(U^) - A, U^ = |p^ - c^|^2, A = sqrt(8) u' (pm + cm).
// U^ = s - 2 p^.c^ + 2 H, H = |c^|^2 / 2; the wave's h^ is within 2^-21 H <= 2^-19 cm^2 of H (eh); the five roundings on any
// path of the multiply-adds and the final add give |t' - (p^.c^ - tau)| <= g (|tau| + 8 pm cm), g = 6 u.  So t' < h^ implies
// U^ > s - 2 tau - 2 g (|tau| + 8 pm cm) - 2 eh, and with
//     tau <= (s - L) / 2 - g (|tau0| + 8 pm cm) - eh,     L = (sqrt(T (1 + 2^-40) / smin) + A)^2
// U^ > L, sqrt(U) > sqrt(T (1 + 2^-40) / smin), P >= smin U > T (the 2^-40 covers the roundings of the double-precision
// partial sum).  Slack: s's own rounding (2^-48 s), flushed subnormals (2^-100), tau rounded DOWN to single precision.
// Refused (-inf: never abandon) unless pm, cm < 2^60 and |tau| < 2^120: no overflow in the products or the sum.
__device__ __forceinline__ float cc_tau32(double T, double s, double pm, double cm, double inv_k)
{
    const float never = -__builtin_inff();
    if (!(T < CC_INF)) return never;
    const double smin = inv_k < 1.0 ? inv_k : 1.0;
    const double A = 2.8284271247461903 * 0x1p-24 * (1.0 + 0x1p-20) * (pm + cm) + 0x1p-120;
    const double r = sqrt(T * (1.0 + 0x1p-40) / smin) * (1.0 + 0x1p-50) + A;
    const double L = r * r * (1.0 + 0x1p-20);
    const double tau0 = 0.5 * (s - L) - 0x1p-48 * s;
    const double tau = tau0 - 6.0 * 0x1p-24 * (__builtin_fabs(tau0) + 8.0 * pm * cm) - 0x1p-19 * cm * cm - 0x1p-100;
    if (!(pm < 0x1p60 && cm < 0x1p60 && __builtin_fabs(tau) < 0x1p120)) return never;
    float t = (float)tau;
    if ((double)t > tau) t = (t > 0.0f) ? __uint_as_float(__float_as_uint(t) - 1u)
                                      : (t < 0.0f ? __uint_as_float(__float_as_uint(t) + 1u) : -0x1p-149f);
    return t;
}

// words per (tile of 128 points, row sub-range): one word of "abandoned a row of the kind" flags, then one per 16-row tile
__host__ __device__ inline int cc_mask_tiles_per_sub(int rows, int nsub) { return (((rows + nsub - 1) / nsub) + CC_SCAN_TM - 1) / CC_SCAN_TM + 1; }

template <int DP, int NW>
__global__ __launch_bounds__(64 * NW, 4) void k_scan_a(const Ctl* __restrict__ ctl, const double* __restrict__ Xt,
                                                     const double* __restrict__ g_cen, const int* __restrict__ g_kind,
                                                     const double* __restrict__ thr, size_t thr_stride,
                                                     unsigned* __restrict__ masks, size_t mask_stride, int tps, int round,
                                                     int mode, int shard_rank, int shard_world, double guess_F)
{
    static_assert(CC_PRE == 8 && CC_PRE <= DP, "prefix dimensions");
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int B = win.B;
    if (B == 0) return;
    const int j0 = (int)blockIdx.x * 128;
    if (j0 >= B) return;
    thr += (size_t)win.q * thr_stride;
    masks += (size_t)win.q * mask_stride;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = gridDim.y;
    const int nsub = S * NW;
    const int sub = blockIdx.y * NW + wv;
    // (the same split of the rows as k_scan_p: the masks are addressed by sub-range and tile)
    int row_lo = 0, row_hi = win.rows;
    if (shard_world > 1) cc_shard_range(win.rows, shard_world, shard_rank, 1, &row_lo, &row_hi);
    const int per = (row_hi - row_lo + nsub - 1) / nsub;
    const int r0 = row_lo + sub * per;
    const int r1 = min(row_hi, r0 + per);
    const size_t n_pts = (size_t)ctl->xt_stride;
    double org[CC_PRE];
#pragma unroll
    for (int i = 0; i < CC_PRE; ++i) org[i] = g_cen[i];  // (wave-uniform: scalar loads)
    // |c - o| <= |c| + |o| <= 2 x the largest |coordinate| of the table's centroids (centroids are means of points)
    const double cmd = 2.0 * __builtin_fmax(ctl->x_absmax, __longlong_as_double((long long)ctl->cen_absmax));
    const double inv_k = ctl->inv_k;
    const bool guessed = guess_F > 0.0;
    cc_f2 p2[2][CC_PRE / 2];
    cc_f2 ninit[2][2];  // [point of the lane][kind]: the accumulator's start {-tau, 0}; no point: -inf (every row "exceeds")
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int jj = j0 + u * 64 + lane;
        const bool valid = jj < B;
        const double* xp = Xt + win.cursor + (valid ? jj : 0);
        double s2 = 0.0;
        float pmf = 0.0f;
        float ph[CC_PRE];
#pragma unroll
        for (int i = 0; i < CC_PRE; ++i) {
            ph[i] = valid ? (float)(xp[(size_t)i * n_pts] - org[i]) : 0.0f;
            s2 += (double)ph[i] * (double)ph[i];
            const float a = __builtin_fabsf(ph[i]);
            pmf = a > pmf ? a : pmf;
        }
#pragma unroll
        for (int i = 0; i < CC_PRE / 2; ++i) p2[u][i] = cc_f2{ph[2 * i], ph[2 * i + 1]};
#pragma unroll
        for (int K = 0; K < 2; ++K) {
            double T;
            if (guessed) T = (ctl->tg_ok[win.q][K] != 0) ? guess_F * ctl->tg[win.q][K] : CC_INF;
            else T = valid ? thr[(size_t)jj * 2 + K] : CC_INF;
            const float tau = valid ? cc_tau32(T, s2, (double)pmf, cmd, inv_k) : __builtin_inff();
            ninit[u][K] = cc_f2{-tau, 0.0f};
        }
    }
    __shared__ __attribute__((aligned(16))) float s_pre[NW * CC_SCAN_TM * CC_PRE];
    __shared__ float s_h[NW * CC_SCAN_TM];
    float* const tile = s_pre + (size_t)wv * CC_SCAN_TM * CC_PRE;
    float* const th = s_h + (size_t)wv * CC_SCAN_TM;
    unsigned* const mrow = masks + ((size_t)blockIdx.x * nsub + sub) * (size_t)tps;
    unsigned dflag = 0u;  // bit 2 u + K: a row of kind K was abandoned for the points of half u
    double tc[2];
    int kdl = CC_KIND_DEAD;
    if (r0 < r1) cc_load_prefix<DP>(g_cen, g_kind, r0, min(CC_SCAN_TM, r1 - r0), lane, tc, kdl);
    int tt = 0;
    for (int rt = r0; rt < r1; rt += CC_SCAN_TM, ++tt) {
        const int tm = __builtin_amdgcn_readfirstlane(min(CC_SCAN_TM, r1 - rt));
        CC_WAVE_SYNC();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float v = (float)(tc[q] - org[lane & 7]);
            tile[lane + q * 64] = v;
            // h of the row: the eight lanes that hold it add their squares (every lane ends up with the sum)
            float hsum = v * v;
            hsum += __shfl_xor(hsum, 1);
            hsum += __shfl_xor(hsum, 2);
            hsum += __shfl_xor(hsum, 4);
            // a prefix beyond single precision's range: never abandoned (t' < -inf is false)
            if ((lane & 7) == 0) th[(lane >> 3) + q * 8] = (hsum < 0x1p120f) ? 0.5f * hsum : -__builtin_inff();
        }
        const unsigned pmask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_PCORE);
        const unsigned omask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_OUTLIER);
        CC_WAVE_SYNC();
        if (rt + CC_SCAN_TM < r1) cc_load_prefix<DP>(g_cen, g_kind, rt + CC_SCAN_TM, min(CC_SCAN_TM, r1 - rt - CC_SCAN_TM), lane, tc, kdl);
        const cc_f4* t4 = reinterpret_cast<const cc_f4*>(__builtin_assume_aligned(tile, 16));
        const unsigned full = (1u << tm) - 1u;
        unsigned surv[2] = {0u, 0u};
        auto rows_of = [&](auto KSELC) {
            constexpr int KSEL = decltype(KSELC)::value;
            auto two = [&](int m0, int n) {
                cc_f4 c01[2], c23[2];
                float h[2];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int m = (v < n) ? m0 + v : m0;
                    c01[v] = t4[m * 2]; c23[v] = t4[m * 2 + 1]; h[v] = th[m];
                }
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    if (v >= n) break;
                    const int m = m0 + v;
                    int K;
                    if constexpr (KSEL == 0) K = 0;
                    else if constexpr (KSEL == 1) K = 1;
                    else K = ((pmask >> m) & 1u) ? 0 : 1;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        cc_f2 acc = __builtin_elementwise_fma(p2[u][0], cc_f2{c01[v].x, c01[v].y}, K == 0 ? ninit[u][0] : ninit[u][1]);
                        acc = __builtin_elementwise_fma(p2[u][1], cc_f2{c01[v].z, c01[v].w}, acc);
                        acc = __builtin_elementwise_fma(p2[u][2], cc_f2{c23[v].x, c23[v].y}, acc);
                        acc = __builtin_elementwise_fma(p2[u][3], cc_f2{c23[v].z, c23[v].w}, acc);
                        float t;  // (written out: the vectoriser pairs the adds of two rows into one packed add behind three moves)
                        asm("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(acc.x), "v"(acc.y));
                        // kept unless t' < h^ (a NaN - an overflow the guards did not see - keeps the row)
                        surv[u] |= (__builtin_amdgcn_ballot_w64(!(t < h[v])) != 0ull) ? (1u << m) : 0u;
                    }
                }
            };
            int m = 0;
            for (; m + 2 <= tm; m += 2) two(m, 2);
            if (m < tm) two(m, 1);
        };
        if (pmask == full) rows_of(std::integral_constant<int, 0>{});
        else if (omask == full) rows_of(std::integral_constant<int, 1>{});
        else rows_of(std::integral_constant<int, -1>{});
        // rows of neither list (dead) are never completed by phase B: dropped from the mask, not counted as abandoned
        const unsigned listed = (pmask | omask) & full;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            surv[u] &= listed;
            if ((~surv[u] & pmask & full) != 0u) dflag |= 1u << (2 * u);
            if ((~surv[u] & omask & full) != 0u) dflag |= 2u << (2 * u);
        }
        if (lane == 0) mrow[1 + tt] = surv[0] | (surv[1] << 16);
    }
    if (lane == 0) mrow[0] = dflag;  // (the sub-range's first word: its flags)
}

template <int DP, int NW, bool MASKED>
__global__ __launch_bounds__(64 * NW, (DP <= 20 ? CC_SCANP_WGS20 : (DP <= 40 ? 3 : 2))) void k_scan_p(
    Ctl* __restrict__ ctl, const double* __restrict__ Xt, const double* __restrict__ g_cen, const double* __restrict__ g_scl,
    const int* __restrict__ g_kind, const int* __restrict__ g_key, const double* __restrict__ thr,
    const float* __restrict__ thr32, size_t thr_stride, Cand* __restrict__ part, int round, int mode, size_t part_stride,
    int shard_rank, int shard_world, unsigned long long* __restrict__ pstat, const int* __restrict__ plist, double guess_F,
    unsigned long long* __restrict__ found, const unsigned* __restrict__ masks, size_t mask_stride, int tps, int q_sub)
{
    static_assert(DP % 2 == 0 && DP > CC_PRE && DP <= 64, "k_scan_p shapes");
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int B = win.B;
    if (B == 0) return;
    const int j0 = (int)blockIdx.x * 64;
    // plist: the points a guessed threshold missed (k_missed's list of the window's parity) instead of the window's tiles
    if (plist) plist += (size_t)win.q * CC_MISSED_CAP;
    const int n_pts_here = plist ? ctl->n_missed[win.q] : B;
    if (j0 >= n_pts_here) return;
    part += (size_t)win.q * part_stride;
    thr += (size_t)win.q * thr_stride;
    thr32 += (size_t)win.q * thr_stride;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = gridDim.y;
    const int nsub = S * NW;
    const int sub = blockIdx.y * NW + wv;
    // exact multi-GPU path: this rank's block of the table rows (the thresholds come from seeds over ALL rows, computed
    // by every rank alike, so every rank abandons against the same T)
    int row_lo = 0, row_hi = win.rows;
    if (shard_world > 1) cc_shard_range(win.rows, shard_world, shard_rank, 1, &row_lo, &row_hi);
    const int per = (row_hi - row_lo + nsub - 1) / nsub;
    const int r0 = row_lo + sub * per;
    const int r1 = min(row_hi, r0 + per);
    const size_t n_pts = (size_t)ctl->xt_stride;
    const bool valid = j0 + lane < n_pts_here;
    const int jj = plist ? (valid ? plist[j0 + lane] : 0) : j0 + lane;

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
    // found != nullptr: GUESSED thresholds - not F x the point's own seed distance (k_seed, k_seed_merge: as much work
    // again as this kernel) but F x the mean distance at which the points of an earlier window joined a MC of the
    // kind (Ctl::tg).  Any threshold is a valid one: what falls under it is evaluated exactly, what is abandoned is
    // bounded by it.  A point whose own MC lies beyond the guess ends with a bound in first place; the wave that does
    // find a pcore MC within the threshold marks its points in `found`, k_missed lists the unmarked ones and the seeded
    // chain runs for them alone (plist).  No mean for the kind yet: +inf, every row of the kind is evaluated.
    const bool guessed = guess_F > 0.0;  // (found may be null: a lean guessed scan - nobody lists the missed points)
    if (guessed) {
        double pm = 0.0;
#pragma unroll
        for (int i = 0; i < CC_PRE; ++i) pm = __builtin_fmax(pm, __builtin_fabs(p[i]));
        const double cmx = __builtin_fmax(ctl->x_absmax, __longlong_as_double((long long)ctl->cen_absmax));
        const double inv_k = ctl->inv_k;
#pragma unroll
        for (int K = 0; K < 2; ++K) {
            const double T = (ctl->tg_ok[win.q][K] != 0) ? guess_F * ctl->tg[win.q][K] : CC_INF;
            th[K] = valid ? T : -CC_INF;
            th32[K] = valid ? cc_thr32(T, pm, cmx, inv_k) : -__builtin_inff();
        }
    } else {
#pragma unroll
        for (int K = 0; K < 2; ++K) {
            th[K] = valid ? thr[(size_t)jj * 2 + K] : -CC_INF;
            th32[K] = valid ? thr32[(size_t)jj * 2 + K] : -__builtin_inff();
        }
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
    // ---- phase B: the rows of a tile that stayed (bit mask `surv`) ----
    // ---- phase B: one surviving row, from its first dimension, with the exact abandon test every eight dimensions ----
    // `cs(i)` / `ss(i)`: the row's centroid coordinate / distance operand of dimension i (wave-uniform).
    auto complete_row = [&](int rowg, bool is_p, auto&& cs, auto&& ss) {
        double acc = 0.0;
        bool gone = false;
        cc_static_for<(DP + 7) / 8>([&](auto CC) {
            constexpr int lo = 8 * decltype(CC)::value, hi = (lo + 8) < DP ? (lo + 8) : DP;
            if (gone) return;
            double c[hi - lo], sc[hi - lo];
#pragma unroll
            for (int i = 0; i < hi - lo; ++i) {
                c[i] = cs(lo + i);
                sc[i] = ss(lo + i);
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
        if (gone) return;
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
    };
    // the one-kernel form: the rows of a tile that stayed (bit mask), centroid and operands as scalar loads
    auto phase_b = [&](int rt, unsigned surv, unsigned pmask, unsigned omask) {
        while (surv != 0u) {
            const int m = __builtin_ctz(surv);
            surv &= surv - 1u;
            const int rowg = rt + m;
            const bool is_p = ((pmask >> m) & 1u) != 0u;
            if (!is_p && ((omask >> m) & 1u) == 0u) continue;  // (neither list)
            const double* __restrict__ rc = g_cen + (size_t)rowg * DP;  // wave-uniform addresses: scalar loads
            const double* __restrict__ rs = g_scl + (size_t)rowg * DP;
            complete_row(rowg, is_p, [&](int i) { return rc[i]; }, [&](int i) { return rs[i]; });
        }
    };
    if constexpr (MASKED) {
        // Phase A has run as a kernel of its own (k_scan_a): per (tile of 128 points, sub-range) one word of "abandoned a row
        // of the kind" flags, then one word per 16-row tile with the rows some lane of either 64-point half keeps.  Most
        // words are zero, but the rows that stay are several per wave (0.6 % of the rows pass an eight-dimension prefix
        // test for SOME point of 64; a fifth of those are completed), each a chain of dependent memory round trips when
        // taken one by one.  So: (1) the wave lists its rows - a vector load of up to 64 mask words, a prefix sum over the
        // lanes' bit counts, every lane writes the rows of its word into the wave's LDS list; (2) it walks the list with
        // the NEXT row's 2 d doubles (and kind) already requested - one pass of coalesced vector loads into the wave's LDS
        // row, read back as broadcasts - while the current one is evaluated.
        constexpr int LCAP = 64 * CC_SCAN_TM;  // one batch of 64 mask words can list this many rows
        __shared__ int s_list[NW * LCAP];
        __shared__ __attribute__((aligned(16))) double s_rowbuf[NW * 2 * DP];
        int* const lst = s_list + (size_t)wv * LCAP;
        double* const srow = s_rowbuf + (size_t)wv * 2 * DP;
        constexpr int RL = (2 * DP + 63) / 64;  // loads per lane for one row
        auto fetch_row = [&](int rowg, double (&rv)[RL], int& rk) {
#pragma unroll
            for (int q = 0; q < RL; ++q) {
                const int e = lane + q * 64;
                rv[q] = (e < DP) ? g_cen[(size_t)rowg * DP + e] : ((e < 2 * DP) ? g_scl[(size_t)rowg * DP + (e - DP)] : 0.0);
            }
            rk = g_kind[rowg];
        };
        // (up to four rows in flight: at 50 000 rows x 40 dims the table is eight times an XCD's L2, a row comes from the memory
        // side in 1-2 us, and a wave has 25 of them to walk)
        constexpr int PF = DP <= 20 ? 4 : 1;  // (d = 40: the points alone are 80 registers - deeper prefetch spills)
        auto walk = [&](int n) {
            if (n <= 0) return;
            double rv[PF][RL];
            int rk[PF];
            CC_WAVE_SYNC();  // (the list is written)
#pragma unroll
            for (int k = 0; k < PF; ++k)
                if (k < n) fetch_row(__builtin_amdgcn_readfirstlane(lst[k]), rv[k], rk[k]);
            for (int base = 0; base < n; base += PF) {
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    const int e = base + k;
                    if (e >= n) break;
                    const int rowg = __builtin_amdgcn_readfirstlane(lst[e]);
                    const int kind = __builtin_amdgcn_readfirstlane(rk[k]);
                    CC_WAVE_SYNC();  // (the previous row's reads are done)
#pragma unroll
                    for (int q = 0; q < RL; ++q) {
                        const int x = lane + q * 64;
                        if (x < 2 * DP) srow[x] = rv[k][q];
                    }
                    CC_WAVE_SYNC();
                    if (e + PF < n) fetch_row(__builtin_amdgcn_readfirstlane(lst[e + PF]), rv[k], rk[k]);
                    if (kind != CC_KIND_PCORE && kind != CC_KIND_OUTLIER) continue;
                    complete_row(rowg, kind == CC_KIND_PCORE, [&](int i) { return srow[i]; }, [&](int i) { return srow[DP + i]; });
                }
            }
            CC_WAVE_SYNC();  // (the list may be rewritten)
        };
        const int half = (int)(blockIdx.x & 1u);
        const int nsub_a = nsub * q_sub;
        const int per_a = (row_hi - row_lo + nsub_a - 1) / nsub_a;
        for (int k = 0; k < q_sub; ++k) {
            const int sa = sub * q_sub + k;
            const int ra0 = row_lo + sa * per_a;
            const int ra1 = min(row_hi, ra0 + per_a);
            if (ra0 >= ra1) break;
            const unsigned* const mrow = masks + (size_t)win.q * mask_stride + ((size_t)(blockIdx.x >> 1) * nsub_a + sa) * (size_t)tps;
            {
                const unsigned fl = __builtin_amdgcn_readfirstlane(mrow[0]) >> (2 * half);
                if ((fl & 1u) != 0u) dropped[0] = true;
                if ((fl & 2u) != 0u) dropped[1] = true;
                n_rows += ra1 - ra0;
            }
            const int ntiles = (ra1 - ra0 + CC_SCAN_TM - 1) / CC_SCAN_TM;
            for (int tb = 0; tb < ntiles; tb += 64) {
                unsigned mw = (tb + lane < ntiles) ? mrow[1 + tb + lane] : 0u;
                const int rt = ra0 + (tb + lane) * CC_SCAN_TM;
                const int tm = min(CC_SCAN_TM, ra1 - rt);  // (<= 0 beyond the sub-range: mw is 0 there)
                mw = (mw >> (16 * half)) & 0xFFFFu & (tm >= CC_SCAN_TM ? 0xFFFFu : ((1u << (tm > 0 ? tm : 0)) - 1u));
                if (__builtin_amdgcn_ballot_w64(mw != 0u) == 0ull) continue;
                // exclusive prefix sum of the lanes' bit counts: where each lane's rows go in the list
                const int cnt = __builtin_popcount(mw);
                int incl = cnt;
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(incl, off);
                    if (lane >= off) incl += o;
                }
                const int total = __builtin_amdgcn_readlane(incl, 63);
                int pos = incl - cnt;
                unsigned m2 = mw;
                while (m2 != 0u) {
                    const int bbit = __builtin_ctz(m2);
                    m2 &= m2 - 1u;
                    lst[pos++] = rt + bbit;
                }
                walk(total);
            }
        }
    } else {
    if (r0 < r1) cc_load_prefix<DP>(g_cen, g_kind, r0, min(CC_SCAN_TM, r1 - r0), lane, tc, kdl);
    for (int rt = r0; rt < r1; rt += CC_SCAN_TM) {
        const int tm = __builtin_amdgcn_readfirstlane(min(CC_SCAN_TM, r1 - rt));
        unsigned surv = 0u;
        unsigned pmask, omask;
        const unsigned full = (1u << tm) - 1u;
        CC_WAVE_SYNC();
#pragma unroll
        for (int q = 0; q < 2; ++q) tile[lane + q * 64] = (float)tc[q];
        pmask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_PCORE);
        omask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_OUTLIER);
        CC_WAVE_SYNC();
        if (rt + CC_SCAN_TM < r1) cc_load_prefix<DP>(g_cen, g_kind, rt + CC_SCAN_TM, min(CC_SCAN_TM, r1 - rt - CC_SCAN_TM), lane, tc, kdl);
        n_rows += tm;
        const cc_f4* t4 = reinterpret_cast<const cc_f4*>(__builtin_assume_aligned(tile, 16));

        // ---- phase A ----
        // straight-line: one compare per row, the wave's verdict as one bit (which kinds lost rows follows from the
        // bits at the end of the tile)
        auto verdict = [&](auto KSELC, int m, float q) {
            constexpr int KSEL = decltype(KSELC)::value;
            float t;
            if constexpr (KSEL == 0) t = th32[0];
            else if constexpr (KSEL == 1) t = th32[1];
            else t = ((pmask >> m) & 1u) ? th32[0] : th32[1];
            // `!(q > t)`, not `q <= t`: coordinates beyond single precision's range (|x| >= 2^128) give inf - inf = NaN in
            // the prefix, and a row whose prefix sum is not a number has to be KEPT (phase B evaluates it exactly); an
            // overflow to +inf is a sum that really exceeds every finite T32 (tests/test_pruned_scan.py, huge coordinates)
            // (lanes without a point hold T32 = -inf and a finite or infinite q, never NaN: their own p is 0)
            surv |= (__builtin_amdgcn_ballot_w64(!(q > t)) != 0ull) ? (1u << m) : 0u;
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
        phase_b(rt, surv, pmask, omask);
    }
    }
    // rows abandoned in phase A: their exact partial sums exceed every lane's T (k_seed_merge), which is all that is
    // recorded of them
    if (dropped[0]) lb[0] = cc_vmin(lb[0], th[0]);
    if (dropped[1]) lb[1] = cc_vmin(lb[1], th[1]);
    if (guessed && found != nullptr) {
        // the points for which this wave evaluated a pcore MC within the guessed threshold: their pcore list's best is exact
        const unsigned long long fm = __builtin_amdgcn_ballot_w64(valid && bs[0][0] >= 0 && bd[0][0] <= th[0]);
        if (lane == 0 && fm != 0ull) atomicOr(found + (size_t)win.q * (CC_MAX_WINDOW / 64) + blockIdx.x, fm);
    }
    // statistics for the host's policy: a sample - the waves of the window's first point tile (atomics of every wave on
    // one address serialise: 30 000 of them cost more than the scan)
    // A split scan's sample describes this rank's rows only, and the host policy that reads the counters has to decide
    // alike on every rank: the sample then goes to the window's own pair of counters, travels with the rank's candidate
    // records (k_merge_partials) and k_decide adds up what all ranks sent - the same sum everywhere.
    // (The chain over a list of missed points adds to the sample too: those points are the ones far from every
    // microcluster, their waves complete most rows, and a stream in which they are many - a table that is still filling -
    // is one on which the pruned chain does not pay.  Leaving them out was tried: the C5-shaped stream, which is all
    // start-up, lost a quarter of its rate.  Split over ranks the list's scan runs behind the gather that carried the
    // sample, so it stays out there.)
    if (lane == 0 && blockIdx.x == 0 && n_rows > 0 && !(plist != nullptr && shard_world > 1)) {
        atomicAdd(pstat + win.q * 2, (unsigned long long)n_rows);
        atomicAdd(pstat + win.q * 2 + 1, (unsigned long long)n_full);
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
    if constexpr (MASKED) {
        // (k_decide merges S x q_sub partials per point: this workgroup's go to the first of its slots, the rest stay empty)
        Cand* o = part + ((size_t)jj * S * q_sub + (size_t)blockIdx.y * q_sub) * 4;
        o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
        const Cand none = Cand{CC_INF, CC_IDX_INF, -1};
        for (int k = 1; k < q_sub; ++k) {
            o[4 * k] = none; o[4 * k + 1] = none; o[4 * k + 2] = none; o[4 * k + 3] = none;
        }
    } else {
        Cand* o = part + ((size_t)jj * S + blockIdx.y) * 4;
        o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
    }
}

// ---------------------------------------------------------------------------------
// k_scan_p2 (round 6): the pruned scan of a window in ONE kernel with phase A the way k_scan_a runs it - TWO points per lane, the
// expanded form, the threshold folded into the accumulator's start (cc_tau32) - and phase B from the same residency.
// k_scan_p's phase A waits on its wave-uniform LDS reads (two 16-byte broadcasts per row and 64 points: VALU 56 % busy, 41 % of the
// waves' time parked on s_waitcnt, profiles/r05_pmc_valu_d20.json); splitting phase A off (k_scan_a) halves the LDS bytes per
// (point, row) and the instructions, but at 5 000 rows the second launch, its prologue and the masks' round trip through L2 cost what
// that saves.  Here a workgroup covers 128 window points: its waves test every row of their sub-range for both 64-point halves
// (7 VALU instructions per (64 points, row) where k_scan_p spends 15) and complete the rows that stay - 1.3 % at C2 - at once,
// half by half.  The points' coordinates are NOT held in registers (2 x d doubles per lane would halve the occupancy): the
// workgroup stages them once in LDS (128 x d doubles, 20 KB at d = 20: all its waves scan the same points) and phase B reads
// them from there; centroid and operands of a surviving row are scalar loads as in k_scan_p.  Same thresholds, same exact
// abandon test every eight dimensions, same candidates and bounds as k_scan_p: a row phase A keeps is evaluated exactly, so only
// "abandoned implies beyond T" matters, and that is cc_tau32's statement (tests/test_pruned_scan.py, the forced-pruning fuzz
// and the full-size plain-scan comparisons run it; CHRONOCLUST_HIP_SCANP2=0 goes back to k_scan_p).
// ---------------------------------------------------------------------------------
#ifndef CC_SCANP2_WGS
#define CC_SCANP2_WGS 4  // workgroups per CU the kernel is compiled for
#endif
template <int DP, int NW>
__global__ __launch_bounds__(64 * NW, CC_SCANP2_WGS) void k_scan_p2(
    Ctl* __restrict__ ctl, const double* __restrict__ Xt, const double* __restrict__ g_cen, const double* __restrict__ g_scl,
    const int* __restrict__ g_kind, const int* __restrict__ g_key, const double* __restrict__ thr, size_t thr_stride,
    Cand* __restrict__ part, int round, int mode, size_t part_stride, int shard_rank, int shard_world,
    unsigned long long* __restrict__ pstat, double guess_F, unsigned long long* __restrict__ found)
{
    static_assert(DP % 2 == 0 && DP > CC_PRE && DP <= 64 && CC_PRE == 8, "k_scan_p2 shapes");
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int B = win.B;
    if (B == 0) return;
    const int j0 = (int)blockIdx.x * 128;
    if (j0 >= B) return;
    part += (size_t)win.q * part_stride;
    thr += (size_t)win.q * thr_stride;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = gridDim.y;
    const int nsub = S * NW;
    const int sub = blockIdx.y * NW + wv;
    int row_lo = 0, row_hi = win.rows;
    if (shard_world > 1) cc_shard_range(win.rows, shard_world, shard_rank, 1, &row_lo, &row_hi);
    const int per = (row_hi - row_lo + nsub - 1) / nsub;
    const int r0 = row_lo + sub * per;
    const int r1 = min(row_hi, r0 + per);
    const size_t n_pts = (size_t)ctl->xt_stride;

    // LDS: the workgroup's 128 points (dimension-major, [i][128]); per wave a tile of 16 centred prefixes + their h; at the end the
    // same bytes carry the candidate exchange of the waves
    constexpr int PTS_DOUBLES = DP * 128;
    constexpr int MERGE_BYTES = (NW > 1 ? NW - 1 : 1) * 8 * 64 * (int)sizeof(Cand);
    constexpr int PTS_BYTES = PTS_DOUBLES * 8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[PTS_BYTES > MERGE_BYTES ? PTS_BYTES : MERGE_BYTES];
    __shared__ __attribute__((aligned(16))) float s_pre[NW * CC_SCAN_TM * CC_PRE];
    __shared__ float s_h[NW * CC_SCAN_TM];
    double* const s_pts = reinterpret_cast<double*>(smem);
    float* const tile = s_pre + (size_t)wv * CC_SCAN_TM * CC_PRE;
    float* const thh = s_h + (size_t)wv * CC_SCAN_TM;
    for (int e = (int)threadIdx.x; e < PTS_DOUBLES; e += 64 * NW) {
        const int i = e >> 7, x = e & 127;
        s_pts[e] = (j0 + x < B) ? Xt[win.cursor + j0 + x + (size_t)i * n_pts] : 0.0;
    }
    // the exact thresholds of the 128 points, per kind (the same for every wave of the workgroup: in LDS, not in eight
    // registers per lane - phase B and the epilogue read them); points beyond the window: -inf, they keep no row alive
    const bool guessed = guess_F > 0.0;
    __shared__ double s_T[2 * 128];
    for (int e = (int)threadIdx.x; e < 2 * 128; e += 64 * NW) {
        const int K = e >> 7, x = e & 127;
        double T;
        if (guessed) T = (ctl->tg_ok[win.q][K] != 0) ? guess_F * ctl->tg[win.q][K] : CC_INF;
        else T = (j0 + x < B) ? thr[(size_t)(j0 + x) * 2 + K] : CC_INF;
        s_T[e] = (j0 + x < B) ? T : -CC_INF;
    }
    double org[CC_PRE];
#pragma unroll
    for (int i = 0; i < CC_PRE; ++i) org[i] = g_cen[i];  // (wave-uniform: scalar loads)
    const double cmd = 2.0 * __builtin_fmax(ctl->x_absmax, __longlong_as_double((long long)ctl->cen_absmax));
    const double inv_k = ctl->inv_k;
    __syncthreads();  // the points and thresholds are staged

    bool valid[2];
    cc_f2 p2[2][CC_PRE / 2];
    cc_f2 ninit[2][2];   // [half][kind]: the accumulator's start {-tau, 0}
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        valid[u] = j0 + u * 64 + lane < B;
        double s2 = 0.0;
        float pmf = 0.0f;
        float ph[CC_PRE];
#pragma unroll
        for (int i = 0; i < CC_PRE; ++i) {
            ph[i] = valid[u] ? (float)(s_pts[i * 128 + u * 64 + lane] - org[i]) : 0.0f;
            s2 += (double)ph[i] * (double)ph[i];
            const float a = __builtin_fabsf(ph[i]);
            pmf = a > pmf ? a : pmf;
        }
#pragma unroll
        for (int i = 0; i < CC_PRE / 2; ++i) p2[u][i] = cc_f2{ph[2 * i], ph[2 * i + 1]};
#pragma unroll
        for (int K = 0; K < 2; ++K) {
            const float tau = valid[u] ? cc_tau32(s_T[K * 128 + u * 64 + lane], s2, (double)pmf, cmd, inv_k) : __builtin_inff();
            ninit[u][K] = cc_f2{-tau, 0.0f};
        }
    }
    auto thx = [&](int u, int K) -> double { return s_T[K * 128 + u * 64 + lane]; };
    double lb[2][2], bd[2][2][2];
    int bs[2][2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int K = 0; K < 2; ++K) {
            lb[u][K] = CC_INF;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                bd[u][K][r] = valid[u] ? CC_INF : -CC_INF;
                bs[u][K][r] = -1;
            }
        }
    unsigned dflag = 0u;  // bit 2 u + K (wave-uniform): phase A abandoned a row of kind K for half u
    int n_rows = 0, n_full = 0;

    // phase B for one row and one half: the whole distance from its first dimension with the reference's four operations per
    // term, the exact abandon test every eight dimensions, then k_scan_u's best-two update
    auto complete_row = [&](auto UC, int rowg, bool is_p) {
        constexpr int u = decltype(UC)::value;
        const double* __restrict__ rc = g_cen + (size_t)rowg * DP;  // wave-uniform addresses: scalar loads
        const double* __restrict__ rs = g_scl + (size_t)rowg * DP;
        const double* __restrict__ px = s_pts + u * 64 + lane;
        double acc = 0.0;
        bool gone = false;
        cc_static_for<(DP + 7) / 8>([&](auto CC) {
            constexpr int lo = 8 * decltype(CC)::value, hi = (lo + 8) < DP ? (lo + 8) : DP;
            if (gone) return;
            double c[hi - lo], sc[hi - lo], pv[hi - lo];
#pragma unroll
            for (int i = 0; i < hi - lo; ++i) {
                c[i] = rc[lo + i];
                sc[i] = rs[lo + i];
                pv[i] = px[(lo + i) * 128];
            }
#pragma unroll
            for (int i = 0; i < hi - lo; ++i) {
                double x = pv[i] - c[i];               // mc_functions.py:37
                x = x * x;                             // :38
                x = x * sc[i];                         // :39 (the divisor is a power of two)
                acc = (lo + i == 0) ? x : acc + x;     // :41
            }
            if constexpr (hi < DP) {
                const double t = thx(u, is_p ? 0 : 1);
                if (__builtin_amdgcn_ballot_w64(acc <= t) == 0ull) {
                    if (is_p) lb[u][0] = cc_vmin(lb[u][0], acc);
                    else lb[u][1] = cc_vmin(lb[u][1], acc);
                    gone = true;
                }
            }
        });
        if (gone) return;
        ++n_full;
        auto update = [&](auto KC) {
            constexpr int K = decltype(KC)::value;
            const double a = acc;
            double& d0 = bd[u][K][0];
            double& d1 = bd[u][K][1];
            int& s0 = bs[u][K][0];
            int& s1 = bs[u][K][1];
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
    };

    double tc[2];
    int kdl = CC_KIND_DEAD;
    if (r0 < r1) cc_load_prefix<DP>(g_cen, g_kind, r0, min(CC_SCAN_TM, r1 - r0), lane, tc, kdl);
    for (int rt = r0; rt < r1; rt += CC_SCAN_TM) {
        const int tm = __builtin_amdgcn_readfirstlane(min(CC_SCAN_TM, r1 - rt));
        CC_WAVE_SYNC();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float v = (float)(tc[q] - org[lane & 7]);
            tile[lane + q * 64] = v;
            float hsum = v * v;
            hsum += __shfl_xor(hsum, 1);
            hsum += __shfl_xor(hsum, 2);
            hsum += __shfl_xor(hsum, 4);
            // a prefix beyond single precision's range: never abandoned (t' < -inf is false)
            if ((lane & 7) == 0) thh[(lane >> 3) + q * 8] = (hsum < 0x1p120f) ? 0.5f * hsum : -__builtin_inff();
        }
        const unsigned pmask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_PCORE);
        const unsigned omask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_OUTLIER);
        CC_WAVE_SYNC();
        if (rt + CC_SCAN_TM < r1) cc_load_prefix<DP>(g_cen, g_kind, rt + CC_SCAN_TM, min(CC_SCAN_TM, r1 - rt - CC_SCAN_TM), lane, tc, kdl);
        n_rows += 2 * tm;
        const cc_f4* t4 = reinterpret_cast<const cc_f4*>(__builtin_assume_aligned(tile, 16));
        const unsigned full = (1u << tm) - 1u;
        unsigned surv[2] = {0u, 0u};
        // ---- phase A (k_scan_a's): t' = p^.c^ - tau < h^ abandons the row for the half ----
        auto rows_of = [&](auto KSELC) {
            constexpr int KSEL = decltype(KSELC)::value;
            auto two = [&](int m0, int n) {
                cc_f4 c01[2], c23[2];
                float h[2];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int m = (v < n) ? m0 + v : m0;
                    c01[v] = t4[m * 2]; c23[v] = t4[m * 2 + 1]; h[v] = thh[m];
                }
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    if (v >= n) break;
                    const int m = m0 + v;
                    int K;
                    if constexpr (KSEL == 0) K = 0;
                    else if constexpr (KSEL == 1) K = 1;
                    else K = ((pmask >> m) & 1u) ? 0 : 1;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        cc_f2 acc = __builtin_elementwise_fma(p2[u][0], cc_f2{c01[v].x, c01[v].y}, K == 0 ? ninit[u][0] : ninit[u][1]);
                        acc = __builtin_elementwise_fma(p2[u][1], cc_f2{c01[v].z, c01[v].w}, acc);
                        acc = __builtin_elementwise_fma(p2[u][2], cc_f2{c23[v].x, c23[v].y}, acc);
                        acc = __builtin_elementwise_fma(p2[u][3], cc_f2{c23[v].z, c23[v].w}, acc);
                        float t;
                        asm("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(acc.x), "v"(acc.y));
                        // kept unless t' < h^ (a NaN - an overflow the guards did not see - keeps the row)
                        surv[u] |= (__builtin_amdgcn_ballot_w64(!(t < h[v])) != 0ull) ? (1u << m) : 0u;
                    }
                }
            };
            int m = 0;
            for (; m + 2 <= tm; m += 2) two(m, 2);
            if (m < tm) two(m, 1);
        };
        if (pmask == full) rows_of(std::integral_constant<int, 0>{});
        else if (omask == full) rows_of(std::integral_constant<int, 1>{});
        else rows_of(std::integral_constant<int, -1>{});
        // rows of neither list (dead) are never completed: dropped from the masks, not counted as abandoned
        const unsigned listed = (pmask | omask) & full;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            surv[u] &= listed;
            if ((~surv[u] & pmask & full) != 0u) dflag |= 1u << (2 * u);
            if ((~surv[u] & omask & full) != 0u) dflag |= 2u << (2 * u);
        }
        // ---- phase B: the rows that stayed, half by half ----
        cc_static_for<2>([&](auto UC) {
            constexpr int u = decltype(UC)::value;
            unsigned sv = surv[u];
            while (sv != 0u) {
                const int m = __builtin_ctz(sv);
                sv &= sv - 1u;
                complete_row(UC, rt + m, ((pmask >> m) & 1u) != 0u);
            }
        });
    }
    // rows abandoned in phase A: their exact partial sums exceed every lane's T, which is all that is recorded of them
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if ((dflag >> (2 * u)) & 1u) lb[u][0] = cc_vmin(lb[u][0], thx(u, 0));
        if ((dflag >> (2 * u)) & 2u) lb[u][1] = cc_vmin(lb[u][1], thx(u, 1));
    }
    if (guessed && found != nullptr) {
        // the points for which this wave evaluated a pcore MC within the guessed threshold: their pcore list's best is exact
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned long long fm = __builtin_amdgcn_ballot_w64(valid[u] && bs[u][0][0] >= 0 && bd[u][0][0] <= thx(u, 0));
            if (lane == 0 && fm != 0ull) atomicOr(found + (size_t)win.q * (CC_MAX_WINDOW / 64) + (size_t)blockIdx.x * 2 + u, fm);
        }
    }
    // statistics for the host's policy: a sample - the waves of the window's first point tile (see k_scan_p)
    if (lane == 0 && blockIdx.x == 0 && n_rows > 0) {
        atomicAdd(pstat + win.q * 2, (unsigned long long)n_rows);
        atomicAdd(pstat + win.q * 2 + 1, (unsigned long long)n_full);
    }
    // the survivors' list-order keys; every kind's pair takes in the bound of what the wave abandoned; the waves' pairs are
    // merged through LDS (the bytes of the staged points: every wave is done with them)
    Cand cnd[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int K = 0; K < 2; ++K) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int sl = bs[u][K][r];
                cnd[u][K * 2 + r] = Cand{bd[u][K][r], sl >= 0 ? g_key[sl] : CC_IDX_INF, sl};
            }
            cc_top2_push(cnd[u][K * 2], cnd[u][K * 2 + 1], Cand{lb[u][K], -1, (valid[u] && lb[u][K] < CC_INF) ? CC_SLOT_BOUND : -1});
        }
    }
    Cand* s_m = reinterpret_cast<Cand*>(smem);
    auto s_m_at = [&](int w, int c) -> Cand& { return s_m[(w * 8 + c) * 64 + lane]; };
    __syncthreads();
    if (wv > 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) s_m_at(wv - 1, u * 4 + c) = cnd[u][c];
    }
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (!valid[u]) continue;
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) {
            cc_top2_push(cnd[u][0], cnd[u][1], s_m_at(w, u * 4 + 0));
            cc_top2_push(cnd[u][0], cnd[u][1], s_m_at(w, u * 4 + 1));
            cc_top2_push(cnd[u][2], cnd[u][3], s_m_at(w, u * 4 + 2));
            cc_top2_push(cnd[u][2], cnd[u][3], s_m_at(w, u * 4 + 3));
        }
        Cand* o = part + ((size_t)(j0 + u * 64 + lane) * S + blockIdx.y) * 4;
        o[0] = cnd[u][0]; o[1] = cnd[u][1]; o[2] = cnd[u][2]; o[3] = cnd[u][3];
    }
}

// ---------------------------------------------------------------------------------
// k_missed: after a snapshot scan with guessed thresholds - the window points no wave found a pcore MC for (their own MC
// lies beyond the guess, or they have none), in point order, for the seeded chain that follows (k_seed / k_seed_merge /
// k_scan_p over `list`).  One workgroup; the marks are cleared for the window's next use of them.  A list that would
// exceed `cap` is cut: the points beyond it keep a bound in first place and k_decide refuses them.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_missed(Ctl* __restrict__ ctl, unsigned long long* __restrict__ found,
                                                 int* __restrict__ list, int cap, int round, int mode)
{
    const ScanWin win = cc_scan_window(ctl, round, mode);
    if (win.B == 0) return;  // (mode 0 and the window was scanned ahead: that scan's list stands)
    found += (size_t)win.q * (CC_MAX_WINDOW / 64);
    list += (size_t)win.q * CC_MISSED_CAP;
    const int tiles = (win.B + 63) / 64;
    const int tid = threadIdx.x;
    constexpr int PER = (CC_MAX_WINDOW / 64 + 1023) / 1024;  // tiles per thread
    unsigned long long miss[PER];
    int mine = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int t = tid * PER + q;
        miss[q] = 0ull;
        if (t < tiles) {
            const int left = win.B - t * 64;
            const unsigned long long all = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            miss[q] = ~found[t] & all;
            if (t == 0 && ctl->seed_at == win.cursor) miss[q] |= 1ull;  // (the point the previous window was cut short at)
            found[t] = 0ull;
            mine += __builtin_popcountll(miss[q]);
        }
    }
    __shared__ int wsum[16];
    __shared__ int total;
    int v = mine;
    const int lane = tid & 63, wid = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    if (lane == 63) wsum[wid] = v;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 16; ++i) { const int x = wsum[i]; wsum[i] = run; run += x; }
        total = run;
    }
    __syncthreads();
    int pos = v - mine + wsum[wid];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        unsigned long long m = miss[q];
        const int t = tid * PER + q;
        while (m != 0ull) {
            const int b = __builtin_ctzll(m);
            m &= m - 1ull;
            if (pos < cap) list[pos] = t * 64 + b;
            ++pos;
        }
    }
    if (tid == 0) {
        ctl->n_missed[win.q] = total < cap ? total : cap;
        ctl->n_missed_all[win.q] = total;  // (k_decide adds it to Ctl::stat_missed)
    }
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
                                                        size_t out_stride, int round, int mode,
                                                        const unsigned long long* __restrict__ pstat, int tail_at,
                                                        const int* __restrict__ plist)
{
    int B, q;
    if (mode == 1) {
        q = round & 1;
        B = ctl->la_b[q];
    } else {
        q = (int)(ctl->window_seq & 1ull);
        B = ctl->win_b;
    }
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (plist != nullptr) {
        // the points a guessed-threshold scan missed (k_missed_g's list): their new partials -> one record each, COMPACT
        // (record e = the list's e-th point), the unit of the second, small all-gather; no parity offset, no tail
        if (B == 0 || j >= ctl->n_missed[q]) return;
        const int e = j;
        j = plist[(size_t)q * CC_MISSED_CAP + e];
        part += (size_t)q * part_stride;
        const Cand none = Cand{CC_INF, CC_IDX_INF, -1};
        Cand p1 = none, p2 = none, o1 = none, o2 = none;
        for (int s2 = 0; s2 < S; ++s2) {
            const Cand* c = part + ((size_t)j * S + s2) * 4;
            cc_top2_push(p1, p2, c[0]);
            cc_top2_push(p1, p2, c[1]);
            cc_top2_push(o1, o2, c[2]);
            cc_top2_push(o1, o2, c[3]);
        }
        Cand* o = out + (size_t)e * 4;
        o[0] = p1; o[1] = p2; o[2] = o1; o[3] = o2;
        return;
    }
    // the record behind the last point's: this rank's sample of its pruned scan (k_scan_p), {rows visited, rows completed},
    // zeros after a plain scan - gathered with the candidates, summed over the ranks by k_decide
    if (j == 0) {
        Cand tail = Cand{0.0, 0, 0};
        if (pstat != nullptr) { tail.key = (int)pstat[q * 2]; tail.slot = (int)pstat[q * 2 + 1]; }
        out[(size_t)q * out_stride + (size_t)tail_at] = tail;
    }
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
// Guessed thresholds on the exact multi-GPU path.  Every rank scanned its share of the rows against the same guess; whether a
// point's own MC was found is only known once the ranks' records are gathered: k_missed_g lists the points whose merged
// pcore list starts with a bound (the same list on every rank: it is a function of the gathered records), the seeded chain
// runs for them (seeds over all rows, replicated - they are few -, phases A / B over the rank's rows), k_merge_partials
// packs their new records compactly, a second, small all-gather (CC_MISSED_CAP records per rank) exchanges those, and
// k_scatter_missed puts every rank's new record in the place of the old one in the gathered buffer k_decide reads.
// ---------------------------------------------------------------------------------
// the sample counters of a split scan with guessed thresholds start at zero (the seeded chain: k_seed_merge does it)
__global__ void k_pstat_zero(const Ctl* __restrict__ ctl, unsigned long long* __restrict__ pstat, int round, int mode)
{
    const ScanWin win = cc_scan_window(ctl, round, mode);
    if (win.B == 0) return;
    pstat[win.q * 2 + threadIdx.x] = 0ull;
}

__global__ __launch_bounds__(1024) void k_missed_g(Ctl* __restrict__ ctl, const Cand* __restrict__ gpart, size_t gpart_stride,
                                                   size_t outer, int world, int* __restrict__ list, int cap, int round, int mode,
                                                   unsigned long long* __restrict__ found)
{
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int B = win.B;
    if (B == 0) return;  // (mode 0 and the window was scanned ahead: that pass's list stands, the chain after it is idle)
    gpart += (size_t)win.q * gpart_stride;
    list += (size_t)win.q * CC_MISSED_CAP;
    const int tid = threadIdx.x;
    // (the scan's own marks are not what decides here; cleared for a later unsplit use of them)
    for (int t = tid; t < (B + 63) / 64; t += 1024) found[(size_t)win.q * (CC_MAX_WINDOW / 64) + t] = 0ull;
    const int per = (B + 1023) >> 10;
    const bool seed_first = ctl->seed_at == win.cursor;  // (the point the previous window was cut short at, Ctl::seed_at)
    auto missed = [&](int j) -> bool {
        Cand best = Cand{CC_INF, CC_IDX_INF, -1};
        for (int r = 0; r < world; ++r) {
            const Cand c = gpart[(size_t)r * outer + (size_t)j * 4];
            if (c.slot != -1 && (best.slot == -1 || cand_less(c.dist, c.key, best.dist, best.key))) best = c;
        }
        return best.slot == CC_SLOT_BOUND || (j == 0 && seed_first);
    };
    int mine = 0;
    for (int k = 0; k < per; ++k) {
        const int j = tid * per + k;
        if (j < B && missed(j)) ++mine;
    }
    __shared__ int wsum[16];
    __shared__ int total;
    int v = mine;
    const int lane = tid & 63, wid = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    if (lane == 63) wsum[wid] = v;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 16; ++i) { const int x = wsum[i]; wsum[i] = run; run += x; }
        total = run;
    }
    __syncthreads();
    int pos = v - mine + wsum[wid];
    for (int k = 0; k < per; ++k) {
        const int j = tid * per + k;
        if (j < B && missed(j)) {
            if (pos < cap) list[pos] = j;
            ++pos;
        }
    }
    if (tid == 0) {
        ctl->n_missed[win.q] = total < cap ? total : cap;
        ctl->n_missed_all[win.q] = total;  // (k_decide adds it to Ctl::stat_missed)
    }
}

__global__ __launch_bounds__(256) void k_scatter_missed(const Ctl* __restrict__ ctl, const int* __restrict__ list,
                                                        const Cand* __restrict__ gathered, int world, Cand* __restrict__ gpart,
                                                        size_t gpart_stride, size_t outer, int round, int mode)
{
    int q;
    if (mode == 1) q = round & 1;
    else q = (int)(ctl->window_seq & 1ull);
    const int n = ctl->n_missed[q];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = t / world, r = t - e * world;
    if (e >= n) return;
    const int j = list[(size_t)q * CC_MISSED_CAP + e];
    const Cand* src = gathered + ((size_t)r * CC_MISSED_CAP + e) * 4;
    Cand* dst = gpart + (size_t)q * gpart_stride + (size_t)r * outer + (size_t)j * 4;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
}
