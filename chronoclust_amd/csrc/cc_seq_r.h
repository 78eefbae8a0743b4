// chronoclust_amd/csrc: k_seq_r, the sequential kernel with the table in REGISTERS.  (included by cc_online.h)
#pragma once
#include "cc_div.h"

// ---------------------------------------------------------------------------------
// k_seq_r: the reference's loop taken literally (hddstream.py:220-237) like k_seq, for the shapes of the reference's own
// data (d <= 4, a handful of pcore MCs, a few hundred rows in all: the bundled d0-d4 files are 7 100 x 3 with 2-15 pcore
// and 21-128 outlier MCs).  k_seq keeps the table in LDS and pays an LDS round trip (~120 cycles, nothing to overlap it
// with in a single wave) eight times per point; here every lane HOLDS rows in registers - one pcore row (slot P) and up
// to SeqRShape::QO outlier rows (slots O[q]) - and a point costs arithmetic only:
//   the point's coordinates are wave-uniform (read from the lane that loaded the point, v_readlane);
//   stage 0 (_add_to_pcore, hddstream.py:288-343): every lane computes, for its own pcore row, the projected distance
//     AND the tentative add (microcluster.py:213-233) - in SIMD the add for all rows costs what the add for the winner
//     costs, and the pdim filter (:317-321) needs it for every row anyway; wave argmin by (distance, list-order key)
//     through DPP row operations; the winner's radius test is read with one ballot; commit = the winner lane keeps its
//     tentative values;
//   stage 1 (_add_to_outlier, :345-395): the same over the outlier slots (distance for every slot, the tentative add for
//     the winner's slot), promotion (:416-430) moves the row into the next free pcore slot by v_readlane;
//   else a new outlier MC (:434-462) in the next free outlier slot.
// The 2 d divisions of a tentative add share their denominator: cc_div.h (three instructions per quotient, bit-exact
// with the compiler's division; operands outside its safe range take the compiler's division).
// Capacity: 64 pcore rows, 64 x SeqRShape::QO outlier slots (slots of promoted rows are not reused).  A table that does not
// fit, or a point that could overflow it, ends the kernel: Ctl::seq_rest tells k_seq, launched right behind, how many
// points of the stint are left for it (0: none - it returns at once).
// ---------------------------------------------------------------------------------

// outlier slots per lane: what the 256 vector registers of the wave hold beside the pcore slot and the tentative values
template <int D> struct SeqRShape { static constexpr int QO = D <= 3 ? 4 : 3; };

template <int D>
struct SeqSlot {
    double cf1[D], cf2[D], cen[D], op[D];
    double w, rw;  // weight; the prepared reciprocal of w + 1 (cc_div_prepare), what the next tentative add divides by
    long long id, uid;
    int key, row;  // list-order key; row of the table in HBM this slot is written back to (-1: empty slot)
};

// minimum over the first `n_lanes` lanes (the others hold +inf; no NaN among the operands), wave-uniform
// (all 64 lanes are active and every source lane of these controls exists: no `old` value to set up, unlike cc_dpp_f64)
template <int CTRL>
__device__ __forceinline__ double cc_seqr_dpp(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cc_seqr_min(double x, int n_lanes)
{
    x = cc_vmin(x, cc_seqr_dpp<0xB1>(x));   // quad_perm [1,0,3,2]
    x = cc_vmin(x, cc_seqr_dpp<0x4E>(x));   // quad_perm [2,3,0,1]
    x = cc_vmin(x, cc_seqr_dpp<0x141>(x));  // row_half_mirror
    x = cc_vmin(x, cc_seqr_dpp<0x140>(x));  // row_mirror: every lane of a row of 16 holds the row's minimum
    const double a = cc_readlane_f64(x, 0);
    if (n_lanes <= 16) return a;
    const double b = cc_readlane_f64(x, 16), c = cc_readlane_f64(x, 32), e = cc_readlane_f64(x, 48);
    const double ab = a < b ? a : b, ce = c < e ? c : e;
    return ab < ce ? ab : ce;
}

template <int D, bool POW2>
__global__ __launch_bounds__(64) void k_seq_r(Ctl* __restrict__ ctl, const double* __restrict__ X, Table tab,
                                              long long* __restrict__ lab_uid, int8_t* __restrict__ lab_path, int n_max)
{
    const long long clk0 = clock64(), wall0 = wall_clock64();
    const Par par = cc_load_par(ctl);
    const int lane = threadIdx.x;
    constexpr int QO = SeqRShape<D>::QO;
    int M = ctl->m_rows;
    const long long cursor0 = ctl->cursor;
    const long long left = ctl->n_points - cursor0;
    const int n = (int)(left < (long long)n_max ? left : (long long)n_max);
    if (n <= 0) {
        if (lane == 0) ctl->seq_rest = 0;
        return;
    }
    int n_pkeys = ctl->n_pkeys, n_okeys = ctl->n_okeys;
    long long pcore_last_id = ctl->pcore_last_id, outlier_last_id = ctl->outlier_last_id;
    const bool filter = par.filter != 0;
    auto op_of = [&](double pr) { return POW2 ? (pr == 1.0 ? 1.0 : par.inv_k) : pr; };
    auto pref_of = [&](double op) { return POW2 ? (op == 1.0 ? 1.0 : par.k) : op; };
    auto scaled = [&](double x, double op) { return POW2 ? x * op : (op == 1.0 ? x : x / op); };  // mc_functions.py:39

    // ---- the rows by kind, in row order: pcore row number t -> lane t, outlier row number t -> slot t / 64 of lane t % 64
    __shared__ int s_rows[2][64 * QO];
    int n_p = 0, n_o = 0;
    bool fits = M <= 64 + 64 * QO;
    if (fits) {
        for (int r0 = 0; r0 < M; r0 += 64) {
            const int r = r0 + lane;
            const int kd = r < M ? tab.kind[r] : CC_KIND_DEAD;
            const unsigned long long mp = __builtin_amdgcn_ballot_w64(kd == CC_KIND_PCORE);
            const unsigned long long mo = __builtin_amdgcn_ballot_w64(kd == CC_KIND_OUTLIER);
            const unsigned long long below = (1ull << lane) - 1ull;
            const int ip = n_p + __builtin_popcountll(mp & below), io = n_o + __builtin_popcountll(mo & below);
            if (kd == CC_KIND_PCORE && ip < 64) s_rows[0][ip] = r;
            if (kd == CC_KIND_OUTLIER && io < 64 * QO) s_rows[1][io] = r;
            n_p += __builtin_popcountll(mp);
            n_o += __builtin_popcountll(mo);
        }
        fits = n_p <= 64 && n_o <= 64 * QO;
    }
    if (!fits) {
        if (lane == 0) ctl->seq_rest = n;  // nothing taken: k_seq behind this kernel takes the stint
        return;
    }
    CC_WAVE_SYNC();

    // The shared-denominator division (cc_div.h) is exact while no numerator needs scaling: 2^-900 <= |CF sum| <= 2^700
    // or +0.  That holds by induction, without a test per quotient, while every operand that enters the sums - the CF
    // entries as loaded, every coordinate of every point - is +0 or has a magnitude in [2^-400, 2^300]: all of them and
    // the squares of the coordinates are then multiples of 2^-852, so is every (rounded) sum of them, a nonzero sum is
    // >= 2^-852, a zero sum is +0 (only (-0) + (-0) gives -0), and 2^32 points cannot pass 2^700.  `exact_ok` is that
    // premise: checked for the table here, for the points as their chunk is loaded; once an operand outside the range
    // has entered a sum, the rest of the kernel divides with the compiler's sequence.
    auto operand_ok = [&](double v) {
        const double a = __builtin_fabs(v);
        return (a >= 0x1p-400 && a <= 0x1p300) || (a == 0.0 && !__builtin_signbit(v));
    };
    bool slot_ok = true;
    SeqSlot<D> P, O[QO];
    auto load_slot = [&](SeqSlot<D>& s, int r) {
        s.row = r;
        if (r >= 0) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                s.cf1[i] = tab.cf1[(size_t)r * D + i]; s.cf2[i] = tab.cf2[(size_t)r * D + i];
                s.cen[i] = tab.cen[(size_t)r * D + i]; s.op[i] = op_of(tab.pref[(size_t)r * D + i]);
                slot_ok = slot_ok && operand_ok(s.cf1[i]) && operand_ok(s.cf2[i]);
            }
            s.w = tab.w[r]; s.key = tab.key[r]; s.id = tab.id[r]; s.uid = tab.uid[r];
            slot_ok = slot_ok && s.w >= 0.0 && s.w < 0x1p59;
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) { s.cf1[i] = 0.0; s.cf2[i] = 0.0; s.cen[i] = 0.0; s.op[i] = 1.0; }
            s.w = 0.0; s.key = CC_IDX_INF; s.id = 0; s.uid = 0;
        }
        s.rw = cc_div_prepare(s.w + 1.0);
    };
    load_slot(P, lane < n_p ? s_rows[0][lane] : -1);
#pragma unroll
    for (int q = 0; q < QO; ++q) load_slot(O[q], q * 64 + lane < n_o ? s_rows[1][q * 64 + lane] : -1);
    int n_os = n_o;  // outlier slots in use (slots of promoted rows stay empty)
    bool exact_ok = __builtin_amdgcn_ballot_w64(!slot_ok) == 0ull;  // (wave-uniform)
    const bool k_ne1 = par.k != 1.0, k_gt1 = par.k > 1.0;  // pref' = k where the variance is low: is that != 1 / > 1?

    // tentative add of the point to slot s (microcluster.py:213-233): CF sums, centroid, operands and the projected radius
    // (mc_functions.py:45-56, the sum over dimensions left to right), count(pref' > 1) and count(pref' != 1)
    // low: per dimension, is the variance of the enlarged MC <= delta^2 (pref' = k there, 1 elsewhere; microcluster.py:109-114)
    struct Tent {
        double c1[D], c2[D], cen[D], op[D], w1, r2;
        double rw;  // cc_div_prepare(w1 + 1): what the slot divides by next if it keeps these values (computed here, beside
                    // the other chains, rather than behind the commit where the next point waits for it)
        bool low[D];
    };
    const double op_k = op_of(par.k);
    auto tentative = [&](const SeqSlot<D>& s, const double (&x)[D]) {
        Tent t;
        t.w1 = s.w + 1.0;
        t.rw = cc_div_prepare(t.w1 + 1.0);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            t.c1[i] = s.cf1[i] + x[i];
            t.c2[i] = s.cf2[i] + x[i] * x[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double qa, qb;
            if (exact_ok) {  // (wave-uniform)
                qa = cc_div_apply(t.c2[i], t.w1, s.rw);
                qb = cc_div_apply(t.c1[i], t.w1, s.rw);
            } else {
                qa = t.c2[i] / t.w1;
                qb = t.c1[i] / t.w1;
            }
            const double var = qa - qb * qb;  // mc_functions.py:14-22
            t.low[i] = var <= par.delta_sq;   // (NaN -> pref' 1.0)
            t.cen[i] = qb;
            t.op[i] = t.low[i] ? op_k : 1.0;
            const double term = POW2 ? var * t.op[i] : ((t.low[i] && k_ne1) ? var / par.k : var);  // mc_functions.py:52
            // :54, left to right from 0.0: 0.0 + term is term except for the sign of a zero, which the comparison
            // with epsilon^2 - the only use of the sum - does not see
            t.r2 = i == 0 ? term : t.r2 + term;
        }
        return t;
    };
    auto count_low = [&](const Tent& t) {
        int c = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) c += t.low[i] ? 1 : 0;
        return c;
    };
    auto dist_to = [&](const SeqSlot<D>& s, const double (&x)[D]) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double t = x[i] - s.cen[i];   // mc_functions.py:37
            t = t * t;                    // :38
            // :39 + :41 (0.0 + the first term is the term itself: a square is never -0.0)
            acc = i == 0 ? scaled(t, s.op[i]) : acc + scaled(t, s.op[i]);
        }
        return acc;
    };
    auto keep = [&](SeqSlot<D>& s, const Tent& t) {
#pragma unroll
        for (int i = 0; i < D; ++i) { s.cf1[i] = t.c1[i]; s.cf2[i] = t.c2[i]; s.cen[i] = t.cen[i]; s.op[i] = t.op[i]; }
        s.w = t.w1;
        s.rw = t.rw;
    };
    // the winner among the lanes whose (distance, key) is (bd, bk), valid lanes only: smallest distance, then smallest key
    auto winner_lane = [&](double bd, int bk, bool valid, int n_lanes) -> int {
        const double Dm = cc_seqr_min(valid ? bd : CC_INF, n_lanes);
        const unsigned long long tied = __builtin_amdgcn_ballot_w64(valid && bd == Dm);
        if (tied == 0ull) return -1;
        int wl = __builtin_ctzll(tied);
        if (tied & (tied - 1ull)) {
            int best_key = CC_IDX_INF;
            for (unsigned long long m = tied; m; m &= m - 1ull) {
                const int l = __builtin_ctzll(m);
                const int k2 = __builtin_amdgcn_readlane(bk, l);
                if (k2 < best_key) { best_key = k2; wl = l; }
            }
        }
        return wl;
    };
    auto bcast_ll = [&](long long v, int l) -> long long {
        const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
        return ((long long)hi << 32) | (unsigned int)lo;
    };

    // ---- the points: lane l loads point c0 + l of a chunk of 64, the next chunk is in flight while this one is processed
    double px[D], pn[D];
    auto fetch = [&](int first, double (&dst)[D]) {
        const int j = first + lane < n ? first + lane : n - 1;
        const double* src = X + (cursor0 + j) * D;
#pragma unroll
        for (int i = 0; i < D; ++i) dst[i] = src[i];
    };
    auto chunk_ok_mask = [&](const double (&p)[D]) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < D; ++i) ok = ok && operand_ok(p[i]);
        return __builtin_amdgcn_ballot_w64(ok);
    };
    fetch(0, px);
    int done = 0;
    bool full = false;
    for (int c0 = 0; c0 < n && !full; c0 += 64) {
        const int cnt = (n - c0) < 64 ? (n - c0) : 64;
        if (c0 + 64 < n) fetch(c0 + 64, pn);
        const unsigned long long pts_ok = chunk_ok_mask(px);  // bit jj: the coordinates of point c0 + jj keep the premise
        long long my_uid = 0;  // lane jj: the label of point c0 + jj
        int my_path = 0;
        int cdone = 0;
        for (int jj = 0; jj < cnt; ++jj) {
            // room for whatever this point does (a promotion takes a pcore slot, a creation an outlier slot)?
            if (n_p >= 64 || n_os >= 64 * QO) { full = true; break; }
            double x[D];
#pragma unroll
            for (int i = 0; i < D; ++i) x[i] = cc_readlane_f64(px[i], jj);
            exact_ok = exact_ok && ((pts_ok >> jj) & 1ull);  // (this point enters a sum whatever happens to it)
            long long uid_out = 0;
            int path = -1;
            bool promoted = false;
            // ---- stage 0: _add_to_pcore (hddstream.py:288-343)
            if (n_p > 0) {
                const Tent t = tentative(P, x);
                const double dq = dist_to(P, x);
                // :317-321: count(pref' != 1) of the enlarged MC must not exceed pi
                const bool adm = lane < n_p && !(filter && k_ne1 && count_low(t) > par.pi);
                const int wl = winner_lane(dq, P.key, adm, n_p);
                if (wl >= 0) {
                    const bool ok = (__builtin_amdgcn_ballot_w64(t.r2 <= par.eps_sq) >> wl) & 1ull;  // :334-337
                    if (ok) {
                        if (lane == wl) keep(P, t);
                        uid_out = bcast_ll(P.uid, wl);
                        path = 0;
                    }
                }
            }
            // ---- stage 1: _add_to_outlier (:345-395)
            if (path < 0 && n_os > 0) {
                double bd = CC_INF;
                int bk = CC_IDX_INF, bq = -1;
#pragma unroll
                for (int q = 0; q < QO; ++q) {
                    if (q * 64 < n_os) {  // (wave-uniform)
                        const double dq = dist_to(O[q], x);
                        if (O[q].row >= 0 && cand_less(dq, O[q].key, bd, bk)) { bd = dq; bk = O[q].key; bq = q; }
                    }
                }
                const int wl = winner_lane(bd, bk, bq >= 0, 64);
                if (wl >= 0) {
                    const int wq = __builtin_amdgcn_readlane(bq, wl);
#pragma unroll
                    for (int q = 0; q < QO; ++q) {
                        if (q == wq) {  // (wave-uniform)
                            const Tent t = tentative(O[q], x);
                            const bool ok = (__builtin_amdgcn_ballot_w64(t.r2 <= par.eps_sq) >> wl) & 1ull;  // :378-381
                            if (ok) {
                                if (lane == wl) keep(O[q], t);
                                uid_out = bcast_ll(O[q].uid, wl);
                                path = 1;
                                // hddstream.py:416-430: heavy enough and few enough preferred dimensions -> a pcore MC
                                const int gt1 = k_gt1 ? __builtin_amdgcn_readlane(count_low(t), wl) : 0;
                                const double w1 = cc_readlane_f64(t.w1, wl);
                                if (w1 >= par.beta_mu && gt1 <= par.pi) {
                                    promoted = true;
                                    // the row moves to the next pcore slot (lane n_p), its outlier slot stays empty
#pragma unroll
                                    for (int i = 0; i < D; ++i) {
                                        const double a = cc_readlane_f64(O[q].cf1[i], wl), b = cc_readlane_f64(O[q].cf2[i], wl);
                                        const double c = cc_readlane_f64(O[q].cen[i], wl), e = cc_readlane_f64(O[q].op[i], wl);
                                        if (lane == n_p) { P.cf1[i] = a; P.cf2[i] = b; P.cen[i] = c; P.op[i] = e; }
                                    }
                                    const double rw = cc_readlane_f64(O[q].rw, wl);
                                    const int row = __builtin_amdgcn_readlane(O[q].row, wl);
                                    if (lane == n_p) {
                                        P.w = w1; P.rw = rw; P.uid = uid_out; P.id = pcore_last_id; P.key = n_pkeys; P.row = row;
                                    }
                                    if (lane == wl) O[q].row = -1;
                                    n_p += 1;
                                    n_pkeys += 1;
                                    pcore_last_id += 1;
                                }
                            }
                        }
                    }
                }
            }
            // ---- hddstream.py:434-462: a new outlier MC holding this point (an add to an empty MC)
            if (path < 0) {
                const int tl = n_os & 63, tq = n_os >> 6;
#pragma unroll
                for (int q = 0; q < QO; ++q) {
                    if (q == tq && lane == tl) {
#pragma unroll
                        for (int i = 0; i < D; ++i) {
                            const double c1 = 0.0 + x[i], c2 = 0.0 + x[i] * x[i];
                            const double qa = c2 / 1.0, qb = c1 / 1.0;
                            const double var = qa - qb * qb;
                            O[q].cf1[i] = c1; O[q].cf2[i] = c2; O[q].cen[i] = qb;
                            O[q].op[i] = op_of((var <= par.delta_sq) ? par.k : 1.0);
                        }
                        O[q].w = 0.0 + 1.0; O[q].rw = cc_div_prepare(2.0);
                        O[q].key = n_okeys; O[q].id = outlier_last_id; O[q].uid = outlier_last_id; O[q].row = M;
                    }
                }
                uid_out = outlier_last_id;
                n_os += 1;
                n_okeys += 1;
                outlier_last_id += 1;
                M += 1;
                path = 2;
            }
            if (lane == jj) {
                my_uid = uid_out;
                my_path = path | (promoted ? 4 : 0);
            }
            cdone = jj + 1;
        }
        if (lane < cdone) {
            lab_uid[cursor0 + c0 + lane] = my_uid;
            lab_path[cursor0 + c0 + lane] = (int8_t)my_path;
        }
        done = c0 + cdone;
#pragma unroll
        for (int i = 0; i < D; ++i) px[i] = pn[i];
    }

    // ---- the rows back to HBM (every column the windowed path reads, scl included)
    auto store_slot = [&](const SeqSlot<D>& s, int kind) {
        const int r = s.row;
        if (r < 0) return;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            tab.cf1[(size_t)r * D + i] = s.cf1[i]; tab.cf2[(size_t)r * D + i] = s.cf2[i];
            tab.cen[(size_t)r * D + i] = s.cen[i]; tab.pref[(size_t)r * D + i] = pref_of(s.op[i]);
            tab.scl[(size_t)r * D + i] = s.op[i];
        }
        tab.w[r] = s.w; tab.kind[r] = kind; tab.key[r] = s.key; tab.id[r] = s.id; tab.uid[r] = s.uid;
    };
    if (lane < n_p) store_slot(P, CC_KIND_PCORE);
#pragma unroll
    for (int q = 0; q < QO; ++q) store_slot(O[q], CC_KIND_OUTLIER);
    if (lane == 0) {
        ctl->cursor = cursor0 + done;
        ctl->m_rows = M;
        ctl->n_pkeys = n_pkeys; ctl->n_okeys = n_okeys;
        ctl->pcore_last_id = pcore_last_id; ctl->outlier_last_id = outlier_last_id;
        ctl->window_seq += 1ull;  // stamps and carry marks of earlier windows are history
        ctl->mode = 0; ctl->car_n = 0;
        ctl->stat_seq_points += done;
        ctl->stat_seq_r_points += done;
        ctl->stat_seq_clk += clock64() - clk0;
        ctl->stat_seq_wall += wall_clock64() - wall0;
        ctl->seq_rest = n - done;  // (capacity reached: k_seq continues on its LDS image, up to its own capacity)
    }
}
