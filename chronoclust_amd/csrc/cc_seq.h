// chronoclust_amd/csrc: k_seq, the sequential kernel.  (included by cc_online.h; one translation unit, cc_api.hip)
#pragma once

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

// (CC_SEQ_DOUBLES - the LDS image of the table: (4 d + 5) doubles per row - and cc_seq_cap_rows: cc_host.h)
#define CC_SEQ_CHUNK_DOUBLES 512  // points staged ahead: 512 / d of them (at most 64), eight doubles per lane in flight

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
                                            long long* __restrict__ lab_uid, int8_t* __restrict__ lab_path, int n_max,
                                            int follow)
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
    // follow: launched behind k_seq_r (cc_seq_r.h) for what that kernel left of the stint - usually nothing
    if (follow) n_max = ctl->seq_rest;
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
