// cc_scan16.h (round 6) - the prefix test of the pruned snapshot scan on the matrix cores.
//
// Phase A of the pruned scan asks, per (window point, table row): does the exact partial sum over the first eight
// dimensions exceed the point's threshold T?  In the expanded form (k_scan_a, cc_scan.h)
//     |p' - c'|^2 = |p'|^2 - 2 (p'.c' - |c'|^2 / 2),      p' = p - o,  c' = c - o   (o = prefix of table row 0)
// that is the SIGN of  p^.c^ - h^ - tau  - a dot product of length eight plus one constant per row and one per point: a
// 32 x 32 x 16 matrix product whose K axis holds the eight dimensions and, in its spare slots, the two constants.  Phase A
// is a FILTER: a row it keeps is evaluated exactly by phase B with the reference's operations, so nothing here touches a
// result; what has to hold is "abandoned implies beyond T", and cc_tau16 states it for half-precision operands.
//
// v_mfma_f32_32x32x16_f16 (32 cycles per instruction and SIMD, products exact in the FP32 accumulator):
//     A (32 x 16) = window points,  B (16 x 32) = table rows,  D[m][n] = sum_k A[m][k] B[k][n]
//     k = 0 .. 7    p^_k                         c^_k                       the prefix, centred on o and scaled by 2^-e
//     k = 8, 9      1, 2^-10                     -h1, -h2 2^10              h^ = |c^|^2 / 2 in two half-precision pieces
//     k = 10, 11    -tauP1, -tauP2 2^10          isP, isP 2^-10             the point's tau against pcore rows ...
//     k = 12, 13    -tauO1, -tauO2 2^10          isO, isO 2^-10             ... and against outlier rows
//     k = 14, 15    0                            0
// Lane l = (r = l & 31, h = l >> 5) holds A[r][8 h .. 8 h + 7] and B[8 h .. 8 h + 7][r]; D: column = l & 31 (the table row),
// row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5) (the point): "some point of the block keeps the row" is a maximum over the
// lane's sixteen registers (v_max3_f32), one compare, and the OR of the two lane halves - 34 VALU instructions per 32 rows
// and 128 points where the packed-FP32 form (k_scan_p2) spends 416 and the same number of scalar ones.
// The 2^-e scaling (2^e >= twice the largest |coordinate| of points and table) puts every operand into [-1, 1]: half precision
// has the range of 2^-14 .. 65504 only.  The second pieces are pre-multiplied by 2^10 so that they stay normal numbers
// (a flushed subnormal would cost 2^-14 of accuracy).
#pragma once

typedef _Float16 cc_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cc_h4 __attribute__((ext_vector_type(4)));
typedef float cc_f16acc __attribute__((ext_vector_type(16)));

#define CC_P16_TM 32        // table rows per tile of the MFMA prefix test
#define CC_P16_BIG 60000.0f  // "never" in half precision (the accumulator then cannot change sign: |p^.c^ - h^| <= 12)

// what k_prefix16 leaves for the scan that follows it (per window parity): the origin, the scale, the table's record count
struct Prefix16Hdr {
    double org[8];
    double sc;      // 2^-e
    int rows;       // rows converted
    int pad;
};

// x rounded toward -inf to a half-precision NORMAL number or zero (no subnormals: their treatment by the matrix cores is
// not relied on); |x| <= 60000
__device__ __forceinline__ _Float16 cc_rd16(double x)
{
    _Float16 h = (_Float16)(float)x;
    unsigned short b = __builtin_bit_cast(unsigned short, h);
    if ((double)h > x) {
        if (h > (_Float16)0.0f) b = (unsigned short)(b - 1u);
        else if (h < (_Float16)0.0f) b = (unsigned short)(b + 1u);
        else b = 0x8400u;  // (+-0 above a negative x: -2^-14)
        h = __builtin_bit_cast(_Float16, b);
    }
    const double a = __builtin_fabs((double)h);
    if (a < 0x1p-14) h = (x >= 0.0) ? (_Float16)0.0f : __builtin_bit_cast(_Float16, (unsigned short)0x8400u);
    return h;
}

// round to nearest, subnormal results flushed to zero (what the conversion error bound of cc_tau16 assumes at worst)
__device__ __forceinline__ _Float16 cc_rn16(double x)
{
    _Float16 h = (_Float16)(float)x;
    if (__builtin_fabs((double)h) < 0x1p-14) h = (_Float16)0.0f;
    return h;
}

// cc_tau16: the point's constant of the MFMA prefix test, in the units of the SCALED operands (coordinates x 2^-e, squared
// distances x 2^-2e).  Ts = T 2^-2e, s = |p^|^2 (exact, from the half-precision values), pm = max |p^_i| <= 1; |c^_i| <= 1.
// Derivation as cc_tau32 (cc_scan.h) with half-precision constants.  Reals: U = sum_{i<8} (p'_i - c'_i)^2 in scaled units; phase
// B's partial sum P >= smin U / 2^-2e.  Every operand is converted double -> single -> half, round to nearest, subnormal halves
// flushed at worst: |x^ - x| <= u |x| + a, u = 2^-11 (1 + 2^-9), a = 2^-14, so sqrt(U) >= sqrt(U^) - A with
// A = sqrt(8) (u (pm' + 1) + 2 a) (pm' = pm (1 + 2^-9) + a bounds the unrounded |p'_i|).  U^ = s - 2 p^.c^ + 2 H, H = |c^|^2 / 2.
// The accumulator holds D = p^.c^ - h^ - tau^ + delta: the products are exact in single precision (11 x 11 bits), the sum of
// the <= 14 terms is taken to be off by at most g (8 pm + H + |tau^|), g = 2^-18 (sixteen additions at two units in the last
// place each, the alignment shifts of a fused adder tree included - tests/hip/prefix16_check.hip measures it: < 2^-21);
// |h^ - H| <= eh = 2^-19 (two pieces: 2^-22 H + a flushed remainder, H <= 4); tau^ <= tau (both pieces rounded down).
// With a prefix of np > 8 dimensions (a second MFMA over dimensions 8 .. np - 1 accumulates into the first one's result):
// sqrt(np) for sqrt(8), np pm for 8 pm, H <= np / 2, up to 30 terms: g = 2^-17, eh = 2^-18 - the constants the code uses for any np.
// D < 0 then implies U^ > s - 2 tau - 2 g (..) - 2 eh >= L = (sqrt(Ts (1 + 2^-40) / smin) + A)^2 for
//     tau <= (s - L) / 2 - g (|tau0| + 8 pm + 4) (1 + 2^-10) - eh,
// hence U > Ts / smin and P > T.  A tau below -60000 is clamped there: the accumulator is then positive whatever the row
// (|p^.c^ - h^| <= 12), i.e. nothing is abandoned, as for T = inf.  Returns the two pieces (tau1, tau2 2^10).
__device__ __forceinline__ void cc_tau16(double Ts, double s, double pm, double inv_k, bool ok, _Float16& t1, _Float16& t2, int np = 8)
{
    t1 = (_Float16)(-CC_P16_BIG);  // never abandon
    t2 = (_Float16)0.0f;
    if (!ok || !(Ts < CC_INF) || !(pm <= 1.0)) return;
    const double smin = inv_k < 1.0 ? inv_k : 1.0;
    const double u = 0x1p-11 * (1.0 + 0x1p-9), a = 0x1p-14;
    const double pmr = pm * (1.0 + 0x1p-9) + a;
    const double A = sqrt((double)np) * (1.0 + 0x1p-20) * (u * (pmr + 1.0) + 2.0 * a);
    const double r = sqrt(Ts * (1.0 + 0x1p-40) / smin) * (1.0 + 0x1p-50) + A;
    const double L = r * r * (1.0 + 0x1p-20);
    const double tau0 = 0.5 * (s - L) - 0x1p-48 * s;
    const double tau = tau0 - 0x1p-17 * (1.0 + 0x1p-10) * (__builtin_fabs(tau0) + (double)np * pm + 0.5 * (double)np) - 0x1p-18 - 0x1p-100;
    if (!(tau > -(double)CC_P16_BIG)) return;  // (NaN included)
    t1 = cc_rd16(tau);
    const double rest = tau - (double)t1;  // >= 0, exact
    t2 = cc_rd16(rest * 1024.0);
}

// ---------------------------------------------------------------------------------
// k_prefix16: the table rows of a snapshot scan as B operands of the prefix test - per row sixteen halves (32 bytes):
// [c^_0 .. c^_7 | -h1, -h2 2^10, isP, isP 2^-10, isO, isO 2^-10, 0, 0].  One thread per row; launched on the scan's stream right
// in front of it with the same (round, mode), so both see the same window slot, row count and rows.  The header carries origin
// and scale to the scan (the scan converts its points with the same values).  A row with a coordinate beyond the scale
// (|c^_i| > 1: the bound it was derived from is read without synchronisation with the other stream's commits) is never abandoned.
// ---------------------------------------------------------------------------------
// dimensions of the prefix test at dimensionality DP: all of them up to 24 (eight in the first MFMA, the rest in the second)
template <int DP>
struct Prefix16 {
    static constexpr int NP = DP < 24 ? DP : 24;
};
template <int DP>
__global__ __launch_bounds__(256) void k_prefix16(const Ctl* __restrict__ ctl, const double* __restrict__ g_cen,
                                                  const int* __restrict__ g_kind, cc_h8* __restrict__ a16,
                                                  Prefix16Hdr* __restrict__ hdr, size_t a16_stride, int round, int mode)
{
    constexpr int NP = Prefix16<DP>::NP;
    const ScanWin win = cc_scan_window(ctl, round, mode);
    if (win.B == 0) return;
    a16 += (size_t)win.q * a16_stride;
    hdr += win.q;
    const double cmd = 2.0 * __builtin_fmax(ctl->x_absmax, __longlong_as_double((long long)ctl->cen_absmax));
    int e = 0;
    (void)frexp(cmd, &e);  // cmd = m 2^e, m in [0.5, 1): 2^e > cmd
    const double sc = ldexp(1.0, -e);
    double org[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) org[i] = (win.rows > 0) ? g_cen[i] : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) hdr->org[i] = org[i];
        hdr->sc = (cmd < CC_INF && e > -480 && e < 480) ? sc : 0.0;  // (0: no prefix test in this launch)
        hdr->rows = win.rows;
    }
    const int row = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    // (the scan reads whole tiles of 32 rows: the rows behind the last one are written as rows of neither list)
    const int rows_pad = ((win.rows + CC_P16_TM - 1) / CC_P16_TM) * CC_P16_TM + CC_P16_TM;  // (a rank's share need not start on a tile)
    if (row >= rows_pad) return;
    cc_h8 lo, hi, x1, x2;
#pragma unroll
    for (int i = 0; i < 8; ++i) { lo[i] = (_Float16)0.0f; hi[i] = (_Float16)0.0f; x1[i] = (_Float16)0.0f; x2[i] = (_Float16)0.0f; }
    if (row < win.rows) {
        const double* c = g_cen + (size_t)row * DP;
        double H = 0.0;
        bool wild = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double x = (c[i] - org[i]) * sc;
            wild = wild || !(__builtin_fabs(x) <= 1.0);
            const _Float16 xh = cc_rn16(x);
            lo[i] = xh;
            H += (double)xh * (double)xh;
        }
        // dimensions 8 .. NP - 1: the second MFMA's operand (centred on the same row 0)
#pragma unroll
        for (int i = 8; i < NP; ++i) {
            const double x = (c[i] - g_cen[i]) * sc;
            wild = wild || !(__builtin_fabs(x) <= 1.0);
            const _Float16 xh = cc_rn16(x);
            if (i < 16) x1[i - 8] = xh;
            else x2[i - 16] = xh;
            H += (double)xh * (double)xh;
        }
        H *= 0.5;
        const int kd = g_kind[row];
        const _Float16 h1 = cc_rn16(H);
        const _Float16 h2 = cc_rn16((H - (double)h1) * 1024.0);
        hi[0] = wild ? (_Float16)CC_P16_BIG : -h1;
        hi[1] = wild ? (_Float16)0.0f : -h2;
        hi[2] = (kd == CC_KIND_PCORE) ? (_Float16)1.0f : (_Float16)0.0f;
        hi[3] = (kd == CC_KIND_PCORE) ? (_Float16)0x1p-10f : (_Float16)0.0f;
        hi[4] = (kd == CC_KIND_OUTLIER) ? (_Float16)1.0f : (_Float16)0.0f;
        hi[5] = (kd == CC_KIND_OUTLIER) ? (_Float16)0x1p-10f : (_Float16)0.0f;
        if (wild) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { lo[i] = (_Float16)0.0f; x1[i] = (_Float16)0.0f; x2[i] = (_Float16)0.0f; }
        }
    }
    a16[(size_t)row * 4] = lo;
    a16[(size_t)row * 4 + 1] = hi;
    a16[(size_t)row * 4 + 2] = x1;
    a16[(size_t)row * 4 + 3] = x2;
}

// ---------------------------------------------------------------------------------
// k_scan_p3: the pruned scan of a window in one kernel (k_scan_p2's structure: a workgroup = NW waves over the same 128 window
// points, each wave a sub-range of the table rows, the points staged once in LDS for phase B) with phase A on the matrix cores.
// Per tile of 32 rows a wave loads its B operand (16 bytes per lane, coalesced, the next tile's in flight), issues four MFMAs
// (the 128 points as four A operands that stay in registers for the whole launch) and reduces the four accumulators to two words:
// the rows some point of either 64-point half keeps.  Those go on a list per wave; phase B walks it after the tiles with four rows
// in flight (k_scan_p<MASKED>'s walk).  Candidates, bounds, marks and statistics: k_scan_p2's.
// ---------------------------------------------------------------------------------
#ifndef CC_SCANP3_WGS
#define CC_SCANP3_WGS 4  // workgroups per CU the kernel is compiled for
#endif
// GENERAL (round 6): the cases the plain scan served with k_scan<FILTER, POW2> - the pdim filter of hddstream.py:317-321 on
// (pi < d) and / or k not a power of two.  Phase A does not depend on either (its bound takes min(1, 1/k) for every operand);
// phase B divides by the preference entry where k is not a power of two (the table's operand column then holds the entries
// themselves), and a pcore row enters a point's list only if the microcluster with the point added keeps pdim <= pi - evaluated
// lazily, for the rows that would enter (cc_tentative_radius: the row's CF1, CF2, W, the point's coordinates from the row-major
// copy).  A row the filter rejects for a point may still be abandoned for it by phase A and leave its bound: sound (a bound
// claims only that nothing closer was overlooked).
template <int DP, int NW, bool LISTED, bool GENERAL>
__global__ __launch_bounds__(64 * NW, CC_SCANP3_WGS) void k_scan_p3(
    Ctl* __restrict__ ctl, const double* __restrict__ Xt, const double* __restrict__ g_cen, const double* __restrict__ g_scl,
    const int* __restrict__ g_kind, const int* __restrict__ g_key, const double* __restrict__ thr, size_t thr_stride,
    Cand* __restrict__ part, int round, int mode, size_t part_stride, int shard_rank, int shard_world,
    unsigned long long* __restrict__ pstat, double guess_F, unsigned long long* __restrict__ found,
    const cc_h8* __restrict__ a16, const Prefix16Hdr* __restrict__ hdr, size_t a16_stride,
    const double* __restrict__ X, const double* __restrict__ g_cf1, const double* __restrict__ g_cf2, const double* __restrict__ g_w)
{
    static_assert(DP % 2 == 0 && DP > 8 && DP <= 64, "k_scan_p3 shapes");
    const Par par = cc_load_par(ctl);
    const bool gen_pow2 = !GENERAL || par.pow2 != 0;
    const bool gen_filter = GENERAL && par.filter != 0;
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int B = win.B;
    if (B == 0) return;
    const int j0 = (int)blockIdx.x * 128;
    if (j0 >= B) return;
    part += (size_t)win.q * part_stride;
    thr += (size_t)win.q * thr_stride;
    a16 += (size_t)win.q * a16_stride;
    hdr += win.q;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = gridDim.y;
    const int nsub = S * NW;
    const int sub = blockIdx.y * NW + wv;
    int row_lo = 0, row_hi = win.rows;
    if (shard_world > 1) cc_shard_range(win.rows, shard_world, shard_rank, 1, &row_lo, &row_hi);
    // sub-ranges are whole tiles of 32 rows (the B operands are read tile by tile)
    const int per = (((row_hi - row_lo + nsub - 1) / nsub + CC_P16_TM - 1) / CC_P16_TM) * CC_P16_TM;
    const int r0 = row_lo + sub * per;
    const int r1 = min(row_hi, r0 + per);
    const size_t n_pts = (size_t)ctl->xt_stride;

    constexpr int PTS_DOUBLES = DP * 128;
    constexpr int MERGE_BYTES = (NW > 1 ? NW - 1 : 1) * 8 * 64 * (int)sizeof(Cand);
    constexpr int PTS_BYTES = PTS_DOUBLES * 8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[PTS_BYTES > MERGE_BYTES ? PTS_BYTES : MERGE_BYTES];
    constexpr int NP = Prefix16<DP>::NP;  // dimensions of the prefix test
    __shared__ __attribute__((aligned(16))) _Float16 s_p16[128 * 24];  // the points' scaled prefixes: [point][24] (zeros beyond NP)
    __shared__ __attribute__((aligned(8))) _Float16 s_tau[128 * 4];   // [point][tauP1, tauP2 2^10, tauO1, tauO2 2^10], negated
    __shared__ double s_T[2 * 128];
    // phase B is deferred: the (row, half) pairs phase A keeps are LISTED per wave and walked afterwards with the next rows'
    // 2 d doubles already requested (k_scan_p<MASKED>'s walk: taken one by one in the middle of the tile loop every completed row
    // is a chain of dependent scalar loads, eight dimensions at a time)
    // (LISTED = false: every kept row is completed at once, its operands by scalar loads - faster while the table sits in L2 and a
    // wave keeps a handful of rows: 54 against 61 us per C2 window; the host picks the form by the table's size)
    constexpr int LCAP = LISTED ? 256 : 2 * CC_P16_TM;
    __shared__ int s_list[LISTED ? NW * LCAP : 1];
    __shared__ __attribute__((aligned(16))) double s_rowbuf[LISTED ? NW * 2 * DP : 2];
    int* const lst = s_list + (size_t)(threadIdx.x >> 6) * LCAP;
    double* const srow = s_rowbuf + (size_t)(threadIdx.x >> 6) * 2 * DP;
    double* const s_pts = reinterpret_cast<double*>(smem);
    for (int e = (int)threadIdx.x; e < PTS_DOUBLES; e += 64 * NW) {
        const int i = e >> 7, x = e & 127;
        s_pts[e] = (j0 + x < B) ? Xt[win.cursor + j0 + x + (size_t)i * n_pts] : 0.0;
    }
    const bool guessed = guess_F > 0.0;
    for (int e = (int)threadIdx.x; e < 2 * 128; e += 64 * NW) {
        const int K = e >> 7, x = e & 127;
        double T;
        if (guessed) T = (ctl->tg_ok[win.q][K] != 0) ? guess_F * ctl->tg[win.q][K] : CC_INF;
        else T = (j0 + x < B) ? thr[(size_t)(j0 + x) * 2 + K] : CC_INF;
        s_T[e] = (j0 + x < B) ? T : -CC_INF;
    }
    const double sc = hdr->sc;
    // (the control block carries 1 / k only where the kernels multiply by it; cc_tau16 needs min(1, 1 / k) - its 2^-40 covers
    // the rounding of the quotient)
    const double inv_k = gen_pow2 ? ctl->inv_k : 1.0 / par.k;
    __syncthreads();  // the points and thresholds are staged
    // the points' prefixes in half precision (four values per thread), then per (point, kind) the constant of the test
    static_assert(NW == 4, "the prologue's work split assumes 256 threads");
    {
        // (thread = (point, half of the 24 slots): twelve values each)
        const int x = (int)threadIdx.x & 127, i0 = 12 * ((int)threadIdx.x >> 7);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            cc_h4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int dim = i0 + 4 * q + i;
                // (centred on table row 0 - hdr->org holds its first eight coordinates, the others are read from the table)
                v[i] = (dim < NP) ? cc_rn16((s_pts[(dim < DP ? dim : 0) * 128 + x] - (dim < 8 ? hdr->org[dim < 8 ? dim : 0] : g_cen[dim < DP ? dim : 0])) * sc)
                                  : (_Float16)0.0f;
            }
            *reinterpret_cast<cc_h4*>(s_p16 + x * 24 + i0 + 4 * q) = v;
        }
    }
    __syncthreads();
    {
        const int x = (int)threadIdx.x & 127, K = (int)threadIdx.x >> 7;
        double s2 = 0.0, pm = 0.0;
        bool fin = true;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const double raw = (s_pts[i * 128 + x] - (i < 8 ? hdr->org[i < 8 ? i : 0] : g_cen[i])) * sc;
            fin = fin && (__builtin_fabs(raw) <= 1.0);
            const double v = (double)s_p16[x * 24 + i];
            s2 += v * v;
            pm = __builtin_fmax(pm, __builtin_fabs(v));
        }
        const double T = s_T[K * 128 + x];
        _Float16 t1, t2;
        if (T == -CC_INF) {  // a lane without a point keeps no row alive
            t1 = (_Float16)CC_P16_BIG;
            t2 = (_Float16)0.0f;
        } else cc_tau16(T * sc * sc, s2, pm, inv_k, fin && sc > 0.0, t1, t2, NP);
        s_tau[x * 4 + K * 2] = -t1;
        s_tau[x * 4 + K * 2 + 1] = -t2;
        if (K == 0 && !(fin && sc > 0.0)) {  // (a point beyond the scale: its prefix must not reach the matrix cores as inf)
#pragma unroll
            for (int i = 0; i < 24; ++i) s_p16[x * 24 + i] = (_Float16)0.0f;
        }
    }
    __syncthreads();
    // the four A operands: points 32 b + (lane & 31); lane half 0 the prefix, half 1 the constants (read again from LDS at the
    // start of every pass over tiles: they need not stay in registers while a list is walked)
    auto load_a = [&](cc_h8 (&afr)[4], cc_h8 (&afx)[4]) {
        const int r = lane & 31;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int x = 32 * b + r;
            // (the second MFMA: dimensions 8 .. 15 on lane half 0, 16 .. 23 on half 1)
            if constexpr (NP > 8) afx[b] = *reinterpret_cast<const cc_h8*>(s_p16 + x * 24 + 8 + 8 * (lane >> 5));
            if (lane < 32) afr[b] = *reinterpret_cast<const cc_h8*>(s_p16 + x * 24);
            else {
                const cc_h4 t = *reinterpret_cast<const cc_h4*>(s_tau + x * 4);
                afr[b] = cc_h8{(_Float16)1.0f, (_Float16)0x1p-10f, t[0], t[1], t[2], t[3], (_Float16)0.0f, (_Float16)0.0f};
            }
        }
    };
    bool valid[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) valid[u] = j0 + u * 64 + lane < B;
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

    // phase B for one row and one half (k_scan_p2's): the reference's four operations per term, the exact abandon test every
    // eight dimensions, then k_scan_u's best-two update
    auto complete_row = [&](auto UC, int rowg, bool is_p) {
        constexpr int u = decltype(UC)::value;
        // the row's centroid and operands: wave-uniform LDS reads (walk() staged them) / scalar loads
        const double* __restrict__ rc = LISTED ? srow : g_cen + (size_t)rowg * DP;
        const double* __restrict__ rs = LISTED ? srow + DP : g_scl + (size_t)rowg * DP;
        const double* __restrict__ px = s_pts + u * 64 + lane;
        double acc = 0.0;
        bool gone = false;
        cc_static_for<(DP + 7) / 8>([&](auto CC) {
            constexpr int lo = 8 * decltype(CC)::value, hi = (lo + 8) < DP ? (lo + 8) : DP;
            if (gone) return;
            // (four dimensions' operands at a time: 24 registers instead of 48)
            cc_static_for<(hi - lo + 3) / 4>([&](auto QC) {
                constexpr int qlo = lo + 4 * decltype(QC)::value, qhi = (qlo + 4) < hi ? (qlo + 4) : hi;
                double c[qhi - qlo], scl[qhi - qlo], pv[qhi - qlo];
#pragma unroll
                for (int i = 0; i < qhi - qlo; ++i) {
                    c[i] = rc[qlo + i];
                    scl[i] = rs[qlo + i];
                    pv[i] = px[(qlo + i) * 128];
                }
#pragma unroll
                for (int i = 0; i < qhi - qlo; ++i) {
                    double x = pv[i] - c[i];               // mc_functions.py:37
                    x = x * x;                             // :38
                    if constexpr (GENERAL) x = gen_pow2 ? x * scl[i] : x / scl[i];  // :39
                    else x = x * scl[i];                   // :39 (the divisor is a power of two)
                    acc = (qlo + i == 0) ? x : acc + x;    // :41
                }
            });
            // (no exact test inside the dimensions phase A has already tested in half precision: a row that passed there passes
            // here but for rounding, and without the test the row's operands are one round trip instead of one per eight dimensions)
            if constexpr (hi < DP && hi >= NP) {
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
            if (GENERAL && K == 0 && gen_filter) {
                if (ins) {
                    // hddstream.py:317-321: pdim of the MC *with the point added* must be <= pi
                    int ne1 = 0;
                    cc_tentative_radius(g_cf1 + (size_t)rowg * DP, g_cf2 + (size_t)rowg * DP, g_w[rowg],
                                        X + (size_t)(win.cursor + j0 + u * 64 + lane) * DP, DP, par, nullptr, &ne1);
                    if (ne1 > par.pi) ins = false;
                }
                first = first && ins;
                d1 = first ? d0 : (ins ? a : d1);
                d0 = first ? a : d0;
            } else {
                d1 = cc_vmin(d1, cc_vmax(d0, a));
                d0 = cc_vmin(d0, a);
            }
            s1 = first ? s0 : (ins ? rowg : s1);
            s0 = first ? rowg : s0;
        };
        if (is_p) update(std::integral_constant<int, 0>{});
        else update(std::integral_constant<int, 1>{});
    };

    // the list's entries: row | half << 30 | is-pcore << 29
    constexpr int RL = (2 * DP + 63) / 64;  // loads per lane for one row
#ifndef CC_P3_PF
#define CC_P3_PF 4
#endif
    constexpr int PF = CC_P3_PF;            // rows in flight
    auto fetch_row = [&](int ent, double (&rv)[RL]) {
        const int rowg = ent & 0x1FFFFFFF;
#pragma unroll
        for (int q = 0; q < RL; ++q) {
            const int e = lane + q * 64;
            rv[q] = (e < DP) ? g_cen[(size_t)rowg * DP + e] : ((e < 2 * DP) ? g_scl[(size_t)rowg * DP + (e - DP)] : 0.0);
        }
    };
    auto walk = [&](int n) {
        if (n <= 0) return;
        double rv[PF][RL];
        CC_WAVE_SYNC();  // (the list is written)
#pragma unroll
        for (int k = 0; k < PF; ++k)
            if (k < n) fetch_row(__builtin_amdgcn_readfirstlane(lst[k]), rv[k]);
        for (int base = 0; base < n; base += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int e = base + k;
                if (e >= n) break;
                const int ent = __builtin_amdgcn_readfirstlane(lst[e]);
                CC_WAVE_SYNC();  // (the previous row's reads are done)
#pragma unroll
                for (int q = 0; q < RL; ++q) {
                    const int x = lane + q * 64;
                    if (x < 2 * DP) srow[x] = rv[k][q];
                }
                CC_WAVE_SYNC();
                if (e + PF < n) fetch_row(__builtin_amdgcn_readfirstlane(lst[e + PF]), rv[k]);
                const int rowg = ent & 0x1FFFFFFF;
                const bool is_p = ((ent >> 29) & 1) != 0;
                if ((ent >> 30) & 1) complete_row(std::integral_constant<int, 1>{}, rowg, is_p);
                else complete_row(std::integral_constant<int, 0>{}, rowg, is_p);
            }
        }
        CC_WAVE_SYNC();  // (the list may be rewritten)
    };
    const bool test_on = sc > 0.0;  // (no usable scale: every row goes to phase B)
    auto load_b = [&](int rt) -> cc_h8 { return a16[(size_t)(rt + (lane & 31)) * 4 + (lane >> 5)]; };
    auto load_bx = [&](int rt) -> cc_h8 { return a16[(size_t)(rt + (lane & 31)) * 4 + 2 + (lane >> 5)]; };
    // passes: tiles until the list could overflow (two halves x 32 rows per tile), then the walk; one pass unless most rows stay
    int rt = r0;
    while (rt < r1) {
    cc_h8 afr[4], afx[4];
    load_a(afr, afx);
    cc_h8 bfr = load_b(rt), bfx;
    if constexpr (NP > 8) bfx = load_bx(rt);
    int kdl = (lane < min(CC_P16_TM, r1 - rt)) ? g_kind[rt + lane] : CC_KIND_DEAD;
    int n_list = 0;
    for (; rt < r1 && (!LISTED || n_list + 2 * CC_P16_TM <= LCAP); rt += CC_P16_TM) {
        const int tm = __builtin_amdgcn_readfirstlane(min(CC_P16_TM, r1 - rt));
        const cc_h8 bcur = bfr, bxcur = bfx;
        const unsigned pmask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_PCORE);
        const unsigned omask = (unsigned)__builtin_amdgcn_ballot_w64(kdl == CC_KIND_OUTLIER);
        if (rt + CC_P16_TM < r1) {
            bfr = load_b(rt + CC_P16_TM);
            if constexpr (NP > 8) bfx = load_bx(rt + CC_P16_TM);
            kdl = (lane < min(CC_P16_TM, r1 - rt - CC_P16_TM)) ? g_kind[rt + CC_P16_TM + lane] : CC_KIND_DEAD;
        }
        n_rows += 2 * tm;
        const unsigned full = (tm >= 32) ? 0xFFFFFFFFu : ((1u << tm) - 1u);
        const unsigned listed = (pmask | omask) & full;
        unsigned surv[2] = {listed, listed};
        if (test_on) {
            // ---- phase A: D = p^.c^ - h^ - tau; a row stays for a half unless D < 0 for all its 64 points ----
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                cc_f16acc z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.0f;
                cc_f16acc d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[2 * u], bcur, z, 0, 0, 0);
                cc_f16acc d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[2 * u + 1], bcur, z, 0, 0, 0);
                if constexpr (NP > 8) {
                    d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afx[2 * u], bxcur, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afx[2 * u + 1], bxcur, d1, 0, 0, 0);
                }
                // "some D >= 0" on the bit patterns: a float with its sign bit clear is a non-negative int (v_max3_i32 needs no
                // canonicalisation of its inputs; -0 cannot arise from a +0 accumulator, and cc_tau16's slack covers D = 0)
                typedef int cc_i16v __attribute__((ext_vector_type(16)));
                const cc_i16v i0 = __builtin_bit_cast(cc_i16v, d0), i1 = __builtin_bit_cast(cc_i16v, d1);
                int m = max(i0[0], i1[0]);
#pragma unroll
                for (int i = 1; i < 16; ++i) m = max(max(m, i0[i]), i1[i]);
                const unsigned long long keep = __builtin_amdgcn_ballot_w64(m >= 0);
                surv[u] = ((unsigned)keep | (unsigned)(keep >> 32)) & listed;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if ((~surv[u] & pmask & full) != 0u) dflag |= 1u << (2 * u);
            if ((~surv[u] & omask & full) != 0u) dflag |= 2u << (2 * u);
        }
        // ---- the rows that stayed go on the wave's list, half by half ----
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned sv = surv[u];
            const int cnt = __builtin_popcount(sv);
            if (cnt == 0) continue;
            if constexpr (!LISTED) {
                unsigned m2 = sv;
                while (m2 != 0u) {
                    const int m = __builtin_ctz(m2);
                    m2 &= m2 - 1u;
                    if (u == 0) complete_row(std::integral_constant<int, 0>{}, rt + m, ((pmask >> m) & 1u) != 0u);
                    else complete_row(std::integral_constant<int, 1>{}, rt + m, ((pmask >> m) & 1u) != 0u);
                }
                continue;
            }
            if (lane < CC_P16_TM && ((sv >> lane) & 1u) != 0u)
                lst[n_list + __builtin_popcount(sv & ((1u << lane) - 1u))] = (rt + lane) | (u << 30) | ((int)((pmask >> lane) & 1u) << 29);
            n_list += cnt;
        }
    }
    if constexpr (LISTED) walk(n_list);
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
// k_seed16 (round 6): the seeds of a pruned scan from the matrix cores - per window point and kind the table row with the
// smallest APPROXIMATE squared distance over the prefix test's dimensions (all of them up to 24), in the format k_seed leaves
// (one SeedCand per point, workgroup sub-range and kind; k_seed_merge evaluates the best three exactly).  k_seed scores eight
// dimensions: enough to find a point's own microcluster, not to rank rows that are all far from it - the pcore list of a point
// whose own microcluster is still an outlier microcluster, every list while the table fills.  With seeds that ARE the nearest
// rows the threshold can be the exact distance of the second nearest (k_seed_merge, F <= 0): the pruned chain then returns the
// exact two best per kind, which is all the plain scan returns, at a fraction of its arithmetic - in the start-up windows too.
// Roles swapped against k_scan_p3: A = 32 table rows (k_prefix16's records as they are), B = 32 points, D[row][point]: a lane
// holds ONE point's scores against 16 rows per MFMA and keeps a running maximum with its row; the slots that carry tau in the
// scan carry + 128 for the rows of the wanted kind here (|p^.c^ - h^| <= 36: the others never win), one pass per kind on top of
// the shared product over dimensions 8 .. 23.
// ---------------------------------------------------------------------------------
template <int DP, int NW>
__global__ __launch_bounds__(64 * NW, 4) void k_seed16(const Ctl* __restrict__ ctl, const double* __restrict__ Xt,
                                                     const double* __restrict__ g_cen, SeedCand* __restrict__ spart, int round, int mode,
                                                     size_t spart_stride, const cc_h8* __restrict__ a16,
                                                     const Prefix16Hdr* __restrict__ hdr, size_t a16_stride)
{
    static_assert(NW == 4 && DP % 2 == 0 && DP > 8 && DP <= 64, "k_seed16 shapes");
    constexpr int NP = Prefix16<DP>::NP;
    const ScanWin win = cc_scan_window(ctl, round, mode);
    const int B = win.B;
    if (B == 0) return;
    const int j0 = (int)blockIdx.x * 128;
    if (j0 >= B) return;
    spart += (size_t)win.q * spart_stride;
    a16 += (size_t)win.q * a16_stride;
    hdr += win.q;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = gridDim.y;
    const int nsub = S * NW;
    const int sub = blockIdx.y * NW + wv;
    const int per = (((win.rows + nsub - 1) / nsub + CC_P16_TM - 1) / CC_P16_TM) * CC_P16_TM;
    const int r0 = sub * per;
    const int r1 = min(win.rows, r0 + per);
    const size_t n_pts = (size_t)ctl->xt_stride;
    const double sc = hdr->sc;
    __shared__ __attribute__((aligned(16))) _Float16 s_p16[128 * 24];
    {
        const int x = (int)threadIdx.x & 127, i0 = 12 * ((int)threadIdx.x >> 7);
        const bool pv = j0 + x < B;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            cc_h4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int dim = i0 + 4 * q + i;
                double raw = 0.0;
                if (dim < NP && pv) raw = (Xt[win.cursor + j0 + x + (size_t)dim * n_pts] - (dim < 8 ? hdr->org[dim < 8 ? dim : 0] : g_cen[dim < DP ? dim : 0])) * sc;
                v[i] = (__builtin_fabs(raw) <= 1.0) ? cc_rn16(raw) : (_Float16)0.0f;  // (a heuristic: a point beyond the scale scores as the origin)
            }
            *reinterpret_cast<cc_h4*>(s_p16 + x * 24 + i0 + 4 * q) = v;
        }
    }
    __syncthreads();
    const int r = lane & 31, hh = lane >> 5;
    cc_h8 bb[4], bx[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int x = 32 * b + r;
        bx[b] = *reinterpret_cast<const cc_h8*>(s_p16 + x * 24 + 8 + 8 * hh);
        if (hh == 0) bb[b] = *reinterpret_cast<const cc_h8*>(s_p16 + x * 24);
        else bb[b] = cc_h8{(_Float16)1.0f, (_Float16)0x1p-10f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
    }
    float best[4][2];
    int idx[4][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int K = 0; K < 2; ++K) { best[b][K] = 64.0f; idx[b][K] = -1; }  // (a row of the wanted kind scores above 128 - 36)
    auto load_a = [&](int rt, int part) -> cc_h8 { return a16[(size_t)(rt + r) * 4 + part + hh]; };
    cc_h8 an, axn;
    if (r0 < r1 && sc > 0.0) { an = load_a(r0, 0); axn = load_a(r0, 2); }
    for (int rt = r0; rt < r1 && sc > 0.0; rt += CC_P16_TM) {
        const cc_h8 ac = an, axc = axn;
        if (rt + CC_P16_TM < r1) { an = load_a(rt + CC_P16_TM, 0); axn = load_a(rt + CC_P16_TM, 2); }
        const int tm = min(CC_P16_TM, r1 - rt);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            cc_f16acc z;
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = 0.0f;
            cc_f16acc xacc = z;
            if constexpr (NP > 8) xacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(axc, bx[b], z, 0, 0, 0);
#pragma unroll
            for (int K = 0; K < 2; ++K) {
                cc_h8 bk = bb[b];
                // (slot 10 pairs with the row's is-pcore flag, slot 12 with its is-outlier flag)
                if (hh == 1) bk[K == 0 ? 2 : 4] = (_Float16)128.0f;
                const cc_f16acc d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ac, bk, xacc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = (i & 3) + 8 * (i >> 2) + 4 * hh;
                    const bool gt = d[i] > best[b][K] && m < tm;  // strict: the first row in scan order keeps a tie (deterministic)
                    best[b][K] = gt ? d[i] : best[b][K];
                    idx[b][K] = gt ? rt + m : idx[b][K];
                }
            }
        }
    }
    // the two lane halves hold disjoint rows of the same points: the better one (ties: the lower row, as a scan in row order would)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int K = 0; K < 2; ++K) {
            const float ob = __shfl_xor(best[b][K], 32);
            const int oi = __shfl_xor(idx[b][K], 32);
            const bool take = oi >= 0 && (idx[b][K] < 0 || ob > best[b][K] || (ob == best[b][K] && oi < idx[b][K]));
            best[b][K] = take ? ob : best[b][K];
            idx[b][K] = take ? oi : idx[b][K];
        }
    // the waves' winners through LDS; wave 0 writes: lanes 0 .. 31 blocks 0, 1 (b = 0: point r, b = 1: point 32 + r), lanes 32 .. 63 blocks 2, 3
    __shared__ float s_b[3 * 8 * 64];
    __shared__ int s_i[3 * 8 * 64];
    if (wv > 0) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int K = 0; K < 2; ++K) {
                s_b[((wv - 1) * 8 + b * 2 + K) * 64 + lane] = best[b][K];
                s_i[((wv - 1) * 8 + b * 2 + K) * 64 + lane] = idx[b][K];
            }
    }
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int K = 0; K < 2; ++K)
#pragma unroll
            for (int w = 0; w < NW - 1; ++w) {
                const float ob = s_b[(w * 8 + b * 2 + K) * 64 + lane];
                const int oi = s_i[(w * 8 + b * 2 + K) * 64 + lane];
                const bool take = oi >= 0 && (idx[b][K] < 0 || ob > best[b][K] || (ob == best[b][K] && oi < idx[b][K]));
                best[b][K] = take ? ob : best[b][K];
                idx[b][K] = take ? oi : idx[b][K];
            }
        // (both lane halves hold block b's 32 points: half 0 writes blocks 0 and 1, half 1 blocks 2 and 3)
        if ((b >> 1) != hh) continue;
        const int x = j0 + 32 * b + r;
        if (x >= B) continue;
        SeedCand* o = spart + ((size_t)x * S + blockIdx.y) * 2;
        // (k_seed_merge ranks by `part`, smaller is nearer: the score's negative)
        o[0] = SeedCand{idx[b][0] >= 0 ? 128.0f - best[b][0] : __builtin_inff(), idx[b][0]};
        o[1] = SeedCand{idx[b][1] >= 0 ? 128.0f - best[b][1] : __builtin_inff(), idx[b][1]};
    }
}
