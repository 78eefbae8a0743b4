// Device check of chronoclust_amd/csrc/cc_scan16.h: the MFMA prefix test of the pruned snapshot scan.
//   (1) operand layout of v_mfma_f32_32x32x16_f16 as the kernels assume it (asymmetric integer data);
//   (2) the accumulator against the exact sum of the half-precision products, relative to the sum of their magnitudes
//       (cc_tau16 assumes 2^-18; the claim checked here: < 2^-21);
//   (3) the statement the scan relies on: D < 0 implies that the exact partial sum over the first eight dimensions, scaled by
//       min(1, 1/k), exceeds the point's threshold - over random points and rows, thresholds from far below to far above the
//       typical distance, clustered pairs whose distance sits at the threshold, coordinates of mixed magnitudes.
// Prints: "pairs <n> abandoned <a> violations <v> max_rel_err <e> layout_errors <l>"; exit status 0 iff v == 0, l == 0, e < 2^-21.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <type_traits>
#include <vector>
#include "../../include/chronoclust_hip.h"
#include "../../chronoclust_amd/csrc/cc_common.h"
#include "../../chronoclust_amd/csrc/cc_online.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// one wave: D = A (32 x 16) B (16 x 32), operands given row-major as halves; D written row-major [32][32]
__global__ void k_mm(const _Float16* __restrict__ A, const _Float16* __restrict__ Bm, float* __restrict__ D)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    cc_h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[r * 16 + 8 * h + j];      // A[row r][k = 8 h + j]
        b[j] = Bm[(8 * h + j) * 32 + r];   // B[k = 8 h + j][col r]
    }
    cc_f16acc z;
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    const cc_f16acc d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = d[i];
}

// the scan's own construction: 32 points (A) x 32 rows (B) from doubles already centred (org = 0), scale sc, thresholds T[2]
__global__ void k_test(const double* __restrict__ P, const double* __restrict__ C, const int* __restrict__ kind, const double* __restrict__ T,
                       double sc, double inv_k, float* __restrict__ D)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    cc_h8 a, b;
    {   // point r
        _Float16 ph[8];
        double s2 = 0.0, pm = 0.0;
        bool fin = true;
        for (int i = 0; i < 8; ++i) {
            const double raw = P[r * 8 + i] * sc;
            fin = fin && fabs(raw) <= 1.0;
            ph[i] = cc_rn16(raw);
            s2 += (double)ph[i] * (double)ph[i];
            pm = fmax(pm, fabs((double)ph[i]));
        }
        _Float16 t[4];
        for (int K = 0; K < 2; ++K) cc_tau16(T[r * 2 + K] * sc * sc, s2, pm, inv_k, fin, t[2 * K], t[2 * K + 1]);
        if (h == 0) for (int i = 0; i < 8; ++i) a[i] = fin ? ph[i] : (_Float16)0.0f;
        else a = cc_h8{(_Float16)1.0f, (_Float16)0x1p-10f, -t[0], -t[1], -t[2], -t[3], (_Float16)0.0f, (_Float16)0.0f};
    }
    {   // row r (k_prefix16's record)
        _Float16 ch[8];
        double H = 0.0;
        bool wild = false;
        for (int i = 0; i < 8; ++i) {
            const double x = C[r * 8 + i] * sc;
            wild = wild || !(fabs(x) <= 1.0);
            ch[i] = cc_rn16(x);
            H += (double)ch[i] * (double)ch[i];
        }
        H *= 0.5;
        const _Float16 h1 = cc_rn16(H), h2 = cc_rn16((H - (double)h1) * 1024.0);
        const int kd = kind[r];
        if (h == 0) for (int i = 0; i < 8; ++i) b[i] = wild ? (_Float16)0.0f : ch[i];
        else b = cc_h8{wild ? (_Float16)CC_P16_BIG : -h1, wild ? (_Float16)0.0f : -h2, (_Float16)(kd == 0 ? 1.0f : 0.0f), (_Float16)(kd == 0 ? 0x1p-10f : 0.0f),
                       (_Float16)(kd == 1 ? 1.0f : 0.0f), (_Float16)(kd == 1 ? 0x1p-10f : 0.0f), (_Float16)0.0f, (_Float16)0.0f};
    }
    cc_f16acc z;
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    const cc_f16acc d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = d[i];  // D[point][row]
}

int main()
{
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    long long layout_errors = 0, pairs = 0, abandoned = 0, violations = 0;
    double max_rel = 0.0;
    _Float16 *dA, *dB;
    float* dD;
    CK(hipMalloc(&dA, 32 * 16 * 2)); CK(hipMalloc(&dB, 16 * 32 * 2)); CK(hipMalloc(&dD, 32 * 32 * 4));
    std::vector<_Float16> A(32 * 16), Bm(16 * 32);
    std::vector<float> D(32 * 32);
    // (1) layout: small integers, asymmetric
    for (int m = 0; m < 32; ++m) for (int k = 0; k < 16; ++k) A[m * 16 + k] = (_Float16)(float)((m * 3 + k * 5) % 17 - 8);
    for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) Bm[k * 32 + n] = (_Float16)(float)((k * 7 + n * 11) % 13 - 6);
    CK(hipMemcpy(dA, A.data(), 32 * 16 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bm.data(), 16 * 32 * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mm, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 32 * 32 * 4, hipMemcpyDeviceToHost));
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        double e = 0.0;
        for (int k = 0; k < 16; ++k) e += (double)(float)A[m * 16 + k] * (double)(float)Bm[k * 32 + n];
        if ((double)D[m * 32 + n] != e) ++layout_errors;
    }
    // (2) accumulation error: random halves of mixed magnitudes, cancelling sums
    for (int rep = 0; rep < 400; ++rep) {
        for (int i = 0; i < 32 * 16; ++i) { const double s = ldexp(1.0, -(int)(rng() % 12)); A[i] = (_Float16)(float)(U(rng) * s); }
        for (int i = 0; i < 16 * 32; ++i) { const double s = ldexp(1.0, -(int)(rng() % 12)); Bm[i] = (_Float16)(float)(U(rng) * s * (rep % 3 == 0 ? 4000.0 : 1.0)); }
        CK(hipMemcpy(dA, A.data(), 32 * 16 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bm.data(), 16 * 32 * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mm, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        CK(hipMemcpy(D.data(), dD, 32 * 32 * 4, hipMemcpyDeviceToHost));
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
            long double e = 0.0L, mag = 0.0L;
            for (int k = 0; k < 16; ++k) { const long double t = (long double)(float)A[m * 16 + k] * (long double)(float)Bm[k * 32 + n]; e += t; mag += fabsl(t); }
            if (mag > 0.0L) max_rel = fmax(max_rel, (double)(fabsl((long double)D[m * 32 + n] - e) / mag));
        }
    }
    // (3) the implication
    double *dP, *dC, *dT;
    int* dK;
    CK(hipMalloc(&dP, 32 * 8 * 8)); CK(hipMalloc(&dC, 32 * 8 * 8)); CK(hipMalloc(&dT, 32 * 2 * 8)); CK(hipMalloc(&dK, 32 * 4));
    std::vector<double> P(32 * 8), C(32 * 8), T(32 * 2);
    std::vector<int> K(32);
    for (int rep = 0; rep < 6000; ++rep) {
        const double inv_k = (rep % 4 == 0) ? 1.0 : ((rep % 4 == 1) ? 0.25 : ((rep % 4 == 2) ? 0x1p-6 : 4.0));
        const double smin = inv_k < 1.0 ? inv_k : 1.0;
        const double ext = ldexp(1.0, (int)(rng() % 40) - 20);           // the data's extent
        const double spread = ldexp(1.0, -(int)(rng() % 14));            // cluster size relative to the extent
        int e = 0;
        (void)frexp(2.0 * ext, &e);
        const double sc = ldexp(1.0, -e);
        for (int r = 0; r < 32; ++r) {
            K[r] = (int)(rng() % 3);  // 0 pcore, 1 outlier, 2 neither
            for (int i = 0; i < 8; ++i) C[r * 8 + i] = U(rng) * ext;
        }
        for (int m = 0; m < 32; ++m) {
            const int near = (int)(rng() % 32);
            for (int i = 0; i < 8; ++i) P[m * 8 + i] = (rep % 2) ? C[near * 8 + i] + U(rng) * ext * spread : U(rng) * ext;
            for (int k = 0; k < 2; ++k) {
                // thresholds around the distance to `near` (the interesting regime), or anywhere
                double u0 = 0.0;
                for (int i = 0; i < 8; ++i) u0 += (P[m * 8 + i] - C[near * 8 + i]) * (P[m * 8 + i] - C[near * 8 + i]);
                const double f = (rng() % 3 == 0) ? ldexp(1.0, (int)(rng() % 30) - 15) : 1.0 + U(rng) * ldexp(1.0, -(int)(rng() % 12));
                T[m * 2 + k] = smin * u0 * f;
            }
        }
        CK(hipMemcpy(dP, P.data(), 32 * 8 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), 32 * 8 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dT, T.data(), 32 * 2 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dK, K.data(), 32 * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dP, dC, dK, dT, sc, inv_k, dD);
        CK(hipMemcpy(D.data(), dD, 32 * 32 * 4, hipMemcpyDeviceToHost));
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
            if (K[n] == 2) continue;
            ++pairs;
            if (!(D[m * 32 + n] < 0.0f)) continue;
            ++abandoned;
            // the smallest partial sum phase B could compute for this row: every term scaled by min(1, 1/k)
            long double u = 0.0L;
            for (int i = 0; i < 8; ++i) { const long double x = (long double)P[m * 8 + i] - (long double)C[n * 8 + i]; u += x * x; }
            if (!((long double)smin * u > (long double)T[m * 2 + K[n]])) ++violations;
        }
    }
    printf("pairs %lld abandoned %lld violations %lld max_rel_err %.3g layout_errors %lld\n", pairs, abandoned, violations, max_rel, layout_errors);
    return (violations == 0 && layout_errors == 0 && max_rel < 0x1p-21) ? 0 : 1;
}
