// chronoclust_amd/csrc: kernels over the uploaded points - transposed copy, finiteness check, MinMax scaler.  (included by cc_online.h; one translation unit, cc_api.hip)
#pragma once

// dimension-major copy of the points for the scan's coalesced loads: xt[i * n + r] = x[r * d + i]
__global__ void k_transpose_points(const double* __restrict__ x, double* __restrict__ xt, long long n, int d)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * d) return;
    const long long r = e / d;
    const int i = (int)(e - r * d);
    xt[(size_t)i * n + r] = x[e];
}

// NaN / Inf check of the uploaded points (cc_points_upload)
// bad[0]: 1 if any value is NaN or Inf; bad[2..3] (one 64-bit word): bits of the largest |value| (Ctl::x_absmax: the
// single-precision prefix of the pruned scans needs a bound on the magnitudes it converts)
__global__ __launch_bounds__(256) void k_check_finite(const double* __restrict__ x, long long n, int* __restrict__ bad)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    int b = 0;
    double m = 0.0;
    for (; i < n; i += stride) {
        const double v = x[i];
        b |= !(v - v == 0.0);
        const double a = __builtin_fabs(v);
        m = a > m ? a : m;  // (NaN never enters)
    }
    if (b) atomicOr(bad, 1);
    __shared__ unsigned long long s_m;
    if (threadIdx.x == 0) s_m = 0ull;
    __syncthreads();
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(&s_m, (unsigned long long)__double_as_longlong(m));
    __syncthreads();
    if (threadIdx.x == 0 && s_m != 0ull) atomicMax(reinterpret_cast<unsigned long long*>(bad + 2), s_m);
}

// ---------------------------------------------------------------------------------
// MinMax scaling on the device (scaling/scaler.py:27-47 = scikit-learn's MinMaxScaler restated, see
// chronoclust_amd/scaling/scaler.py): HBM-bound elementwise passes and one column reduction.
// ---------------------------------------------------------------------------------

// per-column minimum / maximum ignoring NaN (np.nanmin / np.nanmax): one workgroup per (column, row chunk),
// coalescing is across the columns of a row (consecutive threads read consecutive doubles), partials in part[2][chunks][d]
__global__ __launch_bounds__(256) void k_col_minmax(const double* __restrict__ x, long long n, int d,
                                                    double* __restrict__ part, int chunks)
{
    // thread t handles column t % d of rows t / d, t / d + rows_per_pass, ...
    const int rows_per_pass = 256 / d > 0 ? 256 / d : 1;
    const int col = (d <= 256) ? (int)(threadIdx.x % d) : 0;
    const int rsub = (int)(threadIdx.x / d);
    const long long per = (n + chunks - 1) / chunks;
    const long long r0 = (long long)blockIdx.x * per, r1 = (r0 + per < n) ? r0 + per : n;
    double mn = CC_INF, mx = -CC_INF;
    if (rsub < rows_per_pass && d <= 256)
        for (long long r = r0 + rsub; r < r1; r += rows_per_pass) {
            const double v = x[r * d + col];
            mn = __builtin_fmin(mn, v);  // fmin / fmax return the non-NaN operand
            mx = __builtin_fmax(mx, v);
        }
    __shared__ double smn[256], smx[256];
    smn[threadIdx.x] = mn;
    smx[threadIdx.x] = mx;
    __syncthreads();
    if ((int)threadIdx.x < d && d <= 256) {
        for (int q = 1; q < rows_per_pass; ++q) {
            mn = __builtin_fmin(mn, smn[q * d + threadIdx.x]);
            mx = __builtin_fmax(mx, smx[q * d + threadIdx.x]);
        }
        part[(size_t)blockIdx.x * d + threadIdx.x] = mn;
        part[(size_t)(chunks + blockIdx.x) * d + threadIdx.x] = mx;
    }
}

// MinMaxScaler.transform: X * scale_ + min_ (two roundings, in place); inverse: (X - min_) / scale_
__global__ void k_scale_points(double* __restrict__ x, long long tot, int d, const double* __restrict__ scale,
                               const double* __restrict__ mn)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tot) return;
    const int c = (int)(e % d);
    double v = x[e];
    v = v * scale[c];
    v = v + mn[c];
    x[e] = v;
}

__global__ void k_unscale_points(const double* __restrict__ x, double* __restrict__ out, long long tot, int d,
                                 const double* __restrict__ scale, const double* __restrict__ mn)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tot) return;
    const int c = (int)(e % d);
    double v = x[e];
    v = v - mn[c];
    v = v / scale[c];
    out[e] = v;
}
