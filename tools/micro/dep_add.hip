// Micro-benchmark: cycles per dependent v_add_f64 of a lone wave, with and without LDS traffic beside it (the running
// sums of k_chain_long).  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/dep_add.hip -o build_variants/dep_add   (build_variants/ is
// git-ignored and travels to the GPU box; gpurun_out/ does not)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, long long* cyc, int n, int active_waves, double* gout)
{
    __shared__ double s[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += blockDim.x) s[i] = 1.0 + i * 1e-9;
    for (int i = tid; i < 1024; i += blockDim.x) reinterpret_cast<int*>(s + 4096)[i] = (i * 7919) % 32768;
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    long long t0 = 0, t1 = 0;
    double c = out[tid];
    if (wave < active_waves && lane < 14) {
        double* row = s + (size_t)(wave * 14 + lane) * 257 % 4096;
        t0 = clock64();
        if (MODE == 0) {  // adds only
            for (int k = 0; k < n; k += 16) {
#pragma unroll
                for (int u = 0; u < 16; ++u) c = c + 1.25;
            }
        } else if (MODE == 1) {  // read 16, add 16, write 16
            for (int k = 0; k < n; k += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = row[(k & 127) + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) { c = c + v[u]; v[u] = c; }
#pragma unroll
                for (int u = 0; u < 16; ++u) row[(k & 127) + u] = v[u];
            }
        } else if (MODE == 3) {  // read 16, add 16, 16 global stores (row of the step's member: index from LDS, one read per 16 steps)
            const int* q = reinterpret_cast<const int*>(s + 4096);
            for (int k = 0; k < n; k += 16) {
                double v[16];
                const int mine = q[(k & 1023) + (lane & 15)];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = row[(k & 127) + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    c = c + v[u];
                    const int m = __builtin_amdgcn_readlane(mine, u);
                    gout[(size_t)m * 14 + lane] = c;
                }
            }
        } else if (MODE == 4) {  // as mode 1, the reads of the next sixteen issued before the adds of these (two register sets)
            double a[16], b[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] = row[u];
            for (int k = 0; k < n; k += 32) {
#pragma unroll
                for (int u = 0; u < 16; ++u) b[u] = row[((k + 16) & 127) + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) { c = c + a[u]; a[u] = c; }
#pragma unroll
                for (int u = 0; u < 16; ++u) row[(k & 127) + u] = a[u];
#pragma unroll
                for (int u = 0; u < 16; ++u) a[u] = row[((k + 32) & 127) + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) { c = c + b[u]; b[u] = c; }
#pragma unroll
                for (int u = 0; u < 16; ++u) row[((k + 16) & 127) + u] = b[u];
            }
        } else {  // read 16, add 16 (no write)
            for (int k = 0; k < n; k += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = row[(k & 127) + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) { c = c + v[u]; }
            }
        }
        t1 = clock64();
    }
    __syncthreads();
    out[tid] = c;
    if (lane == 0 && wave < active_waves) cyc[wave] = t1 - t0;
}

int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 16 * 8);
    hipMemset(out, 0, 1024 * 8);
    double* gout; hipMalloc(&gout, (size_t)32768 * 14 * 8 + 1024);
    const int n = 4096;
    for (int threads : {64, 256, 1024})
        for (int aw : {1, 3}) {
            if (aw * 64 > threads) continue;
            long long h[16];
            for (int mode = 0; mode < 5; ++mode) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, cyc, n, aw, gout);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, cyc, n, aw, gout);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, out, cyc, n, aw, gout);
                    if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, out, cyc, n, aw, gout);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, out, cyc, n, aw, gout);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
                printf("threads %4d active waves %d mode %d: %.1f cycles per step (wave 0)\n", threads, aw, mode, (double)h[0] / n);
            }
        }
    return 0;
}
