// Test program (built and run by tests/test_div_exact.py on the GPU box): cc_div_prepare / cc_div_apply of
// chronoclust_amd/csrc/cc_div.h against the compiler's IEEE division, bit for bit, for every operand pair the callers'
// guards (cc_div_den_ok / cc_div_num_ok) let through - random operands across the admitted exponent ranges, edge cases
// of both ranges, weights as the online phase sees them (integers, decayed weights) with CF sums of [0, 1] data.
// Prints "pairs <n> guarded <g> mismatches <m>"; exit status 1 if m != 0.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../chronoclust_amd/csrc/cc_div.h"

__global__ void k_check(const double* x, const double* y, int n, unsigned long long* guarded, unsigned long long* bad,
                        double* first_bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i], b = y[i];
    if (!cc_div_den_ok(b) || !cc_div_num_ok(a)) return;
    atomicAdd(guarded, 1ull);
    const double r = cc_div_prepare(b);
    const double fast = cc_div_apply(a, b, r);
    const double ref = a / b;
    if (__double_as_longlong(fast) != __double_as_longlong(ref)) {
        if (atomicAdd(bad, 1ull) == 0ull) { first_bad[0] = a; first_bad[1] = b; first_bad[2] = fast; first_bad[3] = ref; }
    }
}

static double from_bits(uint64_t u)
{
    double d;
    memcpy(&d, &u, 8);
    return d;
}

int main()
{
    std::mt19937_64 rng(12345);
    std::vector<double> xs, ys;
    auto push = [&](double a, double b) { xs.push_back(a); ys.push_back(b); };
    // random mantissas, exponents across (and a little beyond) the admitted ranges
    for (int i = 0; i < 6'000'000; ++i) {
        const uint64_t mx = rng() & ((1ull << 52) - 1), my = rng() & ((1ull << 52) - 1);
        const int ex = (int)(rng() % 1640) - 920, ey = (int)(rng() % 64) - 2;
        const uint64_t sx = (rng() & 1ull) << 63;
        push(from_bits(sx | ((uint64_t)(ex + 1023) << 52) | mx), from_bits(((uint64_t)(ey + 1023) << 52) | my));
    }
    // the online phase's operands: weights 1 .. 2^24 (integers and decayed ones), CF sums of data in [0, 1]
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    for (int i = 0; i < 6'000'000; ++i) {
        const double w = (i & 1) ? (double)(1 + rng() % (1u << 24)) : 1.0 + u01(rng) * (double)(rng() % (1u << 20));
        const double c = u01(rng) * w * ((i & 2) ? 1.0 : u01(rng));
        push(c, w);
    }
    // mantissa edge cases on both sides
    const uint64_t mant[] = {0ull, 1ull, (1ull << 52) - 1, (1ull << 52) - 2, 1ull << 51, (1ull << 51) - 1, (1ull << 51) + 1,
                             0x5555555555555ull, 0xAAAAAAAAAAAAAull};
    const int exs[] = {-900, -899, -500, -53, -1, 0, 1, 52, 53, 54, 500, 699, 700};
    const int eys[] = {0, 1, 2, 30, 52, 53, 59};
    for (uint64_t ma : mant)
        for (uint64_t mb : mant)
            for (int ea : exs)
                for (int eb : eys) {
                    push(from_bits(((uint64_t)(ea + 1023) << 52) | ma), from_bits(((uint64_t)(eb + 1023) << 52) | mb));
                    push(-from_bits(((uint64_t)(ea + 1023) << 52) | ma), from_bits(((uint64_t)(eb + 1023) << 52) | mb));
                }
    for (int eb : eys) push(0.0, from_bits((uint64_t)(eb + 1023) << 52));
    const int n = (int)xs.size();
    double *dx, *dy, *dfirst;
    unsigned long long *dg, *dbad;
    if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&dy, n * 8) != hipSuccess || hipMalloc(&dg, 8) != hipSuccess ||
        hipMalloc(&dbad, 8) != hipSuccess || hipMalloc(&dfirst, 32) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 2;
    }
    (void)hipMemcpy(dx, xs.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dy, ys.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemset(dg, 0, 8);
    (void)hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, n, dg, dbad, dfirst);
    unsigned long long g = 0, bad = 0;
    double first[4] = {0, 0, 0, 0};
    if (hipMemcpy(&g, dg, 8, hipMemcpyDeviceToHost) != hipSuccess) {
        fprintf(stderr, "kernel failed\n");
        return 2;
    }
    (void)hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(first, dfirst, 32, hipMemcpyDeviceToHost);
    printf("pairs %d guarded %llu mismatches %llu\n", n, g, bad);
    if (bad) printf("first: %a / %a -> fast %a, compiler %a\n", first[0], first[1], first[2], first[3]);
    return bad ? 1 : 0;
}
