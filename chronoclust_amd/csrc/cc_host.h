// cc_host.h - the pieces of the library that are plain C++ (no HIP): shared by the one HIP translation unit (cc_api.hip) and
// by the host-only sanitizer build (tests/host_san/: g++ -fsanitize=address,undefined / thread - never the GPU build).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define CC_HOSTDEV __host__ __device__
#else
#define CC_HOSTDEV
#endif

// Exact multi-GPU path: the block [lo, hi) of n rows that rank `rank` of `world` takes, in whole
// units of `unit` rows (1: table rows of a snapshot scan, current pcores of the association argmin; 64: p rows of
// the offline pair matrices = whole words of the adjacency bitmask).  Every rank gets the same share
// ceil(units / world) * unit, so the blocks tile [0, n) in rank order and all-gathers have one block size.
// (64-bit intermediates: n + unit - 1 overflowed an int for n within `unit` of 2^31 - found by the UBSan build; a share
// that would not fit an int is clamped - such a block ends at n anyway)
CC_HOSTDEV inline int cc_shard_share(int n, int world, int unit)
{
    const long long units = ((long long)n + unit - 1) / unit;
    const long long share = ((units + world - 1) / world) * unit;
    return (int)(share < 0x7fffffffLL ? share : 0x7fffffffLL);
}
CC_HOSTDEV inline void cc_shard_range(int n, int world, int rank, int unit, int* lo, int* hi)
{
    const int share = cc_shard_share(n, world, unit);
    const long long a = (long long)rank * share;
    *lo = (int)(a < n ? a : n);
    const long long b = a + share;
    *hi = (int)(b < n ? b : n);
}

// the sequential kernel's LDS image of the table (k_seq, cc_seq.h): (4 d + 5) doubles per row
#define CC_SEQ_DOUBLES 6600
CC_HOSTDEV inline int cc_seq_cap_rows(int d) { return CC_SEQ_DOUBLES / (4 * d + 5); }
