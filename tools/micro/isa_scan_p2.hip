// One kernel of the library compiled alone (seconds instead of the library's 80): ISA and resource usage of k_scan_p2<20, 4>.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-fast-math -c --offload-device-only -S \
//         -Rpass-analysis=kernel-resource-usage -o /tmp/isa/p2.s tools/micro/isa_scan_p2.hip
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/chronoclust_hip.h"
#include "../../chronoclust_amd/csrc/cc_common.h"
#include "../../chronoclust_amd/csrc/cc_online.h"
#ifndef ISA_LISTED
#define ISA_LISTED false
#endif
#ifndef ISA_DP
#define ISA_DP 20
#endif
template __global__ void k_scan_p2<ISA_DP, 4>(Ctl*, const double*, const double*, const double*, const int*, const int*, const double*, size_t, Cand*, int, int, size_t, int, int, unsigned long long*, double, unsigned long long*);
template __global__ void k_scan_p3<ISA_DP, 4, ISA_LISTED>(Ctl*, const double*, const double*, const double*, const int*, const int*, const double*, size_t, Cand*, int, int, size_t, int, int, unsigned long long*, double, unsigned long long*, const cc_h8*, const Prefix16Hdr*, size_t);
template __global__ void k_prefix16<ISA_DP>(const Ctl*, const double*, const int*, cc_h8*, Prefix16Hdr*, size_t, int, int);
