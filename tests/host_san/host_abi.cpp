// The pure-host part of the C-ABI library as a translation unit of its own, for the sanitizer builds ("race
// detection / sanitizers"): the window policy (csrc/cc_policy.h through cc_policy_replay), the text formatter of the per-point
// file (csrc/cc_csv.h through cc_format_points_csv - called from a pool of host threads), cc_shard_rows and the
// sequential-kernel rate guess.  The SAME source text the product compiles (csrc/cc_host_abi.inc is #included by cc_api.hip),
// built by g++ with -fsanitize=address,undefined or -fsanitize=thread.  Never the GPU build: sanitizers are not available
// for device code on this pool.  tests/test_host_sanitizers.py builds and drives it.
#include <algorithm>
#include <cstring>

#include "../../include/chronoclust_hip.h"
#include "../../chronoclust_amd/csrc/cc_host.h"
#include "../../chronoclust_amd/csrc/cc_policy.h"
#include "../../chronoclust_amd/csrc/cc_csv.h"

extern "C" {
#include "../../chronoclust_amd/csrc/cc_host_abi.inc"
}
