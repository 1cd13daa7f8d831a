// Hand-offs between the workgroups of ONE persistent launch through global memory, "the data is the flag"
// (MI355X_MICROARCH.md, visibility): every word is written once per launch with an 8-byte agent-scope atomic store into a
// buffer pre-filled with a bit pattern no arithmetic produces, and consumers re-read it with agent-scope atomic loads until
// it is set.  Spins are bounded; a time-out (or another workgroup's) marks the launch as failed through `status`.
// Used by the Householder reduction (dense64.hip: k_sytrd_wave).
#pragma once
#include <hip/hip_runtime.h>

namespace nlek {
namespace handoff {

typedef unsigned long long u64;
constexpr u64 kUnset = ~0ull;  // a NaN no arithmetic produces (hardware NaNs are 0x7FF8.. / 0xFFF8..)
constexpr unsigned kSpinLimit = 1u << 22;

__device__ __forceinline__ u64 ld_pub(const double* p) {
    return __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_pub(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<u64*>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// re-read one published word until it is set; gives up (fail = true) after kSpinLimit polls or when another workgroup
// has flagged the launch as failed
__device__ __forceinline__ double wait_pub(const double* p, const int* status, bool& fail) {
    u64 b;
    unsigned spins = 0;
    while ((b = ld_pub(p)) == kUnset) {
        if (++spins > kSpinLimit) {
            fail = true;
            break;
        }
        if ((spins & 1023u) == 0 &&
            __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            fail = true;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return __longlong_as_double((long long)b);
}

}  // namespace handoff
}  // namespace nlek
