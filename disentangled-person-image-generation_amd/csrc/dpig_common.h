// Shared host/device helpers for libdpig_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include "dpig_hip.h"

namespace dpig {

// ---- error reporting (thread-local string, C-ABI getter dpig_last_error) ---------------------
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DPIG_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return DPIG_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kNumCU = 256;   // MI355X
constexpr int kNumXCD = 8;

// ---- device helpers --------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
    if (act == DPIG_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DPIG_ACT_LRELU) return v > 0.f ? v : alpha * v;   // tf.maximum(alpha*x, x), alpha<1
    return v;
}
// derivative evaluated at the activation OUTPUT y (sign(y) == sign(pre-activation) for alpha>0)
__device__ __forceinline__ float act_grad(float y, int act, float alpha) {
    if (act == DPIG_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == DPIG_ACT_LRELU) return y > 0.f ? 1.f : alpha;
    return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware bijective remap of a 1-D block id: blocks that the dispatcher places on one XCD
// (id % 8) receive a contiguous range of work items, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / kNumXCD, r = nwg % kNumXCD;
    const int xcd = bid % kNumXCD, idx = bid / kNumXCD;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace dpig
