// What the Winograd kernels of both tile sizes share (dpig_conv_wino.hip: F(2x2, 3x3) / F(3x3, 2x2); dpig_conv_wino4.hip: F(4x4, 3x3)):
// the launch parameter block, the buffer-descriptor / division helpers and the fused epilogue on four channels of one output pixel.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "dpig_common.h"
#include "dpig_conv_plan.h"

namespace dpig {
namespace wino {

constexpr unsigned OOB = 0x7fffffffu;

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lds_void;

struct WParams {
    const float* X;       // gathered activation (x forward, dy for dgrad), NHWC with channel stride ldx
    const float* U;       // transformed filter image [Kout / 64][Cin / 8][16][64][8]
    float* D;             // destination, NHWC with channel stride ldd
    float* D2;            // optional second output (activation before a post-activation residual add)
    const float* bias;    // [Kout] or null
    const float* res;     // residual / accumulate tensor (destination-shaped) or null
    const float* mask;    // activation-output tensor for act' (destination-shaped) or null
    float* partial;       // nsplit > 1: [nsplit][N * H * W][Kout] pre-epilogue partial sums (one per input-channel range)
    int nsplit, cps;      // input-channel splits, chunks per split
    int N, H, W, Cin, Kout;
    int ldx, ldd, ldres, ldmask, ldd2;
    int T, THW, TW;       // tiles in the batch, per image, per tile row
    int nch;              // Cin / 8
    int mtiles, ntiles;
    int act; float alpha; int res_post;
    unsigned x_bytes, u_bytes;
    unsigned mul_thw, shr_thw, mul_tw, shr_tw, mul_th, shr_th;
    int xmajor;           // workgroup order: 1 activation-major, 0 filter-major
    unsigned long long* trace;   // dev aid (dpig_debug_wino_trace): 8 s_memtime stamps per workgroup, or null
};

__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned shr) {
    return mul ? (int)(__umulhi((unsigned)n, mul) >> shr) : n;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Fused epilogue on 4 consecutive channels of one output pixel (the fp32 family's epi_vec4 without its split-K / replicate / class forms)
__device__ __forceinline__ void epi4(const WParams& p, long pix, int col, f32x4 v, f32x4 bv) {
    v += bv;
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
    if (p.res) rv = *reinterpret_cast<const f32x4*>(p.res + pix * p.ldres + col);
    if (p.res && !p.res_post) v += rv;
    if (p.mask) {
        const f32x4 mv = *reinterpret_cast<const f32x4*>(p.mask + pix * p.ldmask + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= act_grad(mv[e], p.act, p.alpha);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act, p.alpha);
    }
    if (p.D2) *reinterpret_cast<f32x4*>(p.D2 + pix * p.ldd2 + col) = v;
    if (p.res && p.res_post) v += rv;
    *reinterpret_cast<f32x4*>(p.D + pix * p.ldd + col) = v;
}


// second pass of a split plan (dpig_conv_wino.hip): sums the nsplit partial outputs in split order and runs the fused epilogue
int launch_reduce(const WParams& p, hipStream_t st);
// cost (cycles, this family's cost model) of the layer without the F(4x4, 3x3) form: the F(2x2, 3x3) plan or the direct family
double alt_cycles(const DpigConvDesc* d, int cin, int kout, int ld_in, int ld_out);
// dev aid: the stamp buffer dpig_debug_wino_trace installed, or null
unsigned long long* trace_buffer();

}  // namespace wino
}  // namespace dpig
