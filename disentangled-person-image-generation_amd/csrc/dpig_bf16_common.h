// Shared device-side pieces of the bf16-STORAGE conv family (dpig_conv_bf16.hip: 128 x 128 tiles; dpig_conv_bf16_q.hip:
// the 8-wave 256 x 256 / 512 x 128 "8-phase" kernels): operand types, the gather-GEMM parameter block, LDS-DMA and
// bf16 pack / unpack helpers, and the fused epilogue on 8 consecutive columns of one GEMM row.
#pragma once
#include <cstdlib>
#include "dpig_common.h"

namespace dpig {
namespace bfk {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int TM = 128, TN = 128, TK = 64;
constexpr int ROWB = 128;                 // bytes of one LDS row: 64 bf16 of k, unpadded (LDS-DMA images are lane-linear)
constexpr int TILE_B = TM * ROWB;         // one operand tile: 16 KB
constexpr int STAGE_B = 2 * TILE_B;       // A + B: 32 KB
constexpr int LDC = TN + 4;               // fp32 accumulator staging stride of the epilogue
constexpr int STATS_RED_BYTES = 16 * TN * 4;                 // scratch of the BN-statistics epilogue: 16 row groups x 128 columns
constexpr int SMEM_BYTES = TM * LDC * 4 + STATS_RED_BYTES;   // 67584 (>= 2 stages = 65536) + 8192; 2 workgroups per CU = 148 KB of 160 KB
constexpr unsigned OOB = 0x7fffffffu;
// split-K partial sums cost relatively more than on the fp32 pipe (the products are ~8x faster, HBM is not)
static inline double split_penalty_bf16() {
    static const double v = getenv("DPIG_BF16_SPLIT_PEN") ? atof(getenv("DPIG_BF16_SPLIT_PEN")) : 700.0;
    return v;
}
#define kSplitPenalty split_penalty_bf16()

struct BGParams {
    const bf16_t* A;      // gathered source activation (x for fwd, dy for dgrad)
    const bf16_t* B;      // filter shadow, [wtap][Ncols][Cs]
    bf16_t* D;            // destination activation
    bf16_t* D2;           // optional second output: the activation BEFORE a post-activation residual add
    const float* bias;    // [Ncols] fp32 or null
    const bf16_t* res;    // residual / accumulate tensor (dest-shaped) or null
    const float* res_cls; // class-indexed residual [images][9][Ncols] fp32 (tiled-embedding collapse) or null
    const bf16_t* mask;   // activation-output tensor for act' (dest-shaped) or null
    float* partial;       // split-K workspace [nsplit][M][Ncols]
    float* stats;         // BN partial statistics [mtiles][2][Ncols] of acc + bias (sum, centred squares per row tile) or null
    int M, Hr, Wr, HrWr;
    int Hs, Ws, lda, Cs, sr;
    int Ncols;
    int Hd, Wd, ldd, dr, dpy, dpx;
    int ldres, ldmask, ldd2;
    int res_post;
    int ntaps, cchunks, ktiles, tiles_per_split, nsplit;
    int mtiles, ntiles;
    int act; float alpha;
    int replicate, identity_rows;
    int tap_nb, oy0, oys, ox0, oxs, w0, wa, wb;     // affine tap family, as GGParams in dpig_conv.hip
    unsigned a_bytes, b_bytes;
    unsigned mul_hrwr, shr_hrwr, mul_wr, shr_wr;
    int tiles_x, tiles_y;  // bh_kernel: 2-D output patches per image
    int halo128;           // host hint for the tile-family choice: the 128 x 128 family would run this layer on its halo-patch kernel
};

__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned shr) {
    return mul ? (int)(__umulhi((unsigned)n, mul) >> shr) : n;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// 16 bytes per lane, HBM/L2 -> LDS (buffer_load_dwordx4 ... lds).  `lds_dst` is wave-uniform; lane l lands at
// lds_dst + 16 * l.  The source is descriptor base + voff (per lane) + soff (scalar); a lane whose voff is OOB fails the
// hardware bounds check and delivers zeros (halo, rows >= M, channels >= C: scripts/ubench/lds_probe.hip).  (The flat
// global_load_lds form with a zero page measured 13 % slower and was dropped.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, char* lds_dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_dst, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ void unpack8(uint4 u, float (&v)[8]) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {      // round-to-nearest-even (v_cvt_pk_bf16_f32)
    bf16x2 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    return make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
}
__device__ __forceinline__ int border_class(int y, int x, int H, int W) {
    const int cy = (y == 0) ? 0 : ((y == H - 1) ? 2 : 1);
    const int cx = (x == 0) ? 0 : ((x == W - 1) ? 2 : 1);
    return cy * 3 + cx;
}

// Fused epilogue on 8 consecutive columns of one GEMM row (16-byte bf16 accesses; fp32 arithmetic).
__device__ __forceinline__ void epi8(const BGParams& p, int row, int col, float (&v)[8], const float (&bv)[8]) {
    long pix = row;
    long crow = 0;
    if (!p.identity_rows || p.res_cls) {
        const int n = row / p.HrWr;
        const int rem = row - n * p.HrWr;
        const int rr = rem / p.Wr;
        const int cc = rem - rr * p.Wr;
        if (!p.identity_rows) pix = ((long)n * p.Hd + (rr * p.dr + p.dpy)) * p.Wd + (cc * p.dr + p.dpx);
        crow = (long)n * 9 + border_class(rr, cc, p.Hr, p.Wr);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bv[e];
    float rv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) rv[e] = 0.f;
    if (p.res_cls) {
        const float4 r0 = *reinterpret_cast<const float4*>(p.res_cls + crow * p.ldres + col);
        const float4 r1 = *reinterpret_cast<const float4*>(p.res_cls + crow * p.ldres + col + 4);
        rv[0] = r0.x; rv[1] = r0.y; rv[2] = r0.z; rv[3] = r0.w; rv[4] = r1.x; rv[5] = r1.y; rv[6] = r1.z; rv[7] = r1.w;
    } else if (p.res) {
        unpack8(*reinterpret_cast<const uint4*>(p.res + pix * p.ldres + col), rv);
    }
    const bool has_res = p.res || p.res_cls;
    if (has_res && !p.res_post) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
    }
    if (p.mask) {
        float mv[8];
        unpack8(*reinterpret_cast<const uint4*>(p.mask + pix * p.ldmask + col), mv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= act_grad(mv[e], p.act, p.alpha);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e], p.act, p.alpha);
    }
    if (p.D2) {
        const uint4 o2 = pack8(v);
        *reinterpret_cast<uint4*>(p.D2 + pix * p.ldd2 + col) = o2;
        if (has_res && p.res_post) {       // the sum is formed from the STORED (rounded) activation: out = c2 + skip
            unpack8(o2, v);
        }
    }
    if (has_res && p.res_post) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
    }
    const uint4 o = pack8(v);
    *reinterpret_cast<uint4*>(p.D + pix * p.ldd + col) = o;
    if (p.replicate) {
        *reinterpret_cast<uint4*>(p.D + (pix + 1) * p.ldd + col) = o;
        *reinterpret_cast<uint4*>(p.D + (pix + p.Wd) * p.ldd + col) = o;
        *reinterpret_cast<uint4*>(p.D + (pix + p.Wd + 1) * p.ldd + col) = o;
    }
}

// wgrad problem: dw[(tap, ci), co] = sum over output pixels of x[src(pixel, tap), ci] * dy[pixel, co]
struct BWParams {
    const bf16_t* X; const bf16_t* DY; float* DW; float* partial;
    int Npix, Ho, Wo, HoWo;
    int H, W, ldx, C, shift, s;
    int K, ldy;
    int ntaps, cblocks, ntiles;
    int ktiles, tiles_per_split, nsplit, wrows;
    float beta;
    int S, pad_t, pad_l;
    unsigned x_bytes, y_bytes;
    unsigned mul_howo, shr_howo, mul_wo, shr_wo;
    float* DB; float* bias_partial; float beta_b;
    int d64_oy, d64_ox, d64_n;
    int q_mtiles, q_ntiles;   // bwq_kernel: item tiles x column tiles
};

// dpig_conv_bf16_wq.hip, the large-tile (8-wave) wgrad of stride-1 SAME layers: bwq_choose = the variant (1: 2 items x 256
// co, 2: 4 items x 128 co) the selection rule picks or 0; bwq_plan = its tile grid and split; bwq_try launches (1 / < 0)
int bwq_choose(int ntaps, int C, int K, long npix, int forced_split);
double bwq_plan(int ntaps, int C, int K, long npix, int variant, int forced_split, int* mtiles, int* ntiles, int* nsplit, int* tps);
int bwq_try(BWParams& p, int variant, hipStream_t st);

// dpig_conv_bf16_q.hip: 1 = launched on the large-tile (8-wave) kernel, 0 = not this layer's case, < 0 = error
int bq_try(BGParams& p, hipStream_t st);

}  // namespace bfk
}  // namespace dpig
