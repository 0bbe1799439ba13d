// dpig_thin.hip -- 3x3 stride-1 SAME convolutions with THREE output channels (the generator's image conv,
// models.py:571-573: 256 -> 3 at full resolution), forward / dgrad / wgrad.
//
// As a GEMM this layer is [N*H*W, 9*C] x [9*C, 3]: on the 128x32 MFMA tile 29 of 32 columns are padding and the
// launch is paced by its per-k-tile fixed costs (340 us for 1.8 GFLOP at B=16).  The layer is HBM-bound -- it
// must stream the C-channel activation (134 MB) once -- so it runs on the vector ALUs instead:
//
//   * one wave walks a strip of one image row; lane l owns input channels 4l..4l+3 (16-byte accesses, C <= 256),
//   * the lane's 9 x 4 x 3 filter coefficients live in registers (108 VGPRs),
//   * the 3x3 input window slides along the row: 3 new 16-byte loads per pixel instead of 9,
//   * fwd: the 3 per-lane partial sums are reduced across the wave with DPP row shifts / broadcasts;
//     dgrad: the 27 dy values of the window are wave-uniform, every lane produces its own 4 dx channels;
//     wgrad: every lane keeps its 108 partial filter gradients (+ the bias gradient) in registers over the strip,
//     waves of a workgroup are summed through LDS in a fixed order and a second kernel sums the workgroups
//     (deterministic: no atomics).
//
// Roofline: HBM.  Algorithmic bytes = 4*(N*H*W*C + N*H*W*3) per call (+ filter), i.e. 135.8 MB at B=16.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "dpig_common.h"
#include "dpig_hip.h"
#include "dpig_thin.h"

namespace dpig {

constexpr int TK = 3;        // output channels
constexpr int XS = 32;       // pixels per strip
constexpr int kWgradBlocks = 2 * kNumCU;   // persistent wgrad workgroups (4 waves each): two per CU
typedef int v4i_t __attribute__((ext_vector_type(4)));

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true);
    return v + __builtin_bit_cast(float, t);
}
// sum over the 64 lanes, result valid in lane 63 (row_shr 1,2,4,8 = inclusive scan inside each row of 16,
// row_bcast:15 folds rows 0->1 and 2->3, row_bcast:31 folds the lower half into the upper)
__device__ __forceinline__ float reduce_to_lane63(float v) {
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return v;
}

struct ThinParams {
    const void* X;               // the C-channel tensor: fp32, or bf16 in the WB kernel variants ('bf16' storage mode)
    const float* W; const float* bias; const float* DY;
    float* Y; void* DX; float* partial;
    int N, H, Wd, C, ldx, ldy;          // ldy: row stride of the 3-channel tensor
    int nstrips, nwaves;
    unsigned x_bytes;
    int act; float alpha;
};

// 4 consecutive channels of the wide tensor: 16 bytes of fp32 or (WB) 8 bytes of bf16, widened exactly
template <bool WB>
__device__ __forceinline__ float4 ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    if constexpr (WB) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
        return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                           __uint_as_float(t.y & 0xffff0000u));
    } else {
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
        return make_float4(t.x, t.y, t.z, t.w);
    }
}
typedef __bf16 thin_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned thin_pack2(float a, float b) {
    thin_bf16x2 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
constexpr unsigned OOBT = 0x7fffffffu;

// filter coefficients of this lane's 4 channels: wr[tap][e][k] = w[(tap*C + 4*lane + e)*3 + k]
__device__ __forceinline__ void load_filter(const float* __restrict__ w, int C, int lane, float (&wr)[9][4][TK]) {
    const bool ok = lane * 4 < C;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < TK; ++k) wr[t][e][k] = ok ? w[((long)t * C + lane * 4 + e) * TK + k] : 0.f;
}

// ---- forward ---------------------------------------------------------------------------------------------
template <bool WB>
__global__ __launch_bounds__(256) void thin3_fwd_kernel(const ThinParams p) {
    constexpr unsigned ES = WB ? 2u : 4u;          // bytes per element of the wide tensor
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wv >= p.nwaves) return;
    const int strip = wv % p.nstrips;
    const int row = wv / p.nstrips;                 // n*H + y
    const int y = row % p.H;
    const int x0 = strip * XS;
    const int x1 = min(p.Wd, x0 + XS);
    float wr[9][4][TK];
    load_filter(p.W, p.C, lane, wr);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.X), 0, (int)p.x_bytes, 0x00020000);
    const bool cok = lane * 4 < p.C;
    // byte offset of pixel (row + dy, ix), or OOB when outside the image
    auto off = [&](int dy, int ix) -> unsigned {
        const bool ok = cok & ((unsigned)(y + dy) < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.Wd);
        return ok ? (unsigned)((((long)(row + dy)) * p.Wd + ix) * p.ldx + lane * 4) * ES : OOBT;
    };
    float4 win[3][3];                               // [column x-1, x, x+1][row y-1, y, y+1]
#pragma unroll
    for (int r = 0; r < 3; ++r) { win[1][r] = ld16<WB>(rs, off(r - 1, x0 - 1)); win[2][r] = ld16<WB>(rs, off(r - 1, x0)); }
    float b[TK];
#pragma unroll
    for (int k = 0; k < TK; ++k) b[k] = p.bias ? p.bias[k] : 0.f;
    const float slope = (p.act == DPIG_ACT_NONE) ? 1.f : ((p.act == DPIG_ACT_RELU) ? 0.f : p.alpha);
    for (int x = x0; x < x1; ++x) {
#pragma unroll
        for (int r = 0; r < 3; ++r) { win[0][r] = win[1][r]; win[1][r] = win[2][r]; win[2][r] = ld16<WB>(rs, off(r - 1, x + 1)); }
        float acc[TK];
#pragma unroll
        for (int k = 0; k < TK; ++k) acc[k] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 v = win[kx][ky];
                const int t = ky * 3 + kx;
#pragma unroll
                for (int k = 0; k < TK; ++k)
                    acc[k] += v.x * wr[t][0][k] + v.y * wr[t][1][k] + v.z * wr[t][2][k] + v.w * wr[t][3][k];
            }
#pragma unroll
        for (int k = 0; k < TK; ++k) acc[k] = reduce_to_lane63(acc[k]);
        if (lane == 63) {
            float* o = p.Y + ((long)row * p.Wd + x) * p.ldy;
#pragma unroll
            for (int k = 0; k < TK; ++k) {
                const float v = acc[k] + b[k];
                o[k] = (v > 0.f) ? v : (v * slope + 0.f);
            }
        }
    }
}

// ---- dgrad: dx[p][c] = sum_{ky,kx,k} dy[p + (1-ky, 1-kx)][k] * w[ky][kx][c][k] --------------------------
template <bool WB>
__global__ __launch_bounds__(256) void thin3_dgrad_kernel(const ThinParams p) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wv >= p.nwaves) return;
    const int strip = wv % p.nstrips;
    const int row = wv / p.nstrips;
    const int y = row % p.H;
    const int x0 = strip * XS;
    const int x1 = min(p.Wd, x0 + XS);
    float wr[9][4][TK];
    load_filter(p.W, p.C, lane, wr);
    // window of dy: g[col][row][k], wave-uniform (scalar loads)
    auto ldg = [&](int dyy, int ix, float (&g)[TK]) {
        const bool ok = ((unsigned)(y + dyy) < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.Wd);
        const float* s = p.DY + ((long)(row + dyy) * p.Wd + (ok ? ix : 0)) * p.ldy;
#pragma unroll
        for (int k = 0; k < TK; ++k) g[k] = ok ? s[k] : 0.f;
    };
    float g[3][3][TK];
#pragma unroll
    for (int r = 0; r < 3; ++r) { ldg(r - 1, x0 - 1, g[1][r]); ldg(r - 1, x0, g[2][r]); }
    const bool cok = lane * 4 < p.C;
    for (int x = x0; x < x1; ++x) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int k = 0; k < TK; ++k) { g[0][r][k] = g[1][r][k]; g[1][r][k] = g[2][r][k]; }
            ldg(r - 1, x + 1, g[2][r]);
        }
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        // output pixel (y, x) receives filter tap (ky, kx) from dy pixel (y + 1 - ky, x + 1 - kx)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int t = ky * 3 + kx;
#pragma unroll
                for (int k = 0; k < TK; ++k) {
                    const float gv = g[2 - kx][2 - ky][k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] += gv * wr[t][e][k];
                }
            }
        if (cok) {
            const long o = ((long)row * p.Wd + x) * p.ldx + lane * 4;
            if constexpr (WB) *reinterpret_cast<uint2*>(static_cast<unsigned short*>(p.DX) + o) = make_uint2(thin_pack2(a[0], a[1]), thin_pack2(a[2], a[3]));
            else *reinterpret_cast<float4*>(static_cast<float*>(p.DX) + o) = make_float4(a[0], a[1], a[2], a[3]);
        }
    }
}

// ---- wgrad: dw[t][c][k] = sum_p x[p + tap t][c] * dy[p][k],  db[k] = sum_p dy[p][k] -------------------------
// partial layout per workgroup: [9][C][3] filter gradient followed by [3] bias gradient
template <bool WB>
__global__ __launch_bounds__(256) void thin3_wgrad_kernel(const ThinParams p) {
    constexpr unsigned ES = WB ? 2u : 4u;
    __shared__ float red[9 * 4 * TK + TK][64];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    float acc[9][4][TK];
    float bs[TK];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < TK; ++k) acc[t][e][k] = 0.f;
#pragma unroll
    for (int k = 0; k < TK; ++k) bs[k] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.X), 0, (int)p.x_bytes, 0x00020000);
    const bool cok = lane * 4 < p.C;
    // a wave owns the strips u = wave, wave + #waves, ... (fixed assignment -> fixed summation order)
    for (int u = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wid); u < p.nwaves; u += gridDim.x * 4) {
        const int strip = u % p.nstrips;
        const int row = u / p.nstrips;
        const int y = row % p.H;
        const int x0 = strip * XS;
        const int x1 = min(p.Wd, x0 + XS);
        auto off = [&](int dy, int ix) -> unsigned {
            const bool ok = cok & ((unsigned)(y + dy) < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.Wd);
            return ok ? (unsigned)((((long)(row + dy)) * p.Wd + ix) * p.ldx + lane * 4) * ES : OOBT;
        };
        float4 win[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { win[1][r] = ld16<WB>(rs, off(r - 1, x0 - 1)); win[2][r] = ld16<WB>(rs, off(r - 1, x0)); }
        for (int x = x0; x < x1; ++x) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { win[0][r] = win[1][r]; win[1][r] = win[2][r]; win[2][r] = ld16<WB>(rs, off(r - 1, x + 1)); }
            const float* s = p.DY + ((long)row * p.Wd + x) * p.ldy;
            float g[TK];
#pragma unroll
            for (int k = 0; k < TK; ++k) { g[k] = s[k]; bs[k] += g[k]; }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 v = win[kx][ky];
                    const int t = ky * 3 + kx;
#pragma unroll
                    for (int k = 0; k < TK; ++k) {
                        acc[t][0][k] += v.x * g[k]; acc[t][1][k] += v.y * g[k];
                        acc[t][2][k] += v.z * g[k]; acc[t][3][k] += v.w * g[k];
                    }
                }
        }
    }
    // waves of the workgroup are added in wave order (fixed summation order)
    for (int ph = 0; ph < 4; ++ph) {
        if (wid == ph) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int k = 0; k < TK; ++k) {
                        const int i = (t * 4 + e) * TK + k;
                        red[i][lane] = (ph == 0) ? acc[t][e][k] : red[i][lane] + acc[t][e][k];
                    }
#pragma unroll
            for (int k = 0; k < TK; ++k) red[108 + k][lane] = (ph == 0) ? bs[k] : red[108 + k][lane] + bs[k];
        }
        __syncthreads();
    }
    // write the workgroup's partial: dw[t][c = 4*lane + e][k]
    float* out = p.partial + (long)blockIdx.x * (9 * p.C * TK + TK);
    for (int i = threadIdx.x; i < 9 * p.C * TK; i += 256) {
        const int k = i % TK;
        const int c = (i / TK) % p.C;
        const int t = i / (TK * p.C);
        out[i] = red[(t * 4 + (c & 3)) * TK + k][c >> 2];
    }
    if (threadIdx.x < TK) out[9 * p.C * TK + threadIdx.x] = red[108 + threadIdx.x][0];   // bias: every lane saw the same dy
}

// dw = beta*dw + sum over workgroups (db likewise; may be null): 32 outputs x 8 slices of the workgroup list per
// block, slices combined through LDS in slice order
__global__ __launch_bounds__(256) void thin3_wgrad_reduce_kernel(const float* __restrict__ partial, int nblk, int wsize,
                                                                 float* __restrict__ dw, float beta,
                                                                 float* __restrict__ db, float beta_b) {
    __shared__ float sm[8][32];
    const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + j;
    const int n = wsize + TK;
    float v = 0.f;
    if (i < n)
        for (int b = g; b < nblk; b += 8) v += partial[(long)b * n + i];
    sm[g][j] = v;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = sm[0][j];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += sm[q][j];
        if (i < wsize) dw[i] = (beta != 0.f) ? beta * dw[i] + t : t;
        else if (db) db[i - wsize] = (beta_b != 0.f) ? beta_b * db[i - wsize] + t : t;
    }
}

// =============================================================================================================
// THREE INPUT channels: the encoder stem (models.py:396, 3x3 s1, 3 -> 128) and the critic's first conv
// (wgan_gp.py DCGANDiscriminator, 5x5 s2, 3 -> 64), forward / wgrad, and the 5x5 s2 dgrad towards the image.
// GEMM-wise K = R*S*3 = 27 / 75: a fraction of one 32-deep k-tile; the layers are bound by streaming the
// Cout-channel tensor.  Lanes own output-channel quads (fwd, wgrad) or single output channels (dgrad); the
// 3-channel image rows a workgroup needs are staged in LDS with zero margins (no border branches), the filter
// sits in LDS (fwd) or registers (dgrad); wgrad keeps R*S*3 (or one filter row of) float4 accumulators per lane.
// =============================================================================================================
struct FewCParams {
    const float* X; const float* W; const float* bias;
    const void* DY; void* Y;     // the K-channel tensor: fp32, or bf16 in the WB kernel variants
    float* DX; float* partial;
    int N, H, Wd, ldx;           // image [N,H,Wd,3], row stride ldx
    int Ho, Wo, K, ldy;          // Cout tensor [N,Ho,Wo,K]
    int pt, pl, act; float alpha;
    int nrows;                   // N*Ho
};
constexpr int FC_MAXW = 264;     // widest staged image row (+ margins)

// stage image row iy of image n into LDS as [ (Wd + S) * 3 ] with pl zero pixels in front: dst[(ix + pl) * 3 + c]
template <int S>
__device__ __forceinline__ void stage_row(const FewCParams& p, int n, int iy, float* dst) {
    const int tot = (p.Wd + S) * 3;
    const bool rok = (unsigned)iy < (unsigned)p.H;
    for (int i = threadIdx.x; i < tot; i += blockDim.x) {
        const int ix = i / 3 - p.pl, c = i - (i / 3) * 3;
        const bool ok = rok & ((unsigned)ix < (unsigned)p.Wd);
        dst[i] = ok ? p.X[(((long)n * p.H + iy) * p.Wd + ix) * p.ldx + c] : 0.f;
    }
}

// ---- forward: one workgroup per output row; thread = (output-channel quad, pixel slot) ------------------------
// 4 consecutive channels of the wide (K-channel) tensor at element offset o
template <bool WB>
__device__ __forceinline__ float4 fewc_ld4(const void* base, long o) {
    if constexpr (WB) {
        const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(base) + o);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u));
    } else {
        return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + o);
    }
}
template <int R, int S, int ST, bool WB>
__global__ __launch_bounds__(256) void fewc_fwd_kernel(const FewCParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* wsm = sm;                                  // [R*S*3][K]
    float* xrow = sm + R * S * 3 * p.K;               // [R][FC_MAXW*3]
    const int row = blockIdx.x;                       // n*Ho + oy
    const int n = row / p.Ho, oy = row - n * p.Ho;
    for (int i = threadIdx.x * 4; i < R * S * 3 * p.K; i += 1024)
        *reinterpret_cast<float4*>(&wsm[i]) = *reinterpret_cast<const float4*>(&p.W[i]);
#pragma unroll
    for (int r = 0; r < R; ++r) stage_row<S>(p, n, oy * ST - p.pt + r, xrow + r * FC_MAXW * 3);
    __syncthreads();
    const int LP = p.K >> 2;                          // lanes per pixel
    const int kq = threadIdx.x % LP, slot = threadIdx.x / LP, nslots = 256 / LP;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + kq * 4);
    const float slope = (p.act == DPIG_ACT_NONE) ? 1.f : ((p.act == DPIG_ACT_RELU) ? 0.f : p.alpha);
    for (int ox = slot; ox < p.Wo; ox += nslots) {
        float4 a = bv;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float* xr = xrow + r * FC_MAXW * 3 + ox * ST * 3;       // (margin: ix + pl folded into staging)
#pragma unroll
            for (int sc = 0; sc < S * 3; ++sc) {
                const float xv = xr[sc];
                const float4 wv = *reinterpret_cast<const float4*>(&wsm[((r * S) * 3 + sc) * p.K + kq * 4]);
                a.x += xv * wv.x; a.y += xv * wv.y; a.z += xv * wv.z; a.w += xv * wv.w;
            }
        }
        a.x = (a.x > 0.f) ? a.x : (a.x * slope + 0.f); a.y = (a.y > 0.f) ? a.y : (a.y * slope + 0.f);
        a.z = (a.z > 0.f) ? a.z : (a.z * slope + 0.f); a.w = (a.w > 0.f) ? a.w : (a.w * slope + 0.f);
        const long o = ((long)row * p.Wo + ox) * p.ldy + kq * 4;
        if constexpr (WB) *reinterpret_cast<uint2*>(static_cast<unsigned short*>(p.Y) + o) = make_uint2(thin_pack2(a.x, a.y), thin_pack2(a.z, a.w));
        else *reinterpret_cast<float4*>(static_cast<float*>(p.Y) + o) = a;
    }
}

// ---- wgrad: persistent workgroups over output rows; blockIdx.y = tap group (TG consecutive taps) ----------------
// partial layout per workgroup (blockIdx.x): [R*S*3][K] filter gradient, then [K] bias gradient
template <int R, int S, int ST, int TG, bool WB>
__global__ __launch_bounds__(256) void fewc_wgrad_kernel(const FewCParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int RG = (TG + S - 1) / S;              // image rows a tap group touches (TG = S: 1, TG = R*S: R)
    float* xrow = sm;                                 // [RG][FC_MAXW*3]
    float* red = sm + RG * FC_MAXW * 3;               // [(TG*3 + 1)][K]
    const int t0 = blockIdx.y * TG;                   // first tap of this group
    const int r0 = t0 / S;                            // its filter row (groups are whole rows or everything)
    const int LP = p.K >> 2;
    const int kq = threadIdx.x % LP, slot = threadIdx.x / LP, nslots = 256 / LP;
    float4 acc[TG * 3];
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < TG * 3; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int row = blockIdx.x; row < p.nrows; row += gridDim.x) {
        const int n = row / p.Ho, oy = row - n * p.Ho;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RG; ++r) stage_row<S>(p, n, oy * ST - p.pt + r0 + r, xrow + r * FC_MAXW * 3);
        __syncthreads();
        for (int ox = slot; ox < p.Wo; ox += nslots) {
            const float4 g = fewc_ld4<WB>(p.DY, ((long)row * p.Wo + ox) * p.ldy + kq * 4);
            bs.x += g.x; bs.y += g.y; bs.z += g.z; bs.w += g.w;
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const float* xr = xrow + (t / S) * FC_MAXW * 3 + (ox * ST + (t % S)) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float xv = xr[c];
                    float4& a = acc[t * 3 + c];
                    a.x += xv * g.x; a.y += xv * g.y; a.z += xv * g.z; a.w += xv * g.w;
                }
            }
        }
    }
    // pixel slots are added in slot order (fixed summation order)
    for (int ph = 0; ph < nslots; ++ph) {
        __syncthreads();
        if (slot == ph) {
#pragma unroll
            for (int i = 0; i < TG * 3; ++i) {
                float4* d = reinterpret_cast<float4*>(&red[i * p.K + kq * 4]);
                float4 v = acc[i];
                if (ph) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *d = v;
            }
            float4* d = reinterpret_cast<float4*>(&red[TG * 3 * p.K + kq * 4]);
            float4 v = bs;
            if (ph) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *d = v;
        }
    }
    __syncthreads();
    float* out = p.partial + (long)blockIdx.x * (R * S * 3 * p.K + p.K);
    for (int i = threadIdx.x; i < TG * 3 * p.K; i += 256) out[t0 * 3 * p.K + i] = red[i];
    if (blockIdx.y == 0)
        for (int i = threadIdx.x; i < p.K; i += 256) out[R * S * 3 * p.K + i] = red[TG * 3 * p.K + i];
}

// dst = beta*dst + sum over workgroups, for the filter ([wsize]) and the bias ([K]) parts of the partials
__global__ __launch_bounds__(256) void fewc_wgrad_reduce_kernel(const float* __restrict__ partial, int nblk, int wsize,
                                                                int K, float* __restrict__ dw, float beta,
                                                                float* __restrict__ db, float beta_b) {
    __shared__ float smr[8][32];
    const int j = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + j;
    const int n = wsize + K;
    float v = 0.f;
    if (i < n)
        for (int b = g; b < nblk; b += 8) v += partial[(long)b * n + i];
    smr[g][j] = v;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = smr[0][j];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += smr[q][j];
        if (i < wsize) dw[i] = (beta != 0.f) ? beta * dw[i] + t : t;
        else if (db) db[i - wsize] = (beta_b != 0.f) ? beta_b * db[i - wsize] + t : t;
    }
}

// ---- dgrad towards the 3-channel image: lane = output channel k (K <= 64), one wave per strip of image pixels --
template <int R, int S, int ST, bool WB>
__global__ __launch_bounds__(256) void fewc_dgrad_kernel(const FewCParams p, int nstrips, int nwaves) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wv >= nwaves) return;
    const int strip = wv % nstrips;
    const int row = wv / nstrips;                     // n*H + iy
    const int n = row / p.H, iy = row - n * p.H;
    const bool kok = lane < p.K;
    float wr[R * S][3];                               // w[t][c][k = lane]
#pragma unroll
    for (int t = 0; t < R * S; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) wr[t][c] = kok ? p.W[((long)t * 3 + c) * p.K + lane] : 0.f;
    const int x0 = strip * XS, x1 = min(p.Wd, x0 + XS);
    for (int ix = x0; ix < x1; ++ix) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int ky = 0; ky < R; ++ky) {
            const int ty = iy + p.pt - ky;            // = oy * ST
            if (ty < 0 || (ty % ST) != 0 || ty / ST >= p.Ho) continue;
#pragma unroll
            for (int kx = 0; kx < S; ++kx) {
                const int tx = ix + p.pl - kx;
                if (tx < 0 || (tx % ST) != 0 || tx / ST >= p.Wo) continue;
                const long go = (((long)n * p.Ho + ty / ST) * p.Wo + tx / ST) * p.ldy + lane;
                const float g = !kok ? 0.f : (WB ? __uint_as_float((unsigned)static_cast<const unsigned short*>(p.DY)[go] << 16)
                                                 : static_cast<const float*>(p.DY)[go]);
                a0 += g * wr[ky * S + kx][0]; a1 += g * wr[ky * S + kx][1]; a2 += g * wr[ky * S + kx][2];
            }
        }
        a0 = reduce_to_lane63(a0); a1 = reduce_to_lane63(a1); a2 = reduce_to_lane63(a2);
        if (lane == 63) {
            float* o = p.DX + ((long)row * p.Wd + ix) * p.ldx;
            o[0] = a0; o[1] = a1; o[2] = a2;
        }
    }
}

// ---- the same gradient for K == 64 with lane = (pixel, 8-channel chunk): a wave takes 8 pixels of ONE row and ONE column parity,
// so the set of taps that reach them is wave-uniform (stride 2: tap (ky, kx) reaches pixel (iy, ix) iff iy + pt - ky and
// ix + pl - kx are even); per tap ONE fully coalesced wave load (8 neighbouring dy pixels x 128 bytes), the lane's three filter
// rows come from LDS, partial sums stay in the lane over all taps and are folded across the 8 chunk lanes once per pixel.
// Persistent workgroups (the 19 KB filter is staged once per workgroup).  HBM traffic: dy once + the image gradient.
template <int R, int S, int ST, bool WB>
__global__ __launch_bounds__(256) void fewc_dgrad_px_kernel(const FewCParams p, int nitems) {
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [R*S][8 chunks][3][8]
    constexpr int K = 64, NCH = 8;
    for (int i = threadIdx.x; i < R * S * 3 * K; i += 256) {
        const int k = i % K, tc = i / K, c = tc % 3, t = tc / 3;
        wl[((t * NCH + (k >> 3)) * 3 + c) * 8 + (k & 7)] = p.W[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int q = lane >> 3, ch = lane & 7;
    const int halfW = (p.Wd + ST - 1) / ST;
    const int segs = (halfW + 7) >> 3;
    for (int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); item < nitems; item += gridDim.x * 4) {
        const int seg = item % segs;
        const int t = item / segs;
        const int px = t % ST;
        const int row = t / ST;                            // n*H + iy
        const int n = row / p.H, iy = row - n * p.H;
        const int ix = (seg * 8 + q) * ST + px;
        const bool valid = ix < p.Wd;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        // taps that reach this parity class: ky = ky0, ky0 + ST, ...; kx likewise (at most NT x NT of them).  Taps that fall off the
        // filter or the dy image are gated to zero with a clamped address instead of a branch, so all NT*NT loads are issued together.
        constexpr int NT = (R + ST - 1) / ST;
        const int ky0 = (iy + p.pt) & (ST - 1), kx0 = (px + p.pl) & (ST - 1);
        uint4 raw[NT * NT];
        float gate[NT * NT];
        int tapi[NT * NT];
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int ky = ky0 + a * ST;
            const int ty = iy + p.pt - ky;                  // even by construction
            const bool oky = ky < R && ty >= 0 && (ty >> 1) < p.Ho;
            const long rbase = ((long)n * p.Ho + (oky ? (ty >> 1) : 0)) * p.Wo;
#pragma unroll
            for (int bq = 0; bq < NT; ++bq) {
                const int kx = kx0 + bq * ST;
                const int tx = ix + p.pl - kx;
                const int ox = tx >> 1;
                const bool ok = oky && valid && kx < S && tx >= 0 && ox < p.Wo;
                const long go = (rbase + (ok ? ox : 0)) * p.ldy + ch * 8;
                if constexpr (WB) raw[a * NT + bq] = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(p.DY) + go);
                else raw[a * NT + bq] = *reinterpret_cast<const uint4*>(static_cast<const float*>(p.DY) + go);   // (first 4 of 8)
                gate[a * NT + bq] = ok ? 1.f : 0.f;
                tapi[a * NT + bq] = ((ky < R ? ky : 0) * S + (kx < S ? kx : 0));
                if constexpr (!WB) {
                    // fp32 dy: second half of the 8 channels, consumed right away (keeps the register budget of the bf16 path)
                    const float4 hi = *reinterpret_cast<const float4*>(static_cast<const float*>(p.DY) + go + 4);
                    const float* w8 = wl + (tapi[a * NT + bq] * NCH + ch) * 24;
                    const float g = gate[a * NT + bq];
                    a0 += g * (hi.x * w8[4] + hi.y * w8[5] + hi.z * w8[6] + hi.w * w8[7]);
                    a1 += g * (hi.x * w8[12] + hi.y * w8[13] + hi.z * w8[14] + hi.w * w8[15]);
                    a2 += g * (hi.x * w8[20] + hi.y * w8[21] + hi.z * w8[22] + hi.w * w8[23]);
                }
            }
        }
#pragma unroll
        for (int t9 = 0; t9 < NT * NT; ++t9) {
            const float* w8 = wl + (tapi[t9] * NCH + ch) * 24;
            const float g = gate[t9];
            const uint4 u = raw[t9];
            if constexpr (WB) {
                float v[8];
                v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
                v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
                v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
                v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ve = v[e] * g;
                    a0 += ve * w8[e]; a1 += ve * w8[8 + e]; a2 += ve * w8[16 + e];
                }
            } else {
                const float v0 = __uint_as_float(u.x) * g, v1 = __uint_as_float(u.y) * g, v2 = __uint_as_float(u.z) * g, v3 = __uint_as_float(u.w) * g;
                a0 += v0 * w8[0] + v1 * w8[1] + v2 * w8[2] + v3 * w8[3];
                a1 += v0 * w8[8] + v1 * w8[9] + v2 * w8[10] + v3 * w8[11];
                a2 += v0 * w8[16] + v1 * w8[17] + v2 * w8[18] + v3 * w8[19];
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64);
        }
        if (valid && ch == 0) {
            float* o = p.DX + ((long)row * p.Wd + ix) * p.ldx;
            o[0] = a0; o[1] = a1; o[2] = a2;
        }
    }
}

// ---- forward on the matrix pipe for a bf16 input (C % 32 == 0): the vector-ALU kernel above needs 108 FMAs per pixel and lane, twice its
// HBM time.  Here D[16 x 16 px] += A[16 x 32 ch] * X[32 ch x 16 px] with v_mfma_f32_16x16x32_bf16 where the 16 rows of A are
// (output channel o, column tap kx) pairs: row = 4 * o + kx (kx = 3: zero).  One UNSHIFTED pixel fragment therefore serves all three
// column taps -- D holds, per pixel column c, the partial results P[o][kx][c] of filter row ky -- and the column shift is applied once
// per output row at the end: y[o][c] = P[o][0][c - 1] + P[o][1][c] + P[o][2][c + 1], i.e. two lane shifts inside the 16-lane rows
// (lane group o holds channel o, its accumulator registers 0..2 the three kx).  A wave owns a strip of 14 output columns (16 fragment
// columns x0 - 1 .. x0 + 14) x RS output rows and walks the RS + 2 input rows once; every input fragment (16 px x 32 ch, one 16-byte
// load per lane) feeds the three output rows it reaches (ky = 0, 1, 2).  The fp32 filter enters as two bf16 terms (hi + lo: 2 MFMAs) --
// x is bf16 already, so the products are those of the fp32 filter to 2^-17.  Per pixel: x read once (HBM-bound), 1/3 of the LDS
// fragment reads and MFMAs of the tap-by-tap formulation.
typedef __bf16 thin_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int T3_RS = 8, T3_OW = 14;
template <int KC>        // KC = C / 32
__global__ __launch_bounds__(256) void thin3_fwd_mfma_kernel(const ThinParams p, int nstrips_x, int nstrips_y, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) unsigned short wq[];       // [hi|lo][3 ky][KC][4 k-groups][16 rows][8]
    constexpr int FRAGS = 3 * KC * 4 * 16;
    // stage the filter: zero the image (rows with kx = 3 / o = 3 stay zero), then stream W ([9][C][3] fp32, contiguous) with 16-byte
    // loads and scatter each coefficient's two bf16 terms to its slot
    for (int i = threadIdx.x; i < FRAGS * 2; i += 256) reinterpret_cast<uint4*>(wq)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const int nw4 = 9 * p.C * TK / 4;                       // (C % 32 == 0: a multiple of 4)
    for (int q = threadIdx.x; q < nw4; q += 256) {
        const float4 w4 = *reinterpret_cast<const float4*>(p.W + q * 4);
        const float wv4[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = q * 4 + u;
            const int o = j % TK, tc = j / TK, ch = tc % p.C, tap = tc / p.C;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int idx = ((((ky * KC + (ch >> 5)) * 4 + ((ch >> 3) & 3)) * 16) + (o * 4 + kx)) * 8 + (ch & 7);
            const __bf16 hi = (__bf16)wv4[u];
            const __bf16 lo = (__bf16)(wv4[u] - (float)hi);
            wq[idx] = __builtin_bit_cast(unsigned short, hi);
            wq[FRAGS * 8 + idx] = __builtin_bit_cast(unsigned short, lo);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wv >= nwaves) return;
    const int sx = wv % nstrips_x;
    const int t = wv / nstrips_x;
    const int sy = t % nstrips_y, n = t / nstrips_y;
    const int x0 = sx * T3_OW, y0 = sy * T3_RS;
    const int col = lane & 15, kg = lane >> 4;
    const int xx = x0 - 1 + col;                            // this lane's pixel column
    const bool xok = (unsigned)xx < (unsigned)p.Wd;
    const unsigned short* xb = static_cast<const unsigned short*>(p.X);
    f32x4 acc[T3_RS];
#pragma unroll
    for (int i = 0; i < T3_RS; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const thin_bf16x8 zero8 = __builtin_bit_cast(thin_bf16x8, make_uint4(0u, 0u, 0u, 0u));
    const int abase = (kg * 16 + col) * 8;                  // this lane's slot inside a [4 k-groups][16 rows][8] fragment
#pragma unroll
    for (int ri = 0; ri < T3_RS + 2; ++ri) {
        const int r = y0 - 1 + ri;                          // input row
        const bool ok = xok && (unsigned)r < (unsigned)p.H;
        const unsigned short* src = xb + ((((long)n * p.H + (ok ? r : 0)) * p.Wd + (ok ? xx : 0)) * p.ldx + kg * 8);
        thin_bf16x8 bf[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const uint4 u = *reinterpret_cast<const uint4*>(src + kc * 32);
            bf[kc] = ok ? __builtin_bit_cast(thin_bf16x8, u) : zero8;
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int oi = ri - ky;                     // output row index within the strip: o = r + 1 - ky = y0 + (ri - ky)
                if (oi < 0 || oi >= T3_RS) continue;        // (compile-time after unrolling)
                const int fo = (ky * KC + kc) * 4 * 16 * 8 + abase;
                const thin_bf16x8 ah = *reinterpret_cast<const thin_bf16x8*>(wq + fo);
                const thin_bf16x8 al = *reinterpret_cast<const thin_bf16x8*>(wq + FRAGS * 8 + fo);
                acc[oi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bf[kc], acc[oi], 0, 0, 0);
                acc[oi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bf[kc], acc[oi], 0, 0, 0);
            }
        }
    }
    // lane group kg = output channel; registers 0..2 = column taps: y[c] = P0[c - 1] + P1[c] + P2[c + 1] (lane shifts inside the 16-lane row)
    const float bk = (p.bias && kg < TK) ? p.bias[kg] : 0.f;
    const float slope = (p.act == DPIG_ACT_NONE) ? 1.f : ((p.act == DPIG_ACT_RELU) ? 0.f : p.alpha);
    const bool writer = kg < TK && col >= 1 && col <= T3_OW && xx < p.Wd;
#pragma unroll
    for (int oi = 0; oi < T3_RS; ++oi) {
        const float left = __shfl_up(acc[oi][0], 1, 16);        // P0 of the pixel one column to the left  (lanes of one 16-lane row)
        const float right = __shfl_down(acc[oi][2], 1, 16);     // P2 of the pixel one column to the right
        const int o = y0 + oi;
        if (writer && o < p.H) {
            const float v = (left + acc[oi][1]) + right + bk;
            p.Y[(((long)n * p.H + o) * p.Wd + xx) * p.ldy + kg] = v > 0.f ? v : v * slope;
        }
    }
}

// ---- dgrad on the matrix pipe for a bf16 dx (C % 32 == 0): dx[p][c] = sum_{tap, o} dy[p - tap + 1][o] * w[tap][c][o] is a GEMM with a
// reduction of only 27 (padded to 32): ONE v_mfma_f32_16x16x32_bf16 per (16 channels x 16 pixels).  A = the filter as [16 channels x 32 k]
// tiles (k = 3 * tap + o) from LDS, B = the 27 dy values around each of 16 consecutive pixels of a row, gathered by the lanes themselves
// (dy is the 3-channel fp32 image gradient: 12 bytes per pixel, cached).  Both operands are fp32 in memory, so each enters as two bf16
// terms and a block is three MFMAs (hi*hi + hi*lo + lo*hi: relative error 2^-16, below dx's bf16 rounding).  The channel -> tile-row map is
// chosen so that a lane's results of two consecutive tiles are 8 consecutive channels: one 16-byte store.  HBM-bound: dx written once.
template <int CT>        // CT = C / 16 channel tiles
__global__ __launch_bounds__(256) void thin3_dgrad_mfma_kernel(const ThinParams p, int groups_per_row, int nitems) {
    extern __shared__ __attribute__((aligned(16))) unsigned short wd[];       // [hi|lo][CT tiles][4 k-groups][16 rows][8]
    constexpr int FR = CT * 4 * 16;
    for (int i = threadIdx.x; i < FR * 2; i += 256) reinterpret_cast<uint4*>(wd)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const int nw4 = 9 * p.C * TK / 4;
    for (int q = threadIdx.x; q < nw4; q += 256) {
        const float4 w4 = *reinterpret_cast<const float4*>(p.W + q * 4);
        const float wv4[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = q * 4 + u;
            const int o = j % TK, tc = j / TK, ch = tc % p.C, tap = tc / p.C;
            const int k = tap * 3 + o;                      // reduction index, < 27
            const int m = ch >> 5, within = ch & 31;
            const int tile = 2 * m + ((within >> 2) & 1), row = (within >> 3) * 4 + (within & 3);
            const int idx = (((tile * 4 + (k >> 3)) * 16) + row) * 8 + (k & 7);
            const __bf16 hi = (__bf16)wv4[u];
            const __bf16 lo = (__bf16)(wv4[u] - (float)hi);
            wd[idx] = __builtin_bit_cast(unsigned short, hi);
            wd[FR * 8 + idx] = __builtin_bit_cast(unsigned short, lo);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int col = lane & 15, kg = lane >> 4;
    const float* dy = p.DY;
    unsigned short* dx = static_cast<unsigned short*>(p.DX);
    const int abase = (kg * 16 + col) * 8;
    for (int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); item < nitems; item += gridDim.x * 4) {
        const int gx = item % groups_per_row;
        const int row = item / groups_per_row;              // n * H + y
        const int y = row % p.H;
        const int x = gx * 16 + col;
        // B fragment: k = kg * 8 + e -> (tap, o); the source pixel of tap (ky, kx) is (y - ky + 1, x - kx + 1)
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kg * 8 + e;
            const int tap = k / 3, o = k - tap * 3;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int sy = y - ky + 1, sx = x - kx + 1;
            const bool ok = k < 27 && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.Wd;
            v[e] = ok ? dy[((long)(row - y + sy) * p.Wd + sx) * p.ldy + o] : 0.f;
        }
        thin_bf16x8 bh, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __bf16 h = (__bf16)v[e];
            bh[e] = h;
            bl[e] = (__bf16)(v[e] - (float)h);
        }
        const bool xok = x < p.Wd;
        unsigned short* orow = dx + ((long)row * p.Wd + (xok ? x : 0)) * p.ldx + kg * 8;
#pragma unroll
        for (int m = 0; m < CT / 2; ++m) {
            f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = f32x4{0.f, 0.f, 0.f, 0.f};
            const thin_bf16x8 a0h = *reinterpret_cast<const thin_bf16x8*>(wd + (2 * m) * 4 * 16 * 8 + abase);
            const thin_bf16x8 a0l = *reinterpret_cast<const thin_bf16x8*>(wd + FR * 8 + (2 * m) * 4 * 16 * 8 + abase);
            const thin_bf16x8 a1h = *reinterpret_cast<const thin_bf16x8*>(wd + (2 * m + 1) * 4 * 16 * 8 + abase);
            const thin_bf16x8 a1l = *reinterpret_cast<const thin_bf16x8*>(wd + FR * 8 + (2 * m + 1) * 4 * 16 * 8 + abase);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bh, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bh, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bl, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bl, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, bh, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, bh, d1, 0, 0, 0);
            // lane (pixel col, k-group kg): tile 2m rows kg*4.. = channels 32m + 8kg + 0..3, tile 2m+1 = channels 32m + 8kg + 4..7
            if (xok)
                *reinterpret_cast<uint4*>(orow + m * 32) = make_uint4(thin_pack2(d0[0], d0[1]), thin_pack2(d0[2], d0[3]),
                                                                      thin_pack2(d1[0], d1[1]), thin_pack2(d1[2], d1[3]));
        }
    }
}

// ---- forward of the 3-channel stems on the matrix pipe for a bf16 output (K % 32 == 0): y[p][k] = sum_{tap, c} x[s*p + tap - pad][c] * w[tap][c][k]
// has a reduction of only R*S*3 (27 / 75): ceil(R*S*3 / 32) k-steps of v_mfma_f32_16x16x32_bf16 per (16 output channels x 16 pixels).
// A = the filter as [16 channels x 32 k] tiles from LDS, B = the image window of 16 consecutive output pixels of a row, gathered by the
// lanes (the fp32 image is 12 bytes per pixel, cached).  Image and filter are fp32: each enters as two bf16 terms, a block is three MFMAs
// (hi*hi + hi*lo + lo*hi, relative error 2^-16: below y's bf16 rounding).  Channel -> tile-row map as in thin3_dgrad_mfma_kernel: two
// tiles' results of a lane are 8 consecutive channels, one 16-byte store.  HBM-bound: y written once.
template <int R, int S, int ST, int KT>        // KT = K / 16 output-channel tiles
__global__ __launch_bounds__(256) void fewc_fwd_mfma_kernel(const FewCParams p, int groups_per_row, int nitems) {
    constexpr int RED = R * S * 3, KS = (RED + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned short wf[];       // [hi|lo][KT tiles][KS][4 k-groups][16 rows][8]
    constexpr int FR = KT * KS * 4 * 16;
    for (int i = threadIdx.x; i < FR * 2; i += 256) reinterpret_cast<uint4*>(wf)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const int K = KT * 16;
    for (int j = threadIdx.x; j < RED * K; j += 256) {                 // W: [R*S][3][K] fp32, k contiguous
        const int k = j % K, red = j / K;                               // red = tap * 3 + c
        const int m = k >> 5, within = k & 31;
        const int tile = 2 * m + ((within >> 2) & 1), row = (within >> 3) * 4 + (within & 3);
        const int idx = ((((tile * KS + (red >> 5)) * 4 + ((red >> 3) & 3)) * 16) + row) * 8 + (red & 7);
        const float wv = p.W[j];
        const __bf16 hi = (__bf16)wv;
        const __bf16 lo = (__bf16)(wv - (float)hi);
        wf[idx] = __builtin_bit_cast(unsigned short, hi);
        wf[FR * 8 + idx] = __builtin_bit_cast(unsigned short, lo);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int col = lane & 15, kg = lane >> 4;
    unsigned short* yb = static_cast<unsigned short*>(p.Y);
    const int abase = (kg * 16 + col) * 8;
    const float slope = (p.act == DPIG_ACT_NONE) ? 1.f : ((p.act == DPIG_ACT_RELU) ? 0.f : p.alpha);
    for (int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); item < nitems; item += gridDim.x * 4) {
        const int gx = item % groups_per_row;
        const int row = item / groups_per_row;              // n * Ho + oy
        const int n = row / p.Ho, oy = row - n * p.Ho;
        const int ox = gx * 16 + col;
        thin_bf16x8 bh[KS], bl[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int red = ks * 32 + kg * 8 + e;
                const int tap = red / 3, c = red - tap * 3;
                const int ky = tap / S, kx = tap - ky * S;
                const int iy = oy * ST - p.pt + ky, ix = ox * ST - p.pl + kx;
                const bool ok = red < RED && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                const float v = ok ? p.X[(((long)n * p.H + iy) * p.Wd + ix) * p.ldx + c] : 0.f;
                const __bf16 h = (__bf16)v;
                bh[ks][e] = h;
                bl[ks][e] = (__bf16)(v - (float)h);
            }
        }
        const bool xok = ox < p.Wo;
        unsigned short* orow = yb + ((long)row * p.Wo + (xok ? ox : 0)) * p.ldy + kg * 8;
#pragma unroll
        for (int m = 0; m < KT / 2; ++m) {
            f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int f0 = ((2 * m) * KS + ks) * 4 * 16 * 8 + abase, f1 = ((2 * m + 1) * KS + ks) * 4 * 16 * 8 + abase;
                const thin_bf16x8 a0h = *reinterpret_cast<const thin_bf16x8*>(wf + f0);
                const thin_bf16x8 a0l = *reinterpret_cast<const thin_bf16x8*>(wf + FR * 8 + f0);
                const thin_bf16x8 a1h = *reinterpret_cast<const thin_bf16x8*>(wf + f1);
                const thin_bf16x8 a1l = *reinterpret_cast<const thin_bf16x8*>(wf + FR * 8 + f1);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bh[ks], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bh[ks], d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, bl[ks], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, bl[ks], d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, bh[ks], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, bh[ks], d1, 0, 0, 0);
            }
            // lane (pixel col, k-group kg): tile 2m rows kg*4.. = channels 32m + 8kg + 0..3, tile 2m+1 = channels 32m + 8kg + 4..7
            float v8[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
            if (p.bias) {
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias + m * 32 + kg * 8), b1 = *reinterpret_cast<const float4*>(p.bias + m * 32 + kg * 8 + 4);
                v8[0] += b0.x; v8[1] += b0.y; v8[2] += b0.z; v8[3] += b0.w; v8[4] += b1.x; v8[5] += b1.y; v8[6] += b1.z; v8[7] += b1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = v8[e] > 0.f ? v8[e] : v8[e] * slope;
            if (xok)
                *reinterpret_cast<uint4*>(orow + m * 32) = make_uint4(thin_pack2(v8[0], v8[1]), thin_pack2(v8[2], v8[3]),
                                                                      thin_pack2(v8[4], v8[5]), thin_pack2(v8[6], v8[7]));
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
static int g_thin_mfma = []() { const char* e = getenv("DPIG_THIN_MFMA"); return e ? atoi(e) : 1; }();     // A/B switch
static bool eligible(const DpigConvDesc* d, int pt, int pl) {
    return d->K == TK && d->R == 3 && d->S == 3 && d->stride == 1 && !d->upsample2x && pt == 1 && pl == 1 &&
           d->C % 4 == 0 && d->C >= 16 && d->C <= 256 && d->ldx % 4 == 0;
}
static bool fill(const DpigConvDesc* d, ThinParams* p, bool wb) {
    p->N = d->N; p->H = d->H; p->Wd = d->W; p->C = d->C; p->ldx = d->ldx; p->ldy = d->ldy;
    p->nstrips = (d->W + XS - 1) / XS;
    p->nwaves = d->N * d->H * p->nstrips;
    const long xe = ((long)d->N * d->H * d->W - 1) * d->ldx + d->C;
    const long es = wb ? 2 : 4;
    if (xe * es >= 0x7fffffffL) return false;
    p->x_bytes = (unsigned)(xe * es);
    p->act = d->act; p->alpha = d->alpha;
    return true;
}

int thin_fwd_try(const DpigConvDesc* d, int pt, int pl, const void* x, const float* w, const float* bias,
                 const float* residual, float* y, float* y_act, hipStream_t st, bool wide_bf16) {
    if (!eligible(d, pt, pl) || residual || y_act || !aligned16(x)) return 0;
    ThinParams p = {};
    if (!fill(d, &p, wide_bf16)) return 0;
    p.X = x; p.W = w; p.bias = bias; p.Y = y;
    if (wide_bf16 && (d->C == 256 || d->C == 128 || d->C == 64) && d->ldx % 8 == 0 && aligned16(w) && g_thin_mfma) {
        const int nsx = (d->W + T3_OW - 1) / T3_OW, nsy = (d->H + T3_RS - 1) / T3_RS;
        const long nw = (long)d->N * nsx * nsy;
        if (nw < 0x7fffffffL) {
            const int kc = d->C / 32;
            const size_t lds = (size_t)2 * 3 * kc * 4 * 16 * 8 * sizeof(unsigned short);
            const dim3 grid((unsigned)((nw + 3) / 4));
            if (kc == 8) hipLaunchKernelGGL((thin3_fwd_mfma_kernel<8>), grid, dim3(256), lds, st, p, nsx, nsy, (int)nw);
            else if (kc == 4) hipLaunchKernelGGL((thin3_fwd_mfma_kernel<4>), grid, dim3(256), lds, st, p, nsx, nsy, (int)nw);
            else hipLaunchKernelGGL((thin3_fwd_mfma_kernel<2>), grid, dim3(256), lds, st, p, nsx, nsy, (int)nw);
            const int rcm = check_launch("thin3_fwd_mfma_kernel");
            return rcm ? rcm : 1;
        }
    }
    if (wide_bf16) hipLaunchKernelGGL(thin3_fwd_kernel<true>, dim3((p.nwaves + 3) / 4), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(thin3_fwd_kernel<false>, dim3((p.nwaves + 3) / 4), dim3(256), 0, st, p);
    const int rc = check_launch("thin3_fwd_kernel");
    return rc ? rc : 1;
}

int thin_dgrad_try(const DpigConvDesc* d, int pt, int pl, const float* dy, const float* w, const float* accum,
                   const float* mask, void* dx, hipStream_t st, bool wide_bf16) {
    if (!eligible(d, pt, pl) || accum || mask || !aligned16(dx)) return 0;
    ThinParams p = {};
    if (!fill(d, &p, wide_bf16)) return 0;
    p.DY = dy; p.W = w; p.DX = dx;
    if (wide_bf16 && (d->C == 256 || d->C == 128 || d->C == 64) && d->ldx % 8 == 0 && aligned16(w) && g_thin_mfma) {
        const int gpr = (d->W + 15) / 16;
        const long nit = (long)d->N * d->H * gpr;
        if (nit < 0x7fffffffL) {
            const int ct = d->C / 16;
            const size_t lds = (size_t)2 * ct * 4 * 16 * 8 * sizeof(unsigned short);
            const long nb = (nit + 3) / 4;
            const dim3 grid((unsigned)(nb < 4 * kNumCU ? nb : 4 * kNumCU));
            if (ct == 16) hipLaunchKernelGGL((thin3_dgrad_mfma_kernel<16>), grid, dim3(256), lds, st, p, gpr, (int)nit);
            else if (ct == 8) hipLaunchKernelGGL((thin3_dgrad_mfma_kernel<8>), grid, dim3(256), lds, st, p, gpr, (int)nit);
            else hipLaunchKernelGGL((thin3_dgrad_mfma_kernel<4>), grid, dim3(256), lds, st, p, gpr, (int)nit);
            const int rcm = check_launch("thin3_dgrad_mfma_kernel");
            return rcm ? rcm : 1;
        }
    }
    if (wide_bf16) hipLaunchKernelGGL(thin3_dgrad_kernel<true>, dim3((p.nwaves + 3) / 4), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(thin3_dgrad_kernel<false>, dim3((p.nwaves + 3) / 4), dim3(256), 0, st, p);
    const int rc = check_launch("thin3_dgrad_kernel");
    return rc ? rc : 1;
}

size_t thin_wgrad_workspace_bytes(const DpigConvDesc* d, int pt, int pl) {
    if (!eligible(d, pt, pl)) return 0;
    return (size_t)kWgradBlocks * (9 * (size_t)d->C * TK + TK) * sizeof(float);
}

int thin_wgrad_try(const DpigConvDesc* d, int pt, int pl, const void* x, const float* dy, float* dw, float beta,
                   float* db, float beta_b, void* ws, size_t ws_bytes, hipStream_t st, bool wide_bf16) {
    if (!eligible(d, pt, pl) || !aligned16(x)) return 0;
    ThinParams p = {};
    if (!fill(d, &p, wide_bf16)) return 0;
    const size_t need = thin_wgrad_workspace_bytes(d, pt, pl);
    if (!ws || ws_bytes < need) return fail(DPIG_ENOMEM, "conv wgrad workspace too small: have %zu", ws_bytes);
    p.X = x; p.DY = dy; p.partial = static_cast<float*>(ws);
    const int nblk = kWgradBlocks;
    if (wide_bf16) hipLaunchKernelGGL(thin3_wgrad_kernel<true>, dim3(nblk), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(thin3_wgrad_kernel<false>, dim3(nblk), dim3(256), 0, st, p);
    int rc = check_launch("thin3_wgrad_kernel");
    if (rc) return rc;
    const int wsize = 9 * d->C * TK;
    hipLaunchKernelGGL(thin3_wgrad_reduce_kernel, dim3((wsize + TK + 31) / 32), dim3(256), 0, st, p.partial, nblk, wsize,
                       dw, beta, db, beta_b);
    rc = check_launch("thin3_wgrad_reduce_kernel");
    return rc ? rc : 1;
}

// ---- three-input-channel layers ---------------------------------------------------------------------------------
static int fewc_kind(const DpigConvDesc* d) {        // 1: 3x3 s1, 2: 5x5 s2, 0: not eligible
    if (d->C != 3 || d->upsample2x || d->W + 5 > FC_MAXW || d->K % 4 != 0 || d->ldy % 4 != 0) return 0;
    const int lp = d->K / 4;
    if (lp < 1 || lp > 64 || (lp & (lp - 1)) != 0) return 0;        // lanes per pixel: a power of two
    if (d->R == 3 && d->S == 3 && d->stride == 1) return 1;
    if (d->R == 5 && d->S == 5 && d->stride == 2) return 2;
    return 0;
}
static void fewc_fill(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, FewCParams* p) {
    p->N = d->N; p->H = d->H; p->Wd = d->W; p->ldx = d->ldx; p->Ho = Ho; p->Wo = Wo; p->K = d->K; p->ldy = d->ldy;
    p->pt = pt; p->pl = pl; p->act = d->act; p->alpha = d->alpha; p->nrows = d->N * Ho;
}
constexpr int kFewCWgradBlocks = 2 * kNumCU;

int fewc_fwd_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const float* x, const float* w,
                 const float* bias, const float* residual, void* y, float* y_act, hipStream_t st, bool wide_bf16) {
    const int kind = fewc_kind(d);
    if (!kind || residual || y_act || !aligned16(w) || !aligned16(y) || (bias && !aligned16(bias))) return 0;
    FewCParams p = {};
    fewc_fill(d, pt, pl, Ho, Wo, &p);
    p.X = x; p.W = w; p.bias = bias; p.Y = y;
    const int R = kind == 1 ? 3 : 5;
    if (wide_bf16 && g_thin_mfma && (d->K == 64 || d->K == 128) && d->ldy % 8 == 0) {
        const int gpr = (Wo + 15) / 16;
        const long nit = (long)d->N * Ho * gpr;
        if (nit < 0x7fffffffL) {
            const int ks = (R * R * 3 + 31) / 32, kt = d->K / 16;
            const size_t ldsm = (size_t)2 * kt * ks * 4 * 16 * 8 * sizeof(unsigned short);
            const long nb = (nit + 3) / 4;
            const dim3 grid((unsigned)(nb < 4 * kNumCU ? nb : 4 * kNumCU));
            if (kind == 1 && kt == 8) hipLaunchKernelGGL((fewc_fwd_mfma_kernel<3, 3, 1, 8>), grid, dim3(256), ldsm, st, p, gpr, (int)nit);
            else if (kind == 1) hipLaunchKernelGGL((fewc_fwd_mfma_kernel<3, 3, 1, 4>), grid, dim3(256), ldsm, st, p, gpr, (int)nit);
            else if (kt == 8) hipLaunchKernelGGL((fewc_fwd_mfma_kernel<5, 5, 2, 8>), grid, dim3(256), ldsm, st, p, gpr, (int)nit);
            else hipLaunchKernelGGL((fewc_fwd_mfma_kernel<5, 5, 2, 4>), grid, dim3(256), ldsm, st, p, gpr, (int)nit);
            const int rcm = check_launch("fewc_fwd_mfma_kernel");
            return rcm ? rcm : 1;
        }
    }
    const size_t lds = ((size_t)R * R * 3 * d->K + (size_t)R * FC_MAXW * 3) * sizeof(float);
    if (kind == 1) {
        if (wide_bf16) hipLaunchKernelGGL((fewc_fwd_kernel<3, 3, 1, true>), dim3(p.nrows), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((fewc_fwd_kernel<3, 3, 1, false>), dim3(p.nrows), dim3(256), lds, st, p);
    } else {
        if (wide_bf16) hipLaunchKernelGGL((fewc_fwd_kernel<5, 5, 2, true>), dim3(p.nrows), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((fewc_fwd_kernel<5, 5, 2, false>), dim3(p.nrows), dim3(256), lds, st, p);
    }
    const int rc = check_launch("fewc_fwd_kernel");
    return rc ? rc : 1;
}

size_t fewc_wgrad_workspace_bytes(const DpigConvDesc* d) {
    if (!fewc_kind(d)) return 0;
    return (size_t)kFewCWgradBlocks * ((size_t)d->R * d->S * 3 * d->K + d->K) * sizeof(float);
}

int fewc_wgrad_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const float* x, const void* dy,
                   float* dw, float beta, float* db, float beta_b, void* ws, size_t ws_bytes, hipStream_t st,
                   bool wide_bf16) {
    const int kind = fewc_kind(d);
    if (!kind || !aligned16(dy)) return 0;
    const size_t need = fewc_wgrad_workspace_bytes(d);
    if (!ws || ws_bytes < need) return fail(DPIG_ENOMEM, "conv wgrad workspace too small: have %zu", ws_bytes);
    FewCParams p = {};
    fewc_fill(d, pt, pl, Ho, Wo, &p);
    p.X = x; p.DY = dy; p.partial = static_cast<float*>(ws);
    const int nblk = p.nrows < kFewCWgradBlocks ? p.nrows : kFewCWgradBlocks;
    if (kind == 1) {
        const size_t lds = ((size_t)3 * FC_MAXW * 3 + (size_t)(27 + 1) * d->K) * sizeof(float);
        if (wide_bf16) hipLaunchKernelGGL((fewc_wgrad_kernel<3, 3, 1, 9, true>), dim3(nblk, 1), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((fewc_wgrad_kernel<3, 3, 1, 9, false>), dim3(nblk, 1), dim3(256), lds, st, p);
    } else {
        const size_t lds = ((size_t)1 * FC_MAXW * 3 + (size_t)(15 + 1) * d->K) * sizeof(float);
        if (wide_bf16) hipLaunchKernelGGL((fewc_wgrad_kernel<5, 5, 2, 5, true>), dim3(nblk, 5), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((fewc_wgrad_kernel<5, 5, 2, 5, false>), dim3(nblk, 5), dim3(256), lds, st, p);
    }
    int rc = check_launch("fewc_wgrad_kernel");
    if (rc) return rc;
    const int wsize = d->R * d->S * 3 * d->K;
    hipLaunchKernelGGL(fewc_wgrad_reduce_kernel, dim3((wsize + d->K + 31) / 32), dim3(256), 0, st, p.partial, nblk, wsize,
                       d->K, dw, beta, db, beta_b);
    rc = check_launch("fewc_wgrad_reduce_kernel");
    return rc ? rc : 1;
}

int fewc_dgrad_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const void* dy, const float* w,
                   const float* accum, const float* mask, float* dx, hipStream_t st, bool wide_bf16) {
    if (fewc_kind(d) != 2 || d->K > 64 || accum || mask) return 0;
    FewCParams p = {};
    fewc_fill(d, pt, pl, Ho, Wo, &p);
    p.DY = dy; p.W = w; p.DX = dx;
    if (d->K == 64 && p.ldy % 8 == 0 && aligned16(dy)) {
        const int halfW = (d->W + 1) / 2, segs = (halfW + 7) / 8;
        const long nit = (long)d->N * d->H * 2 * segs;
        if (nit < 0x7fffffffL) {
            const int nitems = (int)nit;
            const size_t lds = (size_t)25 * 3 * 64 * sizeof(float);
            const int nblk = (nitems + 3) / 4 < 8 * kNumCU ? (nitems + 3) / 4 : 8 * kNumCU;
            if (wide_bf16) hipLaunchKernelGGL((fewc_dgrad_px_kernel<5, 5, 2, true>), dim3(nblk), dim3(256), lds, st, p, nitems);
            else hipLaunchKernelGGL((fewc_dgrad_px_kernel<5, 5, 2, false>), dim3(nblk), dim3(256), lds, st, p, nitems);
            const int rc = check_launch("fewc_dgrad_px_kernel");
            return rc ? rc : 1;
        }
    }
    const int nstrips = (d->W + XS - 1) / XS;
    const int nwaves = d->N * d->H * nstrips;
    if (wide_bf16) hipLaunchKernelGGL((fewc_dgrad_kernel<5, 5, 2, true>), dim3((nwaves + 3) / 4), dim3(256), 0, st, p, nstrips, nwaves);
    else hipLaunchKernelGGL((fewc_dgrad_kernel<5, 5, 2, false>), dim3((nwaves + 3) / 4), dim3(256), 0, st, p, nstrips, nwaves);
    const int rc = check_launch("fewc_dgrad_kernel");
    return rc ? rc : 1;
}

}  // namespace dpig
