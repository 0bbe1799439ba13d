// NHWC implicit-GEMM convolution family for gfx950 (MI355X), fp32 in / fp32 accumulate on the
// matrix cores (v_mfma_f32_32x32x2_f32: exact-f32 fmaf chain, 157 TFLOP/s chip peak).
//
//   fwd    y  = act(conv(x,w) + bias + residual)          "gather-GEMM", B = HWIO filter as [K][N]
//   dgrad  dx = (conv^T(dy,w) + accum) * act'(mask)       same kernel, B read as [N][K]
//   wgrad  dw = x^T (*) dy                                 pixel-reduction GEMM with split-K
//
// One workgroup = 256 threads = 4 waves (2x2), block tile 128(M) x 128(N) x 32(K); each wave owns
// a 64x64 sub-tile = 2x2 MFMA 32x32 accumulators (64 VGPRs).  Operand tiles are staged
// global -> registers -> LDS with a 2-deep LDS ring (loads for k-tile t+1 are issued before the
// MFMAs of k-tile t and written to LDS after them), one barrier per k-tile.
//
// The A operand is never materialised (no im2col): rows are output pixels, a k-tile is
// (filter tap, 32-channel chunk) and each row's 128 contiguous bytes are fetched straight from
// the NHWC activation.  Stride-2 dgrad is decomposed into the 4 output-parity classes so no MFMA
// work is spent on structural zeros; nearest-2x-upsample + 1x1 conv is computed at low
// resolution (the ops commute exactly) with a 2x2 replicating epilogue.
//
// Reference semantics: tf.nn.conv2d 'SAME' (tflib/ops/conv2d.py:106-112), slim.conv2d
// (models.py:396-573) and the gradients TF autodiff derives for them (trainer.py:137-140).
#include <type_traits>
#include "dpig_common.h"
#include "dpig_thin.h"
#include "dpig_conv_plan.h"
#ifdef DPIG_TRACE   // dev aid (never in the shipped build): s_memtime stamps of wave 0 of the first 512 workgroups
__device__ unsigned long long dpig_trace_buf[512 * 256];
__device__ unsigned long long dpig_trace_se[8192 * 4];     // start / end tick of every workgroup
extern "C" int dpig_debug_trace_read_se(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dpig_trace_se), sizeof(unsigned long long) * n);
}
#define DPIG_STAMP(slot) do { if (trace_on && trace_n < 256 && ((slot) == 0 || (slot) >= 5)) dpig_trace_buf[blockIdx.x * 256 + trace_n++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); } while (0)
extern "C" int dpig_debug_trace_read(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dpig_trace_buf), sizeof(unsigned long long) * n);
}
#else
#define DPIG_STAMP(slot) do { } while (0)
#endif

namespace dpig {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDR = BK + 4;        // row-major [row][k] LDS stride (floats): 16B aligned, conflict-free b128
constexpr int LDKN = BN;           // k-major  [k][col] LDS stride
constexpr int TILE_FLOATS = BM * LDR;   // 4608 floats >= 32*128

struct GGParams {
    const float* A;       // gathered source activation (x for fwd, dy for dgrad)
    const float* B;       // HWIO filter
    float* D;             // destination activation
    float* D2;            // optional second output: the activation BEFORE a post-activation residual add
    const float* bias;    // [Ncols] or null
    const float* res;     // residual / accumulate tensor (dest-shaped) or null
    const float* mask;    // activation-output tensor for act' (dest-shaped) or null
    float* partial;       // split-K workspace [nsplit][M][Ncols]
    float* stats;         // BN partial statistics [mtiles][2][Ncols] of v + bias (sum, centred squares per row tile) or null
    const unsigned short* Bs_hi;   // PIPE 3: bf16 hi plane of the filter's split shadow, rows n, k contiguous; the lo plane
    unsigned bs_lo_off, bs_bytes;  //         starts bs_lo_off bytes behind it; bs_bytes = extent of hi..lo for the descriptor
    const unsigned short* As32;    // PIPE 4: the gathered activation in split32 layout [pixel][chunk][32 hi | 32 lo] (dpig_split32)
    unsigned as_bytes;
    int a_nchunk;                  //         32-channel chunks per pixel
    unsigned short* D32;           // split32 image of D, written by the float4 epilogue (x3 mode: the next conv DMAs it) or null
    int d_nchunk;
    int M, Hr, Wr, HrWr;  // row grid (rows = images x Hr x Wr)
    int Hs, Ws, lda, Cs, sr;   // source spatial dims, channel stride, reduction channels, row->src stride
    int Ncols;            // GEMM N
    int Hd, Wd, ldd, dr, dpy, dpx;   // destination pixel = (r*dr+dpy, c*dr+dpx)
    int ldres, ldmask, ldd2;
    int res_post;         // 1: D = act(v + bias) + res  (reference res-blocks, models.py:400,427,536,566)
    int res_class;        // 1: res is [images][9][Ncols], indexed by the 3x3 border class of the output pixel
    int ntaps, cchunks, ktiles, tiles_per_split, nsplit;
    int mtiles, ntiles;
    int act; float alpha;
    int replicate;        // 1: write each result to the 2x2 block (nearest upsample)
    int identity_rows;    // 1: destination pixel index == row index
    // taps are an affine family (no table -> no dynamically indexed kernarg array):
    // tap t -> (a, b) = (t / tap_nb, t % tap_nb); source offset (oy0 + a*oys, ox0 + b*oxs);
    // filter slab w0 + a*wa + b*wb
    int tap_nb, oy0, oys, ox0, oxs, w0, wa, wb;
    unsigned a_bytes, b_bytes;   // byte extents of A and B for the buffer descriptors
    int vec_epi;          // 1: every epilogue operand is 16-byte addressable (float4 path)
    int red_lanes;        // split-K second pass: lanes sharing one float4 of the result (1, or 16 for few-tile many-split plans)
    int vec_a;            // 1: the gathered operand alone is 16-byte loadable (thin-N layers: B is not)
    unsigned mul_hrwr, shr_hrwr, mul_wr, shr_wr;   // magic numbers: row / (Hr*Wr) and rem / Wr without v_rcp sequences
};

// ------------------------------------------------------------------------------------------------
// Epilogue pieces.  Kept out of line and fed scalars only (no struct reference), so that the
// accumulator walk in the kernels unrolls fully (static accumulator indices -> no scratch).
__device__ __attribute__((noinline)) long row_to_pix(int row, int HrWr, int Wr, int Hd, int Wd, int dr, int dpy,
                                                     int dpx) {
    const int n = row / HrWr;
    const int rem = row - n * HrWr;
    const int r = rem / Wr;
    const int c = rem - r * Wr;
    return ((long)n * Hd + (r * dr + dpy)) * Wd + (c * dr + dpx);
}

__device__ __attribute__((noinline)) void epi_store(float* __restrict__ D, const float* __restrict__ bias,
                                                    const float* __restrict__ res,
                                                    const float* __restrict__ mask, long pix, int col, float v,
                                                    int ldd, int ldres, int ldmask, int act, float alpha,
                                                    int replicate, int Wd, float* __restrict__ D2, int ldd2,
                                                    int res_post, long rpix) {
    if (bias) v += bias[col];
    if (!replicate) {
        if (res && !res_post) v += res[rpix * ldres + col];
        if (mask) v *= act_grad(mask[pix * ldmask + col], act, alpha);
        else v = act_apply(v, act, alpha);
        if (D2) D2[pix * ldd2 + col] = v;
        if (res && res_post) v += res[rpix * ldres + col];
        D[pix * ldd + col] = v;
    } else {
        v = act_apply(v, act, alpha);
        D[pix * ldd + col] = v;
        D[(pix + 1) * ldd + col] = v;
        D[(pix + Wd) * ldd + col] = v;
        D[(pix + Wd + 1) * ldd + col] = v;
    }
}

// n / d for 0 <= n < 2^31 with a precomputed (mul, shr); mul == 0 encodes d == 1
__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned shr) {
    return mul ? (int)(__umulhi((unsigned)n, mul) >> shr) : n;
}

// ---- buffer-descriptor loads: 32-bit byte offsets, hardware bounds check -----------------------
// Every gathered element is fetched with buffer_load through an SRD whose num_records is the exact
// byte extent of the tensor.  A lane that must read a structural zero (padding halo, row >= M,
// channel >= C) is simply given the offset OOB (> num_records): the hardware returns 0 -- no
// branches, no 64-bit address math, no select after the load.
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x7fffffffu;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
template <bool VEC>
__device__ __forceinline__ float4 gload4(__amdgpu_buffer_rsrc_t r, unsigned off, bool ok, int first, int limit) {
    // VEC: one 16-byte load (caller folded `first < limit` into ok); else 4 dword loads, element e valid
    // iff first + e < limit
    float4 v;
    if (VEC) {
        // NB: bit_cast the WHOLE vector; element-wise bit_casts of the builtin's result make hipcc
        // (ROCm 7.2) emit a single dword load and splat it
        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(ok ? off : OOB), 0, 0));
        v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w;
    } else {
        v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)((ok & (first + 0 < limit)) ? off + 0 : OOB), 0, 0));
        v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)((ok & (first + 1 < limit)) ? off + 4 : OOB), 0, 0));
        v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)((ok & (first + 2 < limit)) ? off + 8 : OOB), 0, 0));
        v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)((ok & (first + 3 < limit)) ? off + 12 : OOB), 0, 0));
    }
    return v;
}

// 3x3 border class of pixel (y, x) in an H x W image: (top|interior|bottom) x (left|interior|right).
// A SAME 3x3 conv over a spatially constant input only depends on this class.
__device__ __forceinline__ int border_class(int y, int x, int H, int W) {
    const int cy = (y == 0) ? 0 : ((y == H - 1) ? 2 : 1);
    const int cx = (x == 0) ? 0 : ((x == W - 1) ? 2 : 1);
    return cy * 3 + cx;
}
__device__ __forceinline__ long class_row(int row, int HrWr, int Wr, int Hr) {
    const int n = row / HrWr;
    const int rem = row - n * HrWr;
    const int y = rem / Wr;
    return (long)n * 9 + border_class(y, rem - y * Wr, Hr, Wr);
}

// Fused epilogue on 4 consecutive columns of one output row (16-byte accesses throughout).
// The value's two bf16 terms as the split k-loops round them (hi = bf16(v), lo = bf16(v - hi)), 4 channels of one pixel into
// its split32 image [pixel][chunk][32 hi | 32 lo]: what dpig_split32 would write for D, without reading D again.
__device__ __forceinline__ void store_s32(const GGParams& p, long pix, int col, float4 v) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 h0, h1, l0, l1;
    h0[0] = (__bf16)v.x; h0[1] = (__bf16)v.y; h1[0] = (__bf16)v.z; h1[1] = (__bf16)v.w;
    l0[0] = (__bf16)(v.x - (float)h0[0]); l0[1] = (__bf16)(v.y - (float)h0[1]);
    l1[0] = (__bf16)(v.z - (float)h1[0]); l1[1] = (__bf16)(v.w - (float)h1[1]);
    unsigned short* o = p.D32 + (pix * p.d_nchunk + (col >> 5)) * 64 + (col & 31);
    *reinterpret_cast<uint2*>(o) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
    *reinterpret_cast<uint2*>(o + 32) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}
__device__ __forceinline__ void epi_vec4(const GGParams& p, int row, int col, float4 v, float4 bv) {
    long pix = row;
    if (!p.identity_rows) {
        const int n = row / p.HrWr;
        const int rem = row - n * p.HrWr;
        const int rr = rem / p.Wr;
        const int cc = rem - rr * p.Wr;
        pix = ((long)n * p.Hd + (rr * p.dr + p.dpy)) * p.Wd + (cc * p.dr + p.dpx);
    }
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.res) {
        const long rpix = p.res_class ? class_row(row, p.HrWr, p.Wr, p.Hr) : pix;
        rv = *reinterpret_cast<const float4*>(p.res + rpix * p.ldres + col);
    }
    if (p.res && !p.res_post) { v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
    if (p.mask) {
        const float4 mv = *reinterpret_cast<const float4*>(p.mask + pix * p.ldmask + col);
        v.x *= act_grad(mv.x, p.act, p.alpha); v.y *= act_grad(mv.y, p.act, p.alpha);
        v.z *= act_grad(mv.z, p.act, p.alpha); v.w *= act_grad(mv.w, p.act, p.alpha);
    } else {
        v.x = act_apply(v.x, p.act, p.alpha); v.y = act_apply(v.y, p.act, p.alpha);
        v.z = act_apply(v.z, p.act, p.alpha); v.w = act_apply(v.w, p.act, p.alpha);
    }
    if (p.D2) *reinterpret_cast<float4*>(p.D2 + pix * p.ldd2 + col) = v;
    if (p.res && p.res_post) { v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
    *reinterpret_cast<float4*>(p.D + pix * p.ldd + col) = v;
    if (p.D32) store_s32(p, pix, col, v);
    if (p.replicate) {
        *reinterpret_cast<float4*>(p.D + (pix + 1) * p.ldd + col) = v;
        *reinterpret_cast<float4*>(p.D + (pix + p.Wd) * p.ldd + col) = v;
        *reinterpret_cast<float4*>(p.D + (pix + p.Wd + 1) * p.ldd + col) = v;
    }
}

// ---- bf16 matrix-pipe variant of the k loop (DpigConvDesc.compute = DPIG_COMPUTE_BF16; BASELINE configs 3-5) ----
// Tensors stay fp32 in HBM; operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on their way into LDS and fed
// to v_mfma_f32_32x32x16_bf16 (fp32 accumulate, same accumulator layout as the fp32 MFMA, so the epilogue is
// shared).  k-tile = 64: the same 73.7 KB LDS ring holds [128][64] bf16 per operand, rows padded to 144 bytes
// (conflict-free ds_read_b128 fragments: lane = row, 8 consecutive k per lane).  The forward filter is k-major in
// HBM ([k][n]); each thread loads an 8(k) x 4(n) patch and transposes it in registers into four 16-byte LDS rows.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
constexpr int BKH = 64;              // k-tile of the bf16 loop
constexpr int ROWB = 144;            // bytes per LDS row (64 bf16 + 16 pad)
constexpr int TILEB = 128 * ROWB;    // one operand tile
template <bool B_ROWK>
__device__ __forceinline__ void gg_mainloop_bf16(const GGParams& p, char* lds, f32x16 (&acc)[2][2], int m0, int n0,
                                                 int kt_begin, int kt_end, int tid, int wrow, int wcol, int l31,
                                                 int half) {
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B, p.b_bytes);
    const int kq = tid & 15;                                  // 4-float k group of the row-major operands
    unsigned a_rowoff[8];
    int a_iy0[8], a_ix0[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + (tid >> 4) + 16 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = fast_div(mm, p.mul_hrwr, p.shr_hrwr);
        const int rem = mm - n * p.HrWr;
        const int r = fast_div(rem, p.mul_wr, p.shr_wr);
        const int c = rem - r * p.Wr;
        a_iy0[i] = ok ? r * p.sr : -(1 << 24);
        a_ix0[i] = c * p.sr;
        a_rowoff[i] = (unsigned)((((n * p.Hs + r * p.sr) * p.Ws + c * p.sr) * p.lda + kq * 4) * 4);
    }
    unsigned b_off[8];
    bool b_ok[8];
    const int nq = tid & 31, koct = tid >> 5;                 // fwd filter patch: 4 columns x 8 k rows
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (B_ROWK) {
            const int n = n0 + (tid >> 4) + 16 * i;
            b_ok[i] = n < p.Ncols;
            b_off[i] = (unsigned)((n * p.Cs + kq * 4) * 4);
        } else {
            b_ok[i] = n0 + nq * 4 < p.Ncols;
            b_off[i] = (unsigned)((((koct * 8 + i) * p.Ncols) + n0 + nq * 4) * 4);
        }
    }
    int cur_c0, cur_ta, cur_tb;
    {
        const int tap = kt_begin / p.cchunks;
        cur_c0 = (kt_begin - tap * p.cchunks) * BKH;
        cur_ta = tap / p.tap_nb;
        cur_tb = tap - cur_ta * p.tap_nb;
    }
    float4 ra[8], rb[8];
    auto load_tile = [&](bool live) {
        const int c0 = cur_c0, ta = cur_ta, tb = cur_tb;
        cur_c0 += BKH;
        if (cur_c0 >= p.Cs) {
            cur_c0 = 0;
            if (++cur_tb == p.tap_nb) { cur_tb = 0; ++cur_ta; }
        }
        const int wt = p.w0 + ta * p.wa + tb * p.wb;
        const int t_oy = p.oy0 + ta * p.oys, t_ox = p.ox0 + tb * p.oxs;
        const int t_ck = c0 + kq * 4;
        const bool t_kok = t_ck < p.Cs;
        const unsigned t_sA = (unsigned)(((t_oy * p.Ws + t_ox) * p.lda + c0) * 4);
        const unsigned t_sB = B_ROWK ? (unsigned)((wt * p.Ncols * p.Cs + c0) * 4) : (unsigned)(((wt * p.Cs + c0) * p.Ncols) * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = live & t_kok & ((unsigned)(a_iy0[i] + t_oy) < (unsigned)p.Hs) &
                            ((unsigned)(a_ix0[i] + t_ox) < (unsigned)p.Ws);
            ra[i] = gload4<true>(rsA, a_rowoff[i] + t_sA, ok, t_ck, p.Cs);
            if (B_ROWK) rb[i] = gload4<true>(rsB, b_off[i] + t_sB, live & b_ok[i] & t_kok, t_ck, p.Cs);
            else rb[i] = gload4<true>(rsB, b_off[i] + t_sB, live & b_ok[i] & (c0 + koct * 8 + i < p.Cs), 0, 4);
        }
    };
    auto store_tile = [&](int buf) {
        char* As = lds + buf * 2 * TILEB;
        char* Bs = As + TILEB;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (tid >> 4) + 16 * i;
            *reinterpret_cast<uint2*>(As + r * ROWB + kq * 8) = make_uint2(pack_bf16(ra[i].x, ra[i].y), pack_bf16(ra[i].z, ra[i].w));
            if (B_ROWK)
                *reinterpret_cast<uint2*>(Bs + r * ROWB + kq * 8) = make_uint2(pack_bf16(rb[i].x, rb[i].y), pack_bf16(rb[i].z, rb[i].w));
        }
        if (!B_ROWK) {      // register transpose of the 8(k) x 4(n) patch -> 4 rows n of 8 consecutive k
            char* d = Bs + (nq * 4) * ROWB + koct * 16;
            *reinterpret_cast<uint4*>(d + 0 * ROWB) = make_uint4(pack_bf16(rb[0].x, rb[1].x), pack_bf16(rb[2].x, rb[3].x), pack_bf16(rb[4].x, rb[5].x), pack_bf16(rb[6].x, rb[7].x));
            *reinterpret_cast<uint4*>(d + 1 * ROWB) = make_uint4(pack_bf16(rb[0].y, rb[1].y), pack_bf16(rb[2].y, rb[3].y), pack_bf16(rb[4].y, rb[5].y), pack_bf16(rb[6].y, rb[7].y));
            *reinterpret_cast<uint4*>(d + 2 * ROWB) = make_uint4(pack_bf16(rb[0].z, rb[1].z), pack_bf16(rb[2].z, rb[3].z), pack_bf16(rb[4].z, rb[5].z), pack_bf16(rb[6].z, rb[7].z));
            *reinterpret_cast<uint4*>(d + 3 * ROWB) = make_uint4(pack_bf16(rb[0].w, rb[1].w), pack_bf16(rb[2].w, rb[3].w), pack_bf16(rb[4].w, rb[5].w), pack_bf16(rb[6].w, rb[7].w));
        }
    };
    if (kt_begin >= kt_end) return;
    load_tile(true);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    bf16x8 fa[2][2], fb[2][2];                              // [k-step parity][32-row / 32-col block]
    auto load_frag = [&](int b, int ks, bf16x8 (&a)[2], bf16x8 (&bb)[2]) {
        const char* As = lds + b * 2 * TILEB;
        const char* Bs = As + TILEB;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
            a[mb] = *reinterpret_cast<const bf16x8*>(As + (wrow + mb * 32 + l31) * ROWB + ks * 32 + half * 16);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            bb[nb] = *reinterpret_cast<const bf16x8*>(Bs + (wcol + nb * 32 + l31) * ROWB + ks * 32 + half * 16);
    };
    load_frag(0, 0, fa[0], fb[0]);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = (kt + 1) < kt_end;
        load_tile(more);                                   // tile t+1 in flight under this tile's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < BKH / 16; ++ks) {
            if (ks + 1 < BKH / 16) {
                load_frag(buf, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);    // fragments of the next k-step
            } else {
                // last k-step: publish tile t+1 first (its loads have had three k-steps to land), then request the
                // first fragments of t+1; the 4 MFMAs below cover the barrier and the LDS latency
                store_tile(buf ^ 1);                        // (zeros after the last tile: nobody reads them)
                __syncthreads();
                load_frag(buf ^ 1, 0, fa[0], fb[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][mb], fb[ks & 1][nb], acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        buf ^= 1;
    }
}

// ---- split-bf16 variant of the k loop (DpigConvDesc.compute = DPIG_COMPUTE_BF16X3): fp32 accuracy class on the bf16 pipe ----
// Tensors stay fp32 in HBM.  On its way into LDS every operand value v is split into two bf16 terms, hi = bf16(v) and
// lo = bf16(v - hi) (v - hi is exact in fp32; hi + lo carries 16 significand bits, |v - hi - lo| <= 2^-18 |v|), and each
// 16-deep k-step issues THREE v_mfma_f32_32x32x16_bf16 into the same fp32 accumulator: a_hi b_lo + a_lo b_hi + a_hi b_hi
// (the a_lo b_lo term, <= 2^-18 of the product, is dropped).  Products of bf16 numbers are exact in fp32, so the result
// differs from the fp32 pipe's by the operand truncation only: measured <= 5e-6 max|ref| on random-sign data (tests hold
// it to the same 2e-5 max|ref| bar as the exact path) at 3/16 of the fp32 pipe's matrix cycles per flop.
// k-tile = 32 (the fp32 loop's: same split-K plans, same workspace); an LDS row is [32 hi | 32 lo] bf16 = 128 B + 16 pad,
// i.e. the ROWB / TILEB image of the bf16 loop, fragments are ds_read_b128 (8 consecutive k per lane) from either half.
constexpr int BKS = 32;
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pack_bf16(a, b);
    lo = pack_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ void split_store4(char* dst, float a, float b, float c, float d) {   // 4 consecutive k of one row
    unsigned h0, l0, h1, l1;
    split_pair(a, b, h0, l0);
    split_pair(c, d, h1, l1);
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + 64) = make_uint2(l0, l1);
}
// An operand that reaches LDS through the register transpose (the forward filter, both wgrad operands) is written by 32 lanes
// that hold 32 different 4-row groups at ONE k position: with plain rows their 8-byte stores fall on 4 bank groups
// (8-way conflict).  Those tiles keep the 16-byte granule g of row r at position g ^ ((r >> 4) & 3) (2-way); SWA / SWB
// tell the fragment reads which operand is stored that way.
__device__ __forceinline__ int split_tslot(int slot8, int rowquad) { return slot8 ^ (((rowquad >> 2) & 3) << 1); }
struct SplitFrag { bf16x8 ah[2], al[2], bh[2], bl[2]; };
template <bool SWA, bool SWB, bool BDMA = false>
__device__ __forceinline__ void split_load_frag(const char* As, const char* Bs, int ks, int wrow, int wcol, int l31, int half,
                                                SplitFrag& f) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int g = (ks * 2 + half) ^ (SWA ? ((2 * mb + (l31 >> 4)) & 3) : 0);     // wrow, wcol are multiples of 64
        const char* a = As + (wrow + mb * 32 + l31) * ROWB + g * 16;
        f.ah[mb] = *reinterpret_cast<const bf16x8*>(a);
        f.al[mb] = *reinterpret_cast<const bf16x8*>(a + 64);
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        if (BDMA) {          // LDS-DMA image: 128-byte rows, slot = granule ^ ((row >> 1) & 7); the lo granule is 4 slots away
            const int slot = (ks * 2 + half) ^ ((l31 >> 1) & 7);
            const char* b = Bs + (wcol + nb * 32 + l31) * 128;
            f.bh[nb] = *reinterpret_cast<const bf16x8*>(b + slot * 16);
            f.bl[nb] = *reinterpret_cast<const bf16x8*>(b + (slot ^ 4) * 16);
        } else {
            const int g = (ks * 2 + half) ^ (SWB ? ((2 * nb + (l31 >> 4)) & 3) : 0);
            const char* b = Bs + (wcol + nb * 32 + l31) * ROWB + g * 16;
            f.bh[nb] = *reinterpret_cast<const bf16x8*>(b);
            f.bl[nb] = *reinterpret_cast<const bf16x8*>(b + 64);
        }
    }
}
__device__ __forceinline__ void split_mfma(const SplitFrag& f, f32x16 (&acc)[2][2]) {
    // the two cross terms first, the leading term last; the same accumulator is touched every 4th instruction
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mb], f.bl[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[mb], f.bh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[mb], f.bh[nb], acc[mb][nb], 0, 0, 0);
}
// B_DMA (PIPE 3): the FILTER operand's split is taken out of the loop.  Its hi / lo bf16 planes are precomputed once per
// optimizer step (dpig_filter_shadow_split*: rows n, k contiguous -- the transposed shadow for forward, the plain one for
// dgrad, so both are the B_ROWK form) and each wave fills its share of the [128][32 hi | 32 lo] tile with four 1-KB LDS-DMA
// pieces: no VGPRs, no VALU, no ds_write for B (a knock-out build with a free filter operand measured +31 % on forward /
// dgrad).  The DMA image is lane-linear, so the B tile has unpadded 128-byte rows and the bank-conflict swizzle lives in the
// SOURCE address: 16-byte slot s of row r holds granule s ^ ((r >> 1) & 7) (0-3 = hi k-granules, 4-7 = lo).
typedef __attribute__((address_space(3))) void lds_void_t;
template <bool B_ROWK, bool B_DMA = false>
__device__ __forceinline__ void gg_mainloop_split(const GGParams& p, char* lds, f32x16 (&acc)[2][2], int m0, int n0,
                                                  int kt_begin, int kt_end, int tid, int wrow, int wcol, int l31,
                                                  int half) {
    static_assert(!B_DMA || B_ROWK, "split shadows are stored rows n, k contiguous");
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rsB = B_DMA ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Bs_hi), 0, (int)p.bs_bytes, 0x00020000)
                                             : make_rsrc(p.B, p.b_bytes);
    const int kq = tid & 7;                                   // 4-float k group of the row-major operands
    unsigned a_rowoff[4];
    int a_iy0[4], a_ix0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = fast_div(mm, p.mul_hrwr, p.shr_hrwr);
        const int rem = mm - n * p.HrWr;
        const int r = fast_div(rem, p.mul_wr, p.shr_wr);
        const int c = rem - r * p.Wr;
        a_iy0[i] = ok ? r * p.sr : -(1 << 24);
        a_ix0[i] = c * p.sr;
        a_rowoff[i] = (unsigned)((((n * p.Hs + r * p.sr) * p.Ws + c * p.sr) * p.lda + kq * 4) * 4);
    }
    unsigned b_off[4];
    bool b_ok[4];
    const int nq = tid & 31, kgrp = tid >> 5;                 // fwd filter patch: 4 columns x 4 k rows
    int d_kk[4] = {0, 0, 0, 0};                               // B_DMA: this lane's k-granule start inside a tile, per piece
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (B_DMA) {            // piece 4 wave + i = tile rows 8 piece .. + 7; lane -> (row, 16-byte slot)
            const int lane = tid & 63;
            const int r = ((tid >> 6) * 4 + i) * 8 + (lane >> 3);
            const int gs = (lane & 7) ^ ((r >> 1) & 7);
            d_kk[i] = (gs & 3) * 8;
            b_ok[i] = n0 + r < p.Ncols;
            b_off[i] = (unsigned)(((n0 + r) * p.Cs + d_kk[i]) * 2) + ((gs >> 2) ? p.bs_lo_off : 0u);
        } else if (B_ROWK) {
            const int n = n0 + (tid >> 3) + 32 * i;
            b_ok[i] = n < p.Ncols;
            b_off[i] = (unsigned)((n * p.Cs + kq * 4) * 4);
        } else {
            b_ok[i] = n0 + nq * 4 < p.Ncols;
            b_off[i] = (unsigned)((((kgrp * 4 + i) * p.Ncols) + n0 + nq * 4) * 4);
        }
    }
    int cur_c0, cur_ta, cur_tb;
    {
        const int tap = kt_begin / p.cchunks;
        cur_c0 = (kt_begin - tap * p.cchunks) * BKS;
        cur_ta = tap / p.tap_nb;
        cur_tb = tap - cur_ta * p.tap_nb;
    }
    float4 ra[4], rb[4];
    const int dma_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The register-staged operand(s) and the DMA'd filter tile of the SAME k-tile are requested at different moments (below),
    // so each keeps its own (tap, channel-chunk) cursor.
    struct Cursor { int c0, ta, tb; };
    Cursor curA = {cur_c0, cur_ta, cur_tb}, curB = curA;
    auto advance = [&](Cursor& c) {
        c.c0 += BKS;
        if (c.c0 >= p.Cs) {
            c.c0 = 0;
            if (++c.tb == p.tap_nb) { c.tb = 0; ++c.ta; }
        }
    };
    auto load_regs = [&](bool live) {                       // global -> VGPRs: A, and B unless it is DMA'd
        const int c0 = curA.c0, ta = curA.ta, tb = curA.tb;
        advance(curA);
        const int wt = p.w0 + ta * p.wa + tb * p.wb;
        const int t_oy = p.oy0 + ta * p.oys, t_ox = p.ox0 + tb * p.oxs;
        const int t_ck = c0 + kq * 4;
        const bool t_kok = t_ck < p.Cs;
        const unsigned t_sA = (unsigned)(((t_oy * p.Ws + t_ox) * p.lda + c0) * 4);
        const unsigned t_sB = B_ROWK ? (unsigned)((wt * p.Ncols * p.Cs + c0) * 4) : (unsigned)(((wt * p.Cs + c0) * p.Ncols) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = live & t_kok & ((unsigned)(a_iy0[i] + t_oy) < (unsigned)p.Hs) &
                            ((unsigned)(a_ix0[i] + t_ox) < (unsigned)p.Ws);
            ra[i] = gload4<true>(rsA, a_rowoff[i] + t_sA, ok, t_ck, p.Cs);
            if (B_DMA) continue;
            if (B_ROWK) rb[i] = gload4<true>(rsB, b_off[i] + t_sB, live & b_ok[i] & t_kok, t_ck, p.Cs);
            else rb[i] = gload4<true>(rsB, b_off[i] + t_sB, live & b_ok[i] & (c0 + kgrp * 4 + i < p.Cs), 0, 4);
        }
    };
    auto dma_b = [&](bool live, int bufn) {                 // B_DMA: four 1-KB pieces of the filter tile straight into LDS
        const int c0 = curB.c0;
        const int wt = p.w0 + curB.ta * p.wa + curB.tb * p.wb;
        advance(curB);
        const unsigned t_sB = (unsigned)((wt * p.Ncols * p.Cs + c0) * 2);
        char* bdst = lds + bufn * 2 * TILEB + TILEB + dma_wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = live & b_ok[i] & (c0 + d_kk[i] < p.Cs);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(bdst + i * 1024), 16, (int)(ok ? b_off[i] : OOB), (int)t_sB, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
        char* As = lds + buf * 2 * TILEB;
        char* Bs = As + TILEB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 3) + 32 * i;
            split_store4(As + r * ROWB + kq * 8, ra[i].x, ra[i].y, ra[i].z, ra[i].w);
            if (B_ROWK && !B_DMA) split_store4(Bs + r * ROWB + kq * 8, rb[i].x, rb[i].y, rb[i].z, rb[i].w);
        }
        if (!B_ROWK) {      // register transpose of the 4(k) x 4(n) patch -> 4 rows n of 4 consecutive k
            char* d = Bs + (nq * 4) * ROWB + split_tslot(kgrp, nq) * 8;
            split_store4(d + 0 * ROWB, rb[0].x, rb[1].x, rb[2].x, rb[3].x);
            split_store4(d + 1 * ROWB, rb[0].y, rb[1].y, rb[2].y, rb[3].y);
            split_store4(d + 2 * ROWB, rb[0].z, rb[1].z, rb[2].z, rb[3].z);
            split_store4(d + 3 * ROWB, rb[0].w, rb[1].w, rb[2].w, rb[3].w);
        }
    };
    if (kt_begin >= kt_end) return;
    // Pipeline: a k-tile's operands are requested a FULL tile period before they are published: right behind the barrier in
    // the middle of tile t (which retires the LDS buffer the filter DMA of tile t+2 lands in) and consumed by the store in
    // the middle of tile t+1.  (Requested at the head of tile t and consumed in its middle, i.e. 12 MFMAs later, the loads
    // were still on their way: the wait at the store was the longest segment of the loop.)  Nothing is in flight at a
    // barrier: the store has consumed the register loads, and the DMA pieces, issued before them, return before them.
    if (B_DMA) dma_b(true, 0);
    load_regs(true);
    store_tile(0);
    if (B_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (B_DMA) dma_b(kt_begin + 1 < kt_end, 1);
    load_regs(kt_begin + 1 < kt_end);
    int buf = 0;
    SplitFrag f0, f1;
    split_load_frag<false, !B_ROWK, B_DMA>(lds, lds + TILEB, 0, wrow, wcol, l31, half, f0);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more2 = (kt + 2) < kt_end;
        const char* As = lds + buf * 2 * TILEB;
        split_load_frag<false, !B_ROWK, B_DMA>(As, As + TILEB, 1, wrow, wcol, l31, half, f1);
        __builtin_amdgcn_sched_barrier(0);
        split_mfma(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        // second k-step: publish tile t+1 (requested a tile ago), retire buffer t behind the barrier, request tile t+2 (its
        // filter DMA lands in buffer t); the 12 MFMAs below cover the barrier and the LDS latency (zeros after the last
        // tile: nobody reads them)
        store_tile(buf ^ 1);
        if (B_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (B_DMA) dma_b(more2, buf);
        load_regs(more2);
        split_load_frag<false, !B_ROWK, B_DMA>(lds + (buf ^ 1) * 2 * TILEB, lds + (buf ^ 1) * 2 * TILEB + TILEB, 0, wrow, wcol, l31, half, f0);
        __builtin_amdgcn_sched_barrier(0);
        split_mfma(f1, acc);
        __builtin_amdgcn_sched_barrier(0);
        buf ^= 1;
    }
    if (B_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (dead pieces of the tiles past the end)
}

// ---- PIPE 4: split-bf16 with BOTH operands by LDS-DMA ---------------------------------------------------------------------
// The activation's two-term split is taken out of the loop as well: dpig_split32 writes it once per tensor as
// [pixel][32-channel chunk][32 hi | 32 lo] bf16 (4 bytes per element, like the fp32 tensor it mirrors; zero-padded to whole
// chunks), so that one 128-byte tile row is ONE contiguous 128-byte read and a k-tile (tap, chunk) is a scalar offset.  The
// k-loop then is the bf16-storage loop (dpig_conv_bf16.hip) with three MFMAs per fragment pair: eight 1-KB DMA pieces per wave
// per k-tile, no operand ever in a VGPR, 24 MFMAs.  Both tile images have unpadded 128-byte rows, 16-byte slot s of row r
// holding granule s ^ ((r >> 1) & 7) (source-side swizzle: zero LDS bank conflicts).  Same roundings, products and order as
// the in-loop split => bit-identical results.
__device__ __forceinline__ void gg_mainloop_x3dma(const GGParams& p, char* lds, f32x16 (&acc)[2][2], int m0, int n0,
                                                  int kt_begin, int kt_end, int tid, int wrow, int wcol, int l31, int half) {
    constexpr int IMG = 128 * 128;           // one operand image
    constexpr int STG = 2 * IMG;             // a stage: A image, B image
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.As32), 0, (int)p.as_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.Bs_hi), 0, (int)p.bs_bytes, 0x00020000);
    int a_base[4], a_iy0[4], a_ix0[4], a_voff[4], d_kk[4];
    unsigned b_off[4];
    bool b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {            // piece 4 wave + i = tile rows 8 piece .. + 7; lane -> (row, 16-byte slot)
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int gs = (lane & 7) ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = fast_div(mm, p.mul_hrwr, p.shr_hrwr);
        const int rem = mm - n * p.HrWr;
        const int rr = fast_div(rem, p.mul_wr, p.shr_wr);
        const int c = rem - rr * p.Wr;
        a_iy0[i] = ok ? rr * p.sr : -(1 << 24);                // a row beyond M fails every bounds test
        a_ix0[i] = c * p.sr;
        a_base[i] = (((n * p.Hs + rr * p.sr) * p.Ws + c * p.sr) * p.a_nchunk) * 128 + gs * 16;
        d_kk[i] = (gs & 3) * 8;
        b_ok[i] = n0 + r < p.Ncols;
        b_off[i] = (unsigned)(((n0 + r) * p.Cs + d_kk[i]) * 2) + ((gs >> 2) ? p.bs_lo_off : 0u);
    }
    int cur_c0, cur_ta, cur_tb, tap_sB = 0;
    {
        const int tap = kt_begin / p.cchunks;
        cur_c0 = (kt_begin - tap * p.cchunks) * BKS;
        cur_ta = tap / p.tap_nb;
        cur_tb = tap - cur_ta * p.tap_nb;
    }
    auto enter_tap = [&]() {                 // per TAP: the halo test and the tap's pixel shift, folded into one offset per piece
        const int t_oy = p.oy0 + cur_ta * p.oys, t_ox = p.ox0 + cur_tb * p.oxs;
        const int shift = ((t_oy * p.Ws + t_ox) * p.a_nchunk) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = ((unsigned)(a_iy0[i] + t_oy) < (unsigned)p.Hs) & ((unsigned)(a_ix0[i] + t_ox) < (unsigned)p.Ws);
            a_voff[i] = ok ? a_base[i] + shift : (int)OOB;
        }
        tap_sB = ((p.w0 + cur_ta * p.wa + cur_tb * p.wb) * p.Ncols * p.Cs) * 2;
    };
    enter_tap();
    auto issue = [&](int stage) {
        const int c0 = cur_c0;
        char* dst = lds + stage * STG + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(dst + i * 1024), 16, a_voff[i], (c0 >> 5) * 128, 0, 0);
            const bool kok = b_ok[i] & (c0 + d_kk[i] < p.Cs);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(dst + IMG + i * 1024), 16, (int)(kok ? b_off[i] : OOB), tap_sB + c0 * 2, 0, 0);
        }
        cur_c0 += BKS;
        if (cur_c0 >= p.Cs) {                                  // next tap (uniform branch)
            cur_c0 = 0;
            if (++cur_tb == p.tap_nb) { cur_tb = 0; ++cur_ta; }
            enter_tap();
        }
    };
    const int fsw = (l31 >> 1) & 7;
    auto load_frag = [&](int stage, int ks, SplitFrag& f) {
        const int slot = (ks * 2 + half) ^ fsw;
        const char* ab = lds + stage * STG + (wrow + l31) * 128;
        const char* bb = lds + stage * STG + IMG + (wcol + l31) * 128;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            f.ah[mb] = *reinterpret_cast<const bf16x8*>(ab + mb * 32 * 128 + slot * 16);
            f.al[mb] = *reinterpret_cast<const bf16x8*>(ab + mb * 32 * 128 + (slot ^ 4) * 16);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f.bh[nb] = *reinterpret_cast<const bf16x8*>(bb + nb * 32 * 128 + slot * 16);
            f.bl[nb] = *reinterpret_cast<const bf16x8*>(bb + nb * 32 * 128 + (slot ^ 4) * 16);
        }
    };
    SplitFrag f0, f1;
    auto ktile = [&](int stage, bool more) {
        load_frag(stage, 0, f0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(stage ^ 1);          // the stage it fills was last read before the barrier every wave has passed
        load_frag(stage, 1, f1);
        __builtin_amdgcn_sched_barrier(0);
        split_mfma(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        split_mfma(f1, acc);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    if (kt_begin >= kt_end) return;
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int kt = kt_begin;
    for (; kt + 1 < kt_end; kt += 2) {       // two k-tiles per trip: the LDS stage is a compile-time constant
        ktile(0, true);
        ktile(1, kt + 2 < kt_end);
    }
    if (kt < kt_end) ktile(0, false);
}

// NARROW: 128 x 32 block tile (waves stacked 4 x 1, one 32x32 accumulator each) for GEMMs whose N is
// at most 32 (Cout = 3 image conv, dgrad towards a 3-channel image, N = 1 logits): 4x fewer MFMAs than
// masking a 128-wide tile down to 3 columns.
// PIPE: 0 fp32 MFMA (exact), 1 bf16 (operands rounded), 2 split-bf16 (three MFMAs per product block), 3 = 2 with the filter
// operand from precomputed hi / lo shadows by LDS-DMA, 4 = 3 with the activation from its split32 image by LDS-DMA too
template <bool B_ROWK, bool VEC, bool NARROW, int PIPE = 0>
__device__ __forceinline__ void gather_gemm_body(const GGParams& p) {
    constexpr bool BF16 = PIPE == 1;
    constexpr int MB = NARROW ? 1 : 2;          // 32-row blocks per wave
    constexpr int NB = NARROW ? 1 : 2;          // 32-col blocks per wave
    constexpr int BNT = NARROW ? 32 : BN;       // block tile width
    __shared__ __attribute__((aligned(16))) float smem[2][2 * TILE_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wrow = NARROW ? wave * 32 : (wave >> 1) * 64;     // wave's first row / column in the tile
    const int wcol = NARROW ? 0 : (wave & 1) * 64;
    const int l31 = lane & 31, half = lane >> 5;

    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BNT;
    const int split = blockIdx.z;
    const int kt_begin = split * p.tiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.tiles_per_split);

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#ifdef DPIG_TRACE
    const bool trace_on = ((int)blockIdx.x < 512) && (blockIdx.z == 0) && (tid == 0);
    int trace_n = 0;
    if (trace_on) dpig_trace_buf[blockIdx.x * 256 + trace_n++] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    DPIG_STAMP(0);
    if (tid == 0 && blockIdx.x < 8192 && blockIdx.z == 0) { dpig_trace_se[blockIdx.x * 4] = __builtin_amdgcn_s_memtime(); dpig_trace_se[blockIdx.x * 4 + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); }
#endif
    if constexpr (BF16) {
        static_assert(!NARROW && VEC, "the bf16 loop exists for the 128x128 tile, 16-byte loadable operands");
        gg_mainloop_bf16<B_ROWK>(p, reinterpret_cast<char*>(&smem[0][0]), acc, m0, n0, kt_begin, kt_end, tid, wrow, wcol,
                                 l31, half);
    } else if constexpr (PIPE == 2) {
        static_assert(!NARROW && VEC, "the split-bf16 loop exists for the 128x128 tile, 16-byte loadable operands");
        gg_mainloop_split<B_ROWK>(p, reinterpret_cast<char*>(&smem[0][0]), acc, m0, n0, kt_begin, kt_end, tid, wrow, wcol,
                                  l31, half);
    } else if constexpr (PIPE == 3) {
        static_assert(!NARROW && VEC && B_ROWK, "split-bf16 with filter shadows: 128x128 tile, filter rows n / k contiguous");
        gg_mainloop_split<true, true>(p, reinterpret_cast<char*>(&smem[0][0]), acc, m0, n0, kt_begin, kt_end, tid, wrow, wcol,
                                      l31, half);
    } else if constexpr (PIPE == 4) {
        static_assert(!NARROW && VEC && B_ROWK, "split-bf16 with both operands by DMA: 128x128 tile, filter rows n / k contiguous");
        gg_mainloop_x3dma(p, reinterpret_cast<char*>(&smem[0][0]), acc, m0, n0, kt_begin, kt_end, tid, wrow, wcol, l31, half);
    } else {
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B, p.b_bytes);

    // ---- per-thread A rows (fixed for the whole k loop) --------------------------------------
    const int a_kq = tid & 7;
    unsigned a_rowoff[4];
    int a_iy0[4], a_ix0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = fast_div(mm, p.mul_hrwr, p.shr_hrwr);
        const int rem = mm - n * p.HrWr;
        const int r = fast_div(rem, p.mul_wr, p.shr_wr);
        const int c = rem - r * p.Wr;
        a_iy0[i] = ok ? r * p.sr : -(1 << 24);      // a row beyond M fails every bounds test below
        a_ix0[i] = c * p.sr;
        a_rowoff[i] = (unsigned)((((n * p.Hs + r * p.sr) * p.Ws + c * p.sr) * p.lda + a_kq * 4) * 4);
    }
    // ---- per-thread B offsets ------------------------------------------------------------------
    unsigned b_off[4];
    bool b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (B_ROWK) {   // B[(wtap*Ncols + n)*Cs + k]: rows n, k contiguous (dgrad)
            const int n = n0 + (tid >> 3) + 32 * i;
            b_ok[i] = (n < p.Ncols) & ((tid >> 3) + 32 * i < BNT);
            b_off[i] = (unsigned)((n * p.Cs + a_kq * 4) * 4);
        } else {        // B[(wtap*Cs + k)*Ncols + n]: rows k, n contiguous (fwd)
            const int nq = n0 + (tid & 31) * 4;
            b_ok[i] = (VEC ? (nq < p.Ncols) : true) & ((tid & 31) * 4 < BNT);
            b_off[i] = (unsigned)((((tid >> 5) + 8 * i) * p.Ncols + nq) * 4);
        }
    }

    float4 ra[4], rb[4];

    // k-tile cursor (tap row/col, channel chunk) advanced incrementally: no divisions in the loop
    int cur_c0, cur_ta, cur_tb;
    {
        const int tap = kt_begin / p.cchunks;
        cur_c0 = (kt_begin - tap * p.cchunks) * BK;
        cur_ta = tap / p.tap_nb;
        cur_tb = tap - cur_ta * p.tap_nb;
    }

    // The loader of one k-tile is cut into 4 parts (one A row + one B row each) so that the kernel can
    // issue them between the MFMA groups of the previous tile instead of as one ~150-instruction
    // block during which this wave feeds nothing to the matrix pipe.
    int t_oy, t_ox, t_ck, t_c0;
    unsigned t_sA, t_sB;
    bool t_kok;
    auto load_begin = [&]() {            // scalar work: advance the cursor, derive the tap's offsets
        const int c0 = cur_c0, ta = cur_ta, tb = cur_tb;
        cur_c0 += BK;
        if (cur_c0 >= p.Cs) {
            cur_c0 = 0;
            if (++cur_tb == p.tap_nb) { cur_tb = 0; ++cur_ta; }
        }
        const int wt = p.w0 + ta * p.wa + tb * p.wb;
        t_oy = p.oy0 + ta * p.oys;
        t_ox = p.ox0 + tb * p.oxs;
        t_c0 = c0;
        t_ck = c0 + a_kq * 4;
        t_sA = (unsigned)(((t_oy * p.Ws + t_ox) * p.lda + c0) * 4);
        t_kok = VEC ? (t_ck < p.Cs) : true;
        t_sB = B_ROWK ? (unsigned)((wt * p.Ncols * p.Cs + c0) * 4) : (unsigned)(((wt * p.Cs + c0) * p.Ncols) * 4);
    };
    // `live` = false turns the part into loads of structural zeros (offset OOB): the steady-state loop
    // stays branch-free, the last iteration just fetches nothing.
    auto load_part = [&](int i, bool live) {
        // bitwise & on purpose: short-circuit && compiles to exec-mask branches around each load
        const bool ok = live & t_kok & ((unsigned)(a_iy0[i] + t_oy) < (unsigned)p.Hs) &
                        ((unsigned)(a_ix0[i] + t_ox) < (unsigned)p.Ws);
        if (VEC || p.vec_a) ra[i] = gload4<true>(rsA, a_rowoff[i] + t_sA, ok & (t_ck < p.Cs), t_ck, p.Cs);
        else ra[i] = gload4<false>(rsA, a_rowoff[i] + t_sA, ok, t_ck, p.Cs);
        if (B_ROWK) {
            rb[i] = gload4<VEC>(rsB, b_off[i] + t_sB, live & b_ok[i] & t_kok, t_ck, p.Cs);
        } else {
            rb[i] = gload4<VEC>(rsB, b_off[i] + t_sB, live & b_ok[i] & (t_c0 + (tid >> 5) + 8 * i < p.Cs),
                                n0 + (tid & 31) * 4, p.Ncols);
        }
    };
    auto load_tiles = [&]() {
        load_begin();
#pragma unroll
        for (int i = 0; i < 4; ++i) load_part(i, true);
    };

    auto store_tiles = [&](int buf) {
        float* As = smem[buf];
        float* Bs = smem[buf] + TILE_FLOATS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(&As[((tid >> 3) + 32 * i) * LDR + a_kq * 4]) = ra[i];
        if (B_ROWK) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(&Bs[((tid >> 3) + 32 * i) * LDR + a_kq * 4]) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(&Bs[((tid >> 5) + 8 * i) * LDKN + (tid & 31) * 4]) = rb[i];
        }
    };

    auto store_a = [&](int buf, int i) {
        *reinterpret_cast<float4*>(&smem[buf][((tid >> 3) + 32 * i) * LDR + a_kq * 4]) = ra[i];
    };
    auto store_b = [&](int buf, int i) {
        float* Bs = smem[buf] + TILE_FLOATS;
        if (B_ROWK) *reinterpret_cast<float4*>(&Bs[((tid >> 3) + 32 * i) * LDR + a_kq * 4]) = rb[i];
        else *reinterpret_cast<float4*>(&Bs[((tid >> 5) + 8 * i) * LDKN + (tid & 31) * 4]) = rb[i];
    };

    // LDS -> register fragments of k-step kk (8 k values: lane half h takes k = kk*8 + 4h + j)
    auto load_frag = [&](const float* As, const float* Bs, int kk, float4 (&fa)[MB], float4 (&fb)[NB]) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            fa[mb] = *reinterpret_cast<const float4*>(&As[(wrow + mb * 32 + l31) * LDR + kk * 8 + half * 4]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (B_ROWK) {
                fb[nb] = *reinterpret_cast<const float4*>(&Bs[(wcol + nb * 32 + l31) * LDR + kk * 8 + half * 4]);
            } else {
                const float* bp = &Bs[(kk * 8 + half * 4) * LDKN + wcol + nb * 32 + l31];
                fb[nb].x = bp[0 * LDKN]; fb[nb].y = bp[1 * LDKN]; fb[nb].z = bp[2 * LDKN]; fb[nb].w = bp[3 * LDKN];
            }
        }
    };

    if (kt_begin < kt_end) {
        load_tiles();
        store_tiles(0);
        __syncthreads();
        int buf = 0;
        float4 fa[2][MB], fb[2][NB];        // [k-step parity][32-row / 32-col block]
        load_frag(smem[0], smem[0] + TILE_FLOATS, 0, fa[0], fb[0]);
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more = (kt + 1) < kt_end;
            load_begin();
            const float* As = smem[buf];
            const float* Bs = smem[buf] + TILE_FLOATS;
            DPIG_STAMP(1);
            // Schedule of one k-tile = 4 groups (kk) of 4 quads (j) of 4 MFMAs.  The hand-over of tile t+1 is
            // spread BEHIND this tile's MFMAs instead of sitting between two tiles: HBM/L2 loads are issued
            // at the head of groups 0 and 1; the eight 16-byte LDS stores go one or two at a time after the
            // quads of groups 2 and 3 (a ds_write_b128 occupies the store path ~13 cycles, an MFMA quad keeps
            // the matrix pipe busy >= 256); the barrier and the first fragment reads of tile t+1 come before
            // the LAST quad, which covers their latency.  With the whole hand-over after group 3 a wave feeds
            // nothing to the matrix pipe for ~1700 cycles per tile (s_memtime trace), and since the two
            // workgroups of a CU run the same code in phase nobody else does either.
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                if (kk + 1 < BK / 8) load_frag(As, Bs, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
                if (kk < 2) { load_part(2 * kk, more); load_part(2 * kk + 1, more); }
                __builtin_amdgcn_sched_barrier(0);  // keep the memory ops HERE, ahead of the MFMA group
                float av[MB][4], bv[NB][4];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const float4 t = fa[kk & 1][mb];
                    av[mb][0] = t.x; av[mb][1] = t.y; av[mb][2] = t.z; av[mb][3] = t.w;
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float4 t = fb[kk & 1][nb];
                    bv[nb][0] = t.x; bv[nb][1] = t.y; bv[nb][2] = t.z; bv[nb][3] = t.w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (kk == 3 && j == 3) {           // publish tile t+1, request its first fragments
                        DPIG_STAMP(2);
                        __syncthreads();
                        DPIG_STAMP(4);
                        load_frag(smem[buf ^ 1], smem[buf ^ 1] + TILE_FLOATS, 0, fa[0], fb[0]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb][j], bv[nb][j], acc[mb][nb], 0, 0, 0);
                    if (kk == 2) {                     // stores of parts 0,1: one per quad
                        __builtin_amdgcn_sched_barrier(0);
                        if ((j & 1) == 0) store_a(buf ^ 1, j >> 1); else store_b(buf ^ 1, j >> 1);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (kk == 3 && j < 2) {     // parts 2,3: two per quad, done two quads before the barrier
                        __builtin_amdgcn_sched_barrier(0);
                        store_a(buf ^ 1, 2 + j); store_b(buf ^ 1, 2 + j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
    }

    }   // fp32 k loop

    DPIG_STAMP(5);
#ifdef DPIG_TRACE
    if (tid == 0 && blockIdx.x < 8192 && blockIdx.z == 0) dpig_trace_se[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue ------------------------------------------------------------------------------
    // The accumulators go through LDS (the operand ring is dead now) so that every global access of
    // the fused epilogue -- bias, residual, activation mask, the one or two outputs, split-K partials
    // -- is a 16-byte, row-contiguous access issued 16 at a time, instead of 64 dependent 4-byte
    // round trips per lane.
    constexpr int LDC = BN + 4;
    float* Cs = &smem[0][0];                       // 128 x 132 floats = 67.6 KB <= 73.7 KB
    __syncthreads();                               // (the loop's last barrier already passed; cheap)
    DPIG_STAMP(6);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                Cs[(wrow + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDC + wcol + nb * 32 + l31] = acc[mb][nb][r];
    DPIG_STAMP(7);
    __syncthreads();
    DPIG_STAMP(8);

    if (p.vec_epi) {
        const int c = (tid & 31) * 4;
        const int col = n0 + c;
        if (col < p.Ncols && c < BNT) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && p.nsplit == 1) bv = *reinterpret_cast<const float4*>(p.bias + col);
            const int rl0 = tid >> 5;
            if (p.nsplit > 1) {
                float* pp = p.partial + ((long)split * p.M + m0 + rl0) * p.Ncols + col;
#pragma unroll 4
                for (int it = 0; it < 16; ++it) {
                    if (m0 + rl0 + 8 * it < p.M)
                        *reinterpret_cast<float4*>(pp + (long)(8 * it) * p.Ncols) =
                            *reinterpret_cast<const float4*>(&Cs[(rl0 + 8 * it) * LDC + c]);
                }
            } else if (p.identity_rows && !p.res_class && !p.replicate) {
                // Lean path for the common case (output row == pixel): no index arithmetic, no per-element
                // switches -- one slope for all three activations, row pointers advancing by 8 rows.  The four
                // flag combinations the model produces get their own straight-line loop body; the co-resident
                // workgroup is streaming MFMAs meanwhile, and every scalar branch / VALU op issued here waits
                // its turn behind them (the generic epilogue measured ~48k cycles per tile, 7 % of a
                // 72-k-tile workgroup's life).
                const float slope = (p.act == DPIG_ACT_NONE) ? 1.f : ((p.act == DPIG_ACT_RELU) ? 0.f : p.alpha);
                auto run = [&](auto HAS_RES, auto RES_POST, auto HAS_MASK, auto HAS_D2) {
                    const long r0 = (long)(m0 + rl0);
                    float* dp = p.D + r0 * p.ldd + col;
                    const float* rp = HAS_RES ? p.res + r0 * p.ldres + col : nullptr;
                    const float* mp = HAS_MASK ? p.mask + r0 * p.ldmask + col : nullptr;
                    float* d2 = HAS_D2 ? p.D2 + r0 * p.ldd2 + col : nullptr;
#pragma unroll 4
                    for (int it = 0; it < 16; ++it) {
                        if (m0 + rl0 + 8 * it >= p.M) break;
                        float4 v = *reinterpret_cast<const float4*>(&Cs[(rl0 + 8 * it) * LDC + c]);
                        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (HAS_RES) rv = *reinterpret_cast<const float4*>(rp + (long)(8 * it) * p.ldres);
                        if (HAS_RES && !RES_POST) { v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                        if (HAS_MASK) {
                            const float4 mv = *reinterpret_cast<const float4*>(mp + (long)(8 * it) * p.ldmask);
                            v.x *= (mv.x > 0.f) ? 1.f : slope; v.y *= (mv.y > 0.f) ? 1.f : slope;
                            v.z *= (mv.z > 0.f) ? 1.f : slope; v.w *= (mv.w > 0.f) ? 1.f : slope;
                        } else {
                            // (+0.f: RELU of a negative is +0 as with max(v, 0), not -0)
                            v.x = (v.x > 0.f) ? v.x : (v.x * slope + 0.f); v.y = (v.y > 0.f) ? v.y : (v.y * slope + 0.f);
                            v.z = (v.z > 0.f) ? v.z : (v.z * slope + 0.f); v.w = (v.w > 0.f) ? v.w : (v.w * slope + 0.f);
                        }
                        if (HAS_D2) *reinterpret_cast<float4*>(d2 + (long)(8 * it) * p.ldd2) = v;
                        if (HAS_RES && RES_POST) { v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                        *reinterpret_cast<float4*>(dp + (long)(8 * it) * p.ldd) = v;
                        if (p.D32) store_s32(p, r0 + 8 * it, col, v);
                    }
                };
                using T = std::true_type; using F = std::false_type;
                const bool hr = p.res != nullptr, hm = p.mask != nullptr, h2 = p.D2 != nullptr, rpost = p.res_post != 0;
                if (!hr && !hm && !h2) run(F{}, F{}, F{}, F{});                 // bias + activation
                else if (hr && !rpost && !hm && !h2) run(T{}, F{}, F{}, F{});   // + residual before the activation (dgrad accumulate)
                else if (!hr && hm && !h2) run(F{}, F{}, T{}, F{});             // dgrad * activation mask
                else if (hr && rpost && !hm && h2) run(T{}, T{}, F{}, T{});     // res-block tail: act -> D2, + skip -> D
                else {
#pragma unroll 2
                    for (int it = 0; it < 16; ++it) {
                        const int rl = rl0 + 8 * it;
                        if (m0 + rl >= p.M) break;
                        epi_vec4(p, m0 + rl, col, *reinterpret_cast<const float4*>(&Cs[rl * LDC + c]), bv);
                    }
                }
            } else {
#pragma unroll 2
                for (int it = 0; it < 16; ++it) {
                    const int rl = rl0 + 8 * it;
                    if (m0 + rl >= p.M) break;
                    epi_vec4(p, m0 + rl, col, *reinterpret_cast<const float4*>(&Cs[rl * LDC + c]), bv);
                }
            }
        }
        DPIG_STAMP(9);
    } else {
        // generic scalar path (thin / unaligned layers)
        for (int idx = tid; idx < BM * BN; idx += 256) {
            const int rl = idx >> 7, cl = idx & 127;
            const int row = m0 + rl, col = n0 + cl;
            if (row >= p.M || col >= p.Ncols || cl >= BNT) continue;
            const float v = Cs[rl * LDC + cl];
            if (p.nsplit > 1) {
                p.partial[((long)split * p.M + row) * p.Ncols + col] = v;
            } else {
                const long pix = p.identity_rows ? (long)row
                                                 : row_to_pix(row, p.HrWr, p.Wr, p.Hd, p.Wd, p.dr, p.dpy, p.dpx);
                const long rpix = (p.res && p.res_class) ? class_row(row, p.HrWr, p.Wr, p.Hr) : pix;
                epi_store(p.D, p.bias, p.res, p.mask, pix, col, v, p.ldd, p.ldres, p.ldmask, p.act, p.alpha,
                          p.replicate, p.Wd, p.D2, p.ldd2, p.res_post, rpix);
            }
        }
    }
    // ---- batch-norm partial statistics of this row tile (tflib/ops/batchnorm.py:30 after conv2d.py:106-120) -------------
    // The tile is still in LDS: per column its sum and the sum of squared deviations from the TILE's own mean (two passes over
    // LDS, so no E[x^2] - E[x]^2 cancellation); dpig_bn_stats_finalize merges the tiles pairwise-exactly (Chan et al.) in
    // tile order.  Only launched un-split with the float4 epilogue on 128-wide tiles (launch_gg checks).
    if constexpr (!NARROW) {
    if (p.stats) {
        const int c = (tid & 31) * 4, col = n0 + c, rl0 = tid >> 5;
        const bool cok = col < p.Ncols;
        const int nrows = min(BM, p.M - m0);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && cok) bv = *reinterpret_cast<const float4*>(p.bias + col);
        float* red = Cs + BM * LDC;                 // 8 x 128 floats behind the staged tile
        float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int rl = rl0 + 8 * it;
            if (rl < nrows) {
                const float4 v = *reinterpret_cast<const float4*>(&Cs[rl * LDC + c]);
                sm.x += v.x + bv.x; sm.y += v.y + bv.y; sm.z += v.z + bv.z; sm.w += v.w + bv.w;
            }
        }
        *reinterpret_cast<float4*>(&red[rl0 * BN + c]) = sm;
        __syncthreads();
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 t = *reinterpret_cast<const float4*>(&red[g * BN + c]);
            tot.x += t.x; tot.y += t.y; tot.z += t.z; tot.w += t.w;
        }
        const float inv = 1.0f / (float)nrows;
        const float4 mean = make_float4(tot.x * inv, tot.y * inv, tot.z * inv, tot.w * inv);
        __syncthreads();
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int rl = rl0 + 8 * it;
            if (rl < nrows) {
                const float4 v = *reinterpret_cast<const float4*>(&Cs[rl * LDC + c]);
                const float dx = v.x + bv.x - mean.x, dy = v.y + bv.y - mean.y, dz = v.z + bv.z - mean.z, dw = v.w + bv.w - mean.w;
                q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
            }
        }
        *reinterpret_cast<float4*>(&red[rl0 * BN + c]) = q;
        __syncthreads();
        if (rl0 == 0 && cok) {
            float4 qt = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(&red[g * BN + c]);
                qt.x += t.x; qt.y += t.y; qt.z += t.z; qt.w += t.w;
            }
            float* o = p.stats + (long)mt * 2 * p.Ncols + col;
            *reinterpret_cast<float4*>(o) = tot;
            *reinterpret_cast<float4*>(o + p.Ncols) = qt;
        }
    }
    }
#ifdef DPIG_TRACE
    if (tid == 0 && blockIdx.x < 8192 && blockIdx.z == 0) dpig_trace_se[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
#endif
}

template <bool B_ROWK, bool VEC, bool NARROW, int PIPE = 0>
__global__ __launch_bounds__(256, 2) void gather_gemm_kernel(const GGParams p) {
    gather_gemm_body<B_ROWK, VEC, NARROW, PIPE>(p);
}
// Several independent problems of the same kernel variant in ONE launch (blockIdx.y picks the problem): the 4
// output-parity classes of a stride-2 dgrad are 4 small GEMMs (a quarter of the pixels each, 4..9 of the 25 taps);
// as 4 launches + 4 split-K reductions they were launch-bound (~13 us each, 31 TFLOP/s on the critic's layers).
struct GGMulti { GGParams q[4]; };
template <bool B_ROWK, bool VEC, bool NARROW, int PIPE = 0>
__global__ __launch_bounds__(256, 2) void gather_gemm_multi_kernel(const GGMulti m) {
    const GGParams& p = m.q[blockIdx.y];
    if ((int)blockIdx.x >= p.mtiles * p.ntiles || (int)blockIdx.z >= p.nsplit) return;
    gather_gemm_body<B_ROWK, VEC, NARROW, PIPE>(p);
}

// split-K second pass: sum partials in split order (deterministic) and run the fused epilogue
__device__ __forceinline__ void gather_gemm_reduce_body(const GGParams& p);
__global__ __launch_bounds__(256) void gather_gemm_reduce_kernel(const GGParams p) { gather_gemm_reduce_body(p); }
__global__ __launch_bounds__(256) void gather_gemm_reduce_multi_kernel(const GGMulti m) {
    const GGParams& p = m.q[blockIdx.y];
    if (p.nsplit > 1) gather_gemm_reduce_body(p);
}
// A handful of output tiles cut into up to one split per CU (the fully connected layers: 16 .. 112 rows, reductions of 4096 .. 20480) leaves
// the pass few threads with a long chain of dependent loads each (256 splits four at a time: ~19 us for a 16 x 64 result).  Such passes
// give every float4 of the result to 16 lanes: lane j sums splits j, j + 16, ..., the 16 partial sums are combined in a fixed butterfly.
static inline int reduce_lanes(int nsplit, long total4) {
    static const bool off = getenv("DPIG_REDUCE_LANES") && atoi(getenv("DPIG_REDUCE_LANES")) <= 1;      // (A/B switch)
    return (!off && nsplit >= 32 && total4 <= 32768) ? 16 : 1;
}

__device__ __forceinline__ void gather_gemm_reduce_body(const GGParams& p) {
    const long total = (long)p.M * p.Ncols;
    if (p.vec_epi) {
        const int n4 = p.Ncols >> 2;
        const long total4 = total >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(p.partial);
        if (p.red_lanes > 1) {
            const int sub = threadIdx.x & 15;
            const long ngroups = ((long)gridDim.x * blockDim.x) >> 4;
            for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; i < total4; i += ngroups) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
                for (int s = sub; s < p.nsplit; s += 16) {
                    const float4 t = p4[(long)s * total4 + i];
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) {
                    v.x += __shfl_xor(v.x, o, 16); v.y += __shfl_xor(v.y, o, 16);
                    v.z += __shfl_xor(v.z, o, 16); v.w += __shfl_xor(v.w, o, 16);
                }
                if (sub == 0) {
                    const int row = (int)(i / n4);
                    const int col = (int)(i - (long)row * n4) * 4;
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + col);
                    epi_vec4(p, row, col, v, bv);
                }
            }
            return;
        }
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
            float4 v = p4[i];
#pragma unroll 4
            for (int s = 1; s < p.nsplit; ++s) {       // (unrolled: several 16-byte loads in flight per lane)
                const float4 t = p4[(long)s * total4 + i];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const int row = (int)(i / n4);
            const int col = (int)(i - (long)row * n4) * 4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + col);
            epi_vec4(p, row, col, v, bv);
        }
        return;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < p.nsplit; ++s) v += p.partial[(long)s * total + i];
        const int row = (int)(i / p.Ncols);
        const int col = (int)(i - (long)row * p.Ncols);
        const long pix = p.identity_rows ? (long)row
                                         : row_to_pix(row, p.HrWr, p.Wr, p.Hd, p.Wd, p.dr, p.dpy, p.dpx);
        const long rpix = (p.res && p.res_class) ? class_row(row, p.HrWr, p.Wr, p.Hr) : pix;
        epi_store(p.D, p.bias, p.res, p.mask, pix, col, v, p.ldd, p.ldres, p.ldmask, p.act, p.alpha, p.replicate,
                  p.Wd, p.D2, p.ldd2, p.res_post, rpix);
    }
}

// split-K second pass of a forward conv that carries batch-norm statistics: one workgroup per 128 x 128 output tile sums the
// partials in split order (the same sums as gather_gemm_reduce_body), runs the float4 epilogue, and leaves the tile's column sums and
// centred squares exactly as gather_gemm_body's statistics section does for an un-split launch (same thread <-> element map).
__global__ __launch_bounds__(256) void gather_gemm_reduce_stats_kernel(const GGParams p) {
    __shared__ __attribute__((aligned(16))) float red[8 * BN];
    const int tid = threadIdx.x;
    const int mt = blockIdx.x / p.ntiles, nt = blockIdx.x - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int c = (tid & 31) * 4, col = n0 + c, rl0 = tid >> 5;
    const bool cok = col < p.Ncols;
    const int nrows = min(BM, p.M - m0);
    const int n4 = p.Ncols >> 2;
    const long total4 = ((long)p.M * p.Ncols) >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(p.partial);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && cok) bv = *reinterpret_cast<const float4*>(p.bias + col);
    float4 val[16];
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rl = rl0 + 8 * it;
        val[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < nrows && cok) {
            const long i = (long)(m0 + rl) * n4 + (col >> 2);
            float4 v = p4[i];
            for (int sp = 1; sp < p.nsplit; ++sp) {
                const float4 t = p4[(long)sp * total4 + i];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            epi_vec4(p, m0 + rl, col, v, bv);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            val[it] = v;
            sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
        }
    }
    *reinterpret_cast<float4*>(&red[rl0 * BN + c]) = sm;
    __syncthreads();
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float4 t = *reinterpret_cast<const float4*>(&red[g * BN + c]);
        tot.x += t.x; tot.y += t.y; tot.z += t.z; tot.w += t.w;
    }
    const float inv = 1.0f / (float)nrows;
    const float4 mean = make_float4(tot.x * inv, tot.y * inv, tot.z * inv, tot.w * inv);
    __syncthreads();
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int rl = rl0 + 8 * it;
        if (rl < nrows && cok) {
            const float dx = val[it].x - mean.x, dy = val[it].y - mean.y, dz = val[it].z - mean.z, dw = val[it].w - mean.w;
            q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
        }
    }
    *reinterpret_cast<float4*>(&red[rl0 * BN + c]) = q;
    __syncthreads();
    if (rl0 == 0 && cok) {
        float4 qt = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 t = *reinterpret_cast<const float4*>(&red[g * BN + c]);
            qt.x += t.x; qt.y += t.y; qt.z += t.z; qt.w += t.w;
        }
        float* o = p.stats + (long)mt * 2 * p.Ncols + col;
        *reinterpret_cast<float4*>(o) = tot;
        *reinterpret_cast<float4*>(o + p.Ncols) = qt;
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad: dw[(tap, ci), co] = sum over output pixels m of x[src(m,tap), ci] * dy[m, co]
struct WGParams {
    const float* X; const float* DY; float* DW; float* partial;
    int Npix, Ho, Wo, HoWo;
    int H, W, ldx, C, shift, s;
    int K, ldy;
    int ntaps, cblocks, ntiles;
    int ktiles, tiles_per_split, nsplit, wrows;
    float beta;
    int S, pad_t, pad_l;   // tap t = ky*S + kx -> source offset (ky - pad_t, kx - pad_l), filter slab t
    unsigned x_bytes, y_bytes;                     // buffer-descriptor extents
    unsigned mul_howo, shr_howo, mul_wo, shr_wo;   // magic numbers: m / HoWo and rem / Wo without v_rcp
    int vec_epi;                                   // 1: dw / partial rows are 16-byte addressable
    float* DB; float* bias_partial; float beta_b;  // fused bias gradient: db[co] = sum_pixels dy[., co]
    int vec_x;                                     // 1: x alone is 16-byte loadable (Cout = 3: dy is not)
    int d32_oy, d32_ox;                            // S1: (row, col) advance of a pixel index step of BK = 32
    int d64_oy, d64_ox, d64_n;                     //     ... and of the bf16 loop's 64 (+ whole images, generic variant)
    int d32g_oy, d32g_ox, d32g_n;                  //     ... and of the split-bf16 loop's 32, in the bf16 loop's (n, oy, ox) form
};


// bf16 matrix-pipe variant of the stride-1 wgrad pixel loop (see gg_mainloop_bf16 for the operand format).
// Both operands are pixel-major in HBM ([pixel][channel]) while the MFMA wants 8 consecutive k (= pixels) per
// lane: each thread loads an 8(pixel) x 4(channel) patch of x and of dy with the scalar-offset addressing of the
// S1 loader and transposes it in registers into four 16-byte LDS rows per operand; fragments are then plain
// ds_read_b128 (the fp32 loop needs 4-byte LDS reads here).  k-tile = 64 pixels.
template <bool GEN>   // GEN: any stride / the upsampled 1x1 (x offset computed per pixel); else stride-1 SAME (scalar offsets)
__device__ __forceinline__ void wg_mainloop_bf16(const WGParams& p, char* lds, f32x16 (&acc)[2][2], int ci0, int co0,
                                                 int oyoff, int oxoff, int kt_begin, int kt_end, int tid, int wrow,
                                                 int wcol, int l31, int half, bool do_bias, float4& bsum) {
    const int cq = tid & 31, poct = tid >> 5;                  // 4 channels x 8 pixels per thread
    const int padpix = GEN ? 0 : p.pad_t * p.W + p.pad_l;
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X - (long)padpix * p.ldx, p.x_bytes + (unsigned)(padpix * p.ldx * 4));
    const __amdgpu_buffer_rsrc_t rsY = make_rsrc(p.DY, p.y_bytes);
    const bool cx_ok = ci0 + cq * 4 < p.C, cy_ok = co0 + cq * 4 < p.K;
    int s_oy[8], s_ox[8], s_n[8];
    unsigned xv[8], yv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = poct * 8 + e;
        const int m = kt_begin * BKH + r;
        const int n = fast_div(m, p.mul_howo, p.shr_howo);
        const int rem = m - n * p.HoWo;
        s_n[e] = n;
        s_oy[e] = fast_div(rem, p.mul_wo, p.shr_wo);
        s_ox[e] = rem - s_oy[e] * p.Wo;
        xv[e] = cx_ok ? (unsigned)((r * p.ldx + ci0 + cq * 4) * 4) : OOB;
        yv[e] = cy_ok ? (unsigned)((r * p.ldy + co0 + cq * 4) * 4) : OOB;
    }
    float4 ra[8], rb[8];
    auto load_tile = [&](int kt, bool live) {
        const int sx = ((kt * BKH + oyoff * p.W + oxoff + padpix) * p.ldx) * 4;
        const int sy = (kt * BKH * p.ldy) * 4;
        const int left = p.Npix - kt * BKH;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool mok = live & (poct * 8 + e < left);
            f32x4 ta;
            if (GEN) {
                const int py = s_oy[e] * p.s + oyoff, px = s_ox[e] * p.s + oxoff;
                const int iy = py >> p.shift, ix = px >> p.shift;
                const bool ok = mok & cx_ok & (py >= 0) & (px >= 0) & (iy < p.H) & (ix < p.W);
                const unsigned xo = (unsigned)((((s_n[e] * p.H + iy) * p.W + ix) * p.ldx + ci0 + cq * 4) * 4);
                ta = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(ok ? xo : OOB), 0, 0));
            } else {
                const bool ok = mok & ((unsigned)(s_oy[e] + oyoff) < (unsigned)p.H) & ((unsigned)(s_ox[e] + oxoff) < (unsigned)p.W);
                ta = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(ok ? xv[e] : OOB), sx, 0));
            }
            const f32x4 tb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)(mok ? yv[e] : OOB), sy, 0));
            ra[e] = make_float4(ta.x, ta.y, ta.z, ta.w);
            rb[e] = make_float4(tb.x, tb.y, tb.z, tb.w);
            s_ox[e] += p.d64_ox;
            const bool c1 = s_ox[e] >= p.Wo;
            s_ox[e] -= c1 ? p.Wo : 0;
            s_oy[e] += p.d64_oy + (c1 ? 1 : 0);
            const bool c2 = s_oy[e] >= p.Ho;
            s_oy[e] -= c2 ? p.Ho : 0;
            if (GEN) s_n[e] += p.d64_n + (c2 ? 1 : 0);
        }
    };
    auto store_tile = [&](int buf) {
        char* da = lds + buf * 2 * TILEB + (cq * 4) * ROWB + poct * 16;
        char* db = da + TILEB;
        *reinterpret_cast<uint4*>(da + 0 * ROWB) = make_uint4(pack_bf16(ra[0].x, ra[1].x), pack_bf16(ra[2].x, ra[3].x), pack_bf16(ra[4].x, ra[5].x), pack_bf16(ra[6].x, ra[7].x));
        *reinterpret_cast<uint4*>(da + 1 * ROWB) = make_uint4(pack_bf16(ra[0].y, ra[1].y), pack_bf16(ra[2].y, ra[3].y), pack_bf16(ra[4].y, ra[5].y), pack_bf16(ra[6].y, ra[7].y));
        *reinterpret_cast<uint4*>(da + 2 * ROWB) = make_uint4(pack_bf16(ra[0].z, ra[1].z), pack_bf16(ra[2].z, ra[3].z), pack_bf16(ra[4].z, ra[5].z), pack_bf16(ra[6].z, ra[7].z));
        *reinterpret_cast<uint4*>(da + 3 * ROWB) = make_uint4(pack_bf16(ra[0].w, ra[1].w), pack_bf16(ra[2].w, ra[3].w), pack_bf16(ra[4].w, ra[5].w), pack_bf16(ra[6].w, ra[7].w));
        *reinterpret_cast<uint4*>(db + 0 * ROWB) = make_uint4(pack_bf16(rb[0].x, rb[1].x), pack_bf16(rb[2].x, rb[3].x), pack_bf16(rb[4].x, rb[5].x), pack_bf16(rb[6].x, rb[7].x));
        *reinterpret_cast<uint4*>(db + 1 * ROWB) = make_uint4(pack_bf16(rb[0].y, rb[1].y), pack_bf16(rb[2].y, rb[3].y), pack_bf16(rb[4].y, rb[5].y), pack_bf16(rb[6].y, rb[7].y));
        *reinterpret_cast<uint4*>(db + 2 * ROWB) = make_uint4(pack_bf16(rb[0].z, rb[1].z), pack_bf16(rb[2].z, rb[3].z), pack_bf16(rb[4].z, rb[5].z), pack_bf16(rb[6].z, rb[7].z));
        *reinterpret_cast<uint4*>(db + 3 * ROWB) = make_uint4(pack_bf16(rb[0].w, rb[1].w), pack_bf16(rb[2].w, rb[3].w), pack_bf16(rb[4].w, rb[5].w), pack_bf16(rb[6].w, rb[7].w));
        if (do_bias) {          // exact fp32 column sums of dy, as in the fp32 loop (workgroup-uniform branch)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; e < 8; ++e) { bsum.x += rb[e].x; bsum.y += rb[e].y; bsum.z += rb[e].z; bsum.w += rb[e].w; }
        }
    };
    if (kt_begin >= kt_end) return;
    load_tile(kt_begin, true);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    bf16x8 fa[2][2], fb[2][2];
    auto load_frag = [&](int b, int ks, bf16x8 (&a)[2], bf16x8 (&bb)[2]) {
        const char* As = lds + b * 2 * TILEB;
        const char* Bs = As + TILEB;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
            a[mb] = *reinterpret_cast<const bf16x8*>(As + (wrow + mb * 32 + l31) * ROWB + ks * 32 + half * 16);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            bb[nb] = *reinterpret_cast<const bf16x8*>(Bs + (wcol + nb * 32 + l31) * ROWB + ks * 32 + half * 16);
    };
    load_frag(0, 0, fa[0], fb[0]);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        load_tile(kt + 1, (kt + 1) < kt_end);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < BKH / 16; ++ks) {
            if (ks + 1 < BKH / 16) {
                load_frag(buf, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
            } else {                                       // same hand-over as gg_mainloop_bf16
                store_tile(buf ^ 1);
                __syncthreads();
                load_frag(buf ^ 1, 0, fa[0], fb[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][mb], fb[ks & 1][nb], acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        buf ^= 1;
    }
}

// split-bf16 variant of the wgrad pixel loop (see gg_mainloop_split for the operand format): k-tile = 32 pixels, each
// thread loads a 4(pixel) x 4(channel) patch of x and of dy and transposes it in registers into four LDS rows of
// [4 hi | ... | 4 lo] per operand.  The bias gradient stays an exact fp32 column sum.
template <bool GEN>
__device__ __forceinline__ void wg_mainloop_split(const WGParams& p, char* lds, f32x16 (&acc)[2][2], int ci0, int co0,
                                                  int oyoff, int oxoff, int kt_begin, int kt_end, int tid, int wrow,
                                                  int wcol, int l31, int half, bool do_bias, float4& bsum) {
    const int cq = tid & 31, pquad = tid >> 5;                 // 4 channels x 4 pixels per thread
    const int padpix = GEN ? 0 : p.pad_t * p.W + p.pad_l;
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X - (long)padpix * p.ldx, p.x_bytes + (unsigned)(padpix * p.ldx * 4));
    const __amdgpu_buffer_rsrc_t rsY = make_rsrc(p.DY, p.y_bytes);
    const bool cx_ok = ci0 + cq * 4 < p.C, cy_ok = co0 + cq * 4 < p.K;
    int s_oy[4], s_ox[4], s_n[4];
    unsigned xv[4], yv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r = pquad * 4 + e;
        const int m = kt_begin * BKS + r;
        const int n = fast_div(m, p.mul_howo, p.shr_howo);
        const int rem = m - n * p.HoWo;
        s_n[e] = n;
        s_oy[e] = fast_div(rem, p.mul_wo, p.shr_wo);
        s_ox[e] = rem - s_oy[e] * p.Wo;
        xv[e] = cx_ok ? (unsigned)((r * p.ldx + ci0 + cq * 4) * 4) : OOB;
        yv[e] = cy_ok ? (unsigned)((r * p.ldy + co0 + cq * 4) * 4) : OOB;
    }
    float4 ra[4], rb[4];
    auto load_tile = [&](int kt, bool live) {
        const int sx = ((kt * BKS + oyoff * p.W + oxoff + padpix) * p.ldx) * 4;
        const int sy = (kt * BKS * p.ldy) * 4;
        const int left = p.Npix - kt * BKS;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool mok = live & (pquad * 4 + e < left);
            f32x4 ta;
            if (GEN) {
                const int py = s_oy[e] * p.s + oyoff, px = s_ox[e] * p.s + oxoff;
                const int iy = py >> p.shift, ix = px >> p.shift;
                const bool ok = mok & cx_ok & (py >= 0) & (px >= 0) & (iy < p.H) & (ix < p.W);
                const unsigned xo = (unsigned)((((s_n[e] * p.H + iy) * p.W + ix) * p.ldx + ci0 + cq * 4) * 4);
                ta = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(ok ? xo : OOB), 0, 0));
            } else {
                const bool ok = mok & ((unsigned)(s_oy[e] + oyoff) < (unsigned)p.H) & ((unsigned)(s_ox[e] + oxoff) < (unsigned)p.W);
                ta = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(ok ? xv[e] : OOB), sx, 0));
            }
            const f32x4 tb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)(mok ? yv[e] : OOB), sy, 0));
            ra[e] = make_float4(ta.x, ta.y, ta.z, ta.w);
            rb[e] = make_float4(tb.x, tb.y, tb.z, tb.w);
            s_ox[e] += p.d32g_ox;
            const bool c1 = s_ox[e] >= p.Wo;
            s_ox[e] -= c1 ? p.Wo : 0;
            s_oy[e] += p.d32g_oy + (c1 ? 1 : 0);
            const bool c2 = s_oy[e] >= p.Ho;
            s_oy[e] -= c2 ? p.Ho : 0;
            if (GEN) s_n[e] += p.d32g_n + (c2 ? 1 : 0);
        }
    };
    auto store_tile = [&](int buf) {
        char* da = lds + buf * 2 * TILEB + (cq * 4) * ROWB + split_tslot(pquad, cq) * 8;
        char* db = da + TILEB;
        split_store4(da + 0 * ROWB, ra[0].x, ra[1].x, ra[2].x, ra[3].x);
        split_store4(da + 1 * ROWB, ra[0].y, ra[1].y, ra[2].y, ra[3].y);
        split_store4(da + 2 * ROWB, ra[0].z, ra[1].z, ra[2].z, ra[3].z);
        split_store4(da + 3 * ROWB, ra[0].w, ra[1].w, ra[2].w, ra[3].w);
        split_store4(db + 0 * ROWB, rb[0].x, rb[1].x, rb[2].x, rb[3].x);
        split_store4(db + 1 * ROWB, rb[0].y, rb[1].y, rb[2].y, rb[3].y);
        split_store4(db + 2 * ROWB, rb[0].z, rb[1].z, rb[2].z, rb[3].z);
        split_store4(db + 3 * ROWB, rb[0].w, rb[1].w, rb[2].w, rb[3].w);
        if (do_bias) {          // exact fp32 column sums of dy, as in the fp32 loop (workgroup-uniform branch)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; e < 4; ++e) { bsum.x += rb[e].x; bsum.y += rb[e].y; bsum.z += rb[e].z; bsum.w += rb[e].w; }
        }
    };
    if (kt_begin >= kt_end) return;
    load_tile(kt_begin, true);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    SplitFrag f0, f1;
    split_load_frag<true, true>(lds, lds + TILEB, 0, wrow, wcol, l31, half, f0);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const char* As = lds + buf * 2 * TILEB;
        load_tile(kt + 1, (kt + 1) < kt_end);
        __builtin_amdgcn_sched_barrier(0);
        split_load_frag<true, true>(As, As + TILEB, 1, wrow, wcol, l31, half, f1);
        __builtin_amdgcn_sched_barrier(0);
        split_mfma(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        store_tile(buf ^ 1);                               // same hand-over as gg_mainloop_split
        __syncthreads();
        split_load_frag<true, true>(lds + (buf ^ 1) * 2 * TILEB, lds + (buf ^ 1) * 2 * TILEB + TILEB, 0, wrow, wcol, l31, half, f0);
        __builtin_amdgcn_sched_barrier(0);
        split_mfma(f1, acc);
        __builtin_amdgcn_sched_barrier(0);
        buf ^= 1;
    }
}

// NARROW: 128 x 32 tile for Cout <= 32 (the 3-channel image conv).  FLAT: for thin inputs (Cin = 3 stems,
// the 18-channel pose conv) the tile rows are the flattened (tap, ci) index instead of 128 channels of one
// tap, so a 3-channel 5x5 filter gradient is 1 row tile instead of 25 tiles that are 98 % padding.
// S1 (stride 1, SAME, 16-byte loadable): pixel m of dy pairs with pixel m + const of x, so both operands are
// addressed as  per-thread constant voffset + per-k-tile SCALAR soffset; the only per-row vector work left is
// the halo test on an incrementally advanced (oy, ox).  (The generic loader spends ~35 VALU per row per k-tile
// on index arithmetic, which -- not the matrix pipe -- paced groups 0-1 of the k-loop: s_memtime trace.)
template <bool VEC, bool NARROW, bool FLAT, bool S1 = false, int PIPE = 0>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WGParams p) {
    constexpr bool BF16 = PIPE != 0;         // both bf16-pipe loops use the [128][ROWB] LDS image
    constexpr int MB = NARROW ? 1 : 2;
    constexpr int NB = NARROW ? 1 : 2;
    constexpr int BNT = NARROW ? 32 : BN;
    constexpr int SMEM_HALF = BF16 ? (2 * TILEB / 4) : (2 * BK * LDKN);      // floats per buffer of the operand ring
    __shared__ __attribute__((aligned(16))) float smem[2][SMEM_HALF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wrow = NARROW ? wave * 32 : (wave >> 1) * 64;
    const int wcol = NARROW ? 0 : (wave & 1) * 64;
    const int l31 = lane & 31, half = lane >> 5;

    const int mtiles = FLAT ? (p.ntaps * p.C + BM - 1) / BM : p.ntaps * p.cblocks;
    // The (tile, split) pairs are remapped as ONE list with the tile index fastest: the workgroups an XCD receives
    // (linear id % 8) then cover whole pixel ranges -- all taps / channel blocks of one split -- so the dy tile and the
    // overlapping x tiles of a pixel range are fetched into that XCD's L2 once instead of once per XCD.
    const int ntl = mtiles * p.ntiles;
    const int item = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.z), ntl * (int)gridDim.z);
    const int split = item / ntl;
    const int tile = item - split * ntl;
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int tap = FLAT ? 0 : mt / p.cblocks;
    const int ci0 = FLAT ? mt * BM : (mt - tap * p.cblocks) * BM;     // FLAT: first flattened (tap,ci) row
    const int co0 = nt * BNT;
    const int oyoff = tap / p.S - p.pad_t, oxoff = tap % p.S - p.pad_l, wt = tap;
    const int kt_begin = split * p.tiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.tiles_per_split);

    const int q = (tid & 31) * 4;   // column quad inside the 128-wide tile (both operands)
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rsY = make_rsrc(p.DY, p.y_bytes);
    const bool cx_ok = VEC ? (ci0 + q < p.C) : true;
    const bool cy_ok = (VEC ? (co0 + q < p.K) : true) & (q < BNT);
    // fused bias gradient: the row-tile-0 workgroups also sum the dy tile they stage anyway
    const bool do_bias = (p.DB != nullptr) && (mt == 0);
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (PIPE == 1) {
        static_assert(VEC && !NARROW && !FLAT, "the bf16 pixel loop exists for the 16-byte loadable 128x128 variant");
        wg_mainloop_bf16<!S1>(p, reinterpret_cast<char*>(&smem[0][0]), acc, ci0, co0, oyoff, oxoff, kt_begin, kt_end, tid,
                         wrow, wcol, l31, half, do_bias, bsum);
    } else if constexpr (PIPE == 2) {
        static_assert(VEC && !NARROW && !FLAT, "the split-bf16 pixel loop exists for the 16-byte loadable 128x128 variant");
        wg_mainloop_split<!S1>(p, reinterpret_cast<char*>(&smem[0][0]), acc, ci0, co0, oyoff, oxoff, kt_begin, kt_end, tid,
                          wrow, wcol, l31, half, do_bias, bsum);
    } else {
    float4 ra[4], rb[4];
    // FLAT: this thread's 4 tile rows are 4 (tap, ci) pairs, fixed for the whole pixel loop
    int f_oy[4], f_ox[4], f_ci[4];
    bool f_ok[4];
    if (FLAT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = ci0 + q + e;
            f_ok[e] = i < p.ntaps * p.C;
            const int t = f_ok[e] ? i / p.C : 0;
            f_ci[e] = f_ok[e] ? i - t * p.C : 0;
            f_oy[e] = t / p.S - p.pad_t;
            f_ox[e] = t % p.S - p.pad_l;
        }
    }

    // S1 loader state: (oy, ox) of this thread's 4 tile rows at the next k-tile to load, constant voffsets.
    // The x descriptor starts `padpix` pixels BEFORE the tensor so that m + tap shift + padpix >= 0 always
    // (soffset is unsigned); lanes whose tap falls outside the image get voffset = OOB and are never dereferenced.
    int s_oy[4], s_ox[4];
    unsigned s_xv[4], s_yv[4];
    const int padpix = p.pad_t * p.W + p.pad_l;
    const __amdgpu_buffer_rsrc_t rsXs = make_rsrc(p.X - (long)padpix * p.ldx, p.x_bytes + (unsigned)(padpix * p.ldx * 4));
    if (S1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 5) + 8 * i;
            const int m = kt_begin * BK + r;
            const int n = fast_div(m, p.mul_howo, p.shr_howo);
            const int rem = m - n * p.HoWo;
            s_oy[i] = fast_div(rem, p.mul_wo, p.shr_wo);
            s_ox[i] = rem - s_oy[i] * p.Wo;
            s_xv[i] = cx_ok ? (unsigned)((r * p.ldx + ci0 + q) * 4) : OOB;
            s_yv[i] = cy_ok ? (unsigned)((r * p.ldy + co0 + q) * 4) : OOB;
        }
    }
    // one quarter (tile rows i, i+8, ...: one 16-byte load per operand) of the loads of k-tile kt; `live` = false
    // turns it into loads of structural zeros (OOB offset), so the steady-state loop stays branch-free.
    // (S1: must be called exactly once per (kt, i), kt ascending -- it advances the row state.)
    auto load_part = [&](int kt, int i, bool live) {
        if (S1) {
            const bool mok = live & ((tid >> 5) + 8 * i < p.Npix - kt * BK);
            const bool ok = mok & ((unsigned)(s_oy[i] + oyoff) < (unsigned)p.H) & ((unsigned)(s_ox[i] + oxoff) < (unsigned)p.W);
            const int sx = ((kt * BK + oyoff * p.W + oxoff + padpix) * p.ldx) * 4;
            const int sy = (kt * BK * p.ldy) * 4;
            const f32x4 ta = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsXs, (int)(ok ? s_xv[i] : OOB), sx, 0));
            const f32x4 tb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)(mok ? s_yv[i] : OOB), sy, 0));
            ra[i] = make_float4(ta.x, ta.y, ta.z, ta.w);
            rb[i] = make_float4(tb.x, tb.y, tb.z, tb.w);
            s_ox[i] += p.d32_ox;
            const bool c1 = s_ox[i] >= p.Wo;
            s_ox[i] -= c1 ? p.Wo : 0;
            s_oy[i] += p.d32_oy + (c1 ? 1 : 0);
            s_oy[i] -= (s_oy[i] >= p.Ho) ? p.Ho : 0;
            return;
        }
        const int m = kt * BK + (tid >> 5) + 8 * i;
        const bool mok = live & (m < p.Npix);
        const int mm = mok ? m : 0;
        const int n = fast_div(mm, p.mul_howo, p.shr_howo);
        const int rem = mm - n * p.HoWo;
        const int oy = fast_div(rem, p.mul_wo, p.shr_wo);
        const int ox = rem - oy * p.Wo;
        const unsigned yoff = (unsigned)((mm * p.ldy + co0 + q) * 4);
        rb[i] = gload4<VEC>(rsY, yoff, mok & cy_ok, co0 + q, p.K);
        if (FLAT) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int py = oy * p.s + f_oy[e], px = ox * p.s + f_ox[e];
                const bool ok = mok & f_ok[e] & ((unsigned)py < (unsigned)p.H) & ((unsigned)px < (unsigned)p.W);
                const unsigned off = (unsigned)((((n * p.H + py) * p.W + px) * p.ldx + f_ci[e]) * 4);
                v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, (int)(ok ? off : OOB), 0, 0));
            }
            ra[i] = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            const int py = oy * p.s + oyoff, px = ox * p.s + oxoff;
            const int iy = py >> p.shift, ix = px >> p.shift;
            const bool ok = mok & (py >= 0) & (px >= 0) & (iy < p.H) & (ix < p.W);
            const unsigned xoff = (unsigned)((((n * p.H + iy) * p.W + ix) * p.ldx + ci0 + q) * 4);
            if (VEC || p.vec_x) ra[i] = gload4<true>(rsX, xoff, ok & (ci0 + q < p.C), ci0 + q, p.C);
            else ra[i] = gload4<false>(rsX, xoff, ok, ci0 + q, p.C);
        }
    };
    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_part(kt, i, true);
    };
    auto store_tiles = [&](int buf) {
        float* As = smem[buf];
        float* Bs = smem[buf] + BK * LDKN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&As[((tid >> 5) + 8 * i) * LDKN + q]) = ra[i];
            *reinterpret_cast<float4*>(&Bs[((tid >> 5) + 8 * i) * LDKN + q]) = rb[i];
            if (do_bias) { bsum.x += rb[i].x; bsum.y += rb[i].y; bsum.z += rb[i].z; bsum.w += rb[i].w; }
        }
    };

    auto store_a = [&](int buf, int i) {
        *reinterpret_cast<float4*>(&smem[buf][((tid >> 5) + 8 * i) * LDKN + q]) = ra[i];
    };
    auto store_b = [&](int buf, int i) {
        *reinterpret_cast<float4*>(&smem[buf][BK * LDKN + ((tid >> 5) + 8 * i) * LDKN + q]) = rb[i];
        if (do_bias) {      // workgroup-uniform; the empty asm keeps it a real branch (if-converted it costs every
            asm volatile("" ::: "memory");   // workgroup 4 adds + 4 selects per store, all but 1/mtiles for nothing)
            bsum.x += rb[i].x; bsum.y += rb[i].y; bsum.z += rb[i].z; bsum.w += rb[i].w;
        }
    };

    // LDS -> register fragments of k-step kk; both operands are k-major: lane half h takes k = kk*8+4h+j
    auto load_frag = [&](const float* As, const float* Bs, int kk, float (&fa)[MB][4], float (&fb)[NB][4]) {
        const float* ap = &As[(kk * 8 + half * 4) * LDKN + wrow + l31];
        const float* bp = &Bs[(kk * 8 + half * 4) * LDKN + wcol + l31];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) fa[mb][j] = ap[j * LDKN + mb * 32];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) fb[nb][j] = bp[j * LDKN + nb * 32];
        }
    };

#ifdef DPIG_TRACE
    const bool trace_on = (blockIdx.x + gridDim.x * blockIdx.z == 3) && ((tid & 63) == 0);
    int trace_n = 0;
#undef DPIG_STAMP
#define DPIG_STAMP(slot) do { if (trace_on && trace_n < 2000) dpig_trace_buf[(tid >> 6) * 2000 + trace_n++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); } while (0)
    DPIG_STAMP(0);
#endif
    if (kt_begin < kt_end) {
        load_tiles(kt_begin);
        store_tiles(0);
        // The operands of this GEMM stream from HBM (a pixel is used by one k-tile), so a load needs a whole
        // k-tile period to land: the registers of part i are re-loaded with tile t+2 right after the LDS stores
        // that publish their tile t+1 contents (groups 2-3 of tile t), not at the head of tile t+1.  No second
        // register set.  (With the loads at the heads of groups 0-1 the stores of groups 2-3 waited on vmcnt:
        // 148 cycles per MFMA there in the s_memtime trace, against 92 in groups 0-1.)
#pragma unroll
        for (int i = 0; i < 4; ++i) load_part(kt_begin + 1, i, kt_begin + 1 < kt_end);
        __syncthreads();
        int buf = 0;
        float fa[2][MB][4], fb[2][NB][4];    // [k-step parity][32-row/col block][j]
        load_frag(smem[0], smem[0] + BK * LDKN, 0, fa[0], fb[0]);
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more2 = (kt + 2) < kt_end;
            const float* As = smem[buf];
            const float* Bs = smem[buf] + BK * LDKN;
            // hand-over schedule as in gather_gemm_kernel: the LDS stores of tile t+1 ride behind the MFMA quads of
            // groups 2 and 3, the barrier and the first fragment reads sit before the last quad
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                if (kk == 0) DPIG_STAMP(1);
                if (kk == 2) DPIG_STAMP(3);
                if (kk + 1 < BK / 8) load_frag(As, Bs, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (kk == 3 && j == 3) {
                        DPIG_STAMP(2);
                        __syncthreads();
                        DPIG_STAMP(4);
                        load_frag(smem[buf ^ 1], smem[buf ^ 1] + BK * LDKN, 0, fa[0], fb[0]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][mb][j], fb[kk & 1][nb][j],
                                                                               acc[mb][nb], 0, 0, 0);
                    if (kk == 2) {                     // (zeros after the last tile: nobody reads them)
                        __builtin_amdgcn_sched_barrier(0);
                        if ((j & 1) == 0) store_a(buf ^ 1, j >> 1);
                        else { store_b(buf ^ 1, j >> 1); load_part(kt + 2, j >> 1, more2); }
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (kk == 3 && j < 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        store_a(buf ^ 1, 2 + j); store_b(buf ^ 1, 2 + j);
                        load_part(kt + 2, 2 + j, more2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
    }

    }   // fp32 pixel loop

    // ---- epilogue: dw rows are (wtap*C + ci), cols co; staged through LDS for 16-byte stores ----
    const long wsize = (long)p.wrows * p.K;
    float* Cs = &smem[0][0];                       // 128 x 128 floats = the whole 64 KB ring
    __syncthreads();
    if (do_bias) {                                 // 8 thread rows x 128 columns -> 128 column sums
        *reinterpret_cast<float4*>(&Cs[(tid >> 5) * BN + q]) = bsum;
        __syncthreads();
        if (tid < BNT) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) v += Cs[r * BN + tid];
            const int co = co0 + tid;
            if (co < p.K) {
                if (p.nsplit > 1) p.bias_partial[(long)split * p.K + co] = v;
                else p.DB[co] = (p.beta_b != 0.f) ? p.beta_b * p.DB[co] + v : v;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                Cs[(wrow + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * BN + wcol + nb * 32 + l31] = acc[mb][nb][r];
    __syncthreads();
    float* dst = (p.nsplit > 1) ? p.partial + (long)split * wsize : p.DW;
    const float beta = (p.nsplit > 1) ? 0.f : p.beta;
    if (p.vec_epi) {
        const int c = (tid & 31) * 4;
        const int co = co0 + c;
        if (co < p.K && c < BNT) {
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int rl = (tid >> 5) + 8 * it;
                const int ci = ci0 + rl;                       // FLAT: flattened (tap, ci) row
                if (ci >= (FLAT ? p.ntaps * p.C : p.C)) continue;
                float4 v = *reinterpret_cast<const float4*>(&Cs[rl * BN + c]);
                float4* o = reinterpret_cast<float4*>(dst + ((long)wt * p.C + ci) * p.K + co);
                if (beta != 0.f) {
                    const float4 old = *o;
                    v.x += beta * old.x; v.y += beta * old.y; v.z += beta * old.z; v.w += beta * old.w;
                }
                *o = v;
            }
        }
    } else {
        for (int idx = tid; idx < BM * BN; idx += 256) {
            const int rl = idx >> 7, cl = idx & 127;
            const int ci = ci0 + rl, co = co0 + cl;
            if (ci >= (FLAT ? p.ntaps * p.C : p.C) || co >= p.K || cl >= BNT) continue;
            const long o = ((long)wt * p.C + ci) * p.K + co;
            const float v = Cs[rl * BN + cl];
            dst[o] = (beta != 0.f) ? beta * dst[o] + v : v;
        }
    }
}

// out[i] = beta*out[i] + sum_s partial[s][i]   (n multiple of 4, 16B aligned)
// The last block also folds the fused bias-gradient partials (bpart[s][co] -> db[co]) when db != nullptr,
// so a split-K wgrad costs one reduce launch, not two.
__device__ __forceinline__ void splitk_bias_tail(const float* __restrict__ bpart, float* __restrict__ db, int K,
                                                 int nsplit, float beta_b) {
    if (db == nullptr || blockIdx.x != gridDim.x - 1) return;
    for (int co = threadIdx.x; co < K; co += blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < nsplit; ++s) v += bpart[(long)s * K + co];
        db[co] = (beta_b != 0.f) ? beta_b * db[co] + v : v;
    }
}
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                          long n4, int nsplit, float beta,
                                                          const float* __restrict__ bpart, float* __restrict__ db,
                                                          int K, float beta_b) {
    const float4* p4 = reinterpret_cast<const float4*>(partial);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int s = 0; s < nsplit; ++s) {
            const float4 t = p4[(long)s * n4 + i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (beta != 0.f) {
            const float4 o = o4[i];
            v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w;
        }
        o4[i] = v;
    }
    splitk_bias_tail(bpart, db, K, nsplit, beta_b);
}
__global__ __launch_bounds__(256) void splitk_sum_scalar_kernel(const float* __restrict__ partial,
                                                                 float* __restrict__ out, long n, int nsplit,
                                                                 float beta, const float* __restrict__ bpart,
                                                                 float* __restrict__ db, int K, float beta_b) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < nsplit; ++s) v += partial[(long)s * n + i];
        out[i] = (beta != 0.f) ? beta * out[i] + v : v;
    }
    splitk_bias_tail(bpart, db, K, nsplit, beta_b);
}

// ================================================================================================
// host side
// ================================================================================================
static bool vec_ok(const void* a, const void* b, int lda, int Cs, int Ncols) {
    return aligned16(a) && aligned16(b) && (lda % 4 == 0) && (Cs % 4 == 0) && (Ncols % 4 == 0);
}

// k-tile depth of the GEMM loop: 64 when the caller asked for the bf16 matrix pipe and the problem has the shape the
// bf16 loop is written for (16-byte loadable operands, 128-wide tile), else 32 (fp32 MFMA)
static int gg_bk(const DpigConvDesc* d, int lda, int Cs, int Ncols) {
    return (d->compute == DPIG_COMPUTE_BF16 && lda % 4 == 0 && Cs % 4 == 0 && Ncols % 4 == 0 && Ncols > 32) ? BKH : BK;
}
// What a split-K partial-sum round trip costs relative to a k-tile, for choose_split: the split-bf16 k-tile runs ~2.4x
// faster than the exact fp32 one while the fp32 partials cost the same, so splitting pays later there.
static double split_pen(const DpigConvDesc* d) {
    static const double x3 = getenv("DPIG_X3_SPLIT_PEN") ? atof(getenv("DPIG_X3_SPLIT_PEN")) : 120.0;
    static const double f32 = getenv("DPIG_F32_SPLIT_PEN") ? atof(getenv("DPIG_F32_SPLIT_PEN")) : 120.0;
    return d->compute == DPIG_COMPUTE_BF16X3 ? x3 : f32;
}
// matrix pipe of the GEMM loop (the PIPE template argument): the bf16 and split-bf16 loops need the same shape; the split
// loop keeps the fp32 loop's k-tile, so every plan / workspace size of the exact path holds for it
static int gg_pipe(const DpigConvDesc* d, int lda, int Cs, int Ncols, const void* a, const void* b) {
    if (d->compute == DPIG_COMPUTE_F32) return 0;
    const bool shape = lda % 4 == 0 && Cs % 4 == 0 && Ncols % 4 == 0 && Ncols > 32 && aligned16(a) && aligned16(b);
    return shape ? (d->compute == DPIG_COMPUTE_BF16X3 ? 2 : 1) : 0;
}

// derived fields of one problem; *vec / *narrow select the kernel variant
static int prepare_gg(GGParams& p, int nimg, long filter_elems, bool* vec, bool* narrow, int pipe = 0) {
    p.HrWr = p.Hr * p.Wr;
    find_divisor(p.HrWr, &p.mul_hrwr, &p.shr_hrwr);
    find_divisor(p.Wr, &p.mul_wr, &p.shr_wr);
    const long a_elems = ((long)nimg * p.Hs * p.Ws - 1) * p.lda + p.Cs;
    if (a_elems * 4 >= 0x7fffffffL || filter_elems * 4 >= 0x7fffffffL)
        return fail(DPIG_EINVAL, "tensor exceeds the 2 GiB buffer-descriptor range");
    p.a_bytes = (unsigned)(a_elems * 4);
    p.b_bytes = (unsigned)(filter_elems * 4);
    p.vec_epi = (p.Ncols % 4 == 0) && (p.ldd % 4 == 0) && aligned16(p.D) &&
                (!p.bias || aligned16(p.bias)) && (!p.res || (p.ldres % 4 == 0 && aligned16(p.res))) &&
                (!p.mask || (p.ldmask % 4 == 0 && aligned16(p.mask))) &&
                (!p.D2 || (p.ldd2 % 4 == 0 && aligned16(p.D2))) && (!p.partial || aligned16(p.partial));
    p.red_lanes = p.vec_epi ? reduce_lanes(p.nsplit, (long)p.M * p.Ncols / 4) : 1;
    *narrow = p.Ncols <= 32;
    p.mtiles = cdiv(p.M, BM);
    p.ntiles = cdiv(p.Ncols, *narrow ? 32 : BN);
    p.cchunks = cdiv(p.Cs, pipe == 1 ? BKH : BK);
    p.ktiles = p.ntaps * p.cchunks;
    *vec = vec_ok(p.A, p.B, p.lda, p.Cs, p.Ncols);
    p.vec_a = aligned16(p.A) && (p.lda % 4 == 0) && (p.Cs % 4 == 0);
    return DPIG_OK;
}
static int reduce_blocks(const GGParams& p) {
    const long total = (long)p.M * p.Ncols;
    const long threads = p.vec_epi ? (total / 4) * p.red_lanes : total;
    const int blocks = cdiv(threads, 256);
    return blocks > 8 * kNumCU ? 8 * kNumCU : blocks;
}
static int launch_gg(GGParams& p, bool b_rowk, int nimg, long filter_elems, hipStream_t st, int pipe = 0) {
    bool vec, narrow;
    int rc = prepare_gg(p, nimg, filter_elems, &vec, &narrow, pipe);
    if (rc) return rc;
    dim3 grid(p.mtiles * p.ntiles, 1, p.nsplit), block(256);
    if (pipe && (!vec || narrow)) return fail(DPIG_EINVAL, "internal: bf16 loop selected for an ineligible problem");
    if (!p.vec_epi || narrow || p.replicate) p.D32 = nullptr;      // the float4 epilogue (in the kernel, or in the split-K
                                                                   // reduction pass) is what writes the image
    if (p.stats && (!p.vec_epi || narrow || !aligned16(p.stats) || (p.nsplit != 1 && (!p.identity_rows || p.replicate))))
        return fail(DPIG_EINVAL, "conv fwd with BN statistics needs 16-byte aligned operands and more than 32 output channels");
    float* const split_stats = (p.stats && p.nsplit > 1) ? p.stats : nullptr;      // split-K: the reduction pass leaves the statistics
    if (split_stats) p.stats = nullptr;
#define DPIG_GG(BR, VE, NA) hipLaunchKernelGGL((gather_gemm_kernel<BR, VE, NA>), grid, block, 0, st, p)
    if (pipe == 1) {
        if (b_rowk) hipLaunchKernelGGL((gather_gemm_kernel<true, true, false, 1>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gather_gemm_kernel<false, true, false, 1>), grid, block, 0, st, p);
    } else if (pipe == 2) {
        static const bool dbg = getenv("DPIG_DEBUG_PIPE2") != nullptr;       // which layers split the filter in the loop
        if (dbg) fprintf(stderr, "[dpig] pipe 2 %s: M %d Ncols %d Cs %d taps %d nsplit %d\n", b_rowk ? "dgrad" : "fwd", p.M, p.Ncols, p.Cs, p.ntaps, p.nsplit);
        if (b_rowk) hipLaunchKernelGGL((gather_gemm_kernel<true, true, false, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gather_gemm_kernel<false, true, false, 2>), grid, block, 0, st, p);
    } else if (pipe == 3) {
        hipLaunchKernelGGL((gather_gemm_kernel<true, true, false, 3>), grid, block, 0, st, p);
    } else if (pipe == 4) {
        hipLaunchKernelGGL((gather_gemm_kernel<true, true, false, 4>), grid, block, 0, st, p);
    } else if (b_rowk) {
        if (narrow) { if (vec) DPIG_GG(true, true, true); else DPIG_GG(true, false, true); }
        else { if (vec) DPIG_GG(true, true, false); else DPIG_GG(true, false, false); }
    } else {
        if (narrow) { if (vec) DPIG_GG(false, true, true); else DPIG_GG(false, false, true); }
        else { if (vec) DPIG_GG(false, true, false); else DPIG_GG(false, false, false); }
    }
#undef DPIG_GG
    rc = check_launch("gather_gemm_kernel");
    if (rc) return rc;
    if (split_stats) {
        p.stats = split_stats;
        hipLaunchKernelGGL(gather_gemm_reduce_stats_kernel, dim3(p.mtiles * p.ntiles), dim3(256), 0, st, p);
        rc = check_launch("gather_gemm_reduce_stats_kernel");
    } else if (p.nsplit > 1) {
        hipLaunchKernelGGL(gather_gemm_reduce_kernel, dim3(reduce_blocks(p)), dim3(256), 0, st, p);
        rc = check_launch("gather_gemm_reduce_kernel");
    }
    return rc;
}
// n <= 4 problems of the same variant (dgrad: B is [N][K]) in one launch + at most one reduction launch
static int launch_gg_multi(GGParams* q, int n, int nimg, long filter_elems, hipStream_t st, int pipe = 0) {
    GGMulti m = {};
    bool vec = false, narrow = false;
    int max_tiles = 0, max_split = 1, max_red = 0;
    for (int i = 0; i < n; ++i) {
        bool v, na;
        const int rc = prepare_gg(q[i], nimg, filter_elems, &v, &na, pipe);
        if (rc) return rc;
        if (i == 0) { vec = v; narrow = na; }
        else if (v != vec || na != narrow) return fail(DPIG_EINVAL, "internal: multi-launch variants differ");
        if (q[i].mtiles * q[i].ntiles > max_tiles) max_tiles = q[i].mtiles * q[i].ntiles;
        if (q[i].nsplit > max_split) max_split = q[i].nsplit;
        if (q[i].nsplit > 1 && reduce_blocks(q[i]) > max_red) max_red = reduce_blocks(q[i]);
        m.q[i] = q[i];
    }
    dim3 grid(max_tiles, n, max_split), block(256);
    if (pipe && (!vec || narrow)) return fail(DPIG_EINVAL, "internal: bf16 loop selected for an ineligible problem");
#define DPIG_GGM(VE, NA) hipLaunchKernelGGL((gather_gemm_multi_kernel<true, VE, NA>), grid, block, 0, st, m)
    if (pipe == 1) hipLaunchKernelGGL((gather_gemm_multi_kernel<true, true, false, 1>), grid, block, 0, st, m);
    else if (pipe == 2) hipLaunchKernelGGL((gather_gemm_multi_kernel<true, true, false, 2>), grid, block, 0, st, m);
    else if (pipe == 3) hipLaunchKernelGGL((gather_gemm_multi_kernel<true, true, false, 3>), grid, block, 0, st, m);
    else if (pipe == 4) hipLaunchKernelGGL((gather_gemm_multi_kernel<true, true, false, 4>), grid, block, 0, st, m);
    else if (narrow) { if (vec) DPIG_GGM(true, true); else DPIG_GGM(false, true); }
    else { if (vec) DPIG_GGM(true, false); else DPIG_GGM(false, false); }
#undef DPIG_GGM
    int rc = check_launch("gather_gemm_multi_kernel");
    if (rc) return rc;
    if (max_red > 0) {
        hipLaunchKernelGGL(gather_gemm_reduce_multi_kernel, dim3(max_red, n), dim3(256), 0, st, m);
        rc = check_launch("gather_gemm_reduce_multi_kernel");
    }
    return rc;
}

// number of (row tiles x col tiles) and k-tiles for each op, used by both the workspace query and the launch
struct Shape { long M; int Ncols, ktiles; };
static Shape fwd_shape(const DpigConvDesc* d, int Ho, int Wo, int bk = BK) {
    Shape s;
    s.M = d->upsample2x ? (long)d->N * d->H * d->W : (long)d->N * Ho * Wo;
    s.Ncols = d->K;
    s.ktiles = d->R * d->S * cdiv(d->C, bk);
    return s;
}

}  // namespace dpig

using namespace dpig;

extern "C" int dpig_same_pad(int in, int k, int stride, int* out, int* pad_before) {
    const int o = (in + stride - 1) / stride;
    int total = (o - 1) * stride + k - in;
    if (total < 0) total = 0;
    if (out) *out = o;
    if (pad_before) *pad_before = total / 2;
    return DPIG_OK;
}

// shape conditions of the stride-1 SAME wgrad variant (pointer alignment is checked at launch)
static bool wgrad_s1_shape(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo) {
    return d->stride == 1 && !d->upsample2x && d->K > 32 && !(d->C < 32 && d->R * d->S > 1) && Ho == d->H && Wo == d->W &&
           pt >= 0 && pl >= 0 && d->ldx % 4 == 0 && d->ldy % 4 == 0 && d->C % 4 == 0 && d->K % 4 == 0;
}

static size_t conv2d_workspace_bytes_one(const DpigConvDesc* d, int which) {
    int pt, pl, Ho, Wo;
    if (resolve_desc(d, &pt, &pl, &Ho, &Wo)) return 0;
    // (pointer alignment is not known here: when the bf16 loop may be chosen, size for the larger of the two plans)
    if (which == 0) {
        size_t best = 0;
        for (int bk = BK; bk <= gg_bk(d, d->ldx, d->C, d->K); bk += BK) {
            Shape s = fwd_shape(d, Ho, Wo, bk);
            Plan pln = plan_split(cdiv(s.M, BM) * cdiv(s.Ncols, BN), s.ktiles, d->split_k, 32, split_pen(d));
            const size_t b = pln.nsplit > 1 ? (size_t)pln.nsplit * s.M * s.Ncols * sizeof(float) : 0;
            if (b > best) best = b;
        }
        return best;
    } else if (which == 1) {
        size_t best = 0;
        for (int bk = BK; bk <= gg_bk(d, d->ldy, d->K, d->C); bk += BK) {
            size_t b;
            if (d->upsample2x || d->stride == 1) {
                const long M = (long)d->N * d->H * d->W;
                const int ntaps = d->upsample2x ? 4 : d->R * d->S;
                Plan pln = plan_split(cdiv(M, BM) * cdiv(d->C, BN), ntaps * cdiv(d->K, bk), d->split_k, 32, split_pen(d));
                b = pln.nsplit > 1 ? (size_t)pln.nsplit * M * d->C * sizeof(float) : 0;
            } else {
                S2Plan sp;
                plan_dgrad_s2(d, pt, pl, &sp, bk, split_pen(d));
                b = sp.total;
            }
            if (b > best) best = b;
        }
        return best;
    } else if (which == 2) {
        const size_t thin = thin_wgrad_workspace_bytes(d, pt, pl);
        if (thin) return thin;
        const size_t fewc = fewc_wgrad_workspace_bytes(d);
        if (fewc) return fewc;
        const long Npix = (long)d->N * Ho * Wo * (d->upsample2x ? 4 : 1);
        const bool flat = !d->upsample2x && d->C < 32 && d->R * d->S > 1;
        const int tiles = (flat ? cdiv((long)d->R * d->S * d->C, BM) : d->R * d->S * cdiv(d->C, BM)) *
                          cdiv(d->K, d->K <= 32 ? 32 : BN);
        size_t best = 0;
        const bool bf_shape = d->K > 32 && !flat && d->ldx % 4 == 0 && d->ldy % 4 == 0 && d->C % 4 == 0 && d->K % 4 == 0;
        const int bkmax = (d->compute == DPIG_COMPUTE_BF16 && bf_shape) ? BKH : BK;
        for (int bk = BK; bk <= bkmax; bk += BK) {
            Plan pln = plan_split(tiles, cdiv(Npix, bk), d->split_k, 32, split_pen(d));
            const size_t b = pln.nsplit > 1 ? (size_t)pln.nsplit * ((size_t)d->R * d->S * d->C * d->K + d->K) * sizeof(float) : 0;
            if (b > best) best = b;
        }
        return best;
    }
    return 0;
}

// ---- entry points: one launch, or runs of whole images when a tensor exceeds one launch's 2 GiB range (dpig_conv_plan.h) ----
// hi / lo bf16 planes of a filter's split shadow (rows n, k contiguous); null hi = none
struct SplitShadow {
    const unsigned short* hi; const unsigned short* lo;
    const unsigned short* a32;     // the gathered activation in split32 layout (dpig_split32) or null
    unsigned short* out32;         // where to leave the OUTPUT's split32 image (or null); *wrote says whether it was written
    int* wrote;
};
static int conv2d_fwd_one(const DpigConvDesc* d, const float* x, const float* w, const float* bias,
                          const float* residual, float* y, float* y_act, void* ws, size_t ws_bytes, void* stream,
                          float* stats = nullptr, SplitShadow sh = SplitShadow{nullptr, nullptr, nullptr, nullptr, nullptr});
static int conv2d_dgrad_one(const DpigConvDesc* d, const float* dy, const float* w, const float* accum,
                            const float* mask, float* dx, void* ws, size_t ws_bytes, void* stream,
                            SplitShadow sh = SplitShadow{nullptr, nullptr, nullptr, nullptr, nullptr});
// PIPE 3 is PIPE 2 with the filter from its split shadow: same shapes, plus 16-byte k-granules and one descriptor over both planes
static bool shadow_usable(const DpigConvDesc* d, int pipe, int Cs, long filter_elems, SplitShadow sh) {
    if (pipe != 2 || !sh.hi || !sh.lo || Cs % 8 || !aligned16(sh.hi) || !aligned16(sh.lo)) return false;
    const long off = reinterpret_cast<const char*>(sh.lo) - reinterpret_cast<const char*>(sh.hi);
    (void)d;
    return off > 0 && off + filter_elems * 2 < 0x7fffffffL;
}
// PIPE 4 on top of PIPE 3: the gathered activation's split32 image ([pixel][chunk][32 hi | 32 lo]) inside one descriptor
static bool split32_usable(GGParams& p, SplitShadow sh, long pixels, int channels) {
    const long bytes = pixels * cdiv(channels, 32) * 128;
    if (!sh.a32 || !aligned16(sh.a32) || bytes >= 0x7fffffffL) return false;
    p.As32 = sh.a32; p.as_bytes = (unsigned)bytes; p.a_nchunk = cdiv(channels, 32);
    return true;
}
static int conv2d_wgrad_one(const DpigConvDesc* d, const float* x, const float* dy, float* dw, float beta,
                            float* db, float beta_b, void* ws, size_t ws_bytes, void* stream);

extern "C" size_t dpig_conv2d_workspace_bytes(const DpigConvDesc* d, int which) {
    const int per = images_per_launch(d, 4);
    if (!d || per <= 0 || per >= d->N) return conv2d_workspace_bytes_one(d, which);
    DpigConvDesc c = *d;                         // the full runs and the remainder plan their split-K separately
    c.N = per;
    size_t best = conv2d_workspace_bytes_one(&c, which);
    if (d->N % per) {
        c.N = d->N % per;
        const size_t b = conv2d_workspace_bytes_one(&c, which);
        if (b > best) best = b;
    }
    return best;
}

extern "C" int dpig_conv2d_fwd(const DpigConvDesc* d, const float* x, const float* w, const float* bias,
                               const float* residual, float* y, float* y_act, void* ws, size_t ws_bytes,
                               void* stream) {
    const int per = images_per_launch(d, 4);
    if (d && per == 0) return fail(DPIG_EINVAL, "one image exceeds the 2 GiB range of a launch");
    if (!d || per >= d->N || !x || !y) return conv2d_fwd_one(d, x, w, bias, residual, y, y_act, ws, ws_bytes, stream);
    long xpix, ypix;
    image_pixels(d, &xpix, &ypix);
    for (int n0 = 0; n0 < d->N; n0 += per) {
        DpigConvDesc c = *d;
        c.N = d->N - n0 < per ? d->N - n0 : per;
        const float* r = !residual ? nullptr : residual + (d->res_class ? (long)n0 * 9 * d->ldres : (long)n0 * ypix * d->ldres);
        const int rc = conv2d_fwd_one(&c, x + (long)n0 * xpix * d->ldx, w, bias, r, y + (long)n0 * ypix * d->ldy,
                                      y_act ? y_act + (long)n0 * ypix * d->ldy2 : nullptr, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    return DPIG_OK;
}

// DPIG_COMPUTE_BF16X3 with the filter's two-term split precomputed (dpig_filter_shadow_split*): `w` is still the fp32 filter
// (used when a layer cannot take the shadow path: thin layers, <= 32 output columns, C or K not a multiple of 8), w_*_hi / _lo
// the bf16 planes -- transposed [R,S,K,C] for forward, plain [R,S,C,K] for dgrad.  Results equal dpig_conv2d_fwd / _dgrad
// with compute = DPIG_COMPUTE_BF16X3 bit for bit (same products, same order).
extern "C" int dpig_conv2d_fwd_x3(const DpigConvDesc* d, const float* x, const uint16_t* x32, const float* w,
                                  const uint16_t* w_t_hi, const uint16_t* w_t_lo, const float* bias, const float* residual,
                                  float* y, float* y_act, uint16_t* y32, int* y32_written, void* ws, size_t ws_bytes,
                                  void* stream) {
    const int per = images_per_launch(d, 4);
    if (d && per == 0) return fail(DPIG_EINVAL, "one image exceeds the 2 GiB range of a launch");
    if (y32_written) *y32_written = 0;
    SplitShadow sh{w_t_hi, w_t_lo, x32, y32, y32_written};
    if (d && per < d->N) { sh.a32 = nullptr; sh.out32 = nullptr; sh.wrote = nullptr; }   // (batches served in several runs: no images)
    if (!d || per >= d->N || !x || !y) return conv2d_fwd_one(d, x, w, bias, residual, y, y_act, ws, ws_bytes, stream, nullptr, sh);
    long xpix, ypix;
    image_pixels(d, &xpix, &ypix);
    for (int n0 = 0; n0 < d->N; n0 += per) {
        DpigConvDesc c = *d;
        c.N = d->N - n0 < per ? d->N - n0 : per;
        const float* r = !residual ? nullptr : residual + (d->res_class ? (long)n0 * 9 * d->ldres : (long)n0 * ypix * d->ldres);
        const int rc = conv2d_fwd_one(&c, x + (long)n0 * xpix * d->ldx, w, bias, r, y + (long)n0 * ypix * d->ldy,
                                      y_act ? y_act + (long)n0 * ypix * d->ldy2 : nullptr, ws, ws_bytes, stream, nullptr, sh);
        if (rc) return rc;
    }
    return DPIG_OK;
}
extern "C" int dpig_conv2d_dgrad_x3(const DpigConvDesc* d, const float* dy, const uint16_t* dy32, const float* w,
                                    const uint16_t* w_hi, const uint16_t* w_lo, const float* accum, const float* mask,
                                    float* dx, uint16_t* dx32, int* dx32_written, void* ws, size_t ws_bytes, void* stream) {
    const int per = images_per_launch(d, 4);
    if (d && per == 0) return fail(DPIG_EINVAL, "one image exceeds the 2 GiB range of a launch");
    if (dx32_written) *dx32_written = 0;
    SplitShadow sh{w_hi, w_lo, dy32, dx32, dx32_written};
    if (d && per < d->N) { sh.a32 = nullptr; sh.out32 = nullptr; sh.wrote = nullptr; }
    if (!d || per >= d->N || !dy || !dx) return conv2d_dgrad_one(d, dy, w, accum, mask, dx, ws, ws_bytes, stream, sh);
    long xpix, ypix;
    image_pixels(d, &xpix, &ypix);
    for (int n0 = 0; n0 < d->N; n0 += per) {
        DpigConvDesc c = *d;
        c.N = d->N - n0 < per ? d->N - n0 : per;
        const int rc = conv2d_dgrad_one(&c, dy + (long)n0 * ypix * d->ldy, w, accum ? accum + (long)n0 * xpix * d->ldres : nullptr,
                                        mask ? mask + (long)n0 * xpix * d->ldmask : nullptr, dx + (long)n0 * xpix * d->ldx, ws,
                                        ws_bytes, stream, sh);
        if (rc) return rc;
    }
    return DPIG_OK;
}

// Forward conv (+ bias) that also leaves the batch-norm partial statistics of its output: row tiles of 128 output pixels,
// stats[tile][0][K] = sum, stats[tile][1][K] = sum of squared deviations from the tile's mean.  The tile count is 0 when this
// problem's plan cannot carry them (split-K, <= 32 output channels, the upsample fusion, a batch served in several runs):
// the caller then runs dpig_bn_fwd's own statistics passes.
static int bn_stats_tiles(const DpigConvDesc* d, bool allow_split) {
    int pt, pl, Ho, Wo;
    if (resolve_desc(d, &pt, &pl, &Ho, &Wo)) return 0;
    if (d->upsample2x || d->K <= 32 || d->K % 4 || d->ldy % 4 || d->act != DPIG_ACT_NONE || images_per_launch(d, 4) < d->N) return 0;
    const int bk = gg_bk(d, d->ldx, d->C, d->K);
    Shape s = fwd_shape(d, Ho, Wo, bk);
    Plan pln = plan_split(cdiv(s.M, BM) * cdiv(s.Ncols, BN), s.ktiles, d->split_k, 32, split_pen(d));
    return (pln.nsplit == 1 || allow_split) ? (int)cdiv(s.M, BM) : 0;
}
extern "C" int dpig_conv2d_bn_stats_tiles(const DpigConvDesc* d) { return bn_stats_tiles(d, false); }
// ... with a workspace (dpig_conv2d_workspace_bytes(d, 0)) the split-K plans carry the statistics too: their reduction pass leaves them
extern "C" int dpig_conv2d_bn_stats_tiles_ws(const DpigConvDesc* d) { return bn_stats_tiles(d, true); }
extern "C" int dpig_conv2d_fwd_stats(const DpigConvDesc* d, const float* x, const float* w, const float* bias, float* y,
                                     float* stats, void* stream) {
    if (!stats) return fail(DPIG_EINVAL, "conv fwd with BN statistics: null statistics buffer");
    if (dpig_conv2d_bn_stats_tiles(d) <= 0)
        return fail(DPIG_EINVAL, "conv fwd with BN statistics: this problem's plan cannot carry them (dpig_conv2d_bn_stats_tiles == 0)");
    return conv2d_fwd_one(d, x, w, bias, nullptr, y, nullptr, nullptr, 0, stream, stats);
}

extern "C" int dpig_conv2d_fwd_stats_ws(const DpigConvDesc* d, const float* x, const float* w, const float* bias, float* y,
                                        float* stats, void* ws, size_t ws_bytes, void* stream) {
    if (!stats) return fail(DPIG_EINVAL, "conv fwd with BN statistics: null statistics buffer");
    if (dpig_conv2d_bn_stats_tiles_ws(d) <= 0)
        return fail(DPIG_EINVAL, "conv fwd with BN statistics: this problem cannot carry them (dpig_conv2d_bn_stats_tiles_ws == 0)");
    return conv2d_fwd_one(d, x, w, bias, nullptr, y, nullptr, ws, ws_bytes, stream, stats);
}

extern "C" int dpig_conv2d_dgrad(const DpigConvDesc* d, const float* dy, const float* w, const float* accum,
                                 const float* mask, float* dx, void* ws, size_t ws_bytes, void* stream) {
    const int per = images_per_launch(d, 4);
    if (d && per == 0) return fail(DPIG_EINVAL, "one image exceeds the 2 GiB range of a launch");
    if (!d || per >= d->N || !dy || !dx) return conv2d_dgrad_one(d, dy, w, accum, mask, dx, ws, ws_bytes, stream);
    long xpix, ypix;
    image_pixels(d, &xpix, &ypix);
    for (int n0 = 0; n0 < d->N; n0 += per) {
        DpigConvDesc c = *d;
        c.N = d->N - n0 < per ? d->N - n0 : per;
        const int rc = conv2d_dgrad_one(&c, dy + (long)n0 * ypix * d->ldy, w, accum ? accum + (long)n0 * xpix * d->ldres : nullptr,
                                        mask ? mask + (long)n0 * xpix * d->ldmask : nullptr, dx + (long)n0 * xpix * d->ldx, ws,
                                        ws_bytes, stream);
        if (rc) return rc;
    }
    return DPIG_OK;
}

extern "C" int dpig_conv2d_wgrad(const DpigConvDesc* d, const float* x, const float* dy, float* dw, float beta,
                                 float* db, float beta_b, void* ws, size_t ws_bytes, void* stream) {
    const int per = images_per_launch(d, 4);
    if (d && per == 0) return fail(DPIG_EINVAL, "one image exceeds the 2 GiB range of a launch");
    if (!d || per >= d->N || !x || !dy) return conv2d_wgrad_one(d, x, dy, dw, beta, db, beta_b, ws, ws_bytes, stream);
    long xpix, ypix;
    image_pixels(d, &xpix, &ypix);
    for (int n0 = 0; n0 < d->N; n0 += per) {     // image runs accumulate in order: same result on every call
        DpigConvDesc c = *d;
        c.N = d->N - n0 < per ? d->N - n0 : per;
        const int rc = conv2d_wgrad_one(&c, x + (long)n0 * xpix * d->ldx, dy + (long)n0 * ypix * d->ldy, dw, n0 ? 1.0f : beta, db,
                                        n0 ? 1.0f : beta_b, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    return DPIG_OK;
}

// DPIG_COMPUTE_BF16X3 filter gradient from the split32 images of x and dy (dpig_split32, or left by the producing epilogues):
// both operands by LDS-DMA, transposed on their way out of LDS (bw3_kernel, dpig_conv_bf16.hip).  dw is bit-identical to
// dpig_conv2d_wgrad's; db sums dy to its 16 split bits instead of the fp32 values.  Null images, or a layer the kernel does
// not serve (thin layers, <= 32 output channels, a batch served in several runs), run dpig_conv2d_wgrad on x / dy.
extern "C" int dpig_conv2d_wgrad_x3(const DpigConvDesc* d, const float* x, const uint16_t* x32, const float* dy,
                                    const uint16_t* dy32, float* dw, float beta, float* db, float beta_b, void* ws,
                                    size_t ws_bytes, void* stream) {
    if (d && x32 && dy32 && dw && d->compute == DPIG_COMPUTE_BF16X3 && images_per_launch(d, 4) >= d->N) {
        int pt, pl, Ho, Wo;
        int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
        if (rc) return rc;
        rc = wgrad_x3_try(d, pt, pl, Ho, Wo, x32, dy32, dw, beta, db, beta_b, ws, ws_bytes, static_cast<hipStream_t>(stream),
                          split_pen(d));
        if (rc != 0) return rc < 0 ? rc : DPIG_OK;
    }
    return dpig_conv2d_wgrad(d, x, dy, dw, beta, db, beta_b, ws, ws_bytes, stream);
}

static int conv2d_fwd_one(const DpigConvDesc* d, const float* x, const float* w, const float* bias,
                          const float* residual, float* y, float* y_act, void* ws, size_t ws_bytes,
                          void* stream, float* stats, SplitShadow sh) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !w || !y) return fail(DPIG_EINVAL, "null tensor pointer");
    if (d->upsample2x && residual) return fail(DPIG_EINVAL, "residual unsupported with upsample2x");
    GGParams p = {};
    p.A = x; p.B = w; p.D = y; p.bias = bias; p.res = residual; p.mask = nullptr;
    p.D2 = y_act; p.ldd2 = d->ldy2; p.res_post = d->res_after_act;
    p.res_class = (residual && d->res_class) ? 1 : 0;
    if (p.res_class && (d->stride != 1 || d->upsample2x || d->H < 2 || d->W < 2))
        return fail(DPIG_EINVAL, "res_class needs a stride-1 conv on an image of at least 2x2");
    if (y_act && d->ldy2 < d->K) return fail(DPIG_EINVAL, "ldy2 < K");
    if (y_act && d->upsample2x) return fail(DPIG_EINVAL, "y_act unsupported with upsample2x");
    if (residual && d->ldres < d->K) return fail(DPIG_EINVAL, "ldres < K");
    if (!stats) {
        rc = thin_fwd_try(d, pt, pl, x, w, bias, residual, y, y_act, static_cast<hipStream_t>(stream));
        if (rc != 0) return rc < 0 ? rc : DPIG_OK;
        rc = fewc_fwd_try(d, pt, pl, Ho, Wo, x, w, bias, residual, y, y_act, static_cast<hipStream_t>(stream));
        if (rc != 0) return rc < 0 ? rc : DPIG_OK;
    }
    p.partial = static_cast<float*>(ws);
    p.stats = stats;
    const int pipe = gg_pipe(d, d->ldx, d->C, d->K, x, w);
    const bool bf16 = pipe == 1;
    Shape s = fwd_shape(d, Ho, Wo, bf16 ? BKH : BK);
    p.M = (int)s.M;
    p.Hr = d->upsample2x ? d->H : Ho; p.Wr = d->upsample2x ? d->W : Wo;
    p.Hs = d->H; p.Ws = d->W; p.lda = d->ldx; p.Cs = d->C; p.sr = d->stride;
    p.Ncols = d->K;
    if (d->upsample2x) { p.Hd = 2 * d->H; p.Wd = 2 * d->W; p.dr = 2; p.replicate = 1; }
    else { p.Hd = Ho; p.Wd = Wo; p.dr = 1; p.replicate = 0; }
    p.dpy = 0; p.dpx = 0; p.ldd = d->ldy; p.ldres = d->ldres; p.ldmask = 0;
    p.identity_rows = d->upsample2x ? 0 : 1;
    p.act = d->act; p.alpha = d->alpha;
    if (residual && d->ldres < d->K) return fail(DPIG_EINVAL, "ldres < K");
    p.ntaps = d->R * d->S;
    p.tap_nb = d->S; p.oy0 = -pt; p.oys = 1; p.ox0 = -pl; p.oxs = 1; p.w0 = 0; p.wa = d->S; p.wb = 1;
    Plan pln = plan_split(cdiv(s.M, BM) * cdiv(s.Ncols, BN), s.ktiles, d->split_k, 32, split_pen(d));
    p.nsplit = pln.nsplit; p.tiles_per_split = pln.tiles_per_split;
    if (p.nsplit > 1 && ws_bytes < (size_t)p.nsplit * s.M * s.Ncols * sizeof(float))
        return fail(DPIG_ENOMEM, "conv fwd workspace too small: have %zu", ws_bytes);
    if (p.nsplit > 1 && !ws) return fail(DPIG_ENOMEM, "conv fwd needs a workspace");
    if (sh.out32 && !d->upsample2x && d->K % 32 == 0 && aligned16(sh.out32)) { p.D32 = sh.out32; p.d_nchunk = d->K / 32; }
    if (shadow_usable(d, pipe, d->C, (long)d->R * d->S * d->C * d->K, sh)) {     // transposed shadow [tap][K][C]: the B_ROWK form
        p.Bs_hi = sh.hi;
        p.bs_lo_off = (unsigned)(reinterpret_cast<const char*>(sh.lo) - reinterpret_cast<const char*>(sh.hi));
        p.bs_bytes = p.bs_lo_off + (unsigned)((long)d->R * d->S * d->C * d->K * 2);
        const int pp = split32_usable(p, sh, (long)d->N * d->H * d->W, d->C) ? 4 : 3;
        rc = launch_gg(p, true, d->N, (long)d->R * d->S * d->C * d->K, static_cast<hipStream_t>(stream), pp);
    } else {
        rc = launch_gg(p, false, d->N, (long)d->R * d->S * d->C * d->K, static_cast<hipStream_t>(stream), pipe);
    }
    if (sh.wrote) *sh.wrote = (rc == DPIG_OK && p.D32) ? 1 : 0;
    return rc;
}

static int conv2d_dgrad_one(const DpigConvDesc* d, const float* dy, const float* w, const float* accum,
                            const float* mask, float* dx, void* ws, size_t ws_bytes, void* stream, SplitShadow sh) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!dy || !w || !dx) return fail(DPIG_EINVAL, "null tensor pointer");
    if (accum && d->ldres < d->C) return fail(DPIG_EINVAL, "ldres < C");
    if (mask && d->ldmask < d->C) return fail(DPIG_EINVAL, "ldmask < C");
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = thin_dgrad_try(d, pt, pl, dy, w, accum, mask, dx, st);
    if (rc != 0) return rc < 0 ? rc : DPIG_OK;
    rc = fewc_dgrad_try(d, pt, pl, Ho, Wo, dy, w, accum, mask, dx, st);
    if (rc != 0) return rc < 0 ? rc : DPIG_OK;
    GGParams p = {};
    p.A = dy; p.B = w; p.D = dx; p.bias = nullptr; p.res = accum; p.mask = mask;
    p.partial = static_cast<float*>(ws);
    p.lda = d->ldy; p.Cs = d->K; p.Ncols = d->C;
    int pipe = gg_pipe(d, d->ldy, d->K, d->C, dy, w);
    const bool bf16 = pipe == 1;
    if (shadow_usable(d, pipe, d->K, (long)d->R * d->S * d->C * d->K, sh)) {        // plain shadow [tap][C][K]
        pipe = 3;
        p.Bs_hi = sh.hi;
        p.bs_lo_off = (unsigned)(reinterpret_cast<const char*>(sh.lo) - reinterpret_cast<const char*>(sh.hi));
        p.bs_bytes = p.bs_lo_off + (unsigned)((long)d->R * d->S * d->C * d->K * 2);
        if (split32_usable(p, sh, (long)d->N * Ho * Wo * (d->upsample2x ? 4 : 1), d->K)) pipe = 4;
    }
    p.Hd = d->H; p.Wd = d->W; p.ldd = d->ldx; p.ldres = d->ldres; p.ldmask = d->ldmask;
    p.act = mask ? d->act : DPIG_ACT_NONE; p.alpha = d->alpha; p.replicate = 0;
    if (d->upsample2x) {
        // rows = low-res pixels; sum of the 2x2 block of dy, all through filter tap 0
        p.M = d->N * d->H * d->W; p.Hr = d->H; p.Wr = d->W;
        p.Hs = 2 * d->H; p.Ws = 2 * d->W; p.sr = 2;
        p.dr = 1; p.dpy = 0; p.dpx = 0; p.identity_rows = 1;
        p.ntaps = 4;
        p.tap_nb = 2; p.oy0 = 0; p.oys = 1; p.ox0 = 0; p.oxs = 1; p.w0 = 0; p.wa = 0; p.wb = 0;
    } else if (d->stride == 1) {
        p.M = d->N * d->H * d->W; p.Hr = d->H; p.Wr = d->W;
        p.Hs = Ho; p.Ws = Wo; p.sr = 1;
        p.dr = 1; p.dpy = 0; p.dpx = 0; p.identity_rows = 1;
        p.ntaps = d->R * d->S;
        p.tap_nb = d->S; p.oy0 = pt; p.oys = -1; p.ox0 = pl; p.oxs = -1; p.w0 = 0; p.wa = d->S; p.wb = 1;
    } else {
        S2Plan sp;
        plan_dgrad_s2(d, pt, pl, &sp, bf16 ? BKH : BK, split_pen(d));
        if (sp.total > 0 && (!ws || ws_bytes < sp.total))
            return fail(DPIG_ENOMEM, "conv dgrad workspace too small: have %zu", ws_bytes);
        GGParams qs[4];
        for (int i = 0; i < sp.nc; ++i) {
            const DClass& c = sp.cls[i];
            GGParams& q = qs[i];
            q = p;
            q.M = (int)sp.M[i]; q.Hr = c.Hr; q.Wr = c.Wr;
            q.Hs = Ho; q.Ws = Wo; q.sr = 1;
            q.dr = 2; q.dpy = c.py; q.dpx = c.px; q.identity_rows = 0;
            q.ntaps = c.ntaps;
            q.tap_nb = c.nkx > 0 ? c.nkx : 1;
            q.oy0 = c.oy0; q.oys = -1; q.ox0 = c.ox0; q.oxs = -1;
            q.w0 = c.ky0 * d->S + c.kx0; q.wa = d->stride * d->S; q.wb = d->stride;
            q.nsplit = sp.pl[i].nsplit; q.tiles_per_split = sp.pl[i].tiles_per_split;
            q.partial = reinterpret_cast<float*>(static_cast<char*>(ws) + sp.off[i]);
        }
        return launch_gg_multi(qs, sp.nc, d->N, (long)d->R * d->S * d->C * d->K, st, pipe);
    }
    Plan pln = plan_split(cdiv(p.M, BM) * cdiv(p.Ncols, BN), p.ntaps * cdiv(p.Cs, bf16 ? BKH : BK), d->split_k, 32, split_pen(d));
    p.nsplit = pln.nsplit; p.tiles_per_split = pln.tiles_per_split;
    if (p.nsplit > 1 && (!ws || ws_bytes < (size_t)p.nsplit * p.M * p.Ncols * sizeof(float)))
        return fail(DPIG_ENOMEM, "conv dgrad workspace too small: have %zu", ws_bytes);
    // (stride 1 and the upsample fusion only: the stride-2 classes above are separate problems that may split differently)
    if (sh.out32 && d->C % 32 == 0 && aligned16(sh.out32)) { p.D32 = sh.out32; p.d_nchunk = d->C / 32; }
    rc = launch_gg(p, true, d->N, (long)d->R * d->S * d->C * d->K, st, pipe);
    if (sh.wrote) *sh.wrote = (rc == DPIG_OK && p.D32) ? 1 : 0;
    return rc;
}

static int conv2d_wgrad_one(const DpigConvDesc* d, const float* x, const float* dy, float* dw, float beta,
                            float* db, float beta_b, void* ws, size_t ws_bytes, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !dy || !dw) return fail(DPIG_EINVAL, "null tensor pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = thin_wgrad_try(d, pt, pl, x, dy, dw, beta, db, beta_b, ws, ws_bytes, st);
    if (rc != 0) return rc < 0 ? rc : DPIG_OK;
    rc = fewc_wgrad_try(d, pt, pl, Ho, Wo, x, dy, dw, beta, db, beta_b, ws, ws_bytes, st);
    if (rc != 0) return rc < 0 ? rc : DPIG_OK;
    WGParams p = {};
    p.X = x; p.DY = dy; p.DW = dw; p.partial = static_cast<float*>(ws);
    p.H = d->H; p.W = d->W; p.ldx = d->ldx; p.C = d->C; p.K = d->K; p.ldy = d->ldy;
    p.beta = beta;
    if (d->upsample2x) {
        p.Ho = 2 * d->H; p.Wo = 2 * d->W; p.shift = 1; p.s = 1;
        p.ntaps = 1; p.S = 1; p.pad_t = 0; p.pad_l = 0;
    } else {
        p.Ho = Ho; p.Wo = Wo; p.shift = 0; p.s = d->stride;
        p.ntaps = d->R * d->S; p.S = d->S; p.pad_t = pt; p.pad_l = pl;
    }
    p.HoWo = p.Ho * p.Wo;
    p.Npix = d->N * p.HoWo;
    {
        const long xe = ((long)d->N * d->H * d->W - 1) * d->ldx + d->C;
        const long ye = ((long)p.Npix - 1) * d->ldy + d->K;
        if (xe * 4 >= 0x7fffffffL || ye * 4 >= 0x7fffffffL)
            return fail(DPIG_EINVAL, "tensor exceeds the 2 GiB buffer-descriptor range");
        p.x_bytes = (unsigned)(xe * 4); p.y_bytes = (unsigned)(ye * 4);
        find_divisor(p.HoWo, &p.mul_howo, &p.shr_howo);
        find_divisor(p.Wo, &p.mul_wo, &p.shr_wo);
    }
    p.wrows = d->R * d->S * d->C;
    const bool narrow = d->K <= 32;
    const bool flat = !d->upsample2x && d->C < 32 && d->R * d->S > 1;
    const bool vec = aligned16(x) && aligned16(dy) && (d->ldx % 4 == 0) && (d->ldy % 4 == 0) &&
                     (d->C % 4 == 0) && (d->K % 4 == 0);
    const bool s1 = vec && wgrad_s1_shape(d, pt, pl, Ho, Wo) &&
                    ((long)p.x_bytes + (long)(p.pad_t * p.W + p.pad_l) * p.ldx * 4 < 0x7fffffffL);
    const bool bf16 = vec && !flat && !narrow && d->compute == DPIG_COMPUTE_BF16;
    const bool split3 = vec && !flat && !narrow && d->compute == DPIG_COMPUTE_BF16X3;
    p.cblocks = cdiv(d->C, BM);
    p.ntiles = cdiv(d->K, narrow ? 32 : BN);
    p.ktiles = cdiv(p.Npix, bf16 ? BKH : BK);
    const int tiles = (flat ? cdiv((long)p.ntaps * d->C, BM) : p.ntaps * p.cblocks) * p.ntiles;
    Plan pln = plan_split(tiles, p.ktiles, d->split_k, 32, split_pen(d));
    p.nsplit = pln.nsplit; p.tiles_per_split = pln.tiles_per_split;
    const long wsize = (long)p.wrows * d->K;
    if (p.nsplit > 1 && (!ws || ws_bytes < (size_t)p.nsplit * (wsize + d->K) * sizeof(float)))
        return fail(DPIG_ENOMEM, "conv wgrad workspace too small: have %zu", ws_bytes);
    p.DB = db; p.beta_b = beta_b;
    p.bias_partial = p.partial ? p.partial + (long)p.nsplit * wsize : nullptr;
    p.vec_epi = (d->K % 4 == 0) && aligned16(dw) && (p.nsplit == 1 || aligned16(ws));
    p.vec_x = aligned16(x) && (d->ldx % 4 == 0) && (d->C % 4 == 0);
    p.d32_oy = (BK / p.Wo) % p.Ho;
    p.d32_ox = BK % p.Wo;
    p.d64_n = BKH / p.HoWo;
    p.d64_oy = (BKH % p.HoWo) / p.Wo;
    p.d64_ox = (BKH % p.HoWo) % p.Wo;
    p.d32g_n = BKS / p.HoWo;
    p.d32g_oy = (BKS % p.HoWo) / p.Wo;
    p.d32g_ox = (BKS % p.HoWo) % p.Wo;
    dim3 grid(tiles, 1, p.nsplit), block(256);
#define DPIG_WG(VE, NA, FL) hipLaunchKernelGGL((wgrad_kernel<VE, NA, FL>), grid, block, 0, st, p)
    if (flat) { if (narrow) DPIG_WG(false, true, true); else DPIG_WG(false, false, true); }
    else if (narrow) { if (vec) DPIG_WG(true, true, false); else DPIG_WG(false, true, false); }
    else if (bf16 && s1) hipLaunchKernelGGL((wgrad_kernel<true, false, false, true, 1>), grid, block, 0, st, p);
    else if (bf16) hipLaunchKernelGGL((wgrad_kernel<true, false, false, false, 1>), grid, block, 0, st, p);
    else if (split3 && s1) hipLaunchKernelGGL((wgrad_kernel<true, false, false, true, 2>), grid, block, 0, st, p);
    else if (split3) hipLaunchKernelGGL((wgrad_kernel<true, false, false, false, 2>), grid, block, 0, st, p);
    else if (s1) hipLaunchKernelGGL((wgrad_kernel<true, false, false, true>), grid, block, 0, st, p);
    else { if (vec) DPIG_WG(true, false, false); else DPIG_WG(false, false, false); }
#undef DPIG_WG
    rc = check_launch("wgrad_kernel");
    if (rc) return rc;
    if (p.nsplit > 1) {
        if ((wsize % 4 == 0) && aligned16(dw) && aligned16(ws)) {
            const long n4 = wsize / 4;
            int blocks = cdiv(n4, 256);
            if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
            hipLaunchKernelGGL(splitk_sum_kernel, dim3(blocks), dim3(256), 0, st, p.partial, dw, n4, p.nsplit, beta,
                               p.bias_partial, db, d->K, beta_b);
        } else {
            int blocks = cdiv(wsize, 256);
            if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
            hipLaunchKernelGGL(splitk_sum_scalar_kernel, dim3(blocks), dim3(256), 0, st, p.partial, dw, wsize, p.nsplit,
                               beta, p.bias_partial, db, d->K, beta_b);
        }
        rc = check_launch("splitk_sum_kernel");
    }
    return rc;
}
