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
#include "dpig_common.h"

namespace dpig {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDR = BK + 4;        // row-major [row][k] LDS stride (floats): 16B aligned, conflict-free b128
constexpr int LDKN = BN;           // k-major  [k][col] LDS stride
constexpr int TILE_FLOATS = BM * LDR;   // 4608 floats >= 32*128
constexpr int MAX_TAPS = 25;

struct GGParams {
    const float* A;       // gathered source activation (x for fwd, dy for dgrad)
    const float* B;       // HWIO filter
    float* D;             // destination activation
    float* D2;            // optional second output: the activation BEFORE a post-activation residual add
    const float* bias;    // [Ncols] or null
    const float* res;     // residual / accumulate tensor (dest-shaped) or null
    const float* mask;    // activation-output tensor for act' (dest-shaped) or null
    float* partial;       // split-K workspace [nsplit][M][Ncols]
    int M, Hr, Wr, HrWr;  // row grid (rows = images x Hr x Wr)
    int Hs, Ws, lda, Cs, sr;   // source spatial dims, channel stride, reduction channels, row->src stride
    int Ncols;            // GEMM N
    int Hd, Wd, ldd, dr, dpy, dpx;   // destination pixel = (r*dr+dpy, c*dr+dpx)
    int ldres, ldmask, ldd2;
    int res_post;         // 1: D = act(v + bias) + res  (reference res-blocks, models.py:400,427,536,566)
    int ntaps, cchunks, ktiles, tiles_per_split, nsplit;
    int mtiles, ntiles;
    int act; float alpha;
    int replicate;        // 1: write each result to the 2x2 block (nearest upsample)
    int identity_rows;    // 1: destination pixel index == row index
    // taps are an affine family (no table -> no dynamically indexed kernarg array):
    // tap t -> (a, b) = (t / tap_nb, t % tap_nb); source offset (oy0 + a*oys, ox0 + b*oxs);
    // filter slab w0 + a*wa + b*wb
    int tap_nb, oy0, oys, ox0, oxs, w0, wa, wb;
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
                                                    int res_post) {
    if (bias) v += bias[col];
    if (!replicate) {
        if (res && !res_post) v += res[pix * ldres + col];
        if (mask) v *= act_grad(mask[pix * ldmask + col], act, alpha);
        else v = act_apply(v, act, alpha);
        if (D2) D2[pix * ldd2 + col] = v;
        if (res && res_post) v += res[pix * ldres + col];
        D[pix * ldd + col] = v;
    } else {
        v = act_apply(v, act, alpha);
        D[pix * ldd + col] = v;
        D[(pix + 1) * ldd + col] = v;
        D[(pix + Wd) * ldd + col] = v;
        D[(pix + Wd + 1) * ldd + col] = v;
    }
}

template <bool B_ROWK, bool VEC>
__global__ __launch_bounds__(256, 2) void gather_gemm_kernel(const GGParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2][2 * TILE_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int split = blockIdx.z;
    const int kt_begin = split * p.tiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.tiles_per_split);

    // ---- per-thread A rows (fixed for the whole k loop) --------------------------------------
    const int a_kq = tid & 7;
    int a_base[4], a_iy0[4], a_ix0[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int n = mm / p.HrWr;
        const int rem = mm - n * p.HrWr;
        const int r = rem / p.Wr;
        const int c = rem - r * p.Wr;
        a_base[i] = n * p.Hs;
        a_iy0[i] = r * p.sr;
        a_ix0[i] = c * p.sr;
    }

    float4 ra[4], rb[4];

    auto load_tiles = [&](int kt) {
        const int tap = kt / p.cchunks;
        const int c0 = (kt - tap * p.cchunks) * BK;
        const int ta = tap / p.tap_nb, tb = tap - ta * p.tap_nb;
        const int oy = p.oy0 + ta * p.oys, ox = p.ox0 + tb * p.oxs, wt = p.w0 + ta * p.wa + tb * p.wb;
        // A: 128 rows x 32 k, thread -> (row = tid/8 + 32 i, 4 consecutive k)
        const int ck = c0 + a_kq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = a_iy0[i] + oy, ix = a_ix0[i] + ox;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
            const long off = ((long)(a_base[i] + iy) * p.Ws + ix) * p.lda + ck;
            if (VEC) {
                ra[i] = (ok && ck < p.Cs) ? *reinterpret_cast<const float4*>(p.A + off)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                ra[i].x = (ok && ck + 0 < p.Cs) ? p.A[off + 0] : 0.f;
                ra[i].y = (ok && ck + 1 < p.Cs) ? p.A[off + 1] : 0.f;
                ra[i].z = (ok && ck + 2 < p.Cs) ? p.A[off + 2] : 0.f;
                ra[i].w = (ok && ck + 3 < p.Cs) ? p.A[off + 3] : 0.f;
            }
        }
        if (B_ROWK) {
            // B stored [(wtap*Ncols + n)*Cs + k]: rows n, k contiguous (dgrad)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + (tid >> 3) + 32 * i;
                const bool ok = n < p.Ncols;
                const long off = ((long)wt * p.Ncols + n) * p.Cs + ck;
                if (VEC) {
                    rb[i] = (ok && ck < p.Cs) ? *reinterpret_cast<const float4*>(p.B + off)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    rb[i].x = (ok && ck + 0 < p.Cs) ? p.B[off + 0] : 0.f;
                    rb[i].y = (ok && ck + 1 < p.Cs) ? p.B[off + 1] : 0.f;
                    rb[i].z = (ok && ck + 2 < p.Cs) ? p.B[off + 2] : 0.f;
                    rb[i].w = (ok && ck + 3 < p.Cs) ? p.B[off + 3] : 0.f;
                }
            }
        } else {
            // B stored [(wtap*Cs + k)*Ncols + n]: rows k, n contiguous (fwd)
            const int nq = n0 + (tid & 31) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = c0 + (tid >> 5) + 8 * i;
                const bool ok = k < p.Cs;
                const long off = ((long)wt * p.Cs + k) * p.Ncols + nq;
                if (VEC) {
                    rb[i] = (ok && nq < p.Ncols) ? *reinterpret_cast<const float4*>(p.B + off)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    rb[i].x = (ok && nq + 0 < p.Ncols) ? p.B[off + 0] : 0.f;
                    rb[i].y = (ok && nq + 1 < p.Ncols) ? p.B[off + 1] : 0.f;
                    rb[i].z = (ok && nq + 2 < p.Ncols) ? p.B[off + 2] : 0.f;
                    rb[i].w = (ok && nq + 3 < p.Ncols) ? p.B[off + 3] : 0.f;
                }
            }
        }
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

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt_begin < kt_end) {
        load_tiles(kt_begin);
        store_tiles(0);
        __syncthreads();
        int buf = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more = (kt + 1) < kt_end;
            if (more) load_tiles(kt + 1);
            const float* As = smem[buf];
            const float* Bs = smem[buf] + TILE_FLOATS;
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                const float4 a0 =
                    *reinterpret_cast<const float4*>(&As[(wm * 64 + l31) * LDR + kk * 8 + half * 4]);
                const float4 a1 =
                    *reinterpret_cast<const float4*>(&As[(wm * 64 + 32 + l31) * LDR + kk * 8 + half * 4]);
                float4 b0, b1;
                if (B_ROWK) {
                    b0 = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + l31) * LDR + kk * 8 + half * 4]);
                    b1 = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + 32 + l31) * LDR + kk * 8 + half * 4]);
                } else {
                    const float* bp = &Bs[(kk * 8 + half * 4) * LDKN + wn * 64 + l31];
                    b0.x = bp[0 * LDKN]; b0.y = bp[1 * LDKN]; b0.z = bp[2 * LDKN]; b0.w = bp[3 * LDKN];
                    b1.x = bp[0 * LDKN + 32]; b1.y = bp[1 * LDKN + 32];
                    b1.z = bp[2 * LDKN + 32]; b1.w = bp[3 * LDKN + 32];
                }
                const float av0[4] = {a0.x, a0.y, a0.z, a0.w};
                const float av1[4] = {a1.x, a1.y, a1.z, a1.w};
                const float bv0[4] = {b0.x, b0.y, b0.z, b0.w};
                const float bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv0[j], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv1[j], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv0[j], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv1[j], acc[1][1], 0, 0, 0);
                }
            }
            if (more) store_tiles(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            long pix = row;
            if (p.nsplit == 1 && !p.identity_rows && row < p.M)
                pix = row_to_pix(row, p.HrWr, p.Wr, p.Hd, p.Wd, p.dr, p.dpy, p.dpx);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int col = n0 + wn * 64 + nb * 32 + l31;
                const float v = acc[mb][nb][r];
                if (row < p.M && col < p.Ncols) {
                    if (p.nsplit > 1) p.partial[((long)split * p.M + row) * p.Ncols + col] = v;
                    else epi_store(p.D, p.bias, p.res, p.mask, pix, col, v, p.ldd, p.ldres, p.ldmask, p.act,
                                   p.alpha, p.replicate, p.Wd, p.D2, p.ldd2, p.res_post);
                }
            }
        }
    }
}

// split-K second pass: sum partials in split order (deterministic) and run the fused epilogue
__global__ __launch_bounds__(256) void gather_gemm_reduce_kernel(const GGParams p) {
    const long total = (long)p.M * p.Ncols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < p.nsplit; ++s) v += p.partial[(long)s * total + i];
        const int row = (int)(i / p.Ncols);
        const int col = (int)(i - (long)row * p.Ncols);
        const long pix = p.identity_rows ? (long)row
                                         : row_to_pix(row, p.HrWr, p.Wr, p.Hd, p.Wd, p.dr, p.dpy, p.dpx);
        epi_store(p.D, p.bias, p.res, p.mask, pix, col, v, p.ldd, p.ldres, p.ldmask, p.act, p.alpha, p.replicate,
                  p.Wd, p.D2, p.ldd2, p.res_post);
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
};

template <bool VEC>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WGParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2][2 * BK * LDKN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    const int mtiles = p.ntaps * p.cblocks;
    const int tile = xcd_remap(blockIdx.x, mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int tap = mt / p.cblocks;
    const int ci0 = (mt - tap * p.cblocks) * BM;
    const int co0 = nt * BN;
    const int oyoff = tap / p.S - p.pad_t, oxoff = tap % p.S - p.pad_l, wt = tap;
    const int split = blockIdx.z;
    const int kt_begin = split * p.tiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.tiles_per_split);

    const int q = (tid & 31) * 4;   // column quad inside the 128-wide tile (both operands)
    float4 ra[4], rb[4];

    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = kt * BK + (tid >> 5) + 8 * i;
            const bool mok = m < p.Npix;
            const int mm = mok ? m : 0;
            const int n = mm / p.HoWo;
            const int rem = mm - n * p.HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            const int iy = (oy * p.s + oyoff) >> p.shift, ix = (ox * p.s + oxoff) >> p.shift;
            const bool ok = mok && (oy * p.s + oyoff) >= 0 && (ox * p.s + oxoff) >= 0 && iy < p.H && ix < p.W;
            const long xoff = ((long)(n * p.H + iy) * p.W + ix) * p.ldx + ci0 + q;
            const long yoff = (long)mm * p.ldy + co0 + q;
            if (VEC) {
                ra[i] = (ok && ci0 + q < p.C) ? *reinterpret_cast<const float4*>(p.X + xoff)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
                rb[i] = (mok && co0 + q < p.K) ? *reinterpret_cast<const float4*>(p.DY + yoff)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                ra[i].x = (ok && ci0 + q + 0 < p.C) ? p.X[xoff + 0] : 0.f;
                ra[i].y = (ok && ci0 + q + 1 < p.C) ? p.X[xoff + 1] : 0.f;
                ra[i].z = (ok && ci0 + q + 2 < p.C) ? p.X[xoff + 2] : 0.f;
                ra[i].w = (ok && ci0 + q + 3 < p.C) ? p.X[xoff + 3] : 0.f;
                rb[i].x = (mok && co0 + q + 0 < p.K) ? p.DY[yoff + 0] : 0.f;
                rb[i].y = (mok && co0 + q + 1 < p.K) ? p.DY[yoff + 1] : 0.f;
                rb[i].z = (mok && co0 + q + 2 < p.K) ? p.DY[yoff + 2] : 0.f;
                rb[i].w = (mok && co0 + q + 3 < p.K) ? p.DY[yoff + 3] : 0.f;
            }
        }
    };
    auto store_tiles = [&](int buf) {
        float* As = smem[buf];
        float* Bs = smem[buf] + BK * LDKN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&As[((tid >> 5) + 8 * i) * LDKN + q]) = ra[i];
            *reinterpret_cast<float4*>(&Bs[((tid >> 5) + 8 * i) * LDKN + q]) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt_begin < kt_end) {
        load_tiles(kt_begin);
        store_tiles(0);
        __syncthreads();
        int buf = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more = (kt + 1) < kt_end;
            if (more) load_tiles(kt + 1);
            const float* As = smem[buf];
            const float* Bs = smem[buf] + BK * LDKN;
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                const float* ap = &As[(kk * 8 + half * 4) * LDKN + wm * 64 + l31];
                const float* bp = &Bs[(kk * 8 + half * 4) * LDKN + wn * 64 + l31];
                float av0[4], av1[4], bv0[4], bv1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    av0[j] = ap[j * LDKN];
                    av1[j] = ap[j * LDKN + 32];
                    bv0[j] = bp[j * LDKN];
                    bv1[j] = bp[j * LDKN + 32];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv0[j], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv1[j], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv0[j], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv1[j], acc[1][1], 0, 0, 0);
                }
            }
            if (more) store_tiles(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- epilogue: dw rows are (wtap*C + ci), cols co -------------------------------------------
    const long wsize = (long)p.wrows * p.K;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int co = co0 + wn * 64 + nb * 32 + l31;
                const float v = acc[mb][nb][r];
                if (ci < p.C && co < p.K) {
                    const long idx = ((long)wt * p.C + ci) * p.K + co;
                    if (p.nsplit > 1) p.partial[(long)split * wsize + idx] = v;
                    else p.DW[idx] = (p.beta != 0.f) ? p.beta * p.DW[idx] + v : v;
                }
            }
        }
    }
}

// out[i] = beta*out[i] + sum_s partial[s][i]   (n multiple of 4, 16B aligned)
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                          long n4, int nsplit, float beta) {
    const float4* p4 = reinterpret_cast<const float4*>(partial);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
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
}
__global__ __launch_bounds__(256) void splitk_sum_scalar_kernel(const float* __restrict__ partial,
                                                                 float* __restrict__ out, long n, int nsplit,
                                                                 float beta) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < nsplit; ++s) v += partial[(long)s * n + i];
        out[i] = (beta != 0.f) ? beta * out[i] + v : v;
    }
}

// ================================================================================================
// host side
// ================================================================================================
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

static int resolve_desc(const DpigConvDesc* d, int* pt, int* pl, int* Ho, int* Wo) {
    if (!d) return fail(DPIG_EINVAL, "null descriptor");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0) return fail(DPIG_EINVAL, "non-positive dims");
    if (d->R <= 0 || d->S <= 0 || d->R * d->S > MAX_TAPS) return fail(DPIG_EINVAL, "filter %dx%d unsupported", d->R, d->S);
    if (d->stride != 1 && d->stride != 2) return fail(DPIG_EINVAL, "stride %d unsupported", d->stride);
    if (d->ldx < d->C || d->ldy < d->K) return fail(DPIG_EINVAL, "channel stride smaller than channel count");
    if (d->act < 0 || d->act > DPIG_ACT_LRELU) return fail(DPIG_EINVAL, "bad activation %d", d->act);
    if (d->upsample2x && (d->R != 1 || d->S != 1 || d->stride != 1))
        return fail(DPIG_EINVAL, "upsample2x fusion needs a 1x1 stride-1 conv");
    int ho, wo, a, b;
    dpig_same_pad(d->H, d->R, d->stride, &ho, &a);
    dpig_same_pad(d->W, d->S, d->stride, &wo, &b);
    if (d->pad_t >= 0) { a = d->pad_t; }
    if (d->pad_l >= 0) { b = d->pad_l; }
    *pt = a; *pl = b; *Ho = ho; *Wo = wo;
    if ((long)d->N * d->H * d->W * d->ldx >= (1L << 31) || (long)d->N * ho * wo * d->ldy * (d->upsample2x ? 4 : 1) >= (1L << 31))
        return fail(DPIG_EINVAL, "tensor exceeds 2^31 elements");
    return DPIG_OK;
}

static int choose_split(int tiles, int ktiles, int forced) {
    if (forced > 0) return forced < ktiles ? forced : (ktiles > 0 ? ktiles : 1);
    if (ktiles <= 0) return 1;
    const int target = 2 * kNumCU;            // two resident workgroups per CU
    if (tiles >= target / 2) return 1;        // >= 1 block per CU already: splitting only adds traffic
    int s = cdiv(target, tiles);
    const int max_by_k = ktiles / 4 > 0 ? ktiles / 4 : 1;   // keep >= 4 k-tiles per split
    if (s > max_by_k) s = max_by_k;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

struct Plan { int nsplit, tiles_per_split; };
static Plan plan_split(int tiles, int ktiles, int forced) {
    Plan pl;
    pl.nsplit = choose_split(tiles, ktiles, forced);
    pl.tiles_per_split = cdiv(ktiles > 0 ? ktiles : 1, pl.nsplit);
    pl.nsplit = cdiv(ktiles > 0 ? ktiles : 1, pl.tiles_per_split);   // drop empty splits
    return pl;
}

static bool vec_ok(const void* a, const void* b, int lda, int Cs, int Ncols) {
    return aligned16(a) && aligned16(b) && (lda % 4 == 0) && (Cs % 4 == 0) && (Ncols % 4 == 0);
}

static int launch_gg(GGParams& p, bool b_rowk, hipStream_t st) {
    p.HrWr = p.Hr * p.Wr;
    p.mtiles = cdiv(p.M, BM);
    p.ntiles = cdiv(p.Ncols, BN);
    p.cchunks = cdiv(p.Cs, BK);
    p.ktiles = p.ntaps * p.cchunks;
    bool vec = vec_ok(p.A, p.B, p.lda, p.Cs, p.Ncols);
    dim3 grid(p.mtiles * p.ntiles, 1, p.nsplit), block(256);
    if (b_rowk) {
        if (vec) hipLaunchKernelGGL((gather_gemm_kernel<true, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gather_gemm_kernel<true, false>), grid, block, 0, st, p);
    } else {
        if (vec) hipLaunchKernelGGL((gather_gemm_kernel<false, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gather_gemm_kernel<false, false>), grid, block, 0, st, p);
    }
    int rc = check_launch("gather_gemm_kernel");
    if (rc) return rc;
    if (p.nsplit > 1) {
        const long total = (long)p.M * p.Ncols;
        int blocks = cdiv(total, 256);
        if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
        hipLaunchKernelGGL(gather_gemm_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
        rc = check_launch("gather_gemm_reduce_kernel");
    }
    return rc;
}

// number of (row tiles x col tiles) and k-tiles for each op, used by both the workspace query and the launch
struct Shape { long M; int Ncols, ktiles; };
static Shape fwd_shape(const DpigConvDesc* d, int Ho, int Wo) {
    Shape s;
    s.M = d->upsample2x ? (long)d->N * d->H * d->W : (long)d->N * Ho * Wo;
    s.Ncols = d->K;
    s.ktiles = d->R * d->S * cdiv(d->C, BK);
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

// dgrad stride-2 parity class geometry
namespace dpig {
struct DClass { int py, px, Hr, Wr, ntaps, nky, nkx, ky0, kx0, oy0, ox0; };
static int build_dgrad_classes(const DpigConvDesc* d, int pt, int pl, DClass* cls) {
    int nc = 0;
    const int s = d->stride;
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            DClass& c = cls[nc];
            c.py = py; c.px = px;
            c.Hr = (d->H - py + s - 1) / s;
            c.Wr = (d->W - px + s - 1) / s;
            c.ntaps = 0;
            if (c.Hr <= 0 || c.Wr <= 0) continue;
            // valid filter rows: ky = ky0 + s*a with (py + pt - ky) divisible by s
            c.ky0 = ((py + pt) % s + s) % s;
            c.kx0 = ((px + pl) % s + s) % s;
            c.nky = c.ky0 < d->R ? (d->R - c.ky0 + s - 1) / s : 0;
            c.nkx = c.kx0 < d->S ? (d->S - c.kx0 + s - 1) / s : 0;
            c.oy0 = (py + pt - c.ky0) / s;     // exact; a-th valid row has offset oy0 - a
            c.ox0 = (px + pl - c.kx0) / s;
            c.ntaps = c.nky * c.nkx;
            ++nc;
        }
    return nc;
}
}  // namespace dpig

extern "C" size_t dpig_conv2d_workspace_bytes(const DpigConvDesc* d, int which) {
    int pt, pl, Ho, Wo;
    if (resolve_desc(d, &pt, &pl, &Ho, &Wo)) return 0;
    if (which == 0) {
        Shape s = fwd_shape(d, Ho, Wo);
        Plan pln = plan_split(cdiv(s.M, BM) * cdiv(s.Ncols, BN), s.ktiles, d->split_k);
        return pln.nsplit > 1 ? (size_t)pln.nsplit * s.M * s.Ncols * sizeof(float) : 0;
    } else if (which == 1) {
        if (d->upsample2x || d->stride == 1) {
            const long M = (long)d->N * d->H * d->W;
            const int ntaps = d->upsample2x ? 4 : d->R * d->S;
            Plan pln = plan_split(cdiv(M, BM) * cdiv(d->C, BN), ntaps * cdiv(d->K, BK), d->split_k);
            return pln.nsplit > 1 ? (size_t)pln.nsplit * M * d->C * sizeof(float) : 0;
        }
        DClass cls[4];
        const int nc = build_dgrad_classes(d, pt, pl, cls);
        size_t mx = 0;
        for (int i = 0; i < nc; ++i) {
            const long M = (long)d->N * cls[i].Hr * cls[i].Wr;
            Plan pln = plan_split(cdiv(M, BM) * cdiv(d->C, BN), cls[i].ntaps * cdiv(d->K, BK), d->split_k);
            const size_t b = pln.nsplit > 1 ? (size_t)pln.nsplit * M * d->C * sizeof(float) : 0;
            if (b > mx) mx = b;
        }
        return mx;
    } else if (which == 2) {
        const long Npix = (long)d->N * Ho * Wo * (d->upsample2x ? 4 : 1);
        const int tiles = d->R * d->S * cdiv(d->C, BM) * cdiv(d->K, BN);
        Plan pln = plan_split(tiles, cdiv(Npix, BK), d->split_k);
        return pln.nsplit > 1 ? (size_t)pln.nsplit * d->R * d->S * d->C * d->K * sizeof(float) : 0;
    }
    return 0;
}

extern "C" int dpig_conv2d_fwd(const DpigConvDesc* d, const float* x, const float* w, const float* bias,
                               const float* residual, float* y, float* y_act, void* ws, size_t ws_bytes,
                               void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !w || !y) return fail(DPIG_EINVAL, "null tensor pointer");
    if (d->upsample2x && residual) return fail(DPIG_EINVAL, "residual unsupported with upsample2x");
    GGParams p = {};
    p.A = x; p.B = w; p.D = y; p.bias = bias; p.res = residual; p.mask = nullptr;
    p.D2 = y_act; p.ldd2 = d->ldy2; p.res_post = d->res_after_act;
    if (y_act && d->ldy2 < d->K) return fail(DPIG_EINVAL, "ldy2 < K");
    if (y_act && d->upsample2x) return fail(DPIG_EINVAL, "y_act unsupported with upsample2x");
    p.partial = static_cast<float*>(ws);
    Shape s = fwd_shape(d, Ho, Wo);
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
    Plan pln = plan_split(cdiv(s.M, BM) * cdiv(s.Ncols, BN), s.ktiles, d->split_k);
    p.nsplit = pln.nsplit; p.tiles_per_split = pln.tiles_per_split;
    if (p.nsplit > 1 && ws_bytes < (size_t)p.nsplit * s.M * s.Ncols * sizeof(float))
        return fail(DPIG_ENOMEM, "conv fwd workspace too small: have %zu", ws_bytes);
    if (p.nsplit > 1 && !ws) return fail(DPIG_ENOMEM, "conv fwd needs a workspace");
    return launch_gg(p, false, static_cast<hipStream_t>(stream));
}

extern "C" int dpig_conv2d_dgrad(const DpigConvDesc* d, const float* dy, const float* w, const float* accum,
                                 const float* mask, float* dx, void* ws, size_t ws_bytes, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!dy || !w || !dx) return fail(DPIG_EINVAL, "null tensor pointer");
    if (accum && d->ldres < d->C) return fail(DPIG_EINVAL, "ldres < C");
    if (mask && d->ldmask < d->C) return fail(DPIG_EINVAL, "ldmask < C");
    hipStream_t st = static_cast<hipStream_t>(stream);
    GGParams p = {};
    p.A = dy; p.B = w; p.D = dx; p.bias = nullptr; p.res = accum; p.mask = mask;
    p.partial = static_cast<float*>(ws);
    p.lda = d->ldy; p.Cs = d->K; p.Ncols = d->C;
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
        DClass cls[4];
        const int nc = build_dgrad_classes(d, pt, pl, cls);
        for (int i = 0; i < nc; ++i) {
            GGParams q = p;
            q.M = d->N * cls[i].Hr * cls[i].Wr; q.Hr = cls[i].Hr; q.Wr = cls[i].Wr;
            q.Hs = Ho; q.Ws = Wo; q.sr = 1;
            q.dr = 2; q.dpy = cls[i].py; q.dpx = cls[i].px; q.identity_rows = 0;
            q.ntaps = cls[i].ntaps;
            q.tap_nb = cls[i].nkx > 0 ? cls[i].nkx : 1;
            q.oy0 = cls[i].oy0; q.oys = -1; q.ox0 = cls[i].ox0; q.oxs = -1;
            q.w0 = cls[i].ky0 * d->S + cls[i].kx0; q.wa = d->stride * d->S; q.wb = d->stride;
            Plan pln = plan_split(cdiv(q.M, BM) * cdiv(q.Ncols, BN), q.ntaps * cdiv(q.Cs, BK), d->split_k);
            q.nsplit = pln.nsplit; q.tiles_per_split = pln.tiles_per_split;
            if (q.nsplit > 1 && (!ws || ws_bytes < (size_t)q.nsplit * q.M * q.Ncols * sizeof(float)))
                return fail(DPIG_ENOMEM, "conv dgrad workspace too small: have %zu", ws_bytes);
            rc = launch_gg(q, true, st);
            if (rc) return rc;
        }
        return DPIG_OK;
    }
    Plan pln = plan_split(cdiv(p.M, BM) * cdiv(p.Ncols, BN), p.ntaps * cdiv(p.Cs, BK), d->split_k);
    p.nsplit = pln.nsplit; p.tiles_per_split = pln.tiles_per_split;
    if (p.nsplit > 1 && (!ws || ws_bytes < (size_t)p.nsplit * p.M * p.Ncols * sizeof(float)))
        return fail(DPIG_ENOMEM, "conv dgrad workspace too small: have %zu", ws_bytes);
    return launch_gg(p, true, st);
}

extern "C" int dpig_conv2d_wgrad(const DpigConvDesc* d, const float* x, const float* dy, float* dw, float beta,
                                 void* ws, size_t ws_bytes, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !dy || !dw) return fail(DPIG_EINVAL, "null tensor pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
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
    p.wrows = d->R * d->S * d->C;
    p.cblocks = cdiv(d->C, BM);
    p.ntiles = cdiv(d->K, BN);
    p.ktiles = cdiv(p.Npix, BK);
    const int tiles = p.ntaps * p.cblocks * p.ntiles;
    Plan pln = plan_split(tiles, p.ktiles, d->split_k);
    p.nsplit = pln.nsplit; p.tiles_per_split = pln.tiles_per_split;
    const long wsize = (long)p.wrows * d->K;
    if (p.nsplit > 1 && (!ws || ws_bytes < (size_t)p.nsplit * wsize * sizeof(float)))
        return fail(DPIG_ENOMEM, "conv wgrad workspace too small: have %zu", ws_bytes);
    const bool vec = aligned16(x) && aligned16(dy) && (d->ldx % 4 == 0) && (d->ldy % 4 == 0) &&
                     (d->C % 4 == 0) && (d->K % 4 == 0);
    dim3 grid(tiles, 1, p.nsplit), block(256);
    if (vec) hipLaunchKernelGGL((wgrad_kernel<true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((wgrad_kernel<false>), grid, block, 0, st, p);
    rc = check_launch("wgrad_kernel");
    if (rc) return rc;
    if (p.nsplit > 1) {
        if ((wsize % 4 == 0) && aligned16(dw) && aligned16(ws)) {
            const long n4 = wsize / 4;
            int blocks = cdiv(n4, 256);
            if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
            hipLaunchKernelGGL(splitk_sum_kernel, dim3(blocks), dim3(256), 0, st, p.partial, dw, n4, p.nsplit, beta);
        } else {
            int blocks = cdiv(wsize, 256);
            if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
            hipLaunchKernelGGL(splitk_sum_scalar_kernel, dim3(blocks), dim3(256), 0, st, p.partial, dw, wsize, p.nsplit, beta);
        }
        rc = check_launch("splitk_sum_kernel");
    }
    return rc;
}
