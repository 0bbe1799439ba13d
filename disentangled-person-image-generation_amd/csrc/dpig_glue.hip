// Small one-launch kernels for the graph wiring BETWEEN the convolutions of the stage-I step -- everything the builders
// of models.py / trainer.py express with elementwise TF ops and that would otherwise run as a string of torch-native
// launches (BASELINE north_star: torch is plumbing, not compute):
//   mask_split      x_fg = x * m, x_bg = x * (1 - m)                    models.py:402-403   (+ the transposed gradient)
//   roi_boxes       pixel boxes -> normalised, part-major boxes + box_ind models.py:405-413
//   vis_concat      per-part features * visibility, concat with the background feature: models.py:433-442, 467-468
//   emb_class_*     the tiled-embedding collapse of the generator's first conv (trainer.py:588-590, models.py:520-528,
//                   SURVEY F7): per-border-class sums of the filter taps and their transpose for the filter gradient
//   transpose12     [B, A, C] -> [B, C, A]: the critic's tf.reshape on the logical NCHW tensor (wgan_gp.py:433) when the
//                   activations are physically NHWC
//   axpby2d         dst = beta * dst + src on row-strided 2-D views (per-channel gradients into flat gradient slices)
// All are HBM- or latency-bound; 16-byte accesses where the shape allows, fp32 arithmetic.
#include "dpig_common.h"

namespace dpig {

typedef unsigned short glue_bf16;
__device__ __forceinline__ float glue_b2f(glue_bf16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ glue_bf16 glue_f2b(float v) { const __bf16 b = (__bf16)v; return __builtin_bit_cast(glue_bf16, b); }

static inline int glue_blocks(long work, int cap_per_cu = 8) {
    long b = (work + 255) / 256;
    const long cap = (long)cap_per_cu * kNumCU;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// ---- mask split -----------------------------------------------------------------------------------------------------
template <typename T, int VEC>     // VEC elements per thread (16 bytes when aligned)
__global__ __launch_bounds__(256) void mask_split_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ m,
                                                             long rows, int C, T* __restrict__ fg, int ldf,
                                                             T* __restrict__ bg, int ldb) {
    const int cv = C / VEC;
    const long total = rows * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cv;
        const int c = (int)(i - r * cv) * VEC;
        const float mv = m[r], nm = 1.0f - mv;
        T vx[VEC], vf[VEC], vb[VEC];
        if (VEC * sizeof(T) == 16) *reinterpret_cast<uint4*>(vx) = *reinterpret_cast<const uint4*>(x + r * ldx + c);
        else vx[0] = x[r * ldx + c];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (sizeof(T) == 4) {
                const float v = ((const float*)vx)[e];
                ((float*)vf)[e] = v * mv;
                ((float*)vb)[e] = v * nm;
            } else {
                const float v = glue_b2f(((const glue_bf16*)vx)[e]);
                ((glue_bf16*)vf)[e] = glue_f2b(v * mv);
                ((glue_bf16*)vb)[e] = glue_f2b(v * nm);
            }
        }
        if (VEC * sizeof(T) == 16) {
            *reinterpret_cast<uint4*>(fg + r * ldf + c) = *reinterpret_cast<const uint4*>(vf);
            *reinterpret_cast<uint4*>(bg + r * ldb + c) = *reinterpret_cast<const uint4*>(vb);
        } else {
            fg[r * ldf + c] = vf[0];
            bg[r * ldb + c] = vb[0];
        }
    }
}
// dx = dfg * m + dbg * (1 - m)   (either gradient may be absent)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void mask_split_bwd_kernel(const T* __restrict__ dfg, int ldf, const T* __restrict__ dbg,
                                                             int ldb, const float* __restrict__ m, long rows, int C,
                                                             T* __restrict__ dx, int ldx) {
    const int cv = C / VEC;
    const long total = rows * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cv;
        const int c = (int)(i - r * cv) * VEC;
        const float mv = m[r], nm = 1.0f - mv;
        T vf[VEC], vb[VEC], vo[VEC];
        if (VEC * sizeof(T) == 16) {
            *reinterpret_cast<uint4*>(vf) = dfg ? *reinterpret_cast<const uint4*>(dfg + r * ldf + c) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(vb) = dbg ? *reinterpret_cast<const uint4*>(dbg + r * ldb + c) : make_uint4(0, 0, 0, 0);
        } else {
            vf[0] = dfg ? dfg[r * ldf + c] : (T)0;
            vb[0] = dbg ? dbg[r * ldb + c] : (T)0;
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (sizeof(T) == 4) ((float*)vo)[e] = ((const float*)vf)[e] * mv + ((const float*)vb)[e] * nm;
            else ((glue_bf16*)vo)[e] = glue_f2b(glue_b2f(((const glue_bf16*)vf)[e]) * mv + glue_b2f(((const glue_bf16*)vb)[e]) * nm);
        }
        if (VEC * sizeof(T) == 16) *reinterpret_cast<uint4*>(dx + r * ldx + c) = *reinterpret_cast<const uint4*>(vo);
        else dx[r * ldx + c] = vo[0];
    }
}

// ---- ROI boxes: [B][P_total][4] pixel (y1, x1, y2, x2) -> part-major normalised boxes [P*B][4], box_ind [P*B] ---------
template <typename TI>
__global__ __launch_bounds__(256) void roi_boxes_kernel(const TI* __restrict__ bbox, int B, int P_total, int P, float img_H,
                                                        float img_W, float* __restrict__ boxes, int* __restrict__ box_ind) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P * B) return;
    const int p = i / B, b = i - p * B;
    const TI* s = bbox + ((long)b * P_total + p) * 4;
    boxes[i * 4 + 0] = (float)s[0] / img_H;      // models.py:410-413: division by H and W (not H - 1)
    boxes[i * 4 + 1] = (float)s[1] / img_W;
    boxes[i * 4 + 2] = (float)s[2] / img_H;
    boxes[i * 4 + 3] = (float)s[3] / img_W;
    box_ind[i] = b;
}

// ---- visibility multiply + concat ---------------------------------------------------------------------------------------
// all[b][p*z + c] = fea[p*B + b][c] * vis[b][p]  (p < P);  all[b][P*z + c] = bg[b][c]  (c < zbg)
__global__ __launch_bounds__(256) void vis_concat_fwd_kernel(const float* __restrict__ fea, const float* __restrict__ vis,
                                                             int ldvis, const float* __restrict__ bg, int B, int P, int z,
                                                             int zbg, float* __restrict__ all) {
    const int W = P * z + zbg;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * W) return;
    const int b = i / W, c = i - b * W;
    float v;
    if (c < P * z) {
        const int p = c / z, cc = c - p * z;
        v = fea[((long)p * B + b) * z + cc] * vis[b * ldvis + p];
    } else {
        v = bg[(long)b * zbg + (c - P * z)];
    }
    all[i] = v;
}
__global__ __launch_bounds__(256) void vis_concat_bwd_kernel(const float* __restrict__ dall, const float* __restrict__ vis,
                                                             int ldvis, int B, int P, int z, int zbg, float* __restrict__ dfea,
                                                             float* __restrict__ dbg) {
    const int W = P * z + zbg;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * W) return;
    const int b = i / W, c = i - b * W;
    const float g = dall[i];
    if (c < P * z) {
        const int p = c / z, cc = c - p * z;
        dfea[((long)p * B + b) * z + cc] = g * vis[b * ldvis + p];
    } else if (dbg) {
        dbg[(long)b * zbg + (c - P * z)] = g;
    }
}

// ---- tiled-embedding collapse: per-border-class tap sums ----------------------------------------------------------------
// w [3][3][E + P][K] (HWIO).  Border class (cy, cx) of a SAME 3x3 conv sees the taps ky in V(cy), kx in V(cx) with
// V(0) = {1, 2}, V(1) = {0, 1, 2}, V(2) = {0, 1}.  wmat[e][(cy*3 + cx)*K + k] = sum of those taps of w[.][.][e][k].
__global__ __launch_bounds__(256) void emb_class_fwd_kernel(const float* __restrict__ w, int E, int C, int K,
                                                            float* __restrict__ wmat) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)E * K) return;
    const int e = (int)(i / K), k = (int)(i - (long)e * K);
    float t[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) t[ky][kx] = w[(((long)ky * 3 + kx) * C + e) * K + k];
    float ry[3][3];                                    // [cy][kx]: sums over the valid ky
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        ry[0][kx] = t[1][kx] + t[2][kx];
        ry[1][kx] = (t[0][kx] + t[1][kx]) + t[2][kx];
        ry[2][kx] = t[0][kx] + t[1][kx];
    }
#pragma unroll
    for (int cy = 0; cy < 3; ++cy) {
        float* o = wmat + (long)e * 9 * K + (long)cy * 3 * K + k;
        o[0] = ry[cy][1] + ry[cy][2];
        o[K] = (ry[cy][0] + ry[cy][1]) + ry[cy][2];
        o[2 * K] = ry[cy][0] + ry[cy][1];
    }
}
// the transpose: dw[ky][kx][e][k] = beta * dw + sum over the classes (cy, cx) whose valid sets contain (ky, kx) of dwc[e][cy][cx][k]
__global__ __launch_bounds__(256) void emb_class_bwd_kernel(const float* __restrict__ dwc, int E, int C, int K,
                                                            float* __restrict__ dw, float beta) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)E * K) return;
    const int e = (int)(i / K), k = (int)(i - (long)e * K);
    float d[3][3];
#pragma unroll
    for (int cy = 0; cy < 3; ++cy)
#pragma unroll
        for (int cx = 0; cx < 3; ++cx) d[cy][cx] = dwc[(long)e * 9 * K + ((long)cy * 3 + cx) * K + k];
    float ty[3][3];                                    // [ky][cx]: tap ky receives the classes cy with ky in V(cy)
#pragma unroll
    for (int cx = 0; cx < 3; ++cx) {
        ty[0][cx] = d[1][cx] + d[2][cx];
        ty[1][cx] = (d[0][cx] + d[1][cx]) + d[2][cx];
        ty[2][cx] = d[0][cx] + d[1][cx];
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const float v0 = ty[ky][1] + ty[ky][2], v1 = (ty[ky][0] + ty[ky][1]) + ty[ky][2], v2 = ty[ky][0] + ty[ky][1];
        float* o = dw + (((long)ky * 3) * C + e) * K + k;
        const long ts = (long)C * K;
        o[0] = (beta != 0.f ? beta * o[0] : 0.f) + v0;
        o[ts] = (beta != 0.f ? beta * o[ts] : 0.f) + v1;
        o[2 * ts] = (beta != 0.f ? beta * o[2 * ts] : 0.f) + v2;
    }
}

// ---- gradient of act(conv1x1(upsample2x(x))) w.r.t. the conv's LOW-resolution pre-image: the nearest-neighbour upsample commutes
// with a 1x1 conv (models.py:569-570 computed at low resolution), so the backward pass needs sum_{2x2} dy * act'(y) per low-res
// pixel -- one pass over dy / y instead of act_bwd + two 4-tap gathers over the high-resolution gradient.  dz [N,H,W,C] (dense),
// dy / y [N,2H,2W,C] with row strides; VEC channels per thread (16 bytes).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void act_bwd_pool2x_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                                             T* __restrict__ dz, int N, int H, int W, int C, int act, float alpha) {
    const int cv = C / VEC;
    const long total = (long)N * H * W * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * VEC;
        long t = i / cv;
        const int x = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const long n = t / H;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long pix = (n * 2 * H + 2 * yy + (q >> 1)) * (2L * W) + 2 * x + (q & 1);
            T g[VEC], a[VEC];
            if (VEC * sizeof(T) == 16) *reinterpret_cast<uint4*>(g) = *reinterpret_cast<const uint4*>(dy + pix * lddy + c);
            else g[0] = dy[pix * lddy + c];
            if (act != DPIG_ACT_NONE) {
                if (VEC * sizeof(T) == 16) *reinterpret_cast<uint4*>(a) = *reinterpret_cast<const uint4*>(y + pix * ldy + c);
                else a[0] = y[pix * ldy + c];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float gv, av = 1.f;
                if constexpr (sizeof(T) == 2) { gv = glue_b2f(g[e]); if (act != DPIG_ACT_NONE) av = glue_b2f(a[e]); }
                else { gv = g[e]; if (act != DPIG_ACT_NONE) av = a[e]; }
                acc[e] += (act == DPIG_ACT_NONE) ? gv : gv * act_grad(av, act, alpha);
            }
        }
        T o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if constexpr (sizeof(T) == 2) o[e] = glue_f2b(acc[e]); else o[e] = acc[e];
        }
        T* d = dz + (((n * H + yy) * W) + x) * C + c;
        if (VEC * sizeof(T) == 16) *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(o);
        else d[0] = o[0];
    }
}

// ---- dst[r][c] = beta * dst[r][c] + src[r][c] on [outer][rows][cols] views with independent strides ---------------------
__global__ __launch_bounds__(256) void axpby3d_kernel(const float* __restrict__ src, long s_outer, long s_row,
                                                      float* __restrict__ dst, long d_outer, long d_row, int outer, int rows,
                                                      int cols, float beta) {
    const long total = (long)outer * rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cols);
        const long rr = i / cols;
        const int r = (int)(rr % rows), o = (int)(rr / rows);
        float* d = dst + o * d_outer + r * d_row + c;
        const float v = src[o * s_outer + r * s_row + c];
        *d = (beta != 0.f) ? beta * *d + v : v;
    }
}

// ---- y[b][c][a] = x[b][a][c] through a 32 x 32 LDS tile (both sides row-contiguous) -------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose12_kernel(const T* __restrict__ x, T* __restrict__ y, int A, int C) {
    __shared__ T t[32][33];
    const int b = blockIdx.z;
    const int a0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const T* xb = x + (long)b * A * C;
    T* yb = y + (long)b * A * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int a = a0 + ty + 8 * i, c = c0 + tx;
        if (a < A && c < C) t[ty + 8 * i][tx] = xb[(long)a * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, a = a0 + tx;
        if (a < A && c < C) yb[(long)c * A + a] = t[tx][ty + 8 * i];
    }
}

}  // namespace dpig

using namespace dpig;

extern "C" int dpig_mask_split_fwd(const void* x, int ldx, const float* m, int64_t rows, int C, void* fg, int ldfg, void* bg,
                                   int ldbg, int is_bf16, void* stream) {
    if (!x || !m || !fg || !bg || rows < 0 || C <= 0 || ldx < C || ldfg < C || ldbg < C) return fail(DPIG_EINVAL, "mask_split: bad arguments");
    if (rows == 0) return DPIG_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int es = is_bf16 ? 2 : 4, vec = 16 / es;
    const bool v = (C % vec == 0) && (ldx % vec == 0) && (ldfg % vec == 0) && (ldbg % vec == 0) && aligned16(x) && aligned16(fg) && aligned16(bg);
    const int blocks = glue_blocks(rows * (v ? C / vec : C), 16);
    if (is_bf16) {
        if (v) hipLaunchKernelGGL((mask_split_fwd_kernel<glue_bf16, 8>), dim3(blocks), dim3(256), 0, st, (const glue_bf16*)x, ldx, m, (long)rows, C, (glue_bf16*)fg, ldfg, (glue_bf16*)bg, ldbg);
        else hipLaunchKernelGGL((mask_split_fwd_kernel<glue_bf16, 1>), dim3(blocks), dim3(256), 0, st, (const glue_bf16*)x, ldx, m, (long)rows, C, (glue_bf16*)fg, ldfg, (glue_bf16*)bg, ldbg);
    } else {
        if (v) hipLaunchKernelGGL((mask_split_fwd_kernel<float, 4>), dim3(blocks), dim3(256), 0, st, (const float*)x, ldx, m, (long)rows, C, (float*)fg, ldfg, (float*)bg, ldbg);
        else hipLaunchKernelGGL((mask_split_fwd_kernel<float, 1>), dim3(blocks), dim3(256), 0, st, (const float*)x, ldx, m, (long)rows, C, (float*)fg, ldfg, (float*)bg, ldbg);
    }
    return check_launch("mask_split_fwd");
}

extern "C" int dpig_mask_split_bwd(const void* dfg, int ldfg, const void* dbg, int ldbg, const float* m, int64_t rows, int C,
                                   void* dx, int ldx, int is_bf16, void* stream) {
    if ((!dfg && !dbg) || !m || !dx || rows < 0 || C <= 0 || ldx < C || (dfg && ldfg < C) || (dbg && ldbg < C))
        return fail(DPIG_EINVAL, "mask_split_bwd: bad arguments");
    if (rows == 0) return DPIG_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int es = is_bf16 ? 2 : 4, vec = 16 / es;
    const bool v = (C % vec == 0) && (ldx % vec == 0) && (!dfg || (ldfg % vec == 0 && aligned16(dfg))) &&
                   (!dbg || (ldbg % vec == 0 && aligned16(dbg))) && aligned16(dx);
    const int blocks = glue_blocks(rows * (v ? C / vec : C), 16);
    if (is_bf16) {
        if (v) hipLaunchKernelGGL((mask_split_bwd_kernel<glue_bf16, 8>), dim3(blocks), dim3(256), 0, st, (const glue_bf16*)dfg, ldfg, (const glue_bf16*)dbg, ldbg, m, (long)rows, C, (glue_bf16*)dx, ldx);
        else hipLaunchKernelGGL((mask_split_bwd_kernel<glue_bf16, 1>), dim3(blocks), dim3(256), 0, st, (const glue_bf16*)dfg, ldfg, (const glue_bf16*)dbg, ldbg, m, (long)rows, C, (glue_bf16*)dx, ldx);
    } else {
        if (v) hipLaunchKernelGGL((mask_split_bwd_kernel<float, 4>), dim3(blocks), dim3(256), 0, st, (const float*)dfg, ldfg, (const float*)dbg, ldbg, m, (long)rows, C, (float*)dx, ldx);
        else hipLaunchKernelGGL((mask_split_bwd_kernel<float, 1>), dim3(blocks), dim3(256), 0, st, (const float*)dfg, ldfg, (const float*)dbg, ldbg, m, (long)rows, C, (float*)dx, ldx);
    }
    return check_launch("mask_split_bwd");
}

extern "C" int dpig_roi_boxes(const void* bbox, int is_float, int B, int P_total, int P, float img_H, float img_W, float* boxes,
                              int32_t* box_ind, void* stream) {
    if (!bbox || !boxes || !box_ind || B <= 0 || P <= 0 || P > P_total || img_H <= 0.f || img_W <= 0.f)
        return fail(DPIG_EINVAL, "roi_boxes: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int blocks = (P * B + 255) / 256;
    if (is_float) hipLaunchKernelGGL((roi_boxes_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)bbox, B, P_total, P, img_H, img_W, boxes, box_ind);
    else hipLaunchKernelGGL((roi_boxes_kernel<int32_t>), dim3(blocks), dim3(256), 0, st, (const int32_t*)bbox, B, P_total, P, img_H, img_W, boxes, box_ind);
    return check_launch("roi_boxes");
}

extern "C" int dpig_vis_concat_fwd(const float* fea, const float* vis, int ldvis, const float* bg, int B, int P, int z, int zbg,
                                   float* all, void* stream) {
    if (!fea || !vis || !all || B <= 0 || P <= 0 || z <= 0 || zbg < 0 || (zbg > 0 && !bg) || ldvis < P)
        return fail(DPIG_EINVAL, "vis_concat: bad arguments");
    const int total = B * (P * z + zbg);
    hipLaunchKernelGGL(vis_concat_fwd_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), fea, vis,
                       ldvis, bg, B, P, z, zbg, all);
    return check_launch("vis_concat_fwd");
}
extern "C" int dpig_vis_concat_bwd(const float* dall, const float* vis, int ldvis, int B, int P, int z, int zbg, float* dfea,
                                   float* dbg, void* stream) {
    if (!dall || !vis || !dfea || B <= 0 || P <= 0 || z <= 0 || zbg < 0 || ldvis < P) return fail(DPIG_EINVAL, "vis_concat_bwd: bad arguments");
    const int total = B * (P * z + zbg);
    hipLaunchKernelGGL(vis_concat_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), dall, vis,
                       ldvis, B, P, z, zbg, dfea, dbg);
    return check_launch("vis_concat_bwd");
}

extern "C" int dpig_emb_class_weights_fwd(const float* w, int E, int C, int K, float* wmat, void* stream) {
    if (!w || !wmat || E <= 0 || C < E || K <= 0) return fail(DPIG_EINVAL, "emb_class_weights: bad arguments");
    const long n = (long)E * K;
    hipLaunchKernelGGL(emb_class_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, E, C, K, wmat);
    return check_launch("emb_class_weights_fwd");
}
extern "C" int dpig_emb_class_weights_bwd(const float* dwc, int E, int C, int K, float* dw, float beta, void* stream) {
    if (!dwc || !dw || E <= 0 || C < E || K <= 0) return fail(DPIG_EINVAL, "emb_class_weights_bwd: bad arguments");
    const long n = (long)E * K;
    hipLaunchKernelGGL(emb_class_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dwc, E, C, K, dw, beta);
    return check_launch("emb_class_weights_bwd");
}

extern "C" int dpig_act_bwd_pool2x(const void* dy, int lddy, const void* y, int ldy, void* dz, int N, int H, int W, int C, int act,
                                   float alpha, int is_bf16, void* stream) {
    if (!dy || !dz || (act != DPIG_ACT_NONE && !y) || N <= 0 || H <= 0 || W <= 0 || C <= 0 || lddy < C || (y && ldy < C))
        return fail(DPIG_EINVAL, "act_bwd_pool2x: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int vec = is_bf16 ? 8 : 4;
    const bool v = C % vec == 0 && lddy % vec == 0 && (!y || ldy % vec == 0) && aligned16(dy) && aligned16(dz) && (!y || aligned16(y));
    const long work = (long)N * H * W * (v ? C / vec : C);
    const dim3 grid(glue_blocks(work, 16));
    if (is_bf16) {
        if (v) hipLaunchKernelGGL((act_bwd_pool2x_kernel<glue_bf16, 8>), grid, dim3(256), 0, st, (const glue_bf16*)dy, lddy, (const glue_bf16*)y, ldy, (glue_bf16*)dz, N, H, W, C, act, alpha);
        else hipLaunchKernelGGL((act_bwd_pool2x_kernel<glue_bf16, 1>), grid, dim3(256), 0, st, (const glue_bf16*)dy, lddy, (const glue_bf16*)y, ldy, (glue_bf16*)dz, N, H, W, C, act, alpha);
    } else {
        if (v) hipLaunchKernelGGL((act_bwd_pool2x_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)dy, lddy, (const float*)y, ldy, (float*)dz, N, H, W, C, act, alpha);
        else hipLaunchKernelGGL((act_bwd_pool2x_kernel<float, 1>), grid, dim3(256), 0, st, (const float*)dy, lddy, (const float*)y, ldy, (float*)dz, N, H, W, C, act, alpha);
    }
    return check_launch("act_bwd_pool2x");
}

extern "C" int dpig_axpby3d(const float* src, int64_t s_outer, int64_t s_row, float* dst, int64_t d_outer, int64_t d_row, int outer,
                            int rows, int cols, float beta, void* stream) {
    if (!src || !dst || outer <= 0 || rows <= 0 || cols <= 0) return fail(DPIG_EINVAL, "axpby3d: bad arguments");
    const long total = (long)outer * rows * cols;
    hipLaunchKernelGGL(axpby3d_kernel, dim3(glue_blocks(total, 8)), dim3(256), 0, static_cast<hipStream_t>(stream), src, (long)s_outer,
                       (long)s_row, dst, (long)d_outer, (long)d_row, outer, rows, cols, beta);
    return check_launch("axpby3d");
}

extern "C" int dpig_transpose12(const void* x, void* y, int B, int A, int C, int elem_bytes, void* stream) {
    if (!x || !y || B <= 0 || A <= 0 || C <= 0 || (elem_bytes != 2 && elem_bytes != 4) || B > 65535) return fail(DPIG_EINVAL, "transpose12: bad arguments");
    dim3 grid((C + 31) / 32, (A + 31) / 32, B);
    if (grid.y > 65535) return fail(DPIG_EINVAL, "transpose12: too many rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (elem_bytes == 4) hipLaunchKernelGGL((transpose12_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (float*)y, A, C);
    else hipLaunchKernelGGL((transpose12_kernel<glue_bf16>), grid, dim3(256), 0, st, (const glue_bf16*)x, (glue_bf16*)y, A, C);
    return check_launch("transpose12");
}
