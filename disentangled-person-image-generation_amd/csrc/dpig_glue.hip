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

// ---- the generator's first conv fed with KEYPOINTS instead of the rasterised pose map (trainer.py:556-560 builds the map in the
// graph from pose_rcv: coord2channel_simple_rcv + tf_poseInflate, utils.py:237-318).  A pose channel is -1 everywhere except a
// radius-4 disc of 2 * min(v * hits, 1) - 1 around its keypoint, so its contribution to the SAME 3x3 conv is
//     - sum_{valid taps} w[tap][k]            (a border-class constant like the tiled embedding's)
//     + sum_{taps whose source pixel lies in the disc} 2 * min(v * hits, 1) * w[tap][k]      (non-zero within 5 pixels of the keypoint)
// y = act(e9[b][class] + bias - cpos[class] + the sparse sum): the [B,H,W,18] tensor is never formed, the dense 18-channel conv
// (5.4 GFLOP at Market, 123 us on the vector ALUs) becomes one write of y.  Same disc rule as pose_from_rcv_kernel<1>.
__device__ __forceinline__ int stem_disc_half(int a) {
    const int aa = a < 0 ? -a : a;
    return aa == 0 ? 4 : (aa <= 2 ? 3 : (aa == 3 ? 2 : (aa == 4 ? 0 : -1)));
}
__device__ __forceinline__ void stem_keypoint(const float* __restrict__ rcv, int H, int W, int normalized, int* r, int* c, float* v) {
    float R = rcv[0], C = rcv[1];
    if (normalized) {
        R = fminf(fmaxf((R + 1.f) / 2.0f * (float)H, 0.f), (float)(H - 1));
        C = fminf(fmaxf((C + 1.f) / 2.0f * (float)W, 0.f), (float)(W - 1));
    }
    *r = (int)R; *c = (int)C; *v = rcv[2];
}
// amplitude above the -1 background of channel k at pixel (y, x): 0 outside the disc
__device__ __forceinline__ float stem_amp(int r, int c, float v, int H, int W, int y, int x) {
    if ((unsigned)r >= (unsigned)H || (unsigned)c >= (unsigned)W) return 0.f;
    const int a = r - y, bb = c - x;
    const int hw = stem_disc_half(a);
    if (hw < 0 || bb < -hw || bb > hw) return 0.f;
    const float hits = (a == 0 && bb == 0) ? 2.f : 1.f;
    return 2.f * fminf(v * hits, 1.f);
}
constexpr int STEM_MAXK = 32;       // keypoints per image held in LDS
// block = one image row; thread = (pixel, 4 output channels).  w: the whole filter [9][C][K] fp32, pose rows at channel E + k
template <typename T>
__global__ __launch_bounds__(256) void pose_stem_fwd_kernel(const float* __restrict__ rcv, int P, int normalized,
                                                            const float* __restrict__ e9, const float* __restrict__ cpos,
                                                            const float* __restrict__ bias, const float* __restrict__ w, int C, int E,
                                                            int H, int W, int K, int act, float alpha, T* __restrict__ y) {
    __shared__ int s_r[STEM_MAXK], s_c[STEM_MAXK];
    __shared__ float s_v[STEM_MAXK];
    __shared__ int s_near[STEM_MAXK], s_nn;
    const int b = blockIdx.x / H, yy = blockIdx.x - b * H;
    if (threadIdx.x < P) {
        int r, c; float v;
        stem_keypoint(rcv + ((long)b * P + threadIdx.x) * 3, H, W, normalized, &r, &c, &v);
        s_r[threadIdx.x] = r; s_c[threadIdx.x] = c; s_v[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {              // keypoints whose disc (+1 for the 3x3 reach) touches this row
        int n = 0;
        for (int k = 0; k < P; ++k) {
            const int dr = s_r[k] - yy;
            if (s_v[k] != 0.f && dr >= -5 && dr <= 5 && (unsigned)s_r[k] < (unsigned)H && (unsigned)s_c[k] < (unsigned)W) s_near[n++] = k;
        }
        s_nn = n;
    }
    __syncthreads();
    const int nn = s_nn;
    const int KV = K >> 2;
    const int cy = (yy == 0) ? 0 : ((yy == H - 1) ? 2 : 1);
    for (int idx = threadIdx.x; idx < W * KV; idx += 256) {
        const int x = idx / KV, c0 = (idx - x * KV) * 4;
        const int cx = (x == 0) ? 0 : ((x == W - 1) ? 2 : 1);
        const int cls = cy * 3 + cx;
        const float4 e = *reinterpret_cast<const float4*>(e9 + ((long)b * 9 + cls) * K + c0);
        const float4 cp = *reinterpret_cast<const float4*>(cpos + (long)cls * K + c0);
        float4 a = make_float4(e.x - cp.x, e.y - cp.y, e.z - cp.z, e.w - cp.w);
        if (bias) { const float4 bv = *reinterpret_cast<const float4*>(bias + c0); a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w; }
        for (int q = 0; q < nn; ++q) {
            const int k = s_near[q];
            const int r = s_r[k], c = s_c[k];
            if (c - x < -5 || c - x > 5) continue;
            const float v = s_v[k];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int sy = yy + ky - 1;
                if ((unsigned)sy >= (unsigned)H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int sx = x + kx - 1;
                    if ((unsigned)sx >= (unsigned)W) continue;
                    const float amp = stem_amp(r, c, v, H, W, sy, sx);
                    if (amp == 0.f) continue;
                    const float4 wv = *reinterpret_cast<const float4*>(w + ((long)(ky * 3 + kx) * C + E + k) * K + c0);
                    a.x += amp * wv.x; a.y += amp * wv.y; a.z += amp * wv.z; a.w += amp * wv.w;
                }
            }
        }
        a.x = act_apply(a.x, act, alpha); a.y = act_apply(a.y, act, alpha); a.z = act_apply(a.z, act, alpha); a.w = act_apply(a.w, act, alpha);
        T* o = y + (((long)b * H + yy) * W + x) * K + c0;
        if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(o) = a;
        else *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)glue_f2b(a.x) | ((unsigned)glue_f2b(a.y) << 16), (unsigned)glue_f2b(a.z) | ((unsigned)glue_f2b(a.w) << 16));
    }
}
// filter gradient of the pose rows: dwp[tap][k][:] = - sum_b sum_{classes that contain tap} z9[b][class][:]
//                                                  + sum_b sum_{q in disc_k(b)} amp(q) * dz[b][q - tap + 1][:]
// block = (k, tap), 1024 threads; thread = (term slot, 4 output channels): the B * 9 class terms and the B * 81 window positions are
// dealt to the 1024 / (K/4) slots (the gather is a chain of dependent loads: it wants them spread), slots are folded through LDS in slot order.
// z9 = border-class sums of dz.  Fixed summation order.
template <typename T>
__global__ __launch_bounds__(1024) void pose_stem_wgrad_kernel(const float* __restrict__ rcv, int B, int P, int normalized,
                                                              const float* __restrict__ z9, const T* __restrict__ dz, int H, int W,
                                                              int K, float* __restrict__ dwp) {
    __shared__ float red[1024 * 4];
    const int k = blockIdx.x, tap = blockIdx.y, ky = tap / 3, kx = tap - ky * 3;
    const int KV = K >> 2;                                  // <= 256, divides 1024 (checked by the host)
    const int cv = threadIdx.x % KV, slot = threadIdx.x / KV, nslot = 1024 / KV;
    const int c0 = cv * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = slot; i < B * 9; i += nslot) {
        const int b = i / 9, cls = i - b * 9, cy = cls / 3, cx = cls - cy * 3;
        // classes (cy, cx) whose valid taps contain (ky, kx): cy = 0 lacks ky = 0, cy = 2 lacks ky = 2 (same for columns)
        if ((cy == 0 && ky == 0) || (cy == 2 && ky == 2) || (cx == 0 && kx == 0) || (cx == 2 && kx == 2)) continue;
        const float4 z = *reinterpret_cast<const float4*>(z9 + ((long)b * 9 + cls) * K + c0);
        acc.x -= z.x; acc.y -= z.y; acc.z -= z.z; acc.w -= z.w;
    }
    for (int i = slot; i < B * 81; i += nslot) {
        const int b = i / 81, w81 = i - b * 81, a = w81 / 9 - 4, bb = w81 - (w81 / 9) * 9 - 4;
        const int hw = stem_disc_half(a);
        if (bb < -hw || bb > hw) continue;
        int r, c; float v;
        stem_keypoint(rcv + ((long)b * P + k) * 3, H, W, normalized, &r, &c, &v);
        if (v == 0.f || (unsigned)r >= (unsigned)H || (unsigned)c >= (unsigned)W) continue;
        const int sy = r - a, sx = c - bb;                  // source pixel in the disc
        const int py = sy - ky + 1, px = sx - kx + 1;       // the output pixel that reads it through tap (ky, kx)
        if ((unsigned)sy >= (unsigned)H || (unsigned)py >= (unsigned)H || (unsigned)sx >= (unsigned)W || (unsigned)px >= (unsigned)W) continue;
        const float amp = 2.f * fminf(v * ((a == 0 && bb == 0) ? 2.f : 1.f), 1.f);
        const T* g = dz + (((long)b * H + py) * W + px) * K + c0;
        float g0, g1, g2, g3;
        if constexpr (sizeof(T) == 4) { const float4 t = *reinterpret_cast<const float4*>(g); g0 = t.x; g1 = t.y; g2 = t.z; g3 = t.w; }
        else { const uint2 t = *reinterpret_cast<const uint2*>(g); g0 = __uint_as_float(t.x << 16); g1 = __uint_as_float(t.x & 0xffff0000u);
               g2 = __uint_as_float(t.y << 16); g3 = __uint_as_float(t.y & 0xffff0000u); }
        acc.x += amp * g0; acc.y += amp * g1; acc.z += amp * g2; acc.w += amp * g3;
    }
    *reinterpret_cast<float4*>(&red[threadIdx.x * 4]) = acc;
    __syncthreads();
    if (slot == 0) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < nslot; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(&red[(q * KV + cv) * 4]);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        *reinterpret_cast<float4*>(dwp + ((long)tap * P + k) * K + c0) = s;
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

extern "C" int dpig_pose_stem_fwd(const float* rcv, int B, int P, int normalized, const float* e9, const float* cpos, const float* bias,
                                  const float* w, int C, int E, int H, int W, int K, int act, float alpha, void* y, int is_bf16, void* stream) {
    if (!rcv || !e9 || !cpos || !w || !y || C != E + P || E < 0 || B <= 0 || P <= 0 || P > STEM_MAXK || H < 2 || W < 2 || K <= 0 || K % 4)
        return fail(DPIG_EINVAL, "pose_stem_fwd: bad arguments (P <= %d, K %% 4 == 0)", STEM_MAXK);
    if (!aligned16(e9) || !aligned16(cpos) || !aligned16(w) || !aligned16(y) || (bias && !aligned16(bias)))
        return fail(DPIG_EINVAL, "pose_stem_fwd: operands must be 16-byte aligned");
    if ((long)B * H > 0x7fffffffL) return fail(DPIG_EINVAL, "pose_stem_fwd: too many rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (is_bf16) hipLaunchKernelGGL((pose_stem_fwd_kernel<glue_bf16>), dim3(B * H), dim3(256), 0, st, rcv, P, normalized, e9, cpos, bias, w, C, E, H, W, K, act, alpha, (glue_bf16*)y);
    else hipLaunchKernelGGL((pose_stem_fwd_kernel<float>), dim3(B * H), dim3(256), 0, st, rcv, P, normalized, e9, cpos, bias, w, C, E, H, W, K, act, alpha, (float*)y);
    return check_launch("pose_stem_fwd");
}
extern "C" int dpig_pose_stem_wgrad(const float* rcv, int B, int P, int normalized, const float* z9, const void* dz, int H, int W, int K,
                                    float* dwp, int is_bf16, void* stream) {
    if (!rcv || !z9 || !dz || !dwp || B <= 0 || P <= 0 || H < 2 || W < 2 || K <= 0 || K % 4 || K > 1024 || 256 % (K / 4))
        return fail(DPIG_EINVAL, "pose_stem_wgrad: bad arguments (K / 4 must divide 256)");
    if (!aligned16(z9) || !aligned16(dz) || !aligned16(dwp)) return fail(DPIG_EINVAL, "pose_stem_wgrad: operands must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (is_bf16) hipLaunchKernelGGL((pose_stem_wgrad_kernel<glue_bf16>), dim3(P, 9), dim3(1024), 0, st, rcv, B, P, normalized, z9, (const glue_bf16*)dz, H, W, K, dwp);
    else hipLaunchKernelGGL((pose_stem_wgrad_kernel<float>), dim3(P, 9), dim3(1024), 0, st, rcv, B, P, normalized, z9, (const float*)dz, H, W, K, dwp);
    return check_launch("pose_stem_wgrad");
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
