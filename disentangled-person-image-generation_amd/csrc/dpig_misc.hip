// HBM-bound companions of the conv kernels: activation/bias gradients, batch norm, layer norm,
// crop_and_resize, nearest upsample, TF-style Adam and the GAN/L1 losses.  All fp32, NHWC with
// explicit row strides, float4 accesses wherever alignment allows, two-pass (deterministic)
// reductions -- no atomics except the crop_and_resize scatter-add.
#include "dpig_common.h"

namespace dpig {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

static inline int cdivi(long a, long b) { return (int)((a + b - 1) / b); }
static inline int grid_for(long n, int per_block = 256, int cap = 8 * kNumCU) {
    int b = cdivi(n, per_block);
    if (b > cap) b = cap;
    return b < 1 ? 1 : b;
}

// ---------------------------------------------------------------------------------------------
// elementwise activation backward: dz = dy * act'(y)
// ---------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, int lddy,
                                                      const float* __restrict__ y, int ldy,
                                                      float* __restrict__ dz, int lddz, long rows, int cols,
                                                      int act, float alpha, unsigned short* __restrict__ dz32 = nullptr) {
    if (VEC) {
        const int c4 = cols >> 2;
        const long total = rows * c4;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const long r = i / c4;
            const int c = (int)(i - r * c4) * 4;
            const float4 g = *reinterpret_cast<const float4*>(dy + r * lddy + c);
            const float4 o = *reinterpret_cast<const float4*>(y + r * ldy + c);
            float4 v;
            v.x = g.x * act_grad(o.x, act, alpha);
            v.y = g.y * act_grad(o.y, act, alpha);
            v.z = g.z * act_grad(o.z, act, alpha);
            v.w = g.w * act_grad(o.w, act, alpha);
            *reinterpret_cast<float4*>(dz + r * lddz + c) = v;
            if (dz32) {     // the result's split32 image for the split-bf16 conv kernels ([row][chunk][32 hi | 32 lo], cols % 32 == 0)
                typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                b2 h0, h1, l0, l1;
                h0[0] = (__bf16)v.x; h0[1] = (__bf16)v.y; h1[0] = (__bf16)v.z; h1[1] = (__bf16)v.w;
                l0[0] = (__bf16)(v.x - (float)h0[0]); l0[1] = (__bf16)(v.y - (float)h0[1]);
                l1[0] = (__bf16)(v.z - (float)h1[0]); l1[1] = (__bf16)(v.w - (float)h1[1]);
                unsigned short* o = dz32 + (r * (cols >> 5) + (c >> 5)) * 64 + (c & 31);
                *reinterpret_cast<uint2*>(o) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                *reinterpret_cast<uint2*>(o + 32) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
            }
        }
    } else {
        const long total = rows * cols;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const long r = i / cols;
            const int c = (int)(i - r * cols);
            dz[r * lddz + c] = dy[r * lddy + c] * act_grad(y[r * ldy + c], act, alpha);
        }
    }
}

// y = act(x) (stand-alone LeakyReLU / ReLU, wgan_gp.py:23-24) ------------------------------------
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                      int ldy, long rows, int cols, int act, float alpha) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        y[r * ldy + c] = act_apply(x[r * ldx + c], act, alpha);
    }
}

// ---------------------------------------------------------------------------------------------
// column reductions over a [rows, C] matrix.  Block = 64 columns x 4 row groups; grid.y row slabs.
//   MODE 0: s0 = sum a
//   MODE 1: s0 = sum (a - mean[c])^2
//   MODE 2: dz = a*act'(y);  s0 = sum dz ; s1 = sum dz * (x - mean[c]) * rstd[c]          (BN bwd)
//   MODE 3: as MODE 2 with per-sample statistics mean[r / P], rstd[r / P]                  (LN bwd)
// partial layout: [slab][NOUT][C]
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ a, int lda,
                                                          const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ y, int ldy,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, long rows, int C,
                                                          int P, int act, float alpha,
                                                          float* __restrict__ partial) {
    constexpr int NOUT = (MODE >= 2) ? 2 : 1;
    __shared__ float red[NOUT][4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        float mu = 0.f, rs = 0.f;
        if (MODE == 1 || MODE == 2) mu = mean[c];
        if (MODE == 2) rs = rstd[c];
        for (long r = (long)blockIdx.y * 4 + rg; r < rows; r += (long)gridDim.y * 4) {
            const float v = a[r * lda + c];
            if (MODE == 0) {
                s0 += v;
            } else if (MODE == 1) {
                const float d = v - mu;
                s0 += d * d;
            } else {
                if (MODE == 3) { const long n = r / P; mu = mean[n]; rs = rstd[n]; }
                const float dz = (act != DPIG_ACT_NONE) ? v * act_grad(y[r * ldy + c], act, alpha) : v;
                s0 += dz;
                s1 += dz * (x[r * ldx + c] - mu) * rs;
            }
        }
    }
    red[0][rg][cl] = s0;
    if (NOUT == 2) red[NOUT - 1][rg][cl] = s1;
    __syncthreads();
    if (rg == 0 && c < C) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float t = (red[o][0][cl] + red[o][1][cl]) + (red[o][2][cl] + red[o][3][cl]);
            partial[((long)blockIdx.y * NOUT + o) * C + c] = t;
        }
    }
}

// out_o[c] = beta*out_o[c] + scale * sum_slab partial[slab][o][c]      (FIN 0)
// out_0[c] = 1/sqrt(scale * sum + eps)                                  (FIN 1)
// block = 64 columns x 4 slab groups (fixed summation order -> deterministic)
template <int FIN>
__global__ __launch_bounds__(256) void col_final_kernel(const float* __restrict__ partial, int nslab, int nout,
                                                        int C, float* __restrict__ out0,
                                                        float* __restrict__ out1, float scale, float beta,
                                                        float eps) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    for (int o = 0; o < nout; ++o) {
        float s = 0.f;
        if (c < C)
            for (int b = g; b < nslab; b += 4) s += partial[((long)b * nout + o) * C + c];
        red[o][g][cl] = s;
    }
    __syncthreads();
    if (g != 0 || c >= C) return;
    for (int o = 0; o < nout; ++o) {
        float* out = (o == 0) ? out0 : out1;
        if (!out) continue;
        const float s = (red[o][0][cl] + red[o][1][cl]) + (red[o][2][cl] + red[o][3][cl]);
        if (FIN == 1) out[c] = 1.0f / sqrtf(s * scale + eps);
        else out[c] = ((beta != 0.f) ? beta * out[c] : 0.f) + s * scale;
    }
}

static int slabs_for(long rows) {
    long s = rows / 128;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

// ---------------------------------------------------------------------------------------------
// per-image sums of a [N,H,W,C] tensor over the 9 border classes of a 3x3 SAME conv (the gradient
// of the class-indexed residual of dpig_conv2d_fwd).  partial: [N][slab][9][C]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float ld_elem(const float* a, long i) { return a[i]; }
__device__ __forceinline__ float ld_elem(const unsigned short* a, long i) { return __uint_as_float((unsigned)a[i] << 16); }   // bf16
template <typename T>
__global__ __launch_bounds__(256) void class_sum_partial_kernel(const T* __restrict__ a, int lda, int H, int W,
                                                                int C, float* __restrict__ partial) {
    __shared__ float red[9][4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int n = blockIdx.y, slab = blockIdx.z, nslab = gridDim.z;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    if (c < C) {
        const int P = H * W;
        for (int pix = slab * 4 + rg; pix < P; pix += nslab * 4) {
            const int y = pix / W, x = pix - y * W;
            const int cy = (y == 0) ? 0 : ((y == H - 1) ? 2 : 1);
            const int cx = (x == 0) ? 0 : ((x == W - 1) ? 2 : 1);
            const int cls = cy * 3 + cx;
            const float v = ld_elem(a, ((long)n * P + pix) * lda + c);
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] += (cls == k) ? v : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) red[k][rg][cl] = acc[k];
    __syncthreads();
    if (rg == 0 && c < C) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
            partial[(((long)n * nslab + slab) * 9 + k) * C + c] =
                (red[k][0][cl] + red[k][1][cl]) + (red[k][2][cl] + red[k][3][cl]);
    }
}
// The same partial sums with 16-byte loads: thread = (pixel slot, V consecutive channels), a slab = every nslab-th image row.
// A row has ONE row class, so a pixel costs one add per channel (+ two selects for the row's first / last pixel) instead of nine
// conditional adds; pixel slots are folded through LDS in slot order (fixed summation order).
template <typename T, int V>
__global__ __launch_bounds__(256) void class_sum_partial_vec_kernel(const T* __restrict__ a, int lda, int H, int W, int C,
                                                                    float* __restrict__ partial) {
    __shared__ float red[256 * V];
    const int CV = C / V;                                  // threads per pixel (divides 256)
    const int cv = threadIdx.x % CV, slot = threadIdx.x / CV, nslot = 256 / CV;
    const int n = blockIdx.y, slab = blockIdx.z, nslab = gridDim.z;
    float acc[9][V];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[k][e] = 0.f;
    for (int y = slab; y < H; y += nslab) {
        float l[V], m[V], r[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { l[e] = 0.f; m[e] = 0.f; r[e] = 0.f; }
        const T* rowp = a + ((long)n * H + y) * W * lda + cv * V;
#pragma unroll 4
        for (int x = slot; x < W; x += nslot) {
            float v[V];
            if constexpr (sizeof(T) == 2) {
                const uint4 u = *reinterpret_cast<const uint4*>(rowp + (long)x * lda);
                v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
                v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
                v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
                v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
            } else {
                const float4 u = *reinterpret_cast<const float4*>(rowp + (long)x * lda);
                v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
            }
            const bool first = x == 0, last = x == W - 1;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                l[e] += first ? v[e] : 0.f;
                r[e] += last ? v[e] : 0.f;
                m[e] += (first || last) ? 0.f : v[e];
            }
        }
        const int cy = (y == 0) ? 0 : ((y == H - 1) ? 2 : 1);       // (block-uniform)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k == cy) {
#pragma unroll
                for (int e = 0; e < V; ++e) { acc[k * 3][e] += l[e]; acc[k * 3 + 1][e] += m[e]; acc[k * 3 + 2][e] += r[e]; }
            }
    }
    for (int k = 0; k < 9; ++k) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < V; ++e) red[(slot * CV + cv) * V + e] = acc[k][e];
        __syncthreads();
        if (slot == 0) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float s = 0.f;
                for (int q = 0; q < nslot; ++q) s += red[(q * CV + cv) * V + e];
                partial[(((long)n * nslab + slab) * 9 + k) * C + cv * V + e] = s;
            }
        }
    }
}
__global__ __launch_bounds__(256) void class_sum_final_kernel(const float* __restrict__ partial, int nslab, int C,
                                                              long total, float* __restrict__ out) {
    // out[(n*9 + k)*C + c] = sum_slab partial[((n*nslab + slab)*9 + k)*C + c]
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long nk = i / C;
        const int c = (int)(i - nk * C);
        const long n = nk / 9;
        const int k = (int)(nk - n * 9);
        float s = 0.f;
        for (int b = 0; b < nslab; ++b) s += partial[((n * nslab + b) * 9 + k) * C + c];
        out[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// batch norm apply kernels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int ldx, long rows, int C,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ offset,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, int act, float alpha,
                                                       float* __restrict__ y, int ldy) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const float v = (x[r * ldx + c] - mean[c]) * rstd[c] * scale[c] + offset[c];
        y[r * ldy + c] = act_apply(v, act, alpha);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int lddy,
                                                           const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ y, int ldy, long rows,
                                                           int C, const float* __restrict__ scale,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ dscale,
                                                           const float* __restrict__ doffset, int act,
                                                           float alpha, float inv, float* __restrict__ dx, int lddx) {
    const long total = rows * C;      // inv = 1 / (number of rows the statistics were taken over)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        float dz = dy[r * lddy + c];
        if (act != DPIG_ACT_NONE) dz *= act_grad(y[r * ldy + c], act, alpha);
        const float xh = (x[r * ldx + c] - mean[c]) * rstd[c];
        dx[r * lddx + c] = scale[c] * rstd[c] * (dz - doffset[c] * inv - xh * dscale[c] * inv);
    }
}

// (the LayerNorm family and the WGAN-GP penalty reduction live in dpig_norm.hip)
// sum over a 1024-thread workgroup (the loss kernels below); the result is the same in every thread
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = (threadIdx.x < 16) ? red[threadIdx.x] : 0.f;
    if (w == 0) {
        t = wave_sum(t);
        if (l == 0) red[16] = t;
    }
    __syncthreads();
    return red[16];
}

// ---------------------------------------------------------------------------------------------
// tf.image.crop_and_resize (bilinear, extrapolation_value 0) and its image gradient
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool crop_coord(float b1, float b2, int i, int crop, int size, float* pos) {
    // no fma contraction: the forward and both backward passes must take bit-identical in/out-of-image
    // decisions for a sample, whatever code surrounds the inlined expression (TF's CPU kernel does not fuse)
#pragma clang fp contract(off)
    const float in = (crop > 1) ? b1 * (size - 1) + i * ((b2 - b1) * (size - 1) / (float)(crop - 1))
                                : 0.5f * (b1 + b2) * (size - 1);
    *pos = in;
    return !(in < 0.f || in > (float)(size - 1));
}

// forward: one thread per (box, crop pixel, channel quad) -- 16-byte loads/stores when C % 4 == 0
__device__ __forceinline__ void st_elem(float* a, long i, float v) { a[i] = v; }
__device__ __forceinline__ void st_elem(unsigned short* a, long i, float v) {      // bf16, round-to-nearest-even
    const __bf16 b = (__bf16)v;
    a[i] = __builtin_bit_cast(unsigned short, b);
}
template <int V, typename T>
__global__ __launch_bounds__(256) void crop_resize_fwd_kernel(const T* __restrict__ img, int H, int W, int C,
                                                              const float* __restrict__ boxes,
                                                              const int* __restrict__ box_ind, int nbox, int ch,
                                                              int cw, T* __restrict__ out) {
    const int CV = C / V;
    const long total = (long)nbox * ch * cw * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * V;
        long t = i / CV;
        const int j = (int)(t % cw); t /= cw;
        const int ii = (int)(t % ch);
        const int b = (int)(t / ch);
        const float y1 = boxes[b * 4 + 0], x1 = boxes[b * 4 + 1], y2 = boxes[b * 4 + 2], x2 = boxes[b * 4 + 3];
        float in_y, in_x;
        const bool oky = crop_coord(y1, y2, ii, ch, H, &in_y);
        const bool okx = crop_coord(x1, x2, j, cw, W, &in_x);
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = 0.f;
        if (oky && okx) {
            const int ty = (int)floorf(in_y), by = (int)ceilf(in_y);
            const int lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
            const float ly = in_y - ty, lxw = in_x - lx;
            const T* base = img + (long)box_ind[b] * H * W * C + c;
            const T* ptl = base + ((long)ty * W + lx) * C;
            const T* ptr_ = base + ((long)ty * W + rx) * C;
            const T* pbl = base + ((long)by * W + lx) * C;
            const T* pbr = base + ((long)by * W + rx) * C;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float tl = ld_elem(ptl, e), tr = ld_elem(ptr_, e), bl = ld_elem(pbl, e), br = ld_elem(pbr, e);
                const float top = tl + (tr - tl) * lxw;
                const float bot = bl + (br - bl) * lxw;
                v[e] = top + (bot - top) * ly;
            }
        }
        T* o = out + ((((long)b * ch + ii) * cw + j) * C + c);
#pragma unroll
        for (int e = 0; e < V; ++e) st_elem(o, e, v[e]);
    }
}

// backward as two separable GATHER passes (deterministic: no atomics, fixed summation order).
// A sample at continuous coordinate `in` scatters onto integer position p with the bilinear hat weight
// max(0, 1-|in-p|) (floor gets 1-lerp, ceil gets lerp), and the 2-D weight is the product of the two 1-D
// ones, so   dimg[n,y,x] = sum_{b: ind[b]=n} sum_i hat(in_y(b,i)-y) * ( sum_j hat(in_x(b,j)-x) * dout[b,i,j] ).
// Pass X forms the bracket into tmp[b,i,x,:], pass Y finishes.  Out-of-image samples (extrapolation) carry no
// gradient; crop_coord() reproduces the forward's decision bit for bit.  (TF: CropAndResizeGradImage.)
__device__ __forceinline__ void crop_range(float b1, float b2, int crop, int size, int p, int* lo, int* hi) {
    // sample indices whose coordinate can fall in (p-1, p+1); conservative by one on both sides
    if (crop <= 1) { *lo = 0; *hi = 0; return; }
    const float step = (b2 - b1) * (size - 1) / (float)(crop - 1);
    const float org = b1 * (size - 1);
    if (fabsf(step) < 1e-12f) { *lo = 0; *hi = crop - 1; return; }
    float a = ((float)(p - 1) - org) / step, b = ((float)(p + 1) - org) / step;
    if (a > b) { const float t = a; a = b; b = t; }
    a = fmaxf(a, -2.f); b = fminf(b, (float)crop + 1.f);
    const int l = (int)floorf(a) - 1, h = (int)ceilf(b) + 1;
    *lo = l < 0 ? 0 : l;
    *hi = h > crop - 1 ? crop - 1 : h;
}
// V consecutive channels (V = 4: one 16-byte fp32 / 8-byte bf16 access)
template <int V> __device__ __forceinline__ void ld_vec(const float* p, float (&v)[V]) {
    if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else { for (int e = 0; e < V; ++e) v[e] = p[e]; }
}
template <int V> __device__ __forceinline__ void ld_vec(const unsigned short* p, float (&v)[V]) {
    if constexpr (V == 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    } else { for (int e = 0; e < V; ++e) v[e] = __uint_as_float((unsigned)p[e] << 16); }
}
// Image columns a box can touch: samples lie in [min, max] * (W - 1) and reach pixels closer than 1.  Pass X only fills these
// columns of tmp and pass Y only reads them (the rest of a [W]-wide row would be zeros: 3/4 of the traffic at the DeepFashion sizes).
__device__ __forceinline__ void crop_cols(float x1, float x2, int W, int* lo, int* hi) {
    const float a = fminf(x1, x2) * (float)(W - 1), b = fmaxf(x1, x2) * (float)(W - 1);
    const int l = (int)floorf(a) - 1, h = (int)ceilf(b) + 1;
    *lo = l < 0 ? 0 : l;
    *hi = h > W - 1 ? W - 1 : h;
}
constexpr int CROP_MAXC = 128;      // largest crop side the row-table kernels take (reference: 48 / 64)
// pass X, one workgroup per crop row (b, ii): the row's sample coordinates in_x[j] (and whether the sample lies inside the image)
// are computed ONCE into LDS; a thread = (image column of the box's column range, channel group) then walks only the samples
// crop_range() brackets -- no index arithmetic or divisions per sample.  Fixed summation order (j ascending).
template <int V, typename T>
__global__ __launch_bounds__(256) void crop_bwd_x_kernel(const T* __restrict__ dout, int W, int C,
                                                         const float* __restrict__ boxes, int nbox, int ch, int cw,
                                                         float* __restrict__ tmp) {
    __shared__ float s_in[CROP_MAXC];
    const int CV = C / V;
    const long t = blockIdx.x;                                   // b*ch + ii
    const int b = (int)(t / ch);
    const float x1 = boxes[b * 4 + 1], x2 = boxes[b * 4 + 3];
    for (int j = threadIdx.x; j < cw; j += 256) {
        float in_x;
        s_in[j] = crop_coord(x1, x2, j, cw, W, &in_x) ? in_x : -4.f;       // outside the image: weight <= 0 for every pixel
    }
    __syncthreads();
    int xlo, xhi;
    crop_cols(x1, x2, W, &xlo, &xhi);
    const int ncol = xhi - xlo + 1;
    const T* g = dout + (t * cw) * C;
    for (int idx = threadIdx.x; idx < ncol * CV; idx += 256) {
        const int xq = idx / CV, c = (idx - xq * CV) * V;
        const int x = xlo + xq;
        int jlo, jhi;
        crop_range(x1, x2, cw, W, x, &jlo, &jhi);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int j = jlo; j <= jhi; ++j) {
            const float w = 1.f - fabsf(s_in[j] - (float)x);
            if (w <= 0.f) continue;
            float gv[V];
            ld_vec<V>(g + (long)j * C + c, gv);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += w * gv[e];
        }
        float* o = tmp + (t * W + x) * C + c;
        if constexpr (V == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else {
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = acc[e];
        }
    }
}
// pass Y, one workgroup per image row (n, y): the (box, crop row, weight) triples that reach this row are listed ONCE in LDS in
// (box, crop row) order -- the fixed summation order -- together with the box's column range; a thread = (pixel of the row,
// channel group) adds the listed tmp rows whose column range holds its pixel.
constexpr int CROP_MAXENT = 1024;
template <int V, typename T>
__global__ __launch_bounds__(256) void crop_bwd_y_kernel(const float* __restrict__ tmp, int N, int H, int W, int C,
                                                         const float* __restrict__ boxes,
                                                         const int* __restrict__ box_ind, int nbox, int ch,
                                                         T* __restrict__ dimg) {
    __shared__ int s_row[CROP_MAXENT];          // b*ch + ii
    __shared__ float s_w[CROP_MAXENT];
    __shared__ short s_lo[CROP_MAXENT], s_hi[CROP_MAXENT];
    __shared__ int s_n;
    const int CV = C / V;
    const int n = blockIdx.x / H, y = blockIdx.x - n * H;
    // (blockIdx.y: which 256-thread slice of the row's W * C/V work items -- the gather is latency-bound, it wants many threads)
    // list building, thread = box (boxes beyond 256: the per-thread path below): count this box's samples that reach row y, place the
    // box's run after the runs of the boxes before it -- (box, crop row) order, the fixed summation order
    __shared__ int s_cnt[256];
    int my = 0, ilo = 0, ihi = -1, xlo = 0, xhi = -1;
    float y1 = 0.f, y2 = 0.f;
    if (threadIdx.x < nbox && nbox <= 256 && box_ind[threadIdx.x] == n) {
        const int b = threadIdx.x;
        y1 = boxes[b * 4 + 0]; y2 = boxes[b * 4 + 2];
        crop_range(y1, y2, ch, H, y, &ilo, &ihi);
        crop_cols(boxes[b * 4 + 1], boxes[b * 4 + 3], W, &xlo, &xhi);
        for (int ii = ilo; ii <= ihi; ++ii) {
            float in_y;
            if (!crop_coord(y1, y2, ii, ch, H, &in_y)) continue;
            if (1.f - fabsf(in_y - (float)y) > 0.f) ++my;
        }
    }
    s_cnt[threadIdx.x] = my;
    __syncthreads();
    if (nbox <= 256) {
        int off = 0, tot = 0;
        for (int b = 0; b < nbox; ++b) { const int cb = s_cnt[b]; if (b < (int)threadIdx.x) off += cb; tot += cb; }
        if (threadIdx.x == 0) s_n = tot;
        if (my > 0 && tot <= CROP_MAXENT) {
            for (int ii = ilo; ii <= ihi; ++ii) {
                float in_y;
                if (!crop_coord(y1, y2, ii, ch, H, &in_y)) continue;
                const float w = 1.f - fabsf(in_y - (float)y);
                if (w <= 0.f) continue;
                s_row[off] = threadIdx.x * ch + ii; s_w[off] = w; s_lo[off] = (short)xlo; s_hi[off] = (short)xhi;
                ++off;
            }
        }
    } else if (threadIdx.x == 0) {
        s_n = CROP_MAXENT + 1;
    }
    __syncthreads();
    const int cnt = s_n;
    if (cnt > CROP_MAXENT) {
        // more samples reach this row than the table holds (hundreds of boxes on one image): every thread walks the boxes itself,
        // same order, same arithmetic
        for (int idx = blockIdx.y * 256 + threadIdx.x; idx < W * CV; idx += gridDim.y * 256) {
            const int x = idx / CV, c = (idx - x * CV) * V;
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
            for (int b = 0; b < nbox; ++b) {
                if (box_ind[b] != n) continue;
                int xlo, xhi, ilo, ihi;
                crop_cols(boxes[b * 4 + 1], boxes[b * 4 + 3], W, &xlo, &xhi);
                if (x < xlo || x > xhi) continue;
                const float y1 = boxes[b * 4 + 0], y2 = boxes[b * 4 + 2];
                crop_range(y1, y2, ch, H, y, &ilo, &ihi);
                for (int ii = ilo; ii <= ihi; ++ii) {
                    float in_y;
                    if (!crop_coord(y1, y2, ii, ch, H, &in_y)) continue;
                    const float w = 1.f - fabsf(in_y - (float)y);
                    if (w <= 0.f) continue;
                    float gv[V];
                    ld_vec<V>(tmp + (((long)b * ch + ii) * W + x) * C + c, gv);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += w * gv[e];
                }
            }
            T* o = dimg + (((long)n * H + y) * W + x) * C + c;
#pragma unroll
            for (int e = 0; e < V; ++e) st_elem(o, e, acc[e]);
        }
        return;
    }
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < W * CV; idx += gridDim.y * 256) {
        const int x = idx / CV, c = (idx - x * CV) * V;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int k = 0; k < cnt; ++k) {
            if (x < s_lo[k] || x > s_hi[k]) continue;
            float gv[V];
            ld_vec<V>(tmp + ((long)s_row[k] * W + x) * C + c, gv);
            const float w = s_w[k];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += w * gv[e];
        }
        T* o = dimg + (((long)n * H + y) * W + x) * C + c;
        if constexpr (V == 4 && sizeof(T) == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else if constexpr (V == 4) {
            unsigned short h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const __bf16 bb = (__bf16)acc[e]; h[e] = __builtin_bit_cast(unsigned short, bb); }
            *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) st_elem(o, e, acc[e]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// input pipeline: pose target maps (utils.py:237-318)
// ---------------------------------------------------------------------------------------------
// half-width of the inflate stencil at row distance |a| (a disc of radius 4): |b| <= kDiscHalf[|a|]
__device__ __forceinline__ int disc_half(int a) {
    const int aa = a < 0 ? -a : a;
    return aa == 0 ? 4 : (aa <= 2 ? 3 : (aa == 3 ? 2 : (aa == 4 ? 0 : -1)));
}
__device__ __forceinline__ void keypoint_pixel(const float* __restrict__ rcv, int H, int W, int normalized, int* r,
                                               int* c, float* v) {
    float R = rcv[0], C = rcv[1];
    if (normalized) {
        R = fminf(fmaxf((R + 1.f) / 2.0f * (float)H, 0.f), (float)(H - 1));
        C = fminf(fmaxf((C + 1.f) / 2.0f * (float)W, 0.f), (float)(W - 1));
    }
    *r = (int)R; *c = (int)C; *v = rcv[2];                // tf.to_int32 truncates
}
// MODE 0: single points (coord2channel_simple_rcv), MODE 1: points + disc (the chained pipeline)
template <int MODE>
__global__ __launch_bounds__(256) void pose_from_rcv_kernel(const float* __restrict__ rcv, int B, int K, int H, int W,
                                                            int normalized, float* __restrict__ out, int ldo) {
    const long total = (long)B * H * W * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        long t = i / K;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        int r, c; float v;
        keypoint_pixel(rcv + ((long)b * K + k) * 3, H, W, normalized, &r, &c, &v);
        float o = -1.f;
        if (MODE == 0) {
            if ((unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W && r == y && c == x) o = 2.f * v - 1.f;
        } else {
            const int a = r - y, bb = c - x;
            const int hw = disc_half(a);
            if ((unsigned)r < (unsigned)H && (unsigned)c < (unsigned)W && hw >= 0 && bb >= -hw && bb <= hw) {
                const float hits = (a == 0 && bb == 0) ? 2.f : 1.f;      // the shift list contains (0,0) once more
                o = fminf(v * hits, 1.f) * 2.f - 1.f;
            }
        }
        out[((((long)b * H + y) * W + x)) * ldo + k] = o;
    }
}
__global__ __launch_bounds__(256) void pose_inflate_kernel(const float* __restrict__ pose, int ldp, int B, int K, int H,
                                                           int W, float* __restrict__ out, int ldo) {
    const long total = (long)B * H * W * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        long t = i / K;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const float* base = pose + ((long)b * H * W) * ldp + k;
        float s = (base[((long)y * W + x) * ldp] + 1.f) * 0.5f;          // the unshifted map itself
        for (int a = -4; a <= 4; ++a) {
            const int yy = y + a, hw = disc_half(a);
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int bb = -hw; bb <= hw; ++bb) {
                const int xx = x + bb;
                if ((unsigned)xx < (unsigned)W) s += (base[((long)yy * W + xx) * ldp] + 1.f) * 0.5f;
            }
        }
        out[(((long)b * H + y) * W + x) * ldo + k] = fminf(s, 1.f) * 2.f - 1.f;
    }
}

// ---------------------------------------------------------------------------------------------
// SSIM of trainer.generate() (skimage compare_ssim on gray uint8 images), per image
// ---------------------------------------------------------------------------------------------
constexpr int kSsimBlocks = 16;      // workgroups per image in every stage
__device__ __forceinline__ float gray_u8(const float* __restrict__ px) {
    const float r = floorf(fminf(fmaxf(px[0], 0.f), 255.f)), g = floorf(fminf(fmaxf(px[1], 0.f), 255.f)),
                b = floorf(fminf(fmaxf(px[2], 0.f), 255.f));          // clip, astype(uint8)
    return (r * 0.2125f + g * 0.7154f + b * 0.0721f) * (1.0f / 255.0f);
}
__device__ __forceinline__ float block_reduce_256(float v, float* red, int op) {     // op 0 sum, 1 min, 2 max
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o, 64);
        v = op == 0 ? v + t : (op == 1 ? fminf(v, t) : fmaxf(v, t));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < 4; ++i) r = op == 0 ? r + red[i] : (op == 1 ? fminf(r, red[i]) : fmaxf(r, red[i]));
    return r;
}
// stage 1: gray maps ga, gb [B][H*W] and per-workgroup (min, max) of gb
__global__ __launch_bounds__(256) void ssim_gray_kernel(const float* __restrict__ a, const float* __restrict__ b, int HW,
                                                        float* __restrict__ ga, float* __restrict__ gb,
                                                        float* __restrict__ mm) {
    __shared__ float red[4];
    const int img = blockIdx.y;
    float lo = 3.4e38f, hi = -3.4e38f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const long o = (long)img * HW + i;
        const float x = gray_u8(a + o * 3), y = gray_u8(b + o * 3);
        ga[o] = x; gb[o] = y;
        lo = fminf(lo, y); hi = fmaxf(hi, y);
    }
    lo = block_reduce_256(lo, red, 1);
    hi = block_reduce_256(hi, red, 2);
    if (threadIdx.x == 0) { mm[(img * gridDim.x + blockIdx.x) * 2] = lo; mm[(img * gridDim.x + blockIdx.x) * 2 + 1] = hi; }
}
// stage 2: SSIM of every 7x7 window, per-workgroup partial sums
__global__ __launch_bounds__(256) void ssim_map_kernel(const float* __restrict__ ga, const float* __restrict__ gb, int H,
                                                       int W, const float* __restrict__ mm, int nmm,
                                                       float* __restrict__ part) {
    __shared__ float red[4];
    const int img = blockIdx.y;
    float lo = 3.4e38f, hi = -3.4e38f;
    for (int i = 0; i < nmm; ++i) { lo = fminf(lo, mm[(img * nmm + i) * 2]); hi = fmaxf(hi, mm[(img * nmm + i) * 2 + 1]); }
    const float R = hi - lo;
    const float C1 = (0.01f * R) * (0.01f * R), C2 = (0.03f * R) * (0.03f * R);
    const int Hv = H - 6, Wv = W - 6;
    const float* A = ga + (long)img * H * W;
    const float* Bm = gb + (long)img * H * W;
    float sum = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Hv * Wv; i += gridDim.x * 256) {
        const int y = i / Wv, x = i - y * Wv;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
        for (int dy = 0; dy < 7; ++dy)
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) {
                const float u = A[(y + dy) * W + x + dx], v = Bm[(y + dy) * W + x + dx];
                sx += u; sy += v; sxx += u * u; syy += v * v; sxy += u * v;
            }
        const float n = 49.f, ux = sx / n, uy = sy / n;
        const float cn = n / (n - 1.f);
        const float vx = cn * (sxx / n - ux * ux), vy = cn * (syy / n - uy * uy), vxy = cn * (sxy / n - ux * uy);
        sum += ((2.f * ux * uy + C1) * (2.f * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
    sum = block_reduce_256(sum, red, 0);
    if (threadIdx.x == 0) part[img * gridDim.x + blockIdx.x] = sum;
}
__global__ void ssim_final_kernel(const float* __restrict__ part, int npart, int B, float inv_count,
                                  float* __restrict__ out) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= B) return;
    float s = 0.f;
    for (int i = 0; i < npart; ++i) s += part[img * npart + i];
    out[img] = s * inv_count;
}

// ---------------------------------------------------------------------------------------------
// WGAN-GP gradient penalty: interpolation, and penalty + double-backward seed in one pass
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_interpolate_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                             const float* __restrict__ alpha, long D, long total,
                                                             float* __restrict__ xhat) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float a = alpha[i / D];
        const float r = real[i];
        xhat[i] = r + a * (fake[i] - r);
    }
}

// ---------------------------------------------------------------------------------------------
// nearest-neighbour 2x upsample (align_corners=False -> exact 2x2 replication) and its gradient
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ x, int N, int H, int W,
                                                             int C, float* __restrict__ y) {
    const long total = (long)N * 2 * H * 2 * W * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long t = i / C;
        const int ox = (int)(t % (2 * W)); t /= (2 * W);
        const int oy = (int)(t % (2 * H));
        const long n = t / (2 * H);
        y[i] = x[((n * H + (oy >> 1)) * W + (ox >> 1)) * C + c];
    }
}
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, int N, int H, int W,
                                                             int C, float* __restrict__ dx) {
    const long total = (long)N * H * W * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long t = i / C;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const long n = t / H;
        const float* p = dy + ((n * 2 * H + 2 * iy) * (2L * W) + 2 * ix) * C + c;
        dx[i] = (p[0] + p[C]) + (p[2L * W * C] + p[2L * W * C + C]);
    }
}

// ---------------------------------------------------------------------------------------------
// TensorFlow Adam (epsilon outside the bias-corrected sqrt): trainer.py:137-140
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_t, float b1, float b2,
                                         float eps) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= lr_t * m / (sqrtf(v) + eps);
}

// Two float4 groups in flight per thread, non-temporal accesses (every byte is touched once per step): 5.5 -> 5.7-5.9 TB/s on the
// generator's 118.5 M parameters (scripts/ubench/adam_bw.py; 28 B per parameter and step).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n,
                                                    const float* __restrict__ lr_dev, float b1, float b2,
                                                    float eps, float corr, float gscale,
                                                    const float* __restrict__ corr_dev) {
    const float lr_t = lr_dev[0] * (corr_dev ? corr_dev[0] : corr);
    const long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    typedef float f4v __attribute__((ext_vector_type(4)));
    auto ld = [&](const float4* q) -> float4 {
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(q));
        return make_float4(t[0], t[1], t[2], t[3]);
    };
    auto st = [&](float4* q, const float4& x) {
        const f4v t = {x.x, x.y, x.z, x.w};
        __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(q));
    };
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const long j = i + stride;
        float4 pa = ld(p4 + i), ma = ld(m4 + i), va = ld(v4 + i), ga = ld(g4 + i);
        float4 pb = ld(p4 + j), mb = ld(m4 + j), vb = ld(v4 + j), gb = ld(g4 + j);
        adam_one(pa.x, ga.x * gscale, ma.x, va.x, lr_t, b1, b2, eps);
        adam_one(pa.y, ga.y * gscale, ma.y, va.y, lr_t, b1, b2, eps);
        adam_one(pa.z, ga.z * gscale, ma.z, va.z, lr_t, b1, b2, eps);
        adam_one(pa.w, ga.w * gscale, ma.w, va.w, lr_t, b1, b2, eps);
        adam_one(pb.x, gb.x * gscale, mb.x, vb.x, lr_t, b1, b2, eps);
        adam_one(pb.y, gb.y * gscale, mb.y, vb.y, lr_t, b1, b2, eps);
        adam_one(pb.z, gb.z * gscale, mb.z, vb.z, lr_t, b1, b2, eps);
        adam_one(pb.w, gb.w * gscale, mb.w, vb.w, lr_t, b1, b2, eps);
        st(p4 + i, pa); st(m4 + i, ma); st(v4 + i, va);
        st(p4 + j, pb); st(m4 + j, mb); st(v4 + j, vb);
    }
    if (i < n4) {
        float4 pa = ld(p4 + i), ma = ld(m4 + i), va = ld(v4 + i), ga = ld(g4 + i);
        adam_one(pa.x, ga.x * gscale, ma.x, va.x, lr_t, b1, b2, eps);
        adam_one(pa.y, ga.y * gscale, ma.y, va.y, lr_t, b1, b2, eps);
        adam_one(pa.z, ga.z * gscale, ma.z, va.z, lr_t, b1, b2, eps);
        adam_one(pa.w, ga.w * gscale, ma.w, va.w, lr_t, b1, b2, eps);
        st(p4 + i, pa); st(m4 + i, ma); st(v4 + i, va);
    }
    for (long t = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        adam_one(p[t], g[t] * gscale, m[t], v[t], lr_t, b1, b2, eps);
}
static void launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float b1, float b2,
                        float eps, float corr, float gscale, const float* corr_dev) {
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, st, p, g, m, v, n, lr_dev, b1, b2, eps, corr, gscale, corr_dev);
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const void* const* __restrict__ ptrs,
                                                         const long* __restrict__ sizes,
                                                         const float* __restrict__ lr_dev, float b1, float b2,
                                                         float eps, float corr, float gscale) {
    const int t = blockIdx.y;
    const long n = sizes[t];
    float* p = (float*)ptrs[4 * t + 0];
    const float* g = (const float*)ptrs[4 * t + 1];
    float* m = (float*)ptrs[4 * t + 2];
    float* v = (float*)ptrs[4 * t + 3];
    const float lr_t = lr_dev[0] * corr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        adam_one(p[i], g[i] * gscale, m[i], v[i], lr_t, b1, b2, eps);
}

// ---------------------------------------------------------------------------------------------
// tf.train.RMSPropOptimizer (trainer.py:119-122, wgan / lsgan modes): ms = d*ms + (1-d)*g^2;
// mom = mu*mom + lr*g/sqrt(ms+eps); p -= mom.  (TF initialises ms to ONES, mom to zeros.)
// and the WGAN weight clipping of trainer.py:124-128.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ ms, float* __restrict__ mom, long n,
                                                      const float* __restrict__ lr_dev, float decay, float mu,
                                                      float eps, float gscale) {
    const float lr = lr_dev[0];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gg = g[i] * gscale;
        const float m2 = decay * ms[i] + (1.f - decay) * gg * gg;
        const float mo = mu * mom[i] + lr * gg / sqrtf(m2 + eps);
        ms[i] = m2;
        mom[i] = mo;
        p[i] -= mo;
    }
}
__global__ __launch_bounds__(256) void clip_kernel(float* __restrict__ p, long n, float lo, float hi) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        p[i] = fminf(fmaxf(p[i], lo), hi);
}

// ---------------------------------------------------------------------------------------------
// losses
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sce_mean_kernel(const float* __restrict__ x, int n, float label,
                                                        float* __restrict__ out, float* __restrict__ dx,
                                                        float scale) {
    __shared__ float red[17];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float v = x[i];
        // max(x,0) - x*z + log1p(exp(-|x|))   (tf.nn.sigmoid_cross_entropy_with_logits)
        s += fmaxf(v, 0.f) - v * label + log1pf(expf(-fabsf(v)));
        if (dx) {
            const float sg = 1.f / (1.f + expf(-v));
            dx[i] = scale * (sg - label) / (float)n;
        }
    }
    const float t = block_sum_1024(s, red);
    if (threadIdx.x == 0) out[0] = t / (float)n;
}

// wgan / lsgan critic-output losses (trainer.py:218-220, 246-248): mean(x) or mean((x - target)^2) of a logit vector,
// and the gradient scale * d(mean)/dx
__global__ __launch_bounds__(1024) void logit_mean_kernel(const float* __restrict__ x, int n, int squared, float target,
                                                          float* __restrict__ out, float* __restrict__ dx, float scale) {
    __shared__ float red[17];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float v = x[i] - (squared ? target : 0.f);
        s += squared ? v * v : v;
        if (dx) dx[i] = scale * (squared ? 2.f * v : 1.f) / (float)n;
    }
    const float t = block_sum_1024(s, red);
    if (threadIdx.x == 0) out[0] = t / (float)n;
}

__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         long n, float* __restrict__ da, float gs,
                                                         float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        s += fabsf(d);
        if (da) da[i] = (d > 0.f) ? gs : ((d < 0.f) ? -gs : 0.f);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ partial, int nb, float inv_n,
                                                       float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
}

}  // namespace dpig

using namespace dpig;

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int dpig_version(void) { return DPIG_VERSION; }
extern "C" const char* dpig_last_error(void) { return err_buf(); }

extern "C" int dpig_act_bwd(const float* dy, int lddy, const float* y, int ldy, float* dz, int lddz, int64_t rows,
                            int cols, int act, float alpha, void* stream) {
    if (!dy || !y || !dz) return fail(DPIG_EINVAL, "act_bwd: null pointer");
    if (rows <= 0 || cols <= 0) return fail(DPIG_EINVAL, "act_bwd: empty");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = aligned16(dy) && aligned16(y) && aligned16(dz) && cols % 4 == 0 && lddy % 4 == 0 &&
                     ldy % 4 == 0 && lddz % 4 == 0;
    if (vec)
        hipLaunchKernelGGL((act_bwd_kernel<true>), dim3(grid_for(rows * (cols / 4))), dim3(256), 0, st, dy, lddy, y,
                           ldy, dz, lddz, (long)rows, cols, act, alpha);
    else
        hipLaunchKernelGGL((act_bwd_kernel<false>), dim3(grid_for(rows * cols)), dim3(256), 0, st, dy, lddy, y, ldy,
                           dz, lddz, (long)rows, cols, act, alpha);
    return check_launch("act_bwd_kernel");
}

// dpig_act_bwd that also leaves dz's split32 image (dpig_split32 layout) for the split-bf16 conv kernels that read dz next
extern "C" int dpig_act_bwd_s32(const float* dy, int lddy, const float* y, int ldy, float* dz, int lddz, int64_t rows, int cols,
                                int act, float alpha, uint16_t* dz32, void* stream) {
    if (!dy || !y || !dz || !dz32) return fail(DPIG_EINVAL, "act_bwd_s32: null pointer");
    if (rows <= 0 || cols <= 0) return fail(DPIG_EINVAL, "act_bwd_s32: empty");
    if ((cols % 32) || (lddy % 4) || (ldy % 4) || (lddz % 4) || ((uintptr_t)dy % 16) || ((uintptr_t)y % 16) || ((uintptr_t)dz % 16) ||
        ((uintptr_t)dz32 % 16))
        return fail(DPIG_EINVAL, "act_bwd_s32: needs a multiple of 32 columns and 16-byte addressable rows");
    hipLaunchKernelGGL((act_bwd_kernel<true>), dim3(grid_for(rows * (cols / 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dy, lddy, y, ldy, dz, lddz, (long)rows, cols, act, alpha, dz32);
    return check_launch("act_bwd_kernel");
}

extern "C" int dpig_act_fwd(const float* x, int ldx, float* y, int ldy, int64_t rows, int cols, int act, float alpha,
                            void* stream) {
    if (!x || !y) return fail(DPIG_EINVAL, "act_fwd: null pointer");
    if (rows <= 0 || cols <= 0) return fail(DPIG_EINVAL, "act_fwd: empty");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       ldx, y, ldy, (long)rows, cols, act, alpha);
    return check_launch("act_fwd_kernel");
}

extern "C" size_t dpig_colsum_workspace_bytes(int64_t rows, int cols) {
    return (size_t)slabs_for(rows) * 2 * cols * sizeof(float);
}
extern "C" int dpig_colsum(const float* a, int lda, int64_t rows, int cols, float* out, float beta, void* ws,
                           size_t ws_bytes, void* stream) {
    if (!a || !out) return fail(DPIG_EINVAL, "colsum: null pointer");
    if (!ws || ws_bytes < dpig_colsum_workspace_bytes(rows, cols)) return fail(DPIG_ENOMEM, "colsum: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nslab = slabs_for(rows);
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL((col_partial_kernel<0>), dim3(cdivi(cols, 64), nslab), dim3(256), 0, st, a, lda, nullptr, 0,
                       nullptr, 0, nullptr, nullptr, (long)rows, cols, 1, 0, 0.f, partial);
    hipLaunchKernelGGL((col_final_kernel<0>), dim3(cdivi(cols, 64)), dim3(256), 0, st, partial, nslab, 1, cols, out,
                       nullptr, 1.0f, beta, 0.f);
    return check_launch("colsum");
}

static int class_slabs(int N, int H, int W) {
    int s = (H * W) / 256;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    if (s > H) s = H;            // (the vector kernel deals whole image rows to slabs)
    return s;
}
extern "C" size_t dpig_border_class_sum_workspace_bytes(int N, int H, int W, int C) {
    return (size_t)N * class_slabs(N, H, W) * 9 * C * sizeof(float);
}
template <typename T>
static int border_class_sum_impl(const T* a, int lda, int N, int H, int W, int C, float* out, void* ws, size_t ws_bytes,
                                 void* stream) {
    if (!a || !out) return fail(DPIG_EINVAL, "border_class_sum: null pointer");
    if (N <= 0 || H < 2 || W < 2 || C <= 0 || lda < C) return fail(DPIG_EINVAL, "border_class_sum: bad shape");
    if (!ws || ws_bytes < dpig_border_class_sum_workspace_bytes(N, H, W, C))
        return fail(DPIG_ENOMEM, "border_class_sum: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nslab = class_slabs(N, H, W);
    float* partial = static_cast<float*>(ws);
    constexpr int V = 16 / (int)sizeof(T);
    if (C % V == 0 && lda % V == 0 && C / V <= 256 && 256 % (C / V) == 0 && aligned16(a))
        hipLaunchKernelGGL((class_sum_partial_vec_kernel<T, V>), dim3(1, N, nslab), dim3(256), 0, st, a, lda, H, W, C, partial);
    else
        hipLaunchKernelGGL(class_sum_partial_kernel<T>, dim3(cdivi(C, 64), N, nslab), dim3(256), 0, st, a, lda, H, W, C, partial);
    const long total = (long)N * 9 * C;
    hipLaunchKernelGGL(class_sum_final_kernel, dim3(grid_for(total)), dim3(256), 0, st, partial, nslab, C, total, out);
    return check_launch("border_class_sum");
}
extern "C" int dpig_border_class_sum(const float* a, int lda, int N, int H, int W, int C, float* out, void* ws,
                                     size_t ws_bytes, void* stream) {
    return border_class_sum_impl<float>(a, lda, N, H, W, C, out, ws, ws_bytes, stream);
}
extern "C" int dpig_border_class_sum_bf16(const uint16_t* a, int lda, int N, int H, int W, int C, float* out, void* ws,
                                          size_t ws_bytes, void* stream) {
    return border_class_sum_impl<unsigned short>(a, lda, N, H, W, C, out, ws, ws_bytes, stream);
}

extern "C" size_t dpig_bn_workspace_bytes(int64_t rows, int C) {
    return (size_t)slabs_for(rows) * 2 * C * sizeof(float);
}
extern "C" int dpig_bn_fwd(const float* x, int ldx, int64_t rows, int C, const float* scale, const float* offset,
                           float eps, int act, float alpha, float* y, int ldy, float* save_mean, float* save_rstd,
                           void* ws, size_t ws_bytes, void* stream) {
    if (!x || !scale || !offset || !y || !save_mean || !save_rstd) return fail(DPIG_EINVAL, "bn_fwd: null pointer");
    if (!ws || ws_bytes < dpig_bn_workspace_bytes(rows, C)) return fail(DPIG_ENOMEM, "bn_fwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nslab = slabs_for(rows);
    float* partial = static_cast<float*>(ws);
    const dim3 g1(cdivi(C, 64), nslab), g2(cdivi(C, 64));
    hipLaunchKernelGGL((col_partial_kernel<0>), g1, dim3(256), 0, st, x, ldx, nullptr, 0, nullptr, 0, nullptr,
                       nullptr, (long)rows, C, 1, 0, 0.f, partial);
    hipLaunchKernelGGL((col_final_kernel<0>), g2, dim3(256), 0, st, partial, nslab, 1, C, save_mean, nullptr,
                       1.0f / (float)rows, 0.f, 0.f);
    hipLaunchKernelGGL((col_partial_kernel<1>), g1, dim3(256), 0, st, x, ldx, nullptr, 0, nullptr, 0, save_mean,
                       nullptr, (long)rows, C, 1, 0, 0.f, partial);
    hipLaunchKernelGGL((col_final_kernel<1>), g2, dim3(256), 0, st, partial, nslab, 1, C, save_rstd, nullptr,
                       1.0f / (float)rows, 0.f, eps);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * C)), dim3(256), 0, st, x, ldx, (long)rows, C, scale,
                       offset, save_mean, save_rstd, act, alpha, y, ldy);
    return check_launch("bn_fwd");
}

extern "C" int dpig_bn_bwd(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy, int64_t rows,
                           int C, const float* scale, const float* save_mean, const float* save_rstd, int act,
                           float alpha, float* dx, int lddx, float* dscale, float* doffset, void* ws,
                           size_t ws_bytes, void* stream) {
    if (!dy || !x || !scale || !save_mean || !save_rstd || !dx || !dscale || !doffset)
        return fail(DPIG_EINVAL, "bn_bwd: null pointer");
    if (act != DPIG_ACT_NONE && !y) return fail(DPIG_EINVAL, "bn_bwd: activation output required");
    if (!ws || ws_bytes < dpig_bn_workspace_bytes(rows, C)) return fail(DPIG_ENOMEM, "bn_bwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nslab = slabs_for(rows);
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL((col_partial_kernel<2>), dim3(cdivi(C, 64), nslab), dim3(256), 0, st, dy, lddy, x, ldx, y, ldy,
                       save_mean, save_rstd, (long)rows, C, 1, act, alpha, partial);
    hipLaunchKernelGGL((col_final_kernel<0>), dim3(cdivi(C, 64)), dim3(256), 0, st, partial, nslab, 2, C, doffset,
                       dscale, 1.0f, 0.f, 0.f);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * C)), dim3(256), 0, st, dy, lddy, x, ldx, y, ldy,
                       (long)rows, C, scale, save_mean, save_rstd, dscale, doffset, act, alpha, 1.0f / (float)rows, dx, lddx);
    return check_launch("bn_bwd");
}

// ---- statistics left by the producing conv's epilogue (dpig_conv2d_fwd_stats) -------------------------------------------
// stats[tile][0][C] = sum over the tile's rows, stats[tile][1][C] = sum of squared deviations from the tile's own mean; tiles
// hold `rows_per_tile` rows (the last one the remainder).  Merged with the exact pairwise update (Chan, Golub, LeVeque):
// 16 lanes per column take every 16th tile in order, then the 16 partial results are merged in lane order -- fixed order,
// no atomics.  mean and rstd = 1/sqrt(biased variance + eps) are what dpig_bn_apply / dpig_bn_bwd expect.
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ stats, int tiles, long rows,
                                                                int rows_per_tile, int C, float eps,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    // 16 columns per block, 16 lanes per column: lane g takes tiles g, g + 16, ... in order (the loads of a tile do not
    // depend on the running merge, so they pipeline), then the 16 partial results are merged in lane order
    __shared__ float sn[16][16], sm[16][16], sq[16][16];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (c < C) {
        for (int t = g; t < tiles; t += 16) {
            const long left = rows - (long)t * rows_per_tile;
            const float nb = (float)(left < rows_per_tile ? left : rows_per_tile);
            const float mb = stats[((long)t * 2) * C + c] / nb, qb = stats[((long)t * 2 + 1) * C + c];
            const float nt = n + nb, delta = mb - mean;
            mean += delta * (nb / nt);
            m2 += qb + delta * delta * (n * nb / nt);
            n = nt;
        }
    }
    sn[g][cl] = n; sm[g][cl] = mean; sq[g][cl] = m2;
    __syncthreads();
    if (g == 0 && c < C) {
        for (int k = 1; k < 16; ++k) {
            const float nb = sn[k][cl];
            if (nb > 0.f) {
                const float nt = n + nb, delta = sm[k][cl] - mean;
                mean += delta * (nb / nt);
                m2 += sq[k][cl] + delta * delta * (n * nb / nt);
                n = nt;
            }
        }
        mean_out[c] = mean;
        rstd_out[c] = 1.0f / sqrtf(m2 / n + eps);
    }
}
extern "C" int dpig_bn_stats_finalize(const float* stats, int tiles, int64_t rows, int rows_per_tile, int C, float eps,
                                      float* mean, float* rstd, void* stream) {
    if (!stats || !mean || !rstd) return fail(DPIG_EINVAL, "bn_stats_finalize: null pointer");
    if (tiles <= 0 || rows <= 0 || rows_per_tile <= 0 || C <= 0 || (long)tiles * rows_per_tile < rows ||
        (long)(tiles - 1) * rows_per_tile >= rows)
        return fail(DPIG_EINVAL, "bn_stats_finalize: %d tiles of %d rows do not cover %ld rows", tiles, rows_per_tile, (long)rows);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(cdivi(C, 16)), dim3(256), 0, static_cast<hipStream_t>(stream), stats,
                       tiles, (long)rows, rows_per_tile, C, eps, mean, rstd);
    return check_launch("bn_stats_finalize");
}

// ---- staged form (synchronised BN over data-parallel ranks): the caller all-reduces the [C] vectors ----------
extern "C" int dpig_bn_sqdev(const float* x, int ldx, int64_t rows, int C, const float* mean, float* sq_out,
                             void* ws, size_t ws_bytes, void* stream) {
    if (!x || !mean || !sq_out) return fail(DPIG_EINVAL, "bn_sqdev: null pointer");
    if (!ws || ws_bytes < dpig_bn_workspace_bytes(rows, C)) return fail(DPIG_ENOMEM, "bn_sqdev: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nslab = slabs_for(rows);
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL((col_partial_kernel<1>), dim3(cdivi(C, 64), nslab), dim3(256), 0, st, x, ldx, nullptr, 0, nullptr, 0,
                       mean, nullptr, (long)rows, C, 1, 0, 0.f, partial);
    hipLaunchKernelGGL((col_final_kernel<0>), dim3(cdivi(C, 64)), dim3(256), 0, st, partial, nslab, 1, C, sq_out, nullptr,
                       1.0f, 0.f, 0.f);
    return check_launch("bn_sqdev");
}
extern "C" int dpig_bn_apply(const float* x, int ldx, int64_t rows, int C, const float* scale, const float* offset,
                             const float* mean, const float* rstd, int act, float alpha, float* y, int ldy,
                             void* stream) {
    if (!x || !scale || !offset || !mean || !rstd || !y) return fail(DPIG_EINVAL, "bn_apply: null pointer");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * C)), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx,
                       (long)rows, C, scale, offset, mean, rstd, act, alpha, y, ldy);
    return check_launch("bn_apply");
}
extern "C" int dpig_bn_bwd_sums(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy,
                                int64_t rows, int C, const float* mean, const float* rstd, int act, float alpha,
                                float* dscale, float* doffset, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !mean || !rstd || !dscale || !doffset) return fail(DPIG_EINVAL, "bn_bwd_sums: null pointer");
    if (act != DPIG_ACT_NONE && !y) return fail(DPIG_EINVAL, "bn_bwd_sums: activation output required");
    if (!ws || ws_bytes < dpig_bn_workspace_bytes(rows, C)) return fail(DPIG_ENOMEM, "bn_bwd_sums: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nslab = slabs_for(rows);
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL((col_partial_kernel<2>), dim3(cdivi(C, 64), nslab), dim3(256), 0, st, dy, lddy, x, ldx, y, ldy,
                       mean, rstd, (long)rows, C, 1, act, alpha, partial);
    hipLaunchKernelGGL((col_final_kernel<0>), dim3(cdivi(C, 64)), dim3(256), 0, st, partial, nslab, 2, C, doffset,
                       dscale, 1.0f, 0.f, 0.f);
    return check_launch("bn_bwd_sums");
}
extern "C" int dpig_bn_bwd_apply(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy,
                                 int64_t rows, int C, const float* scale, const float* mean, const float* rstd,
                                 const float* dscale, const float* doffset, int act, float alpha, float inv_count,
                                 float* dx, int lddx, void* stream) {
    if (!dy || !x || !scale || !mean || !rstd || !dscale || !doffset || !dx)
        return fail(DPIG_EINVAL, "bn_bwd_apply: null pointer");
    if (act != DPIG_ACT_NONE && !y) return fail(DPIG_EINVAL, "bn_bwd_apply: activation output required");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * C)), dim3(256), 0, static_cast<hipStream_t>(stream), dy,
                       lddy, x, ldx, y, ldy, (long)rows, C, scale, mean, rstd, dscale, doffset, act, alpha, inv_count,
                       dx, lddx);
    return check_launch("bn_bwd_apply");
}

// ---- fully connected layers ride on the conv kernels (a [M,K] matrix is an M x 1 x 1 x K image) -
static DpigConvDesc linear_desc(int M, int Kin, int Nout, int act, float alpha) {
    DpigConvDesc d = {};
    d.N = M; d.H = 1; d.W = 1; d.C = Kin; d.K = Nout; d.R = 1; d.S = 1; d.stride = 1;
    d.pad_t = 0; d.pad_l = 0; d.ldx = Kin; d.ldy = Nout; d.ldres = 0; d.ldmask = 0;
    d.act = act; d.alpha = alpha; d.upsample2x = 0; d.split_k = 0;
    return d;
}
extern "C" size_t dpig_linear_workspace_bytes(int M, int Kin, int Nout, int which) {
    DpigConvDesc d = linear_desc(M, Kin, Nout, 0, 0.f);
    return dpig_conv2d_workspace_bytes(&d, which);
}
extern "C" int dpig_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int Kin, int Nout,
                               int act, float alpha, void* ws, size_t ws_bytes, void* stream) {
    if (M <= 0 || Kin <= 0 || Nout <= 0) return fail(DPIG_EINVAL, "linear: non-positive dims");
    DpigConvDesc d = linear_desc(M, Kin, Nout, act, alpha);
    return dpig_conv2d_fwd(&d, x, w, bias, nullptr, y, nullptr, ws, ws_bytes, stream);
}
extern "C" int dpig_linear_dgrad(const float* dy, const float* w, float* dx, int M, int Kin, int Nout, void* ws,
                                 size_t ws_bytes, void* stream) {
    if (M <= 0 || Kin <= 0 || Nout <= 0) return fail(DPIG_EINVAL, "linear: non-positive dims");
    DpigConvDesc d = linear_desc(M, Kin, Nout, 0, 0.f);
    return dpig_conv2d_dgrad(&d, dy, w, nullptr, nullptr, dx, ws, ws_bytes, stream);
}
extern "C" int dpig_linear_wgrad(const float* x, const float* dy, float* dw, float beta, int M, int Kin, int Nout,
                                 void* ws, size_t ws_bytes, void* stream) {
    if (M <= 0 || Kin <= 0 || Nout <= 0) return fail(DPIG_EINVAL, "linear: non-positive dims");
    DpigConvDesc d = linear_desc(M, Kin, Nout, 0, 0.f);
    return dpig_conv2d_wgrad(&d, x, dy, dw, beta, nullptr, 0.f, ws, ws_bytes, stream);
}

template <typename T>
static int crop_resize_fwd_impl(const T* img, int N, int H, int W, int C, const float* boxes, const int32_t* box_ind,
                                int nbox, int ch, int cw, T* out, void* stream) {
    if (!img || !boxes || !box_ind || !out) return fail(DPIG_EINVAL, "crop_resize: null pointer");
    if (N <= 0 || nbox <= 0 || ch <= 0 || cw <= 0) return fail(DPIG_EINVAL, "crop_resize: empty");
    if (C % 4 == 0 && aligned16(img) && aligned16(out))
        hipLaunchKernelGGL((crop_resize_fwd_kernel<4, T>), dim3(grid_for((long)nbox * ch * cw * (C / 4))), dim3(256), 0,
                           static_cast<hipStream_t>(stream), img, H, W, C, boxes, box_ind, nbox, ch, cw, out);
    else
        hipLaunchKernelGGL((crop_resize_fwd_kernel<1, T>), dim3(grid_for((long)nbox * ch * cw * C)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), img, H, W, C, boxes, box_ind, nbox, ch, cw, out);
    return check_launch("crop_resize_fwd");
}
extern "C" int dpig_crop_resize_fwd(const float* img, int N, int H, int W, int C, const float* boxes,
                                    const int32_t* box_ind, int nbox, int ch, int cw, float* out, void* stream) {
    return crop_resize_fwd_impl<float>(img, N, H, W, C, boxes, box_ind, nbox, ch, cw, out, stream);
}
extern "C" int dpig_crop_resize_fwd_bf16(const uint16_t* img, int N, int H, int W, int C, const float* boxes,
                                         const int32_t* box_ind, int nbox, int ch, int cw, uint16_t* out, void* stream) {
    return crop_resize_fwd_impl<unsigned short>(img, N, H, W, C, boxes, box_ind, nbox, ch, cw, out, stream);
}
extern "C" size_t dpig_crop_resize_bwd_workspace_bytes(int W, int C, int nbox, int ch) {
    if (W <= 0 || C <= 0 || nbox <= 0 || ch <= 0) return 0;
    return (size_t)nbox * ch * W * C * sizeof(float);
}
template <typename T>
static int crop_resize_bwd_impl(const T* dout, int N, int H, int W, int C, const float* boxes, const int32_t* box_ind,
                                int nbox, int ch, int cw, T* dimg, void* ws, size_t ws_bytes, void* stream) {
    if (!dout || !boxes || !box_ind || !dimg) return fail(DPIG_EINVAL, "crop_resize: null pointer");
    if (N <= 0 || nbox <= 0 || ch <= 0 || cw <= 0) return fail(DPIG_EINVAL, "crop_resize: empty");
    if (!ws || ws_bytes < dpig_crop_resize_bwd_workspace_bytes(W, C, nbox, ch))
        return fail(DPIG_ENOMEM, "crop_resize_bwd workspace too small: have %zu", ws_bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* tmp = static_cast<float*>(ws);
    const bool v4 = (C % 4 == 0) && aligned16(dout) && aligned16(dimg) && aligned16(ws);
    if (cw > CROP_MAXC || W > 32767 || (long)nbox * ch > 0x7fffffffL || (long)N * H > 0x7fffffffL)
        return fail(DPIG_EINVAL, "crop_resize_bwd: crop wider than %d or image too large", CROP_MAXC);
    const int per_row = W * (v4 ? C / 4 : C);
    int ysplit = (per_row + 255) / 256;
    if (ysplit > 64) ysplit = 64;
    const dim3 gx((unsigned)((long)nbox * ch)), gy((unsigned)((long)N * H), (unsigned)ysplit);
    if (v4) {
        hipLaunchKernelGGL((crop_bwd_x_kernel<4, T>), gx, dim3(256), 0, st, dout, W, C, boxes, nbox, ch, cw, tmp);
        hipLaunchKernelGGL((crop_bwd_y_kernel<4, T>), gy, dim3(256), 0, st, tmp, N, H, W, C, boxes, box_ind, nbox, ch, dimg);
    } else {
        hipLaunchKernelGGL((crop_bwd_x_kernel<1, T>), gx, dim3(256), 0, st, dout, W, C, boxes, nbox, ch, cw, tmp);
        hipLaunchKernelGGL((crop_bwd_y_kernel<1, T>), gy, dim3(256), 0, st, tmp, N, H, W, C, boxes, box_ind, nbox, ch, dimg);
    }
    return check_launch("crop_resize_bwd");
}
extern "C" int dpig_crop_resize_bwd(const float* dout, int N, int H, int W, int C, const float* boxes,
                                    const int32_t* box_ind, int nbox, int ch, int cw, float* dimg, void* ws,
                                    size_t ws_bytes, void* stream) {
    return crop_resize_bwd_impl<float>(dout, N, H, W, C, boxes, box_ind, nbox, ch, cw, dimg, ws, ws_bytes, stream);
}
extern "C" int dpig_crop_resize_bwd_bf16(const uint16_t* dout, int N, int H, int W, int C, const float* boxes,
                                         const int32_t* box_ind, int nbox, int ch, int cw, uint16_t* dimg, void* ws,
                                         size_t ws_bytes, void* stream) {
    return crop_resize_bwd_impl<unsigned short>(dout, N, H, W, C, boxes, box_ind, nbox, ch, cw, dimg, ws, ws_bytes, stream);
}

extern "C" int dpig_pose_points(const float* rcv, int B, int K, int H, int W, int is_normalized, float* out, int ldo,
                                void* stream) {
    if (!rcv || !out) return fail(DPIG_EINVAL, "pose_points: null pointer");
    if (B <= 0 || K <= 0 || H <= 0 || W <= 0 || ldo < K) return fail(DPIG_EINVAL, "pose_points: bad shape");
    hipLaunchKernelGGL((pose_from_rcv_kernel<0>), dim3(grid_for((long)B * H * W * K)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), rcv, B, K, H, W, is_normalized, out, ldo);
    return check_launch("pose_points");
}
extern "C" int dpig_pose_rasterize(const float* rcv, int B, int K, int H, int W, int is_normalized, float* out,
                                   int ldo, void* stream) {
    if (!rcv || !out) return fail(DPIG_EINVAL, "pose_rasterize: null pointer");
    if (B <= 0 || K <= 0 || H <= 0 || W <= 0 || ldo < K) return fail(DPIG_EINVAL, "pose_rasterize: bad shape");
    hipLaunchKernelGGL((pose_from_rcv_kernel<1>), dim3(grid_for((long)B * H * W * K)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), rcv, B, K, H, W, is_normalized, out, ldo);
    return check_launch("pose_rasterize");
}
extern "C" int dpig_pose_inflate(const float* pose, int ldp, int B, int K, int H, int W, float* out, int ldo,
                                 void* stream) {
    if (!pose || !out) return fail(DPIG_EINVAL, "pose_inflate: null pointer");
    if (B <= 0 || K <= 0 || H <= 0 || W <= 0 || ldo < K || ldp < K) return fail(DPIG_EINVAL, "pose_inflate: bad shape");
    if (pose == out) return fail(DPIG_EINVAL, "pose_inflate: in-place not supported");
    hipLaunchKernelGGL(pose_inflate_kernel, dim3(grid_for((long)B * H * W * K)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), pose, ldp, B, K, H, W, out, ldo);
    return check_launch("pose_inflate");
}

extern "C" size_t dpig_ssim_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H < 7 || W < 7) return 0;
    return ((size_t)2 * B * H * W + (size_t)3 * B * kSsimBlocks) * sizeof(float);
}
extern "C" int dpig_ssim_gray_u8(const float* a, const float* b, int B, int H, int W, float* out, void* ws,
                                 size_t ws_bytes, void* stream) {
    if (!a || !b || !out) return fail(DPIG_EINVAL, "ssim: null pointer");
    if (B <= 0 || H < 7 || W < 7) return fail(DPIG_EINVAL, "ssim: images must be at least 7x7");
    if (!ws || ws_bytes < dpig_ssim_workspace_bytes(B, H, W)) return fail(DPIG_ENOMEM, "ssim: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* ga = static_cast<float*>(ws);
    float* gb = ga + (size_t)B * H * W;
    float* mm = gb + (size_t)B * H * W;
    float* part = mm + (size_t)2 * B * kSsimBlocks;
    hipLaunchKernelGGL(ssim_gray_kernel, dim3(kSsimBlocks, B), dim3(256), 0, st, a, b, H * W, ga, gb, mm);
    hipLaunchKernelGGL(ssim_map_kernel, dim3(kSsimBlocks, B), dim3(256), 0, st, ga, gb, H, W, mm, kSsimBlocks, part);
    hipLaunchKernelGGL(ssim_final_kernel, dim3((B + 63) / 64), dim3(64), 0, st, part, kSsimBlocks, B,
                       1.0f / (float)((H - 6) * (W - 6)), out);
    return check_launch("ssim");
}

extern "C" int dpig_gp_interpolate(const float* real, const float* fake, const float* alpha, int B, int64_t D,
                                   float* xhat, void* stream) {
    if (!real || !fake || !alpha || !xhat) return fail(DPIG_EINVAL, "gp_interpolate: null pointer");
    if (B <= 0 || D <= 0) return fail(DPIG_EINVAL, "gp_interpolate: empty");
    hipLaunchKernelGGL(gp_interpolate_kernel, dim3(grid_for((long)B * D)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       real, fake, alpha, (long)D, (long)B * D, xhat);
    return check_launch("gp_interpolate");
}
extern "C" int dpig_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, void* stream) {
    if (!x || !y) return fail(DPIG_EINVAL, "upsample2x: null pointer");
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for((long)N * H * W * C * 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, N, H, W, C, y);
    return check_launch("upsample2x_fwd");
}
extern "C" int dpig_upsample2x_bwd(const float* dy, int N, int H, int W, int C, float* dx, void* stream) {
    if (!dy || !dx) return fail(DPIG_EINVAL, "upsample2x: null pointer");
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((long)N * H * W * C)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dy, N, H, W, C, dx);
    return check_launch("upsample2x_bwd");
}

static float adam_corr(float b1, float b2, int step) {
    // lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t), t >= 1
    const double t = (double)step;
    return (float)(sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
}
extern "C" int dpig_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev,
                              float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || !lr_dev) return fail(DPIG_EINVAL, "adam: null pointer");
    if (step < 1) return fail(DPIG_EINVAL, "adam: step must be >= 1");
    if (!(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v))) return fail(DPIG_EALIGN, "adam: 16B alignment required");
    launch_adam(static_cast<hipStream_t>(stream), p, g, m, v, (long)n, lr_dev, beta1, beta2, eps, adam_corr(beta1, beta2, step), grad_scale,
                (const float*)nullptr);
    return check_launch("adam");
}

// Graph-replayable form: the step counter and the bias correction live in device memory
// (state_dev = {int32 t; float corr}); one tick kernel advances them, so a captured hipGraph replays
// the right lr_t every step (a host-side `step` argument would be frozen into the graph).
__global__ void adam_tick_kernel(int* __restrict__ state, float b1, float b2) {
    const int t = state[0] + 1;
    state[0] = t;
    reinterpret_cast<float*>(state)[1] = (float)(sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
}
extern "C" int dpig_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev,
                                  void* state_dev, float beta1, float beta2, float eps, float grad_scale,
                                  void* stream) {
    if (!p || !g || !m || !v || !lr_dev || !state_dev) return fail(DPIG_EINVAL, "adam: null pointer");
    if (!(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v))) return fail(DPIG_EALIGN, "adam: 16B alignment required");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, static_cast<int*>(state_dev), beta1, beta2);
    launch_adam(st, p, g, m, v, (long)n, lr_dev, beta1, beta2, eps, 1.0f, grad_scale, reinterpret_cast<const float*>(state_dev) + 1);
    return check_launch("adam_dev");
}
extern "C" int dpig_adam_multi(const void* const* ptrs_dev, const int64_t* sizes_dev, int ntensors, int64_t max_size,
                               const float* lr_dev, float beta1, float beta2, float eps, int step, float grad_scale,
                               void* stream) {
    if (!ptrs_dev || !sizes_dev || !lr_dev || ntensors <= 0) return fail(DPIG_EINVAL, "adam_multi: bad arguments");
    if (step < 1) return fail(DPIG_EINVAL, "adam: step must be >= 1");
    int bx = grid_for(max_size, 256, 512);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(bx, ntensors), dim3(256), 0, static_cast<hipStream_t>(stream), ptrs_dev,
                       reinterpret_cast<const long*>(sizes_dev), lr_dev, beta1, beta2, eps,
                       adam_corr(beta1, beta2, step), grad_scale);
    return check_launch("adam_multi");
}

extern "C" int dpig_rmsprop_step(float* p, const float* g, float* ms, float* mom, int64_t n, const float* lr_dev,
                                 float decay, float momentum, float eps, float grad_scale, void* stream) {
    if (!p || !g || !ms || !mom || !lr_dev || n <= 0) return fail(DPIG_EINVAL, "rmsprop: bad arguments");
    hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, ms, mom,
                       (long)n, lr_dev, decay, momentum, eps, grad_scale);
    return check_launch("rmsprop");
}
extern "C" int dpig_clip(float* p, int64_t n, float lo, float hi, void* stream) {
    if (!p || n <= 0 || !(lo <= hi)) return fail(DPIG_EINVAL, "clip: bad arguments");
    hipLaunchKernelGGL(clip_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, (long)n, lo, hi);
    return check_launch("clip");
}

extern "C" int dpig_sce_mean(const float* logits, int n, float label, float* out, float* dlogits, float scale,
                             void* stream) {
    if (!logits || !out || n <= 0) return fail(DPIG_EINVAL, "sce: bad arguments");
    hipLaunchKernelGGL(sce_mean_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), logits, n, label, out,
                       dlogits, scale);
    return check_launch("sce_mean");
}
extern "C" int dpig_logit_mean(const float* logits, int n, int squared, float target, float* out, float* dlogits,
                               float scale, void* stream) {
    if (!logits || !out || n <= 0) return fail(DPIG_EINVAL, "logit_mean: bad arguments");
    hipLaunchKernelGGL(logit_mean_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), logits, n, squared,
                       target, out, dlogits, scale);
    return check_launch("logit_mean");
}
extern "C" size_t dpig_l1_workspace_bytes(int64_t n) { return (size_t)grid_for(n, 1024, 1024) * sizeof(float); }
extern "C" int dpig_l1_mean(const float* a, const float* b, int64_t n, float* out, float* da, float scale, void* ws,
                            size_t ws_bytes, void* stream) {
    if (!a || !b || !out || n <= 0) return fail(DPIG_EINVAL, "l1: bad arguments");
    if (!ws || ws_bytes < dpig_l1_workspace_bytes(n)) return fail(DPIG_ENOMEM, "l1: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = grid_for(n, 1024, 1024);
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, st, a, b, (long)n, da, scale / (float)n, partial);
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, st, partial, nb, 1.0f / (float)n, out);
    return check_launch("l1_mean");
}
