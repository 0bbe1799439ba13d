// LayerNorm over (H, W, C) per sample (tflib/ops/layernorm.py:6-20: tf.nn.moments over [1,2,3] + tf.nn.batch_normalization with a
// per-channel scale / offset, eps 1e-5 inside the square root), its gradient, and the second-order piece the WGAN-GP penalty needs
// (trainer.py:222-236 differentiates the critic's backward pass once more); plus the penalty's own reduction (dpig_gp_penalty).
//
// Round 4 rewrite.  The round-1 kernels gave each SAMPLE one workgroup: at the critic's sizes (8-16 samples of 0.5-8 MB) that is
// 8-16 of 256 CUs pulling 2-3 passes each -- 213-610 us per call, one third of the DeepFashion wgan-gp step
// (profiles/r04a_df256_wgan_gp_bf16_kernel_stats.md).  Here a sample is cut into chunks of <= 8192 elements, one workgroup per
// (sample, chunk):
//   statistics : every chunk leaves (sum, centred sum of squares about ITS OWN mean); consumers merge the chunks of their sample with
//                the exact pairwise update in chunk order (no E[x^2] - E[x]^2 cancellation, deterministic, identical in every block);
//   other sums : per-chunk partial sums, added in chunk order by the consumer.
// HBM-bound, 16-byte accesses when the sample length and channel count allow (scalar otherwise), any element type of
// {fp32, bf16}: arithmetic is fp32 either way, bf16 tensors are read / written directly ('bf16' storage mode: no cvt passes).
#include <initializer_list>
#include "dpig_common.h"

namespace dpig {
namespace nrm {

constexpr int NT = 256;            // threads per workgroup
constexpr int CHUNK_MAX = 8192;    // elements per (sample, chunk) workgroup
constexpr int MAX_CHUNKS = 1024;

typedef unsigned short bf16_t;

// ---- element access: V consecutive elements as fp32 ----------------------------------------------------------------------------
template <int V> struct Vec { float v[V]; };
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {          // round-to-nearest-even (v_cvt_pk_bf16_f32)
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
template <int V> __device__ __forceinline__ void ldv(const float* p, long i, float (&v)[V]) {
    if (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p + i); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else {
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = p[i + e];
    }
}
template <int V> __device__ __forceinline__ void ldv(const bf16_t* p, long i, float (&v)[V]) {
    if (V == 8) {
        const uint4 u = *reinterpret_cast<const uint4*>(p + i);
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
        v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = bf2f(p[i + e]);
    }
}
template <int V> __device__ __forceinline__ void stv(float* p, long i, const float (&v)[V]) {
    if (V == 4) *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
        for (int e = 0; e < V; ++e) p[i + e] = v[e];
    }
}
template <int V> __device__ __forceinline__ void stv(bf16_t* p, long i, const float (&v)[V]) {
    if (V == 8) {
        uint4 u;
        u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        u.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16); u.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
        *reinterpret_cast<uint4*>(p + i) = u;
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) p[i + e] = f2bf(v[e]);
    }
}

// sum of NS values per thread over the workgroup (4 waves); the result is the same in every thread
template <int NS>
__device__ __forceinline__ void block_sums(float (&s)[NS], float (*red)[4]) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NS; ++k) s[k] = wave_sum(s[k]);
    __syncthreads();                                   // (red may still be read from a previous call)
    if (l == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) red[k][w] = s[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NS; ++k) s[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
}

struct Geo {          // a sample of L elements in nch chunks of CH (the last one shorter)
    long L;
    int CH, nch, C, cmask;       // cmask = C - 1 when C is a power of two, else 0
};
__device__ __forceinline__ int chan(const Geo& g, long i) { return g.cmask ? (int)(i & g.cmask) : (int)(i % g.C); }

// mean / biased variance of sample n from its chunks' (sum, M2): exact pairwise merge in chunk order.  fp32 with one reciprocal per
// chunk (every thread of every workgroup of the sample runs this loop: fp64 divisions here cost 35 us per launch at 64 chunks); all chunks
// but the last hold CH elements, so the running count is j * CH.
__device__ __forceinline__ void merge_stats(const float* __restrict__ part, const Geo& g, int n, float* mean, float* var) {
    const float* p = part + (long)n * g.nch * 2;
    float cnt = 0.f, mu = 0.f, m2 = 0.f;
    for (int j = 0; j < g.nch; ++j) {
        const long left = g.L - (long)j * g.CH;
        const float nb = (float)(left < g.CH ? left : g.CH);
        const float tot = cnt + nb, rtot = 1.0f / tot;
        const float mb = p[2 * j] * (1.0f / nb), d = mb - mu;
        mu += d * (nb * rtot);
        m2 += p[2 * j + 1] + d * d * (cnt * nb * rtot);
        cnt = tot;
    }
    *mean = mu;
    *var = m2 / cnt;
}
template <int NS>
__device__ __forceinline__ void merge_sums(const float* __restrict__ part, const Geo& g, int n, float (&s)[NS]) {
    const float* p = part + (long)n * g.nch * NS;
#pragma unroll
    for (int k = 0; k < NS; ++k) s[k] = 0.f;
    for (int j = 0; j < g.nch; ++j)
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] += p[j * NS + k];
}

// this workgroup's (sample, chunk): elements [lo_, hi_) of sample n_, which starts at element base_ of the tensor
#define NRM_CHUNK_DECL                                                                                      \
    const int n_ = blockIdx.x / g.nch, j_ = blockIdx.x - n_ * g.nch;                                         \
    const long base_ = (long)n_ * g.L;                                                                       \
    const long lo_ = (long)j_ * g.CH, hi_ = (lo_ + g.CH < g.L) ? lo_ + g.CH : g.L;
#define NRM_CHUNK_LOOP(i) for (long i = lo_ + (long)threadIdx.x * V; i < hi_; i += (long)NT * V)

// ---- forward ----------------------------------------------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(NT) void ln_stats_kernel(const T* __restrict__ x, Geo g, float* __restrict__ part) {
    NRM_CHUNK_DECL
    __shared__ float red[1][4];
    float s[1] = {0.f};
    NRM_CHUNK_LOOP(i) {
        float v[V];
        ldv<V>(x, base_ + i, v);
#pragma unroll
        for (int e = 0; e < V; ++e) s[0] += v[e];
    }
    block_sums<1>(s, red);
    const float sum = s[0];
    const float mu = sum / (float)(hi_ - lo_);
    s[0] = 0.f;
    NRM_CHUNK_LOOP(i) {                                 // second pass over <= 32 KB: L2 / L1 hits
        float v[V];
        ldv<V>(x, base_ + i, v);
#pragma unroll
        for (int e = 0; e < V; ++e) { const float d = v[e] - mu; s[0] += d * d; }
    }
    block_sums<1>(s, red);
    if (threadIdx.x == 0) { part[2 * (long)blockIdx.x] = sum; part[2 * (long)blockIdx.x + 1] = s[0]; }
}

template <typename T, int V>
__global__ __launch_bounds__(NT) void ln_apply_kernel(const T* __restrict__ x, Geo g, const float* __restrict__ part,
                                                      const float* __restrict__ scale, const float* __restrict__ offset, float eps,
                                                      int act, float alpha, T* __restrict__ y, float* __restrict__ save_mean,
                                                      float* __restrict__ save_rstd) {
    NRM_CHUNK_DECL
    float mean, var;
    merge_stats(part, g, blockIdx.x / g.nch, &mean, &var);
    const float rstd = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0 && blockIdx.x % g.nch == 0) { save_mean[blockIdx.x / g.nch] = mean; save_rstd[blockIdx.x / g.nch] = rstd; }
    NRM_CHUNK_LOOP(i) {
        float v[V], o[V];
        ldv<V>(x, base_ + i, v);
        const int c = chan(g, i);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int ce = (V == 1) ? c : c + e;        // (V > 1 only when C % V == 0: a vector never wraps the channel axis)
            o[e] = act_apply((v[e] - mean) * rstd * scale[ce] + offset[ce], act, alpha);
        }
        stv<V>(y, base_ + i, o);
    }
}

// ---- first-order backward: dx = r * (g - mean(g) - xh * mean(g * xh)),  g = dy * act'(y) * scale -----------------------------------
template <typename T, int V>
__global__ __launch_bounds__(NT) void ln_bwd_sums_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                         Geo g, const float* __restrict__ scale, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, int act, float alpha,
                                                         float* __restrict__ part) {
    NRM_CHUNK_DECL
    __shared__ float red[2][4];
    const float mu = mean[blockIdx.x / g.nch], rs = rstd[blockIdx.x / g.nch];
    float s[2] = {0.f, 0.f};
    NRM_CHUNK_LOOP(i) {
        float d[V], xv[V], yv[V];
        ldv<V>(dy, base_ + i, d);
        ldv<V>(x, base_ + i, xv);
        if (act != DPIG_ACT_NONE) ldv<V>(y, base_ + i, yv);
        const int c = chan(g, i);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float dz = d[e];
            if (act != DPIG_ACT_NONE) dz *= act_grad(yv[e], act, alpha);
            const float gg = dz * scale[(V == 1) ? c : c + e];
            s[0] += gg;
            s[1] += gg * (xv[e] - mu) * rs;
        }
    }
    block_sums<2>(s, red);
    if (threadIdx.x == 0) { part[2 * (long)blockIdx.x] = s[0]; part[2 * (long)blockIdx.x + 1] = s[1]; }
}

template <typename T, int V>
__global__ __launch_bounds__(NT) void ln_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                          Geo g, const float* __restrict__ scale, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, int act, float alpha,
                                                          const float* __restrict__ part, T* __restrict__ dx) {
    NRM_CHUNK_DECL
    const float mu = mean[blockIdx.x / g.nch], rs = rstd[blockIdx.x / g.nch];
    float S[2];
    merge_sums<2>(part, g, blockIdx.x / g.nch, S);
    const float S1 = S[0] / (float)g.L, S2 = S[1] / (float)g.L;
    NRM_CHUNK_LOOP(i) {
        float d[V], xv[V], yv[V], o[V];
        ldv<V>(dy, base_ + i, d);
        ldv<V>(x, base_ + i, xv);
        if (act != DPIG_ACT_NONE) ldv<V>(y, base_ + i, yv);
        const int c = chan(g, i);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float dz = d[e];
            if (act != DPIG_ACT_NONE) dz *= act_grad(yv[e], act, alpha);
            const float xh = (xv[e] - mu) * rs;
            o[e] = rs * (dz * scale[(V == 1) ? c : c + e] - S1 - xh * S2);
        }
        stv<V>(dx, base_ + i, o);
    }
}

// per-channel parameter gradients: doffset[c] = sum dz, dscale[c] = sum dz * xh over all samples and pixels.
// partial [slab][2][C]; a workgroup = 128 channels (V per thread) x row groups, rows strided by the slab count (fixed order).
template <typename T, int V>
__global__ __launch_bounds__(NT) void ln_param_partial_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd, long rows,
                                                              int C, int P, int act, float alpha, float* __restrict__ partial) {
    constexpr int LPR = 128 / V, RG = NT / LPR;            // lanes per row of 128 channels, row groups
    __shared__ float red[2][RG][128 + 4];
    const int cg = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    const int c0 = blockIdx.x * 128 + cg * V;
    float s0[V], s1[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
    if (c0 < C) {
        for (long r = (long)blockIdx.y * RG + rg; r < rows; r += (long)gridDim.y * RG) {
            const long n = r / P;
            const float mu = mean[n], rs = rstd[n];
            float d[V], xv[V], yv[V];
            ldv<V>(dy, r * C + c0, d);
            ldv<V>(x, r * C + c0, xv);
            if (act != DPIG_ACT_NONE) ldv<V>(y, r * C + c0, yv);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                if (V > 1 || c0 + e < C) {
                    const float dz = (act != DPIG_ACT_NONE) ? d[e] * act_grad(yv[e], act, alpha) : d[e];
                    s0[e] += dz;
                    s1[e] += dz * (xv[e] - mu) * rs;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { red[0][rg][cg * V + e] = s0[e]; red[1][rg][cg * V + e] = s1[e]; }
    __syncthreads();
    if (threadIdx.x < 128 && blockIdx.x * 128 + (int)threadIdx.x < C) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < RG; ++q) t += red[o][q][threadIdx.x];
            partial[((long)blockIdx.y * 2 + o) * C + blockIdx.x * 128 + threadIdx.x] = t;
        }
    }
}
// out_o[c] = sum over slabs of partial[slab][o][c], o = 0 -> out0, 1 -> out1.  16 channels x 16 slab groups per workgroup (a short
// dependent-load chain per thread), the group sums added in group order: deterministic.
__global__ __launch_bounds__(NT) void ln_param_final_kernel(const float* __restrict__ partial, int nslab, int C, float* __restrict__ out0,
                                                            float* __restrict__ out1) {
    __shared__ float red[2][16][16];
    const int cl = threadIdx.x & 15, gq = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        float sum = 0.f;
        if (c < C)
            for (int b = gq; b < nslab; b += 16) sum += partial[((long)b * 2 + o) * C + c];
        red[o][gq][cl] = sum;
    }
    __syncthreads();
    if (gq != 0 || c >= C) return;
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { t0 += red[0][q][cl]; t1 += red[1][q][cl]; }
    out0[c] = t0;
    out1[c] = t1;
}

// ---- second order: the backward of ln_bwd.  With g = dz * gamma, dx = r (g - mean(g) - xh mean(g xh)) and
// upstream u = dP/d(dx):
//   dP/dg_i  = r (u_i - mean(u) - xh_i mean(u xh))
//   dP/dxh_i = -r (mean(g xh) u_i + mean(u xh) g_i) =: q_i,   dP/dr = sum_i u_i (g_i - mean(g) - xh_i mean(g xh)) =: U1
//   dP/dx_j  = r (q_j - mean(q) - xh_j mean(q xh)) - U1 r^2 xh_j / L
// gs = dP/dg * dz is written (fp32) for the per-channel gamma gradient (a column sum over samples and pixels).
// Three stages: A -> (sum u, sum u xh, sum g, sum g xh);  B -> (U1, sum q, sum q xh) from A's means;  C applies.
template <typename T, int V>
__device__ __forceinline__ void ln2_load(const T* u, const T* dy, const T* x, const T* y, long idx, int act, float alpha,
                                         const float* scale, int c, float mu, float r, float (&uu)[V], float (&dz)[V], float (&gg)[V],
                                         float (&xh)[V], float (&ag)[V]) {
    float d[V], xv[V], yv[V];
    ldv<V>(u, idx, uu);
    ldv<V>(dy, idx, d);
    ldv<V>(x, idx, xv);
    if (act != DPIG_ACT_NONE) ldv<V>(y, idx, yv);
#pragma unroll
    for (int e = 0; e < V; ++e) {
        ag[e] = (act != DPIG_ACT_NONE) ? act_grad(yv[e], act, alpha) : 1.f;
        dz[e] = d[e] * ag[e];
        gg[e] = dz[e] * scale[(V == 1) ? c : c + e];
        xh[e] = (xv[e] - mu) * r;
    }
}

template <typename T, int V>
__global__ __launch_bounds__(NT) void ln_bwd2_a_kernel(const T* __restrict__ u, const T* __restrict__ dy, const T* __restrict__ x,
                                                       const T* __restrict__ y, Geo g, const float* __restrict__ scale,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                                       float alpha, float* __restrict__ partA) {
    NRM_CHUNK_DECL
    __shared__ float red[4][4];
    const float mu = mean[blockIdx.x / g.nch], r = rstd[blockIdx.x / g.nch];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    NRM_CHUNK_LOOP(i) {
        float uu[V], dz[V], gg[V], xh[V], ag[V];
        ln2_load<T, V>(u, dy, x, y, base_ + i, act, alpha, scale, chan(g, i), mu, r, uu, dz, gg, xh, ag);
#pragma unroll
        for (int e = 0; e < V; ++e) { s[0] += uu[e]; s[1] += uu[e] * xh[e]; s[2] += gg[e]; s[3] += gg[e] * xh[e]; }
    }
    block_sums<4>(s, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) partA[4 * (long)blockIdx.x + k] = s[k];
    }
}

template <typename T, int V>
__global__ __launch_bounds__(NT) void ln_bwd2_b_kernel(const T* __restrict__ u, const T* __restrict__ dy, const T* __restrict__ x,
                                                       const T* __restrict__ y, Geo g, const float* __restrict__ scale,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                                       float alpha, const float* __restrict__ partA, float* __restrict__ partB) {
    NRM_CHUNK_DECL
    __shared__ float red[3][4];
    const float mu = mean[blockIdx.x / g.nch], r = rstd[blockIdx.x / g.nch];
    const float invL = 1.0f / (float)g.L;
    float A[4];
    merge_sums<4>(partA, g, blockIdx.x / g.nch, A);
    const float cc = A[1] * invL, a = A[2] * invL, b = A[3] * invL;
    float s[3] = {0.f, 0.f, 0.f};
    NRM_CHUNK_LOOP(i) {
        float uu[V], dz[V], gg[V], xh[V], ag[V];
        ln2_load<T, V>(u, dy, x, y, base_ + i, act, alpha, scale, chan(g, i), mu, r, uu, dz, gg, xh, ag);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float q = -r * (b * uu[e] + cc * gg[e]);
            s[0] += uu[e] * (gg[e] - a - xh[e] * b);
            s[1] += q;
            s[2] += q * xh[e];
        }
    }
    block_sums<3>(s, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) partB[3 * (long)blockIdx.x + k] = s[k];
    }
}

template <typename T, int V>
__global__ __launch_bounds__(NT) void ln_bwd2_c_kernel(const T* __restrict__ u, const T* __restrict__ dy, const T* __restrict__ x,
                                                       const T* __restrict__ y, Geo g, const float* __restrict__ scale,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                                       float alpha, const float* __restrict__ partA, const float* __restrict__ partB,
                                                       T* __restrict__ d_dy, T* __restrict__ d_x, float* __restrict__ gs) {
    NRM_CHUNK_DECL
    const float mu = mean[blockIdx.x / g.nch], r = rstd[blockIdx.x / g.nch];
    const float invL = 1.0f / (float)g.L;
    float A[4], Bq[3];
    merge_sums<4>(partA, g, blockIdx.x / g.nch, A);
    merge_sums<3>(partB, g, blockIdx.x / g.nch, Bq);
    const float mu_u = A[0] * invL, cc = A[1] * invL, b = A[3] * invL;
    const float U1 = Bq[0], mq = Bq[1] * invL, mqx = Bq[2] * invL;
    NRM_CHUNK_LOOP(i) {
        float uu[V], dz[V], gg[V], xh[V], ag[V], o1[V], o2[V], o3[V];
        const int c = chan(g, i);
        ln2_load<T, V>(u, dy, x, y, base_ + i, act, alpha, scale, c, mu, r, uu, dz, gg, xh, ag);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float dg = r * (uu[e] - mu_u - xh[e] * cc);
            const float q = -r * (b * uu[e] + cc * gg[e]);
            o1[e] = dg * scale[(V == 1) ? c : c + e] * ag[e];
            o2[e] = r * (q - mq - xh[e] * mqx) - U1 * r * r * xh[e] * invL;
            o3[e] = dg * dz[e];
        }
        stv<V>(d_dy, base_ + i, o1);
        stv<V>(d_x, base_ + i, o2);
#pragma unroll
        for (int e = 0; e < V; ++e) gs[base_ + i + e] = o3[e];
    }
}

// ---- WGAN-GP penalty: slope_b = ||g_b||_2 from per-chunk sums of squares; dg_b = coef_b * g_b -----------------------------------------
template <int V>
__global__ __launch_bounds__(NT) void gp_sq_kernel(const float* __restrict__ gsrc, Geo g, float* __restrict__ part) {
    NRM_CHUNK_DECL
    __shared__ float red[1][4];
    float s[1] = {0.f};
    NRM_CHUNK_LOOP(i) {
        float v[V];
        ldv<V>(gsrc, base_ + i, v);
#pragma unroll
        for (int e = 0; e < V; ++e) s[0] += v[e] * v[e];
    }
    block_sums<1>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}
template <int V>
__global__ __launch_bounds__(NT) void gp_scale_kernel(const float* __restrict__ gsrc, Geo g, const float* __restrict__ part, int B,
                                                      float lambda, float* __restrict__ dg, float* __restrict__ slopes) {
    NRM_CHUNK_DECL
    float S[1];
    merge_sums<1>(part, g, blockIdx.x / g.nch, S);
    const float slope = sqrtf(S[0]);
    if (threadIdx.x == 0 && blockIdx.x % g.nch == 0) slopes[blockIdx.x / g.nch] = slope;
    const float coef = slope > 0.f ? lambda * 2.f * (slope - 1.f) / ((float)B * slope) : 0.f;
    NRM_CHUNK_LOOP(i) {
        float v[V];
        ldv<V>(gsrc, base_ + i, v);
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] *= coef;
        stv<V>(dg, base_ + i, v);
    }
}
__global__ void gp_final_kernel(const float* __restrict__ slopes, int B, float lambda, float* __restrict__ penalty) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) { const float d = slopes[b] - 1.f; s += d * d; }       // fixed order
    penalty[0] = lambda * s / (float)B;
}

// ---- training-mode batch norm on bf16 tensors (tflib/ops/batchnorm.py:30: tf.nn.fused_batch_norm; statistics over N, H, W per channel) ------
// The 'bf16' storage mode used to bracket the fp32 kernels of dpig_misc.hip with conversion passes (3 tensors in, 1 out per call: 36
// conversion launches per DeepFashion step).  These read / write the bf16 tensors directly: 8 channels (16 bytes) per thread, fp32
// arithmetic, the same two-pass statistics (mean, then centred squares) and the same fixed summation order idea (row slabs, then slabs).
// MODE 0: s0 = sum a;  1: s0 = sum (a - mean[c])^2;  2: dz = a * act'(y), s0 = sum dz, s1 = sum dz * (x - mean[c]) * rstd[c]
template <int MODE>
__global__ __launch_bounds__(NT) void bn16_partial_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ x, int ldx,
                                                          const bf16_t* __restrict__ y, int ldy, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, long rows, int C, int act, float alpha,
                                                          float* __restrict__ partial) {
    constexpr int NOUT = MODE == 2 ? 2 : 1;
    __shared__ float red[NOUT][16][128 + 4];
    const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 128 + cg * 8;
    float s0[8], s1[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; mu[e] = 0.f; rs[e] = 0.f; }
    if (c0 < C) {
        if (MODE >= 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) mu[e] = mean[c0 + e];
        }
        if (MODE == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) rs[e] = rstd[c0 + e];
        }
        for (long r = (long)blockIdx.y * 16 + rg; r < rows; r += (long)gridDim.y * 16) {
            float v[8];
            ldv<8>(a, r * lda + c0, v);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s0[e] += v[e];
            } else if (MODE == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[e] - mu[e]; s0[e] += d * d; }
            } else {
                float xv[8], yv[8];
                ldv<8>(x, r * ldx + c0, xv);
                if (act != DPIG_ACT_NONE) ldv<8>(y, r * ldy + c0, yv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dz = (act != DPIG_ACT_NONE) ? v[e] * act_grad(yv[e], act, alpha) : v[e];
                    s0[e] += dz;
                    s1[e] += dz * (xv[e] - mu[e]) * rs[e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[0][rg][cg * 8 + e] = s0[e];
        if (NOUT == 2) red[NOUT - 1][rg][cg * 8 + e] = s1[e];
    }
    __syncthreads();
    if (threadIdx.x < 128 && blockIdx.x * 128 + (int)threadIdx.x < C) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float t = 0.f;
#pragma unroll
            for (int g16 = 0; g16 < 16; ++g16) t += red[o][g16][threadIdx.x];
            partial[((long)blockIdx.y * NOUT + o) * C + blockIdx.x * 128 + threadIdx.x] = t;
        }
    }
}
// FIN 0: out_o[c] = scale * sum_slab partial[slab][o][c];  FIN 1: out_0[c] = 1 / sqrt(scale * sum + eps).  A block = 16 channels x 16 slab
// groups (group g takes slabs g, g + 16, ...; the group sums are added in group order: deterministic)
template <int FIN>
__global__ __launch_bounds__(NT) void bn16_final_kernel(const float* __restrict__ partial, int nslab, int nout, int C, float* __restrict__ out0,
                                                        float* __restrict__ out1, float scale, float eps) {
    __shared__ float red[2][16][16];
    const int cl = threadIdx.x & 15, gq = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    for (int o = 0; o < nout; ++o) {
        float sum = 0.f;
        if (c < C)
            for (int b = gq; b < nslab; b += 16) sum += partial[((long)b * nout + o) * C + c];
        red[o][gq][cl] = sum;
    }
    __syncthreads();
    if (gq != 0 || c >= C) return;
    for (int o = 0; o < nout; ++o) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[o][q][cl];
        float* out = o == 0 ? out0 : out1;
        if (FIN == 1) out[c] = 1.0f / sqrtf(t * scale + eps);
        else out[c] = t * scale;
    }
}
// y = act((x - mean) * rstd * scale + offset).  FIXED (C / 8 divides the block size, every model width): a thread keeps its 8 channels for
// all its rows, the per-channel constants live in registers; otherwise they are fetched per element group.
template <bool FIXED>
__global__ __launch_bounds__(NT) void bn16_apply_kernel(const bf16_t* __restrict__ x, int ldx, long rows, int C, const float* __restrict__ scale,
                                                        const float* __restrict__ offset, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, int act, float alpha, bf16_t* __restrict__ y, int ldy) {
    const int c8 = C >> 3;
    if (FIXED) {
        const int c0 = (threadIdx.x % c8) * 8, rpb = NT / c8;
        float ka[8], kb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ka[e] = rstd[c0 + e] * scale[c0 + e]; kb[e] = offset[c0 + e] - mean[c0 + e] * ka[e]; }
        for (long r = (long)blockIdx.x * rpb + threadIdx.x / c8; r < rows; r += (long)gridDim.x * rpb) {
            float v[8], o[8];
            ldv<8>(x, r * ldx + c0, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = act_apply(v[e] * ka[e] + kb[e], act, alpha);
            stv<8>(y, r * ldy + c0, o);
        }
        return;
    }
    const long total = rows * c8;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / c8;
        const int c0 = (int)(i - r * c8) * 8;
        float v[8], o[8];
        ldv<8>(x, r * ldx + c0, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ka = rstd[c0 + e] * scale[c0 + e];
            o[e] = act_apply(v[e] * ka + (offset[c0 + e] - mean[c0 + e] * ka), act, alpha);
        }
        stv<8>(y, r * ldy + c0, o);
    }
}
template <bool FIXED>
__global__ __launch_bounds__(NT) void bn16_bwd_apply_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx,
                                                            const bf16_t* __restrict__ y, int ldy, long rows, int C,
                                                            const float* __restrict__ scale, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dscale,
                                                            const float* __restrict__ doffset, int act, float alpha, float inv,
                                                            bf16_t* __restrict__ dx, int lddx) {
    const int c8 = C >> 3;
    auto one = [&](long r, int c0, const float (&mu)[8], const float (&rs)[8], const float (&k1)[8], const float (&k2)[8], const float (&k3)[8]) {
        float d[8], xv[8], yv[8], o[8];
        ldv<8>(dy, r * lddy + c0, d);
        ldv<8>(x, r * ldx + c0, xv);
        if (act != DPIG_ACT_NONE) ldv<8>(y, r * ldy + c0, yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dz = (act != DPIG_ACT_NONE) ? d[e] * act_grad(yv[e], act, alpha) : d[e];
            const float xh = (xv[e] - mu[e]) * rs[e];
            o[e] = k1[e] * (dz - k2[e] - xh * k3[e]);
        }
        stv<8>(dx, r * lddx + c0, o);
    };
    auto consts = [&](int c0, float (&mu)[8], float (&rs)[8], float (&k1)[8], float (&k2)[8], float (&k3)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mu[e] = mean[c0 + e]; rs[e] = rstd[c0 + e];
            k1[e] = scale[c0 + e] * rs[e]; k2[e] = doffset[c0 + e] * inv; k3[e] = dscale[c0 + e] * inv;
        }
    };
    float mu[8], rs[8], k1[8], k2[8], k3[8];
    if (FIXED) {
        const int c0 = (threadIdx.x % c8) * 8, rpb = NT / c8;
        consts(c0, mu, rs, k1, k2, k3);
        for (long r = (long)blockIdx.x * rpb + threadIdx.x / c8; r < rows; r += (long)gridDim.x * rpb) one(r, c0, mu, rs, k1, k2, k3);
        return;
    }
    const long total = rows * c8;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / c8;
        const int c0 = (int)(i - r * c8) * 8;
        consts(c0, mu, rs, k1, k2, k3);
        one(r, c0, mu, rs, k1, k2, k3);
    }
}
static int bn16_slabs(long rows) {
    long s = rows / 128;
    return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}
static int bn16_grid(long n) {
    long b = (n + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 8 * kNumCU ? 8 * kNumCU : b));
}
static int bn16_check(const void* a, int lda, int C) {
    if (C % 8 || lda % 8 || lda < C || !aligned16(a)) return fail(DPIG_EALIGN, "bf16 batch norm: 16-byte channel vectors (C, strides multiples of 8)");
    return DPIG_OK;
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static Geo make_geo(int N, long L, int C) {
    Geo g;
    g.L = L;
    g.C = C;
    g.cmask = (C & (C - 1)) == 0 ? C - 1 : 0;
    // enough workgroups to fill the chip (~4 per CU) without going below 2048 elements per chunk
    long want = (1024 + N - 1) / N;
    long ch = (L + want - 1) / want;
    if (ch < 2048) ch = 2048;
    if (ch > CHUNK_MAX) ch = CHUNK_MAX;
    ch = (ch + 2047) / 2048 * 2048;
    long nch = (L + ch - 1) / ch;
    if (nch > MAX_CHUNKS) { nch = MAX_CHUNKS; ch = ((L + nch - 1) / nch + 2047) / 2048 * 2048; nch = (L + ch - 1) / ch; }
    g.CH = (int)ch;
    g.nch = (int)nch;
    return g;
}
static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
static int param_slabs(long rows) {
    long s = rows / 64;
    return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}
template <typename T> constexpr int vec_width() { return sizeof(T) == 4 ? 4 : 8; }
template <typename T>
static bool vec_ok(long L, int C, std::initializer_list<const void*> ptrs) {
    constexpr int V = vec_width<T>();
    if (L % V || C % V) return false;
    for (const void* p : ptrs)
        if (p && !aligned16(p)) return false;
    return true;
}

template <typename T>
static int ln_fwd_impl(const T* x, int N, int P, int C, const float* scale, const float* offset, float eps, int act, float alpha, T* y,
                       float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!x || !scale || !offset || !y || !save_mean || !save_rstd) return fail(DPIG_EINVAL, "ln_fwd: null pointer");
    if (N <= 0 || P <= 0 || C <= 0) return fail(DPIG_EINVAL, "ln_fwd: empty");
    if (!ws || ws_bytes < dpig_ln_fwd_workspace_bytes(N, P, C)) return fail(DPIG_ENOMEM, "ln_fwd: workspace too small");
    const Geo g = make_geo(N, (long)P * C, C);
    float* part = static_cast<float*>(ws);
    constexpr int V = vec_width<T>();
    const dim3 grid(N * g.nch), block(NT);
    if (vec_ok<T>(g.L, C, {x, y})) {
        hipLaunchKernelGGL((ln_stats_kernel<T, V>), grid, block, 0, st, x, g, part);
        hipLaunchKernelGGL((ln_apply_kernel<T, V>), grid, block, 0, st, x, g, part, scale, offset, eps, act, alpha, y, save_mean, save_rstd);
    } else {
        hipLaunchKernelGGL((ln_stats_kernel<T, 1>), grid, block, 0, st, x, g, part);
        hipLaunchKernelGGL((ln_apply_kernel<T, 1>), grid, block, 0, st, x, g, part, scale, offset, eps, act, alpha, y, save_mean, save_rstd);
    }
    return check_launch("ln_fwd");
}

template <typename T>
static int ln_bwd_impl(const T* dy, const T* x, const T* y, int N, int P, int C, const float* scale, const float* save_mean,
                       const float* save_rstd, int act, float alpha, T* dx, float* dscale, float* doffset, void* ws, size_t ws_bytes,
                       hipStream_t st) {
    if (!dy || !x || !scale || !save_mean || !save_rstd || !dx) return fail(DPIG_EINVAL, "ln_bwd: null pointer");
    if ((dscale == nullptr) != (doffset == nullptr)) return fail(DPIG_EINVAL, "ln_bwd: dscale and doffset come together");
    if (act != DPIG_ACT_NONE && !y) return fail(DPIG_EINVAL, "ln_bwd: activation output required");
    if (N <= 0 || P <= 0 || C <= 0) return fail(DPIG_EINVAL, "ln_bwd: empty");
    if (!ws || ws_bytes < dpig_ln_workspace_bytes(N, P, C)) return fail(DPIG_ENOMEM, "ln_bwd: workspace too small");
    const Geo g = make_geo(N, (long)P * C, C);
    const long rows = (long)N * P;
    float* part = static_cast<float*>(ws);
    float* ppart = reinterpret_cast<float*>(static_cast<char*>(ws) + up256((size_t)N * g.nch * 2 * sizeof(float)));
    if (dscale) {
        const int nslab = param_slabs(rows);
        constexpr int VP = vec_width<T>();
        if (C % VP == 0 && aligned16(dy) && aligned16(x) && (act == DPIG_ACT_NONE || aligned16(y)))
            hipLaunchKernelGGL((ln_param_partial_kernel<T, VP>), dim3((C + 127) / 128, nslab), dim3(NT), 0, st, dy, x, y, save_mean, save_rstd,
                               rows, C, P, act, alpha, ppart);
        else
            hipLaunchKernelGGL((ln_param_partial_kernel<T, 1>), dim3((C + 127) / 128, nslab), dim3(NT), 0, st, dy, x, y, save_mean, save_rstd,
                               rows, C, P, act, alpha, ppart);
        hipLaunchKernelGGL(ln_param_final_kernel, dim3((C + 15) / 16), dim3(NT), 0, st, ppart, nslab, C, doffset, dscale);
    }
    constexpr int V = vec_width<T>();
    const dim3 grid(N * g.nch), block(NT);
    if (vec_ok<T>(g.L, C, {dy, x, y, dx})) {
        hipLaunchKernelGGL((ln_bwd_sums_kernel<T, V>), grid, block, 0, st, dy, x, y, g, scale, save_mean, save_rstd, act, alpha, part);
        hipLaunchKernelGGL((ln_bwd_apply_kernel<T, V>), grid, block, 0, st, dy, x, y, g, scale, save_mean, save_rstd, act, alpha, part, dx);
    } else {
        hipLaunchKernelGGL((ln_bwd_sums_kernel<T, 1>), grid, block, 0, st, dy, x, y, g, scale, save_mean, save_rstd, act, alpha, part);
        hipLaunchKernelGGL((ln_bwd_apply_kernel<T, 1>), grid, block, 0, st, dy, x, y, g, scale, save_mean, save_rstd, act, alpha, part, dx);
    }
    return check_launch("ln_bwd");
}

template <typename T>
static int ln_bwd2_impl(const T* u, const T* dy, const T* x, const T* y, int N, int P, int C, const float* scale, const float* save_mean,
                        const float* save_rstd, int act, float alpha, T* d_dy, T* d_x, float* d_scale, void* ws, size_t ws_bytes,
                        void* stream) {
    if (!u || !dy || !x || !scale || !save_mean || !save_rstd || !d_dy || !d_x || !d_scale)
        return fail(DPIG_EINVAL, "ln_bwd2: null pointer");
    if (act != DPIG_ACT_NONE && !y) return fail(DPIG_EINVAL, "ln_bwd2: activation output required");
    if (N <= 0 || P <= 0 || C <= 0) return fail(DPIG_EINVAL, "ln_bwd2: empty");
    if (!ws || ws_bytes < dpig_ln_bwd2_workspace_bytes(N, P, C)) return fail(DPIG_ENOMEM, "ln_bwd2: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Geo g = make_geo(N, (long)P * C, C);
    char* cur = static_cast<char*>(ws);
    float* gs = reinterpret_cast<float*>(cur); cur += up256((size_t)N * P * C * sizeof(float));
    float* partA = reinterpret_cast<float*>(cur); cur += up256((size_t)N * g.nch * 4 * sizeof(float));
    float* partB = reinterpret_cast<float*>(cur); cur += up256((size_t)N * g.nch * 3 * sizeof(float));
    constexpr int V = vec_width<T>();
    const dim3 grid(N * g.nch), block(NT);
    if (vec_ok<T>(g.L, C, {u, dy, x, y, d_dy, d_x}) && (sizeof(T) == 4 || C % 8 == 0)) {
        hipLaunchKernelGGL((ln_bwd2_a_kernel<T, V>), grid, block, 0, st, u, dy, x, y, g, scale, save_mean, save_rstd, act, alpha, partA);
        hipLaunchKernelGGL((ln_bwd2_b_kernel<T, V>), grid, block, 0, st, u, dy, x, y, g, scale, save_mean, save_rstd, act, alpha, partA, partB);
        hipLaunchKernelGGL((ln_bwd2_c_kernel<T, V>), grid, block, 0, st, u, dy, x, y, g, scale, save_mean, save_rstd, act, alpha, partA, partB,
                           d_dy, d_x, gs);
    } else {
        hipLaunchKernelGGL((ln_bwd2_a_kernel<T, 1>), grid, block, 0, st, u, dy, x, y, g, scale, save_mean, save_rstd, act, alpha, partA);
        hipLaunchKernelGGL((ln_bwd2_b_kernel<T, 1>), grid, block, 0, st, u, dy, x, y, g, scale, save_mean, save_rstd, act, alpha, partA, partB);
        hipLaunchKernelGGL((ln_bwd2_c_kernel<T, 1>), grid, block, 0, st, u, dy, x, y, g, scale, save_mean, save_rstd, act, alpha, partA, partB,
                           d_dy, d_x, gs);
    }
    int rc = check_launch("ln_bwd2");
    if (rc) return rc;
    const size_t off = (size_t)(cur - static_cast<char*>(ws));
    return dpig_colsum(gs, C, (int64_t)N * P, C, d_scale, 0.f, cur, ws_bytes - off, stream);
}

}  // namespace nrm
}  // namespace dpig

using namespace dpig;
using namespace dpig::nrm;

extern "C" size_t dpig_ln_fwd_workspace_bytes(int N, int P, int C) {
    if (N <= 0 || P <= 0 || C <= 0) return 0;
    return up256((size_t)N * make_geo(N, (long)P * C, C).nch * 2 * sizeof(float));
}
extern "C" int dpig_ln_fwd(const float* x, int N, int P, int C, const float* scale, const float* offset, float eps, int act, float alpha,
                           float* y, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream) {
    return ln_fwd_impl<float>(x, N, P, C, scale, offset, eps, act, alpha, y, save_mean, save_rstd, ws, ws_bytes, static_cast<hipStream_t>(stream));
}
extern "C" int dpig_ln_fwd_bf16(const uint16_t* x, int N, int P, int C, const float* scale, const float* offset, float eps, int act,
                                float alpha, uint16_t* y, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream) {
    return ln_fwd_impl<bf16_t>(x, N, P, C, scale, offset, eps, act, alpha, y, save_mean, save_rstd, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

extern "C" size_t dpig_ln_workspace_bytes(int N, int P, int C) {
    if (N <= 0 || P <= 0 || C <= 0) return 0;
    return up256((size_t)N * make_geo(N, (long)P * C, C).nch * 2 * sizeof(float)) + up256((size_t)param_slabs((long)N * P) * 2 * C * sizeof(float));
}
extern "C" int dpig_ln_bwd(const float* dy, const float* x, const float* y, int N, int P, int C, const float* scale, const float* save_mean,
                           const float* save_rstd, int act, float alpha, float* dx, float* dscale, float* doffset, void* ws, size_t ws_bytes,
                           void* stream) {
    return ln_bwd_impl<float>(dy, x, y, N, P, C, scale, save_mean, save_rstd, act, alpha, dx, dscale, doffset, ws, ws_bytes,
                              static_cast<hipStream_t>(stream));
}
extern "C" int dpig_ln_bwd_bf16(const uint16_t* dy, const uint16_t* x, const uint16_t* y, int N, int P, int C, const float* scale,
                                const float* save_mean, const float* save_rstd, int act, float alpha, uint16_t* dx, float* dscale,
                                float* doffset, void* ws, size_t ws_bytes, void* stream) {
    return ln_bwd_impl<bf16_t>(dy, x, y, N, P, C, scale, save_mean, save_rstd, act, alpha, dx, dscale, doffset, ws, ws_bytes,
                               static_cast<hipStream_t>(stream));
}

extern "C" size_t dpig_ln_bwd2_workspace_bytes(int N, int P, int C) {
    if (N <= 0 || P <= 0 || C <= 0) return 0;
    const int nch = make_geo(N, (long)P * C, C).nch;
    return up256((size_t)N * P * C * sizeof(float)) + up256((size_t)N * nch * 4 * sizeof(float)) + up256((size_t)N * nch * 3 * sizeof(float)) +
           dpig_colsum_workspace_bytes((int64_t)N * P, C);
}
extern "C" int dpig_ln_bwd2(const float* u, const float* dy, const float* x, const float* y, int N, int P, int C, const float* scale,
                            const float* save_mean, const float* save_rstd, int act, float alpha, float* d_dy, float* d_x, float* d_scale,
                            void* ws, size_t ws_bytes, void* stream) {
    return ln_bwd2_impl<float>(u, dy, x, y, N, P, C, scale, save_mean, save_rstd, act, alpha, d_dy, d_x, d_scale, ws, ws_bytes, stream);
}
extern "C" int dpig_ln_bwd2_bf16(const uint16_t* u, const uint16_t* dy, const uint16_t* x, const uint16_t* y, int N, int P, int C,
                                 const float* scale, const float* save_mean, const float* save_rstd, int act, float alpha, uint16_t* d_dy,
                                 uint16_t* d_x, float* d_scale, void* ws, size_t ws_bytes, void* stream) {
    return ln_bwd2_impl<bf16_t>(u, dy, x, y, N, P, C, scale, save_mean, save_rstd, act, alpha, d_dy, d_x, d_scale, ws, ws_bytes, stream);
}

extern "C" size_t dpig_gp_penalty_workspace_bytes(int B, int64_t D) {
    if (B <= 0 || D <= 0) return 0;
    return up256((size_t)B * make_geo(B, (long)D, 1).nch * sizeof(float));
}
extern "C" int dpig_gp_penalty(const float* g, int B, int64_t D, float lambda, float* penalty, float* dg, float* slopes, void* ws,
                               size_t ws_bytes, void* stream) {
    if (!g || !penalty || !dg || !slopes) return fail(DPIG_EINVAL, "gp_penalty: null pointer");
    if (B <= 0 || D <= 0) return fail(DPIG_EINVAL, "gp_penalty: empty");
    if (!ws || ws_bytes < dpig_gp_penalty_workspace_bytes(B, D)) return fail(DPIG_ENOMEM, "gp_penalty: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Geo geo = make_geo(B, (long)D, 1);
    float* part = static_cast<float*>(ws);
    const dim3 grid(B * geo.nch), block(NT);
    if (D % 4 == 0 && aligned16(g) && aligned16(dg)) {
        hipLaunchKernelGGL((gp_sq_kernel<4>), grid, block, 0, st, g, geo, part);
        hipLaunchKernelGGL((gp_scale_kernel<4>), grid, block, 0, st, g, geo, part, B, lambda, dg, slopes);
    } else {
        hipLaunchKernelGGL((gp_sq_kernel<1>), grid, block, 0, st, g, geo, part);
        hipLaunchKernelGGL((gp_scale_kernel<1>), grid, block, 0, st, g, geo, part, B, lambda, dg, slopes);
    }
    hipLaunchKernelGGL(gp_final_kernel, dim3(1), dim3(64), 0, st, slopes, B, lambda, penalty);
    return check_launch("gp_penalty");
}

// ---- batch norm on bf16 tensors --------------------------------------------------------------------------------------------------------
extern "C" size_t dpig_bn_bf16_workspace_bytes(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return up256((size_t)bn16_slabs(rows) * 2 * C * sizeof(float));
}
extern "C" int dpig_bn_apply_bf16(const uint16_t* x, int ldx, int64_t rows, int C, const float* scale, const float* offset, const float* mean,
                                  const float* rstd, int act, float alpha, uint16_t* y, int ldy, void* stream) {
    if (!x || !scale || !offset || !mean || !rstd || !y || rows <= 0 || C <= 0) return fail(DPIG_EINVAL, "bn_apply_bf16: bad arguments");
    int rc = bn16_check(x, ldx, C);
    if (!rc) rc = bn16_check(y, ldy, C);
    if (rc) return rc;
    const dim3 grid(bn16_grid(rows * (C / 8)));
    if (NT % (C / 8) == 0) hipLaunchKernelGGL((bn16_apply_kernel<true>), grid, dim3(NT), 0, static_cast<hipStream_t>(stream), x, ldx, (long)rows, C,
                                              scale, offset, mean, rstd, act, alpha, y, ldy);
    else hipLaunchKernelGGL((bn16_apply_kernel<false>), grid, dim3(NT), 0, static_cast<hipStream_t>(stream), x, ldx, (long)rows, C, scale, offset,
                            mean, rstd, act, alpha, y, ldy);
    return check_launch("bn_apply_bf16");
}
extern "C" int dpig_bn_fwd_bf16(const uint16_t* x, int ldx, int64_t rows, int C, const float* scale, const float* offset, float eps, int act,
                                float alpha, uint16_t* y, int ldy, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes,
                                void* stream) {
    if (!x || !scale || !offset || !y || !save_mean || !save_rstd || rows <= 0 || C <= 0) return fail(DPIG_EINVAL, "bn_fwd_bf16: bad arguments");
    int rc = bn16_check(x, ldx, C);
    if (rc) return rc;
    if (!ws || ws_bytes < dpig_bn_bf16_workspace_bytes(rows, C)) return fail(DPIG_ENOMEM, "bn_fwd_bf16: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nslab = bn16_slabs(rows);
    float* partial = static_cast<float*>(ws);
    const dim3 g1((C + 127) / 128, nslab), g2((C + 15) / 16);
    hipLaunchKernelGGL((bn16_partial_kernel<0>), g1, dim3(NT), 0, st, x, ldx, (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, 0,
                       (const float*)nullptr, (const float*)nullptr, (long)rows, C, 0, 0.f, partial);
    hipLaunchKernelGGL((bn16_final_kernel<0>), g2, dim3(NT), 0, st, partial, nslab, 1, C, save_mean, (float*)nullptr, 1.0f / (float)rows, 0.f);
    hipLaunchKernelGGL((bn16_partial_kernel<1>), g1, dim3(NT), 0, st, x, ldx, (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, 0,
                       (const float*)save_mean, (const float*)nullptr, (long)rows, C, 0, 0.f, partial);
    hipLaunchKernelGGL((bn16_final_kernel<1>), g2, dim3(NT), 0, st, partial, nslab, 1, C, save_rstd, (float*)nullptr, 1.0f / (float)rows, eps);
    rc = check_launch("bn_fwd_bf16");
    if (rc) return rc;
    return dpig_bn_apply_bf16(x, ldx, rows, C, scale, offset, save_mean, save_rstd, act, alpha, y, ldy, stream);
}
extern "C" int dpig_bn_bwd_bf16(const uint16_t* dy, int lddy, const uint16_t* x, int ldx, const uint16_t* y, int ldy, int64_t rows, int C,
                                const float* scale, const float* save_mean, const float* save_rstd, int act, float alpha, uint16_t* dx,
                                int lddx, float* dscale, float* doffset, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !scale || !save_mean || !save_rstd || !dx || !dscale || !doffset || rows <= 0 || C <= 0)
        return fail(DPIG_EINVAL, "bn_bwd_bf16: bad arguments");
    if (act != DPIG_ACT_NONE && !y) return fail(DPIG_EINVAL, "bn_bwd_bf16: activation output required");
    int rc = bn16_check(dy, lddy, C);
    if (!rc) rc = bn16_check(x, ldx, C);
    if (!rc && act != DPIG_ACT_NONE) rc = bn16_check(y, ldy, C);
    if (!rc) rc = bn16_check(dx, lddx, C);
    if (rc) return rc;
    if (!ws || ws_bytes < dpig_bn_bf16_workspace_bytes(rows, C)) return fail(DPIG_ENOMEM, "bn_bwd_bf16: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nslab = bn16_slabs(rows);
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL((bn16_partial_kernel<2>), dim3((C + 127) / 128, nslab), dim3(NT), 0, st, dy, lddy, x, ldx, y, ldy, save_mean, save_rstd,
                       (long)rows, C, act, alpha, partial);
    hipLaunchKernelGGL((bn16_final_kernel<0>), dim3((C + 15) / 16), dim3(NT), 0, st, partial, nslab, 2, C, doffset, dscale, 1.0f, 0.f);
    const dim3 grid(bn16_grid(rows * (C / 8)));
    if (NT % (C / 8) == 0) hipLaunchKernelGGL((bn16_bwd_apply_kernel<true>), grid, dim3(NT), 0, st, dy, lddy, x, ldx, y, ldy, (long)rows, C, scale,
                                              save_mean, save_rstd, dscale, doffset, act, alpha, 1.0f / (float)rows, dx, lddx);
    else hipLaunchKernelGGL((bn16_bwd_apply_kernel<false>), grid, dim3(NT), 0, st, dy, lddy, x, ldx, y, ldy, (long)rows, C, scale, save_mean,
                            save_rstd, dscale, doffset, act, alpha, 1.0f / (float)rows, dx, lddx);
    return check_launch("bn_bwd_bf16");
}
