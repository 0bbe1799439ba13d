// bf16-STORAGE filter gradient, large-tile form: the wgrad sibling of dpig_conv_bf16_q.hip for stride-1 SAME layers
// (models.py:398-400, 534-536, 564-566: the 3x3 convs of the residual blocks -- 87 % of the wgrad time).
//
//   dw[(tap, ci), co] = sum over output pixels m of  x[m + shift(tap), ci] * dy[m, co]          (fp32 result)
//
// GEMM rows are ITEMS = (filter tap, 128-channel block of ci), columns = co, the reduction runs over pixels (64 per
// k-tile) and is split across workgroups (fp32 partial slabs + bw_splitk_sum_kernel, as the 128 x 128 kernel).  8 waves,
// one workgroup per CU: bwq_kernel<2, 4> = 2 items x 256 co, bwq_kernel<4, 2> = 4 items x 128 co; every wave 128 ci x 64
// co = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16.
//  * Both operands are pixel-major in HBM while a fragment wants 8 consecutive k (= pixels) of one channel: tiles are
//    staged as they lie ([64 pixels][128 channels], 256-byte rows, LDS-DMA pieces of 4 pixel rows) and read with
//    ds_read_b64_tr_b16 (the transpose happens in the LDS read; granule g of pixel row r at slot g ^ ((r & 3) << 2), as
//    bw_kernel).  The reads are INLINE ASM: hipcc orders every transposing-read builtin behind pending LDS-DMA with
//    s_waitcnt vmcnt(0), which would drain the pipeline each phase; each read's destination is released to the compiler
//    by the waiting statement that names it (guide section 5.7, form ii).
//  * Schedule = dpig_conv_bf16_q.hip's: two LDS slots, two phases of 16 MFMAs per k-tile, raw barriers, the two wave
//    groups one barrier apart, counted vmcnt -- with the staging units cut along k instead of along the tile's rows (a
//    pixel row holds all 128 channels of its block): phase PA reads pixels 0..31 of the k-tile ({XA01, XB01}), phase PB
//    pixels 32..63 ({XA23, XB23});  PA(t) stages X23(t+1) into the other slot, PB(t) stages X01(t+2) into its own:
//      RAW  a unit is retired (counted vmcnt, N = the two units issued after it) one phase before it is read;
//      WAR  a unit is re-staged one phase after its last read, whose lgkmcnt(0) sits before that phase's first barrier.
//  * The MFMAs are fed dy-fragment first, so a lane owns one ci row and runs of 4 consecutive co: 16-byte fp32 stores.
//  * Bias gradient (tf.nn.bias_add's BiasAddGrad): the four waves of item 0 add up the dy fragments they hold anyway.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "dpig_bf16_common.h"
#include "dpig_conv_plan.h"

namespace dpig {
namespace bfk {

typedef __attribute__((address_space(3))) char wq_lds_char;
typedef int wq_v2i __attribute__((ext_vector_type(2)));
typedef int wq_v4i __attribute__((ext_vector_type(4)));

constexpr int WQ_ROW = 256;                       // bytes of one pixel row of a [64][128] bf16 tile
constexpr int WQ_SLOT = 64 * WQ_ROW;              // 16 KB: one slot of one 128-channel block

template <int WM, int WN>
struct WQGeom {
    static constexpr int NP = WN / 2;                              // 128-column blocks of co
    static constexpr int B_OFF = WM * 2 * WQ_SLOT;
    static constexpr int SMEM = B_OFF + NP * 2 * WQ_SLOT;
    static constexpr int NA = WM, NB = NP;                         // DMA pieces per wave per unit (32 pixel rows = 8 pieces per block)
    static constexpr int VMC = 2 * (NA + NB);
    static_assert(WM * WN == 8 && (WN == 2 || WN == 4) && SMEM <= 163840, "8 waves, LDS plan");
};

template <int N>
__device__ __forceinline__ void wq_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wq_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wq_dma16(__amdgpu_buffer_rsrc_t rs, int voff, int soff, wq_lds_char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, voff, soff, 0, 0);
}
template <int OFF>
__device__ __forceinline__ wq_v2i wq_tr_read(int addr) {          // 4 consecutive pixels of this lane's channel
    wq_v2i r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

template <int WM, int WN>
__global__ __launch_bounds__(512, 2) void bwq_kernel(const BWParams p) {
    using G = WQGeom<WM, WN>;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM];        // the ONLY LDS object
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int grp = wave >> 2;
    const int l31 = lane & 31, half = lane >> 5;

    const int nitems = p.ntaps * p.cblocks;
    const int ntl = p.q_mtiles * p.q_ntiles;
    const int wi = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.z), ntl * (int)gridDim.z);   // (tile, split), tile fastest
    const int split = wi / ntl;
    const int tile = wi - split * ntl;
    const int mt = tile / p.q_ntiles, nt = tile - mt * p.q_ntiles;
    const int co0 = nt * (WN * 64);
    const int kt_begin = split * p.tiles_per_split;
    const int nkt = min(p.ktiles, kt_begin + p.tiles_per_split) - kt_begin;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the x descriptor starts `padpix` pixels BEFORE the tensor: the scalar pixel offset m + tap shift + padpix is never negative
    const int padpix = p.pad_t * p.W + p.pad_l;
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X - (long)padpix * p.ldx, p.x_bytes + (unsigned)(padpix * p.ldx * 2));
    const __amdgpu_buffer_rsrc_t rsY = make_rsrc(p.DY, p.y_bytes);

    // ---- DMA roles: a piece = 4 pixel rows x 256 B of one 128-channel block; this wave fills pixel rows 4*wave .. +3 (unit
    // 01) and 32 + 4*wave .. +3 (unit 23) of EVERY block; lane -> (pixel row = lane >> 4, 16-byte granule = lane & 15, swizzled)
    const int prow = 4 * wave + (lane >> 4);
    const int gran = (lane & 15) ^ ((lane >> 4) << 2);
    int it_oy[WM], it_ox[WM];                      // (scalar) tap shift of item row j
    int x_voff[WM], x_soff[WM], y_voff[G::NB];
#pragma unroll
    for (int j = 0; j < WM; ++j) {
        const int item = mt * WM + j;
        const bool iok = item < nitems;
        const int tap = iok ? item / p.cblocks : 0;
        const int ci0 = (item - tap * p.cblocks) * 128;
        it_oy[j] = tap / p.S - p.pad_t;
        it_ox[j] = tap % p.S - p.pad_l;
        x_voff[j] = (iok && ci0 + gran * 8 < p.C) ? (prow * p.ldx + ci0 + gran * 8) * 2 : (int)OOB;
        x_soff[j] = ((kt_begin * TK + it_oy[j] * p.W + it_ox[j] + padpix) * p.ldx) * 2;
    }
#pragma unroll
    for (int j = 0; j < G::NB; ++j)
        y_voff[j] = (co0 + j * 128 + gran * 8 < p.K) ? (prow * p.ldy + co0 + j * 128 + gran * 8) * 2 : (int)OOB;
    // this lane's two pixels (units 01 / 23) of the k-tile being staged: (oy, ox), advanced by 64 pixels per k-tile
    int s_oy[2], s_ox[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int m = kt_begin * TK + 32 * u + prow;
        const int n = fast_div(m, p.mul_howo, p.shr_howo);
        const int rem = m - n * p.HoWo;
        s_oy[u] = fast_div(rem, p.mul_wo, p.shr_wo);
        s_ox[u] = rem - s_oy[u] * p.Wo;
    }
    int st_left = nkt, st_pix = kt_begin * TK, st_y = (kt_begin * TK * p.ldy) * 2, st_dead = 0;
    const int u32x = 32 * p.ldx * 2, u32y = 32 * p.ldy * 2;
    wq_lds_char* const L = (wq_lds_char*)smem;
    int dcur = (4 * wave) * WQ_ROW;                // byte offset in a block's CURRENT slot of this wave's piece (unit 01)
    int sdir = WQ_SLOT;                            // current slot -> other slot (sign flips per k-tile)
    auto issue = [&](int other, auto U) {          // unit U (0: pixels 0..31, 1: 32..63) of the cursor's k-tile
        constexpr int u = decltype(U)::value;
        const int d = (other ? dcur + sdir : dcur) + u * 32 * WQ_ROW;
        const bool rok = 32 * u + prow < p.Npix - st_pix;
#pragma unroll
        for (int j = 0; j < WM; ++j) {
            const bool ok = rok & ((unsigned)(s_oy[u] + it_oy[j]) < (unsigned)p.H) & ((unsigned)(s_ox[u] + it_ox[j]) < (unsigned)p.W);
            wq_dma16(rsX, (ok ? x_voff[j] : (int)OOB) | st_dead, x_soff[j] + u * u32x, L + (d + j * 2 * WQ_SLOT));
        }
#pragma unroll
        for (int j = 0; j < G::NB; ++j)
            wq_dma16(rsY, (rok ? y_voff[j] : (int)OOB) | st_dead, st_y + u * u32y, L + (G::B_OFF + d + j * 2 * WQ_SLOT));
    };
    auto advance = [&]() {                         // cursor -> next k-tile
        --st_left;
        st_pix += TK;
        st_y += TK * p.ldy * 2;
#pragma unroll
        for (int j = 0; j < WM; ++j) x_soff[j] += TK * p.ldx * 2;
        if (st_left <= 0) st_dead = (int)OOB;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            s_ox[u] += p.d64_ox;
            const bool c1 = s_ox[u] >= p.Wo;
            s_ox[u] -= c1 ? p.Wo : 0;
            s_oy[u] += p.d64_oy + (c1 ? 1 : 0);
            s_oy[u] -= (s_oy[u] >= p.Ho) ? p.Ho : 0;
        }
    };

    // ---- transposed fragment reads.  One ds_read_b64_tr_b16 serves a [4 pixels][16 channels] block per 16 lanes: lane q of
    // the group supplies the 8-byte address of pixel q / 4, channels 4 (q % 4) .. +3 and receives the 4 pixels of channel q;
    // two reads (pixels kb .. kb+3, kb+4 .. kb+7) make the 8 consecutive k of a 32x32x16 fragment.
    const int q = lane & 15, cgrp = (lane >> 4) & 1, pq = q >> 2;
    int pa[4], pb[2];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int ca = b * 32 + cgrp * 16 + (q & 3) * 4;
        pa[b] = wr * 2 * WQ_SLOT + (half * 8 + pq) * WQ_ROW + (((ca >> 3) ^ (pq << 2)) * 16) + (ca & 7) * 2;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int cb = (wc & 1) * 64 + b * 32 + cgrp * 16 + (q & 3) * 4;
        pb[b] = G::B_OFF + (wc >> 1) * 2 * WQ_SLOT + (half * 8 + pq) * WQ_ROW + (((cb >> 3) ^ (pq << 2)) * 16) + (cb & 7) * 2;
    }
    const int lbase = (int)(size_t)L;               // LDS byte address of the array (0: it is the only LDS object)
    const bool do_bias = (p.DB != nullptr) && (mt == 0) && (wr == 0);          // (wave-uniform)
    float bsum[2] = {0.f, 0.f};

    // One phase: the 24 transposed reads of 32 pixels (k-steps ks0, ks0 + 1), the staging of one unit, the waits, 16 MFMAs.
    auto phase = [&](auto KS0, auto stage) {
        constexpr int ks0 = decltype(KS0)::value;
        wq_v2i ra[4][2][2], rb[2][2][2];           // [block][k-step][pixels 0..3 / 4..7]
        auto reads = [&](auto KS) {
            constexpr int ks = decltype(KS)::value;
            constexpr int o = (ks0 + ks) * 16 * WQ_ROW;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                rb[b][ks][0] = wq_tr_read<o>(lbase + pb[b]);
                rb[b][ks][1] = wq_tr_read<o + 4 * WQ_ROW>(lbase + pb[b]);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                ra[b][ks][0] = wq_tr_read<o>(lbase + pa[b]);
                ra[b][ks][1] = wq_tr_read<o + 4 * WQ_ROW>(lbase + pa[b]);
            }
        };
        reads(std::integral_constant<int, 0>{});
        reads(std::integral_constant<int, 1>{});
        stage();
        wq_wait_vm<G::VMC>();
        // fragments home BEFORE the barrier (the WAR rule); each statement releases the destinations it names
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(rb[0][0][0]), "+v"(rb[0][0][1]), "+v"(rb[1][0][0]), "+v"(rb[1][0][1]), "+v"(rb[0][1][0]), "+v"(rb[0][1][1]),
                       "+v"(rb[1][1][0]), "+v"(rb[1][1][1])
                     :: "memory");
        asm volatile("" : "+v"(ra[0][0][0]), "+v"(ra[0][0][1]), "+v"(ra[1][0][0]), "+v"(ra[1][0][1]), "+v"(ra[2][0][0]), "+v"(ra[2][0][1]),
                          "+v"(ra[3][0][0]), "+v"(ra[3][0][1]));
        asm volatile("" : "+v"(ra[0][1][0]), "+v"(ra[0][1][1]), "+v"(ra[1][1][0]), "+v"(ra[1][1][1]), "+v"(ra[2][1][0]), "+v"(ra[2][1][1]),
                          "+v"(ra[3][1][0]), "+v"(ra[3][1][1]));
        wq_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fb[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const wq_v4i v = {rb[b][ks][0][0], rb[b][ks][0][1], rb[b][ks][1][0], rb[b][ks][1][1]};
                fb[b] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const wq_v4i v = {ra[mb][ks][0][0], ra[mb][ks][0][1], ra[mb][ks][1][0], ra[mb][ks][1][1]};
                const bf16x8 fa = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[nb], fa, acc[mb][nb], 0, 0, 0);
            }
            if (do_bias) {                        // this lane holds dy[8 pixels][co = nb*32 + l31]
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    float v8[8];
                    unpack8(__builtin_bit_cast(uint4, fb[nb]), v8);
                    bsum[nb] += ((v8[0] + v8[1]) + (v8[2] + v8[3])) + ((v8[4] + v8[5]) + (v8[6] + v8[7]));
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        wq_barrier();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- prologue: k-tile 0 complete + unit 01 of k-tile 1 in flight ---------------------------------------------------
    issue(0, I0{});
    issue(0, I1{});
    advance();
    issue(1, I0{});
    wq_wait_vm<G::NA + G::NB>();
    wq_barrier();
    if (grp == 1) wq_barrier();                    // stagger: this group runs one barrier behind
    for (int t = 0; t < nkt; ++t) {
        phase(I0{}, [&]() { issue(1, I1{}); });                       // PA: pixels 0..31; stage X23(t+1) -> other slot
        phase(I2{}, [&]() { advance(); issue(0, I0{}); });            // PB: pixels 32..63; stage X01(t+2) -> this slot
        dcur += sdir;
#pragma unroll
        for (int b = 0; b < 4; ++b) pa[b] += sdir;
#pragma unroll
        for (int b = 0; b < 2; ++b) pb[b] += sdir;
        sdir = -sdir;
    }
    if (grp == 0) wq_barrier();
    wq_wait_vm<0>();                               // zero fills issued past the last k-tile

    // ---- epilogue: accumulators hold D^T -- lane (l31, half): ci row mb*32 + l31, co runs 8q + 4*half + 0..3 -------------
    const long wsize = (long)p.wrows * p.K;
    float* dst = (p.nsplit > 1) ? p.partial + (long)split * wsize : p.DW;
    const float beta = (p.nsplit > 1) ? 0.f : p.beta;
    const int item = mt * WM + wr;
    if (item < nitems) {
        const int tap = item / p.cblocks;
        const int ci0 = (item - tap * p.cblocks) * 128;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int ci = ci0 + mb * 32 + l31;
            if (ci >= p.C) continue;
            float* row = dst + ((long)tap * p.C + ci) * p.K + co0 + wc * 64;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int c = nb * 32 + 8 * qq + 4 * half;
                    if (co0 + wc * 64 + c >= p.K) continue;
                    float4 v = make_float4(acc[mb][nb][4 * qq], acc[mb][nb][4 * qq + 1], acc[mb][nb][4 * qq + 2], acc[mb][nb][4 * qq + 3]);
                    float4* o = reinterpret_cast<float4*>(row + c);
                    if (beta != 0.f) {
                        const float4 old = *o;
                        v.x += beta * old.x; v.y += beta * old.y; v.z += beta * old.z; v.w += beta * old.w;
                    }
                    *o = v;
                }
        }
    }
    if (do_bias) {                                 // sum of the two half-waves' pixels; column co0 + wc*64 + nb*32 + l31
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float v = bsum[nb] + __shfl_xor(bsum[nb], 32, 64);
            const int co = co0 + wc * 64 + nb * 32 + l31;
            if (half == 0 && co < p.K) {
                if (p.nsplit > 1) p.bias_partial[(long)split * p.K + co] = v;
                else p.DB[co] = (p.beta_b != 0.f) ? p.beta_b * p.DB[co] + v : v;
            }
        }
    }
}

// ================================================================================================
// host side
// ================================================================================================
static int g_wq_mode = -1;       // 0 off, 1 automatic, 2 whenever the layer is legal (tests); environment DPIG_BF16_WQ
static int g_wq_variant = 0;

// Tile grid and split of the large-tile wgrad for (ntaps, C, K, pixels); returns the fraction of launched MFMA work that is real
// x the fraction of the chip's 256 slots the grid fills (0 when the layer is not legal for the kernel).
double bwq_plan(int ntaps, int C, int K, long npix, int variant, int forced_split, int* mtiles, int* ntiles, int* nsplit, int* tps) {
    const int wm = variant == 1 ? 2 : 4, bn = variant == 1 ? 256 : 128;
    const int nitems = ntaps * cdiv(C, 128);
    const int mt = cdiv(nitems, wm), nt = cdiv(K, bn);
    const int ktiles = cdiv(npix, TK);
    const int tiles = mt * nt;
    int s = forced_split > 0 ? forced_split : (tiles >= kNumCU ? 1 : kNumCU / tiles);
    const int smax = ktiles / 16 > 0 ? ktiles / 16 : 1;          // >= 16 k-tiles per workgroup: prologue / epilogue stay small
    if (s > smax) s = smax;
    const int per = cdiv(ktiles, s);
    s = cdiv(ktiles, per);
    *mtiles = mt; *ntiles = nt; *nsplit = s; *tps = per;
    const long wgs = (long)tiles * s;
    const long rounds = (wgs + kNumCU - 1) / kNumCU;
    const double real = ((double)ntaps * C * K) / ((double)mt * wm * 128 * nt * bn);
    return real * (double)wgs / (double)(rounds * kNumCU);
}

void wq_init() {
    if (g_wq_mode >= 0) return;
    const char* e = getenv("DPIG_BF16_WQ");
    g_wq_mode = e ? atoi(e) : 1;
    const char* v = getenv("DPIG_BF16_WQ_VARIANT");
    g_wq_variant = v ? atoi(v) : 0;
}

// Variant (1 / 2) the automatic rule picks for this stride-1 SAME layer, or 0: keep the 128 x 128 kernel.
int bwq_choose(int ntaps, int C, int K, long npix, int forced_split) {
    wq_init();
    if (!g_wq_mode || C % 8 || K % 8 || C < 64 || K < 64) return 0;
    int a, b, c, d;
    const double e1 = bwq_plan(ntaps, C, K, npix, 1, forced_split, &a, &b, &c, &d);
    const double e2 = bwq_plan(ntaps, C, K, npix, 2, forced_split, &a, &b, &c, &d);
    // Measured (profiles/r03_conv_bf16_tile_ab.txt): 2 items x 256 co is +8 ... +29 % over the 128 x 128 kernel wherever its grid
    // fills the chip (256-channel layers and up); 4 items x 128 co never beats it by more than 2 %: automatic mode uses variant 1 only.
    if (g_wq_mode == 1) return (g_wq_variant != 2 && e1 >= 0.85) ? 1 : 0;
    return g_wq_variant ? g_wq_variant : (e2 > e1 * 1.03 ? 2 : 1);
}

int bwq_try(BWParams& p, int variant, hipStream_t st) {
    if (variant == 1) hipLaunchKernelGGL((bwq_kernel<2, 4>), dim3(p.q_mtiles * p.q_ntiles, 1, p.nsplit), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((bwq_kernel<4, 2>), dim3(p.q_mtiles * p.q_ntiles, 1, p.nsplit), dim3(512), 0, st, p);
    const int rc = check_launch("bwq_kernel");
    return rc ? rc : 1;
}

}  // namespace bfk
}  // namespace dpig

extern "C" int dpig_conv_bf16_set_large_tile_wgrad(int mode, int variant) {
    if (mode < 0 || mode > 2 || variant < 0 || variant > 2) return dpig::fail(DPIG_EINVAL, "large-tile mode / variant out of range");
    dpig::bfk::wq_init();
    dpig::bfk::g_wq_mode = mode;
    dpig::bfk::g_wq_variant = variant;
    return DPIG_OK;
}
