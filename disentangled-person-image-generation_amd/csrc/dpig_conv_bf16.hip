// bf16-STORAGE convolution family for gfx950 (MI355X): activations, activation gradients and the filter shadows are
// bfloat16 in HBM, products run on v_mfma_f32_32x32x16_bf16 (dense peak 2.5 PFLOP/s), accumulation, bias, residual
// adds and every epilogue are fp32, filter gradients are written in fp32 (BASELINE configs 3-5: "bf16").
//
//   fwd    y  = act(conv(x, w) + bias + residual)     B = filter shadow  [tap][Cout][Cin]  (rows = GEMM N, k contiguous)
//   dgrad  dx = (conv^T(dy, w) + accum) * act'(mask)  B = filter shadow  [tap][Cin][Cout]  (the HWIO layout itself)
//   wgrad  dw = x^T (*) dy   (fp32 out)               both operands pixel-major; fragments by ds_read_b64_tr_b16
//
// What differs from the fp32 family (dpig_conv.hip), and why:
//  * operand tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds / buffer_load ... lds, 16 bytes per lane): no staging
//    registers, no ds_write pass, no conversions in the loop.  The DMA writes LDS lane-linearly (wave base + 16 * lane)
//    while the SOURCE address is per lane, so the implicit-GEMM gather (pixel + filter tap, zero halo) and the
//    bank-conflict swizzle are both expressed on the source side: lane l of a DMA instruction fills LDS row l / 8, slot
//    l % 8 and fetches the row's 16-byte chunk  slot ^ ((row >> 1) & 7).  A 128-byte row (64 bf16 of k) is fetched by
//    8 neighbouring lanes in permuted order: still one full line per row.
//  * fragments are ds_read_b128 (8 consecutive k per lane) at slot chunk ^ ((row >> 1) & 7): the 16 lanes of every
//    ds_read_b128 group hit 16 different 16-byte bank groups (conflict-free without padding, which LDS-DMA forbids).
//  * block tile 128 x 128 x 64, 4 waves (2 x 2) of 64 x 64, two 32 KB LDS stages, one barrier per k-tile, 2 workgroups
//    per CU; the DMA of k-tile t+1 is in flight under the 16 MFMAs per wave of k-tile t.
// Layer shapes the loops are not written for (fewer than 32 in/out channels, channel counts not multiples of 8) are not
// accepted here (dpig_conv2d_bf16_supported); the host converts those few thin layers and runs them on the fp32 kernels.
//
// Reference semantics as in dpig_conv.hip: tf.nn.conv2d 'SAME' (tflib/ops/conv2d.py:106-112), slim.conv2d
// (models.py:396-573) and the gradients TF autodiff derives for them (trainer.py:137-140).
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "dpig_common.h"
#include "dpig_conv_plan.h"
#include "dpig_thin.h"

#ifdef DPIG_TRACE   // dev aid (scripts/ubench/trace_bf16.py; never in the shipped build): s_memtime stamps of one lane per wave
__device__ unsigned long long dpig_bf_trace[256 * 4 * 160];
extern "C" int dpig_debug_bf_trace_read(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dpig_bf_trace), sizeof(unsigned long long) * n);
}
#define BF_STAMP(slot) do { if (trace_on && trace_n < 160) dpig_bf_trace[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 160 + trace_n++] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); } while (0)
#else
#define BF_STAMP(slot) do { } while (0)
#endif

#include "dpig_bf16_common.h"

namespace dpig {
namespace bfk {

// ---- epilogue shared by the k-loop variants: accumulators -> LDS (fp32) -> 16-byte row-contiguous global accesses ----
// `rowof(rl)` maps tile row rl (0..127) to the GEMM row it holds, or -1 when the tile row is padding: m0 + rl for the
// linear row tiles of bg_kernel, the NHW pixel index of the 2-D patch position for the halo tiles of bh_kernel.
// NB = 32-column blocks of a wave's sub-tile (2: four waves of 64 x 64; 1: eight waves of 64 x 32), RG = row groups of the
// workgroup's store pass (threads / 16).
template <int NB, int RG, typename RowOf>
__device__ __forceinline__ void bg_epilogue(const BGParams& p, char* smem, f32x16 (&acc)[2][NB], int m0, int n0, int split,
                                            int tid, int wrow, int wcol, int l31, int half, RowOf rowof) {
    constexpr int NIT = TM / RG;                 // rows a thread stores
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                Cs[(wrow + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDC + wcol + nb * 32 + l31] = acc[mb][nb][r];
    __syncthreads();

    const int c = (tid & 15) * 8;
    const int col = n0 + c;
    const int rl0 = tid >> 4;
    // ---- batch-norm partial statistics of this row tile (as dpig_conv.hip's: per column the sum and the sum of squared
    // deviations from the TILE's mean) of the values AS STORED: accumulator + bias rounded to bf16, so that bn_apply / bn_bwd, which
    // read the bf16 tensor, see exactly the mean and variance of what they normalise -- and the fallback that takes the
    // statistics from the stored tensor (split-K plans, multi-run batches) agrees with this path.  Only set by
    // dpig_conv2d_fwd_bf16_stats: un-split bg_kernel launch, rows m0 .. m0 + 127 are pixels m0 .. (every thread takes part).
    if (RG == 16 && p.stats) {                       // (the host launches the statistics epilogue on the four-wave kernel only)
        const bool cok = col < p.Ncols;
        const int nrows = min(TM, p.M - m0);
        float b8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[e] = (p.bias && cok) ? p.bias[col + e] : 0.f;
        float* red = Cs + TM * LDC;
        float sm[8], q8[8], tot[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sm[e] = 0.f; q8[e] = 0.f; tot[e] = 0.f; }
        for (int it = 0; it < 8; ++it) {
            const int rl = rl0 + 16 * it;
            if (rl < nrows) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sm[e] += (float)(__bf16)(Cs[rl * LDC + c + e] + b8[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rl0 * TN + c + e] = sm[e];
        __syncthreads();
        for (int g = 0; g < 16; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) tot[e] += red[g * TN + c + e];
        const float inv = 1.0f / (float)nrows;
        __syncthreads();
        for (int it = 0; it < 8; ++it) {
            const int rl = rl0 + 16 * it;
            if (rl < nrows) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dv = (float)(__bf16)(Cs[rl * LDC + c + e] + b8[e]) - tot[e] * inv; q8[e] += dv * dv; }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rl0 * TN + c + e] = q8[e];
        __syncthreads();
        if (rl0 == 0 && cok) {
            float* o = p.stats + ((long)(m0 / TM) * 2) * p.Ncols + col;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float qt = 0.f;
                for (int g = 0; g < 16; ++g) qt += red[g * TN + c + e];
                o[e] = tot[e];
                o[p.Ncols + e] = qt;
            }
        }
    }
    if (col >= p.Ncols) return;
    if (p.nsplit > 1) {
        // pre-epilogue partial sums [split][GEMM row][column]; the row of tile row rl comes from rowof (m0 + rl for the linear tiles, the
        // pixel index of the patch position for the halo tiles): bg_reduce_kernel is the second pass of both
        float* const pbase = p.partial + (long)split * p.M * p.Ncols + col;
#pragma unroll 4
        for (int it = 0; it < NIT; ++it) {
            const long r0 = rowof(rl0 + RG * it);
            if (r0 >= 0) {
                float* pp = pbase + r0 * p.Ncols;
                *reinterpret_cast<float4*>(pp) = *reinterpret_cast<const float4*>(&Cs[(rl0 + RG * it) * LDC + c]);
                *reinterpret_cast<float4*>(pp + 4) = *reinterpret_cast<const float4*>(&Cs[(rl0 + RG * it) * LDC + c + 4]);
            }
        }
        return;
    }
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    auto load_c = [&](int rl, float (&v)[8]) {
        const float4 v0 = *reinterpret_cast<const float4*>(&Cs[rl * LDC + c]);
        const float4 v1 = *reinterpret_cast<const float4*>(&Cs[rl * LDC + c + 4]);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    };
    if (p.identity_rows && !p.res_cls && !p.replicate) {
        // lean bodies for the flag combinations the models produce (one slope for all three activations, row pointers
        // advancing by 16 rows, no per-element switches): the co-resident workgroup is streaming MFMAs meanwhile
        const float slope = (p.act == DPIG_ACT_NONE) ? 1.f : ((p.act == DPIG_ACT_RELU) ? 0.f : p.alpha);
        auto run = [&](auto HAS_RES, auto RES_POST, auto HAS_MASK, auto HAS_D2) {
#pragma unroll 2
            for (int it = 0; it < NIT; ++it) {
                const long r0 = rowof(rl0 + RG * it);
                if (r0 < 0) continue;
                bf16_t* dp = p.D + r0 * p.ldd + col;
                const bf16_t* rp = HAS_RES ? p.res + r0 * p.ldres + col : nullptr;
                const bf16_t* mp = HAS_MASK ? p.mask + r0 * p.ldmask + col : nullptr;
                bf16_t* d2 = HAS_D2 ? p.D2 + r0 * p.ldd2 + col : nullptr;
                float v[8], rv[8];
                load_c(rl0 + RG * it, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bv[e];
                if (HAS_RES) unpack8(*reinterpret_cast<const uint4*>(rp), rv);
                if (HAS_RES && !RES_POST) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                if (HAS_MASK) {
                    float mv[8];
                    unpack8(*reinterpret_cast<const uint4*>(mp), mv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= (mv[e] > 0.f) ? 1.f : slope;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (v[e] > 0.f) ? v[e] : (v[e] * slope + 0.f);
                }
                if (HAS_D2) {
                    const uint4 o2 = pack8(v);
                    *reinterpret_cast<uint4*>(d2) = o2;
                    if (HAS_RES && RES_POST) unpack8(o2, v);
                }
                if (HAS_RES && RES_POST) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                *reinterpret_cast<uint4*>(dp) = pack8(v);
            }
        };
        using T = std::true_type; using F = std::false_type;
        const bool hr = p.res != nullptr, hm = p.mask != nullptr, h2 = p.D2 != nullptr, rpost = p.res_post != 0;
        if (!hr && !hm && !h2) { run(F{}, F{}, F{}, F{}); return; }                 // bias + activation
        if (hr && !rpost && !hm && !h2) { run(T{}, F{}, F{}, F{}); return; }        // + residual before the activation
        if (!hr && hm && !h2) { run(F{}, F{}, T{}, F{}); return; }                  // dgrad * activation mask
        if (hr && rpost && !hm && h2) { run(T{}, T{}, F{}, T{}); return; }          // res-block tail
    }
#pragma unroll 2
    for (int it = 0; it < NIT; ++it) {
        const int rl = rl0 + RG * it;
        const int row = rowof(rl);
        if (row < 0) continue;
        float v[8];
        load_c(rl, v);
        epi8(p, row, col, v, bv);
    }
}

// ------------------------------------------------------------------------------------------------
// gather-GEMM: D[M x Ncols] = gather(A)[M x K] * B^T, K = taps x channels
// NW = waves of the workgroup: 4 (sub-tiles 64 x 64) or 8 (64 x 32: each wave issues HALF the LDS-DMA pieces of a k-tile -- their issue
// cost, not their bytes, is what bounds this loop (profiles/r02_bf16_kloop_timeline_knockout.md) -- for 1.5x the fragment reads).  The
// k order of every output element is the same, so the two give identical bits.
// ST = LDS stages of the k-loop.  2 (rounds 2-5): the DMA of k-tile t + 1 flies under the MFMAs of k-tile t and is waited for in full
// before the barrier -- fine with two workgroups per CU, whose loops interleave.  4 (round 6, bg8d_kernel): THREE k-tiles in flight,
// counted waits (s_waitcnt vmcnt(2 P): only k-tile t's own pieces must have landed), raw barriers, one 128-KB workgroup per CU -- for
// the small layers whose split plans give the chip ONE round of <= 256 workgroups: alone on its CU a two-stage loop exposes the whole
// L2 -> LDS latency every k-tile (32 x 16 C384: 0.83 us per k-tile against 0.21 us of MFMAs).  Same k order per output element: same bits.
template <int NW, int ST = 2>
__device__ __forceinline__ void bg_body(const BGParams& p) {
    constexpr int NB = NW == 4 ? 2 : 1;          // 32-column blocks per wave
    constexpr int RPR = 8 * NW;                  // tile rows one DMA round of the workgroup fills
    constexpr int NJ = TM / RPR;                 // DMA rounds per operand tile
    constexpr int SMEM_ST = ST * STAGE_B > SMEM_BYTES ? ST * STAGE_B : SMEM_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_ST];        // the ONLY LDS object (two would make hipcc
                                                                       // drain the DMA queue before every ds_read)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = NW == 4 ? (wave >> 1) * 64 : (wave >> 2) * 64, wcol = NW == 4 ? (wave & 1) * 64 : (wave & 3) * 32;
    const int l31 = lane & 31, half = lane >> 5;
#ifdef DPIG_TRACE
    const bool trace_on = ((int)blockIdx.x < 256) && (blockIdx.z == 0) && (lane == 0);
    int trace_n = 0;
#endif

    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * TM, n0 = nt * TN;
    const int split = blockIdx.z;
    const int kt_begin = split * p.tiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.tiles_per_split);

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B, p.b_bytes);

    // ---- DMA roles: instruction j of this wave fills tile rows RPR j + 8*wave .. +7 (RPR = 32 / 64); lane -> (row, slot) ----
    // Per-lane work is kept OUT of the k-tile loop (the loop is issue-bound: 16 MFMAs of 32 cycles per k-tile leave ~500
    // cycles for everything else): the halo test and the tap's pixel shift are folded into a per-lane offset once per
    // TAP (a_voff, OOB when the tap falls outside the image for this row); the channel chunk of a k-tile is a SCALAR
    // offset of the DMA instruction, as is the whole filter-slab offset of the B operand.
    const int lrow = 8 * wave + (lane >> 3);                   // (+ RPR j)
    const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);          // the row's 16-byte chunk this lane fetches
    int a_base[NJ], a_iy0[NJ], a_ix0[NJ], a_voff[NJ], b_voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int m = m0 + lrow + RPR * j;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = fast_div(mm, p.mul_hrwr, p.shr_hrwr);
        const int rem = mm - n * p.HrWr;
        const int r = fast_div(rem, p.mul_wr, p.shr_wr);
        const int c = rem - r * p.Wr;
        a_iy0[j] = ok ? r * p.sr : -(1 << 24);                 // a row beyond M fails every bounds test
        a_ix0[j] = c * p.sr;
        a_base[j] = ((((n * p.Hs + r * p.sr) * p.Ws + c * p.sr) * p.lda) + chunk * 8) * 2;
        const int nn = n0 + lrow + RPR * j;
        b_voff[j] = (nn < p.Ncols) ? (nn * p.Cs + chunk * 8) * 2 : (int)OOB;
    }
    const bool ktail = (p.Cs & (TK - 1)) != 0;                 // the last channel chunk of a tap is partial (uniform)
    int cur_c0, cur_ta, cur_tb, tap_sB = 0;
    {
        const int tap = kt_begin / p.cchunks;
        cur_c0 = (kt_begin - tap * p.cchunks) * TK;
        cur_ta = tap / p.tap_nb;
        cur_tb = tap - cur_ta * p.tap_nb;
    }
    auto enter_tap = [&]() {
        const int t_oy = p.oy0 + cur_ta * p.oys, t_ox = p.ox0 + cur_tb * p.oxs;
        const int shift = ((t_oy * p.Ws + t_ox) * p.lda) * 2;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // bitwise & on purpose: && would become exec-mask branches
            const bool ok = ((unsigned)(a_iy0[j] + t_oy) < (unsigned)p.Hs) & ((unsigned)(a_ix0[j] + t_ox) < (unsigned)p.Ws);
            a_voff[j] = ok ? a_base[j] + shift : (int)OOB;
        }
        tap_sB = ((p.w0 + cur_ta * p.wa + cur_tb * p.wb) * p.Ncols * p.Cs) * 2;
    };
    enter_tap();
    auto issue_dummy = [&](int stage) {      // (ST > 2) past the last k-tile: the same DMA instructions with out-of-range offsets -- zero fill of a
        char* dst = smem + stage * STAGE_B + (8 * wave) * ROWB;           // free stage -- so that the counted waits stay uniform
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            dma16(rsA, (int)OOB, 0, dst + RPR * j * ROWB);
            dma16(rsB, (int)OOB, 0, dst + TILE_B + RPR * j * ROWB);
        }
    };
    auto issue = [&](int stage) {
        const int c0 = cur_c0;
        char* dst = smem + stage * STAGE_B + (8 * wave) * ROWB;
        if (ktail) {
            const bool kok = c0 + chunk * 8 < p.Cs;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                dma16(rsA, kok ? a_voff[j] : (int)OOB, c0 * 2, dst + RPR * j * ROWB);
                dma16(rsB, kok ? b_voff[j] : (int)OOB, tap_sB + c0 * 2, dst + TILE_B + RPR * j * ROWB);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                dma16(rsA, a_voff[j], c0 * 2, dst + RPR * j * ROWB);
                dma16(rsB, b_voff[j], tap_sB + c0 * 2, dst + TILE_B + RPR * j * ROWB);
            }
        }
        cur_c0 += TK;
        if (cur_c0 >= p.Cs) {                                  // next tap (uniform branch)
            cur_c0 = 0;
            if (++cur_tb == p.tap_nb) { cur_tb = 0; ++cur_ta; }
            enter_tap();
        }
    };

    // ---- fragment addresses: row = wave row/col + 32 mb + l31, 8 consecutive k = chunk 2 ks + half -----------------
    const int fsw = (l31 >> 1) & 7;
    const char* fa_base = smem + (wrow + l31) * ROWB;
    const char* fb_base = smem + TILE_B + (wcol + l31) * ROWB;
    int so[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) so[ks] = ((2 * ks + half) ^ fsw) * 16;

    bf16x8 fa[2][2], fb[2][NB];                    // [k-step parity][32-row / 32-col block]
    auto load_frag = [&](int stage, int ks, bf16x8 (&a)[2], bf16x8 (&b)[NB]) {
        const char* ab = fa_base + stage * STAGE_B;
        const char* bb = fb_base + stage * STAGE_B;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) a[mb] = *reinterpret_cast<const bf16x8*>(ab + mb * 32 * ROWB + so[ks]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = *reinterpret_cast<const bf16x8*>(bb + nb * 32 * ROWB + so[ks]);
    };
    // One k-tile: the first fragments are requested right behind the barrier, the DMA of the NEXT tile (address
    // arithmetic + 8 LDS-DMA instructions) is issued under their latency, the fragments of k-step ks+1 are read under the
    // MFMAs of ks.  The stage the DMA fills was last read before the barrier every wave has passed.
    auto ktile = [&](int stage, bool more) {
        BF_STAMP(1);
        load_frag(stage, 0, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(stage ^ 1);
        BF_STAMP(2);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) load_frag(stage, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][mb], fb[ks & 1][nb], acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        BF_STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BF_STAMP(4);
        __syncthreads();
    };

    BF_STAMP(0);
    if constexpr (ST == 2) {
        if (kt_begin < kt_end) {
            issue(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int kt = kt_begin;
            // two k-tiles per trip so that the LDS stage is a compile-time constant
            for (; kt + 1 < kt_end; kt += 2) {
                ktile(0, true);
                ktile(1, kt + 2 < kt_end);
            }
            if (kt < kt_end) ktile(0, false);
        }
    } else if (kt_begin < kt_end) {
        // ---- ST-stage ring: k-tiles t + 1 .. t + ST - 1 in flight under k-tile t ------------------------------------------------------
        constexpr int P = 2 * NJ;                          // DMA instructions of one k-tile per wave
        int nissued = kt_begin;                            // next k-tile to issue
        auto issue_next = [&](int stage) {
            if (nissued < kt_end) issue(stage); else issue_dummy(stage);
            ++nissued;
        };
#pragma unroll
        for (int i = 0; i < ST - 1; ++i) issue_next(i);
        // one k-tile: wait for ITS pieces only (the ST - 2 younger k-tiles stay in flight), raw barrier (every wave's pieces of k-tile t
        // are in LDS; every wave has finished reading k-tile t - 1, whose stage the next issue overwrites), issue k-tile t + ST - 1,
        // fragments + MFMAs of k-tile t.  __syncthreads() would drain the DMA queue (it waits for vmcnt(0)).
        auto ktile_deep = [&](auto STAGE) {
            constexpr int stage = decltype(STAGE)::value;
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(P * (ST - 2)) : "memory");
            load_frag(stage, 0, fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            issue_next((stage + ST - 1) % ST);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) load_frag(stage, ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][mb], fb[ks & 1][nb], acc[mb][nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int kt = kt_begin;
        for (; kt + ST <= kt_end; kt += ST) {
            ktile_deep(std::integral_constant<int, 0>{});
            ktile_deep(std::integral_constant<int, 1>{});
            ktile_deep(std::integral_constant<int, 2 % ST>{});
            ktile_deep(std::integral_constant<int, 3 % ST>{});
        }
        const int rem = kt_end - kt;                       // 0 .. ST - 1 k-tiles left: stages 0, 1, 2 in order
        if (rem > 0) ktile_deep(std::integral_constant<int, 0>{});
        if (rem > 1) ktile_deep(std::integral_constant<int, 1>{});
        if (rem > 2) ktile_deep(std::integral_constant<int, 2 % ST>{});
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");    // the dummy pieces too: the epilogue reuses the LDS
    }

    BF_STAMP(5);
    bg_epilogue<NB, 4 * NW>(p, smem, acc, m0, n0, split, tid, wrow, wcol, l31, half,
                            [&](int rl) { return (m0 + rl < p.M) ? m0 + rl : -1; });
    BF_STAMP(6);
}

__global__ __launch_bounds__(256, 2) void bg_kernel(const BGParams p) { bg_body<4>(p); }
__global__ __launch_bounds__(512, 2) void bg8_kernel(const BGParams p) { bg_body<8>(p); }
__global__ __launch_bounds__(512, 1) void bg8d_kernel(const BGParams p) { bg_body<8, 4>(p); }       // deep ring, one workgroup per CU

struct BGMulti { BGParams q[4]; };
__global__ __launch_bounds__(256, 2) void bg_multi_kernel(const BGMulti m) {
    const BGParams& p = m.q[blockIdx.y];
    if ((int)blockIdx.x >= p.mtiles * p.ntiles || (int)blockIdx.z >= p.nsplit) return;
    bg_body<4>(p);
}
__global__ __launch_bounds__(512, 2) void bg8_multi_kernel(const BGMulti m) {
    const BGParams& p = m.q[blockIdx.y];
    if ((int)blockIdx.x >= p.mtiles * p.ntiles || (int)blockIdx.z >= p.nsplit) return;
    bg_body<8>(p);
}

__global__ __launch_bounds__(512, 1) void bg8d_multi_kernel(const BGMulti m) {                      // the deep ring for a multi-problem launch
    const BGParams& p = m.q[blockIdx.y];
    if ((int)blockIdx.x >= p.mtiles * p.ntiles || (int)blockIdx.z >= p.nsplit) return;
    bg_body<8, 4>(p);
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-1 SAME convolutions (forward, and the stride-1 dgrad = the same window with flipped taps): ~90 % of the
// model's FLOPs.  bg_kernel re-fetches its A tile for each of the 9 taps -- 9 x 16 KB per 64-channel chunk -- and measures
// L2 -> LDS bound (the L2 serves ~60 % of its peak request rate while the matrix pipe is ~50 % busy).  Here the 128 output
// pixels of a workgroup are a 2-D patch (8 x 16 or 16 x 8) of ONE image and the input patch WITH ITS HALO ((8+2) x (16+2)
// = 180 pixels x 64 channels = 23 KB) is staged once per channel chunk; the 9 taps read shifted windows of it straight
// into MFMA A fragments (an address offset per tap), only the 16 KB filter tile streams per (tap, chunk): L2 -> LDS bytes
// per chunk 288 KB -> 167 KB, DMA instructions 288 -> 167.  The halo of the NEXT chunk is fetched one 1 KB piece per
// wave per tap under taps 0..5 of the current chunk, so the per-tile DMA load stays even (18.6 KB).
// LDS: 2 halo buffers (184 pixel rows of 128 B) + 2 filter tiles = 79872 B; two workgroups per CU.
// Bank conflicts: pixel (hy, hx) of the halo keeps its 16-byte chunk c at slot c ^ ((hx >> 1) + (TW / 2) hy) & 7: the 16
// lanes of a ds_read_b128 group (consecutive hx of one or two patch rows) then hit 16 different 16-byte bank groups
// for every tap shift.
template <int TWL, int NW>   // log2 of the patch width: 4 (8 rows x 16) or 3 (16 rows x 8); waves: 4 (64 x 64 sub-tiles) or 8 (64 x 32, as bg_body)
__device__ __forceinline__ void bh_body(const BGParams& p) {
    constexpr int NB = NW == 4 ? 2 : 1;          // 32-column blocks per wave
    constexpr int RPR = 8 * NW, NJ = TM / RPR;   // filter DMA: tile rows per round of the workgroup, rounds
    constexpr int NQ = 24 / NW;                  // halo DMA pieces per wave (23 pieces in all)
    constexpr int TW = 1 << TWL, TH = TM / TW, P = TW + 2, NPIX = (TH + 2) * P;       // 180 halo pixels
    constexpr int HROWS = 184, HALO_B = HROWS * ROWB;                                 // 23 DMA pieces of 8 pixels
    constexpr int HSMEM = 2 * HALO_B + 2 * TILE_B;                                    // 79872
    static_assert(NPIX <= HROWS && HSMEM >= SMEM_BYTES, "LDS plan");
    __shared__ __attribute__((aligned(16))) char smem[HSMEM];
    char* const bbuf = smem + 2 * HALO_B;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = NW == 4 ? (wave >> 1) * 64 : (wave >> 2) * 64, wcol = NW == 4 ? (wave & 1) * 64 : (wave & 3) * 32;
    const int l31 = lane & 31, half = lane >> 5;

    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int n0 = nt * TN;
    const int per_img = p.tiles_x * p.tiles_y;
    const int img = mt / per_img;
    const int trem = mt - img * per_img;
    const int tyi = trem / p.tiles_x;
    const int y0 = tyi * TH, x0 = (trem - tyi * p.tiles_x) * TW;

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B, p.b_bytes);
    const bool ktail = (p.Cs & (TK - 1)) != 0;

    // ---- halo DMA roles: piece i = wave + NW q (q < NQ, i < 23) covers halo pixels 8 i .. 8 i + 7; lane -> (pixel, slot)
    int h_voff[NQ], h_gran[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int hp = 8 * (wave + NW * q) + (lane >> 3);
        const int hy = hp / P, hx = hp - hy * P;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = (hp < NPIX) & ((unsigned)y < (unsigned)p.Hs) & ((unsigned)x < (unsigned)p.Ws);
        const int g = (lane & 7) ^ (((hx >> 1) + (TW / 2) * hy) & 7);
        h_gran[q] = g;
        h_voff[q] = ok ? ((((img * p.Hs + y) * p.Ws + x) * p.lda) + g * 8) * 2 : (int)OOB;
    }
    auto issue_halo = [&](int q, int buf, int c0) {            // piece (wave + NW q) of the chunk starting at channel c0
        if (wave + NW * q >= 23) return;                       // (wave-uniform)
        int v = h_voff[q];
        if (ktail) v = (c0 + h_gran[q] * 8 < p.Cs) ? v : (int)OOB;
        dma16(rsA, v, c0 * 2, smem + buf * HALO_B + (wave + NW * q) * 8 * ROWB);
    };
    // ---- filter DMA roles (as bg_kernel): instruction j fills tile rows RPR j + 8 wave .. +7 -----------------------
    const int lrow = 8 * wave + (lane >> 3);
    const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);
    int b_voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nn = n0 + lrow + RPR * j;
        b_voff[j] = (nn < p.Ncols) ? (nn * p.Cs + chunk * 8) * 2 : (int)OOB;
    }
    auto issue_b = [&](int stage, int tap, int c0) {
        const int ta = tap / 3, tb = tap - ta * 3;
        const int sB = ((p.w0 + ta * p.wa + tb * p.wb) * p.Ncols * p.Cs + c0) * 2;
        char* dst = bbuf + stage * TILE_B + (8 * wave) * ROWB;
        const bool kok = !ktail | (c0 + chunk * 8 < p.Cs);
#pragma unroll
        for (int j = 0; j < NJ; ++j) dma16(rsB, kok ? b_voff[j] : (int)OOB, sB, dst + RPR * j * ROWB);
    };

    // ---- fragment addressing ---------------------------------------------------------------------------------------
    int f_ty[2], f_tx[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int r = wrow + mb * 32 + l31;
        f_ty[mb] = r >> TWL;
        f_tx[mb] = r & (TW - 1);
    }
    const int fsw = (l31 >> 1) & 7;
    const char* fb_base = bbuf + (wcol + l31) * ROWB;
    int sob[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sob[ks] = ((2 * ks + half) ^ fsw) * 16;

    bf16x8 fa[2][2], fb[2][NB];
    // split plan (round 6): blockIdx.z owns the channel chunks [c_begin, nch) of a range of tiles_per_split chunks -- a layer with fewer
    // patches than the chip has slots (32 x 16 C384, 16 x 8 C512, DeepFashion's 32 x 32 C512 / 16 x 16 C768) keeps the halo staging
    // (167 instead of 288 KB of L2 -> LDS traffic per chunk) instead of falling back to the tap-major kernel for its split-K
    const int split = blockIdx.z;
    const int c_begin = split * p.tiles_per_split;
    const int nch = min(p.cchunks, c_begin + p.tiles_per_split);
    const int ktiles = 9 * nch;
    // one k-tile = (chunk c, tap): A fragments from the resident halo of chunk c shifted by the tap, B from bbuf[stage]
    auto ktile = [&](int kt, int stage) {
        const int c = kt / 9, tap = kt - c * 9;
        const int ta = tap / 3, tb = tap - ta * 3;
        const int dyy = 1 + p.oy0 + ta * p.oys, dxx = 1 + p.ox0 + tb * p.oxs;       // halo shift of this tap (0..2)
        const char* hb = smem + (c & 1) * HALO_B;
        const char* bb = fb_base + stage * TILE_B;
        const char* abase[2];
        int sw[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int hy = f_ty[mb] + dyy, hx = f_tx[mb] + dxx;
            abase[mb] = hb + (hy * P + hx) * ROWB;
            sw[mb] = ((hx >> 1) + (TW / 2) * hy) & 7;
        }
        auto load_frag = [&](int ks, bf16x8 (&a)[2], bf16x8 (&b)[NB]) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) a[mb] = *reinterpret_cast<const bf16x8*>(abase[mb] + (((2 * ks + half) ^ sw[mb]) << 4));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = *reinterpret_cast<const bf16x8*>(bb + nb * 32 * ROWB + sob[ks]);
        };
        load_frag(0, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < ktiles) {
            const int kn = kt + 1, cn = kn / 9;
            issue_b(stage ^ 1, kn - cn * 9, cn * TK);
        }
        if (tap < NQ && c + 1 < nch) issue_halo(tap, (c + 1) & 1, (c + 1) * TK);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) load_frag(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][mb], fb[ks & 1][nb], acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

#pragma unroll
    for (int q = 0; q < NQ; ++q) issue_halo(q, c_begin & 1, c_begin * TK);
    issue_b(0, 0, c_begin * TK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int kt = 9 * c_begin;
    for (; kt + 1 < ktiles; kt += 2) {
        ktile(kt, 0);
        ktile(kt + 1, 1);
    }
    if (kt < ktiles) ktile(kt, 0);

    bg_epilogue<NB, 4 * NW>(p, smem, acc, 0, n0, split, tid, wrow, wcol, l31, half, [&](int rl) {
        const int y = y0 + (rl >> TWL), x = x0 + (rl & (TW - 1));
        return (y < p.Hs && x < p.Ws) ? (img * p.Hs + y) * p.Ws + x : -1;
    });
}
template <int TWL> __global__ __launch_bounds__(256, 2) void bh_kernel(const BGParams p) { bh_body<TWL, 4>(p); }
template <int TWL> __global__ __launch_bounds__(512, 2) void bh8_kernel(const BGParams p) { bh_body<TWL, 8>(p); }

// split-K second pass: sum the fp32 partials in split order (deterministic), run the fused epilogue, write bf16
__device__ __forceinline__ void bg_reduce_body(const BGParams& p) {
    const int n8 = p.Ncols >> 3;
    const long total8 = (long)p.M * n8;
    const long total = (long)p.M * p.Ncols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
        const float4* p4 = reinterpret_cast<const float4*>(p.partial) + 2 * i;
        float4 a = p4[0], b = p4[1];
#pragma unroll 4
        for (int s = 1; s < p.nsplit; ++s) {
            const float4 t0 = p4[(long)s * (total >> 2)], t1 = p4[(long)s * (total >> 2) + 1];
            a.x += t0.x; a.y += t0.y; a.z += t0.z; a.w += t0.w;
            b.x += t1.x; b.y += t1.y; b.z += t1.z; b.w += t1.w;
        }
        const int row = (int)(i / n8);
        const int col = (int)(i - (long)row * n8) * 8;
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = p.bias ? p.bias[col + e] : 0.f;
        epi8(p, row, col, v, bv);
    }
}
__global__ __launch_bounds__(256) void bg_reduce_kernel(const BGParams p) { bg_reduce_body(p); }
__global__ __launch_bounds__(256) void bg_reduce_multi_kernel(const BGMulti m) {
    const BGParams& p = m.q[blockIdx.y];
    if (p.nsplit > 1) bg_reduce_body(p);
}

// ------------------------------------------------------------------------------------------------
// wgrad: dw[(tap, ci), co] = sum over output pixels m of x[src(m, tap), ci] * dy[m, co]   (fp32 result)
// GEMM rows = ci (128 per workgroup), columns = co (128), reduction = pixels (64 per k-tile).  Both operands are
// pixel-major in HBM ([pixel][channel]) while an MFMA lane wants 8 consecutive k (= pixels) of ONE channel: the tiles
// are staged as they are ([64 pixels][128 channels], one 256-byte row per pixel = 16 DMA granules of 8 channels) and
// the fragments are read with ds_read_b64_tr_b16, which hands each lane 4 consecutive pixels of its channel out of a
// [4 pixels][16 channels] block (the transpose happens inside the LDS read).  Granule g of pixel row r sits at slot
// g ^ ((r & 3) << 2): the 32 lanes served together (4 pixels x 4 granules) then cover all 64 banks once.
constexpr int WROWB = 256;                // bytes of one pixel row of a wgrad operand tile (128 channels)
constexpr int WTILE_B = TK * WROWB;       // 16 KB
constexpr int WSTAGE_B = 2 * WTILE_B;
constexpr int WSMEM_BYTES = 2 * WSTAGE_B; // 64 KB = the [128][128] fp32 staging of the epilogue

// NW = 4 waves of 64 x 64 or 8 waves of 64 x 32 (as bg_body: half the DMA instructions and halo tests per wave; same k order, same bits)
template <bool S1, int NW>   // S1: stride-1 SAME layer -- output pixel m pairs with input pixel m + const
__device__ __forceinline__ void bw_body(const BWParams& p) {
    constexpr int NB = NW == 4 ? 2 : 1;          // 32-column blocks per wave
    constexpr int PPR = 4 * NW;                  // pixel rows one DMA round of the workgroup fills
    constexpr int NJ = TK / PPR;                 // DMA rounds per operand tile
    __shared__ __attribute__((aligned(16))) char smem[WSMEM_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = NW == 4 ? (wave >> 1) * 64 : (wave >> 2) * 64, wcol = NW == 4 ? (wave & 1) * 64 : (wave & 3) * 32;
    const int l31 = lane & 31, half = lane >> 5;

    const int mtiles = p.ntaps * p.cblocks;
    const int ntl = mtiles * p.ntiles;
    // (tile, split) pairs remapped as one list, tile fastest: an XCD receives whole pixel ranges (see dpig_conv.hip)
    const int item = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.z), ntl * (int)gridDim.z);
    const int split = item / ntl;
    const int tile = item - split * ntl;
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int tap = mt / p.cblocks;
    const int ci0 = (mt - tap * p.cblocks) * TM;
    const int co0 = nt * TN;
    const int oyoff = tap / p.S - p.pad_t, oxoff = tap % p.S - p.pad_l;
    const int kt_begin = split * p.tiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.tiles_per_split);
    const bool do_bias = (p.DB != nullptr) && (mt == 0);

    f32x16 acc[2][NB];
    f32x16 accb[NB];                        // bias gradient: ones x dy on the matrix pipe (rows all equal)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { acc[0][nb][r] = 0.f; acc[1][nb][r] = 0.f; accb[nb][r] = 0.f; }
    // S1: the x descriptor starts `padpix` pixels BEFORE the tensor so that the scalar pixel offset m + tap shift + padpix is
    // never negative; lanes whose tap falls outside the image carry voffset = OOB and are never dereferenced
    const int padpix = S1 ? p.pad_t * p.W + p.pad_l : 0;
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X - (long)padpix * p.ldx, p.x_bytes + (unsigned)(padpix * p.ldx * 2));
    const __amdgpu_buffer_rsrc_t rsY = make_rsrc(p.DY, p.y_bytes);

    // ---- DMA roles: instruction j of this wave fills pixel rows PPR j + 4*wave .. +3 (1 KB; PPR = 16 / 32); lane -> (pixel, slot) ---
    // dy: per-lane constant offset + SCALAR k-tile offset; pixels >= Npix lie beyond the descriptor (zeros for free).
    // x (S1): likewise, plus the halo test on an incrementally advanced (oy, ox); other layers (stride 2, the upsampled
    // 1x1) compute the source pixel per row.
    const int prow = 4 * wave + (lane >> 4);                   // (+ PPR j); (prow & 3) == lane >> 4
    const int gran = (lane & 15) ^ ((lane >> 4) << 2);         // the 8-channel granule this lane fetches
    const bool cx_ok = ci0 + gran * 8 < p.C, cy_ok = co0 + gran * 8 < p.K;
    int s_oy[NJ], s_ox[NJ], s_n[NJ], y_voff[NJ], x_voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = prow + PPR * j;
        const int m = kt_begin * TK + r;
        const int n = fast_div(m, p.mul_howo, p.shr_howo);
        const int rem = m - n * p.HoWo;
        s_n[j] = n;
        s_oy[j] = fast_div(rem, p.mul_wo, p.shr_wo);
        s_ox[j] = rem - s_oy[j] * p.Wo;
        y_voff[j] = cy_ok ? (r * p.ldy + co0 + gran * 8) * 2 : (int)OOB;
        x_voff[j] = cx_ok ? (r * p.ldx + ci0 + gran * 8) * 2 : (int)OOB;
    }
    auto issue = [&](int kt, int stage) {
        char* dst = smem + stage * WSTAGE_B + (4 * wave) * WROWB;
        const int left = p.Npix - kt * TK;
        const int sy = (kt * TK * p.ldy) * 2;
        const int sx = S1 ? ((kt * TK + oyoff * p.W + oxoff + padpix) * p.ldx) * 2 : 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = prow + PPR * j;
            if (S1) {
                const bool ok = (r < left) & ((unsigned)(s_oy[j] + oyoff) < (unsigned)p.H) & ((unsigned)(s_ox[j] + oxoff) < (unsigned)p.W);
                dma16(rsX, ok ? x_voff[j] : (int)OOB, sx, dst + PPR * j * WROWB);
            } else {
                const int py = s_oy[j] * p.s + oyoff, px = s_ox[j] * p.s + oxoff;
                const int iy = py >> p.shift, ix = px >> p.shift;
                const bool ok = (r < left) & cx_ok & (py >= 0) & (px >= 0) & (iy < p.H) & (ix < p.W);
                const int xo = ((((s_n[j] * p.H + iy) * p.W + ix) * p.ldx) + ci0 + gran * 8) * 2;
                dma16(rsX, ok ? xo : (int)OOB, 0, dst + PPR * j * WROWB);
            }
            dma16(rsY, y_voff[j], sy, dst + WTILE_B + PPR * j * WROWB);
            // advance this row's pixel by one k-tile (64 pixels)
            s_ox[j] += p.d64_ox;
            const bool c1 = s_ox[j] >= p.Wo;
            s_ox[j] -= c1 ? p.Wo : 0;
            s_oy[j] += p.d64_oy + (c1 ? 1 : 0);
            const bool c2 = s_oy[j] >= p.Ho;
            s_oy[j] -= c2 ? p.Ho : 0;
            if (!S1) s_n[j] += p.d64_n + (c2 ? 1 : 0);
        }
    };

    // ---- transposed fragment reads.  One ds_read_b64_tr_b16 serves a [4 pixels][16 channels] block per 16 lanes: lane
    // q of the group supplies the 8-byte address of pixel q / 4, channels 4 (q % 4) .. +3 and receives the 4 pixels of
    // channel q.  Two reads (pixels kb .. kb+3 and kb+4 .. kb+7) make the 8 consecutive k of a 32x32x16 fragment.
    const int q = lane & 15;
    const int cgrp = (lane >> 4) & 1;                          // which 16 of the fragment's 32 channels
    const int pq = q >> 2;                                     // pixel (0..3) whose address this lane supplies
    // per-lane base addresses (operand tile offset, stage, k-step and the +4 pixel step are compile-time immediates)
    const char* pa[2];
    const char* pb[NB];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int ca = wrow + b * 32 + cgrp * 16 + (q & 3) * 4;       // first of this lane's 4 address channels
        // granule = chan / 8, swizzled by the pixel row: every pixel read below is 4 t + pq, so (pixel & 3) == pq
        pa[b] = smem + (half * 8 + pq) * WROWB + (((ca >> 3) ^ (pq << 2)) * 16) + (ca & 7) * 2;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int cb = wcol + b * 32 + cgrp * 16 + (q & 3) * 4;
        pb[b] = smem + WTILE_B + (half * 8 + pq) * WROWB + (((cb >> 3) ^ (pq << 2)) * 16) + (cb & 7) * 2;
    }
    auto tr_read = [&](const char* a) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    };
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    // One k-tile.  ALL 32 transposed reads of the tile are requested first and the DMA of the next tile is issued behind
    // them: hipcc treats the tr-read intrinsic as possibly aliasing a pending LDS-DMA and puts `s_waitcnt vmcnt(0)` in
    // front of the first tr read that follows a DMA in program order -- with the DMA issued last that wait is the one
    // the tile's own barrier needed anyway, and the DMA stays in flight under the 16 MFMAs.
    auto ktile = [&](int kt, int stage, bool more) {
        bf16x8 fa[4][2], fb[4][NB];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                       // k-step ks: pixels 16 ks + 8 half .. + 7 of the tile
            const int o = stage * WSTAGE_B + ks * 16 * WROWB;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const s16x4 a0 = tr_read(pa[b] + o), a1 = tr_read(pa[b] + o + 4 * WROWB);
                const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                fa[ks][b] = __builtin_bit_cast(bf16x8, av);
                if (b < NB) {
                    const s16x4 b0 = tr_read(pb[b] + o), b1 = tr_read(pb[b] + o + 4 * WROWB);
                    const s16x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    fb[ks][b] = __builtin_bit_cast(bf16x8, bv);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(kt + 1, stage ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mb], fb[ks][nb], acc[mb][nb], 0, 0, 0);
            if (do_bias) {                                     // workgroup-uniform
                const s16x8 one8 = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
                const bf16x8 ones = __builtin_bit_cast(bf16x8, one8);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) accb[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, fb[ks][nb], accb[nb], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    if (kt_begin < kt_end) {
        issue(kt_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int kt = kt_begin;
        for (; kt + 1 < kt_end; kt += 2) {
            ktile(kt, 0, true);
            ktile(kt + 1, 1, kt + 2 < kt_end);
        }
        if (kt < kt_end) ktile(kt, 0, false);
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------
    const long wsize = (long)p.wrows * p.K;
    float* Cs = reinterpret_cast<float*>(smem);        // [128][128] fp32 = the whole 64 KB
    if (do_bias && wrow == 0 && l31 < 32 && half == 0) {
        // row 0 of the ones x dy product: accumulator register 0 of lanes 0..31 (half 0) holds (row 0, column l31)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int co = co0 + wcol + nb * 32 + l31;
            if (co < p.K) {
                const float v = accb[nb][0];
                if (p.nsplit > 1) p.bias_partial[(long)split * p.K + co] = v;
                else p.DB[co] = (p.beta_b != 0.f) ? p.beta_b * p.DB[co] + v : v;
            }
        }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                Cs[(wrow + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * TN + wcol + nb * 32 + l31] = acc[mb][nb][r];
    __syncthreads();
    float* dst = (p.nsplit > 1) ? p.partial + (long)split * wsize : p.DW;
    const float beta = (p.nsplit > 1) ? 0.f : p.beta;
    const int c = (tid & 31) * 4;
    const int co = co0 + c;
    if (co < p.K) {
#pragma unroll 4
        for (int it = 0; it < 64 / NW; ++it) {
            const int rl = (tid >> 5) + 2 * NW * it;
            const int ci = ci0 + rl;
            if (ci >= p.C) continue;
            float4 v = *reinterpret_cast<const float4*>(&Cs[rl * TN + c]);
            float4* o = reinterpret_cast<float4*>(dst + ((long)tap * p.C + ci) * p.K + co);
            if (beta != 0.f) {
                const float4 old = *o;
                v.x += beta * old.x; v.y += beta * old.y; v.z += beta * old.z; v.w += beta * old.w;
            }
            *o = v;
        }
    }
}
template <bool S1> __global__ __launch_bounds__(256, 2) void bw_kernel(const BWParams p) { bw_body<S1, 4>(p); }
template <bool S1> __global__ __launch_bounds__(512, 2) void bw8_kernel(const BWParams p) { bw_body<S1, 8>(p); }

// ------------------------------------------------------------------------------------------------
// Split-bf16 filter gradient with BOTH operands from their split32 images (dpig_split32: [pixel][32-channel chunk][32 hi |
// 32 lo]) -- DPIG_COMPUTE_BF16X3's wgrad without the register path.  x and dy are the images the forward / dgrad launches of
// the same layer already consumed, so this kernel adds no pass over the tensors.  Structure = bw_kernel (LDS-DMA tiles as
// they lie in memory, pixel-major; ds_read_b64_tr_b16 hands each lane 4 consecutive pixels of its channel) with three MFMAs
// per fragment pair in the order of the in-loop split (x_hi dy_lo + x_lo dy_hi + x_hi dy_hi per 16-pixel step) and the same
// k-tile (32 pixels): the filter gradient is bit-identical to wgrad_kernel<..., PIPE 2>'s.  The bias gradient is ones x
// (dy_hi + dy_lo) on the matrix pipe (dy to 16 significand bits; the register path sums the fp32 dy).
// A pixel of an operand tile is 512 B = 4 chunks x [32 hi | 32 lo]; granule (16 B) g of pixel r is kept at g ^ swz(r & 3),
// swz = ((r & 1) << 1) | ((r & 2) << 2): the four pixels a transposed read touches then hit four different 32-byte windows.
struct BW3Params {
    const bf16_t* X32; const bf16_t* DY32; float* DW; float* partial;
    int Npix, Ho, Wo, HoWo;
    int H, W, C, K, shift, s, nchx, nchy;
    int ntaps, cblocks, ntiles;
    int ktiles, tiles_per_split, nsplit, wrows;
    float beta;
    int S, pad_t, pad_l;
    unsigned x_bytes, y_bytes;
    unsigned mul_howo, shr_howo, mul_wo, shr_wo;
    float* DB; float* bias_partial; float beta_b;
    int d32_oy, d32_ox, d32_n;
};
constexpr int W3_TK = 32;
constexpr int W3_PIX_B = 512;
constexpr int W3_TILE_B = W3_TK * W3_PIX_B;      // 16 KB
constexpr int W3_STAGE_B = 2 * W3_TILE_B;
constexpr int W3_SMEM = 2 * W3_STAGE_B;          // 64 KB = the [128][128] fp32 staging of the epilogue

template <bool S1>
__global__ __launch_bounds__(256, 2) void bw3_kernel(const BW3Params p) {
    __shared__ __attribute__((aligned(16))) char smem[W3_SMEM];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * 64;
    const int l31 = lane & 31, half = lane >> 5;

    const int mtiles = p.ntaps * p.cblocks;
    const int ntl = mtiles * p.ntiles;
    const int item = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.z), ntl * (int)gridDim.z);
    const int split = item / ntl;
    const int tile = item - split * ntl;
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int tap = mt / p.cblocks;
    const int ci0 = (mt - tap * p.cblocks) * TM;
    const int co0 = nt * TN;
    const int oyoff = tap / p.S - p.pad_t, oxoff = tap % p.S - p.pad_l;
    const int kt_begin = split * p.tiles_per_split;
    const int kt_end = min(p.ktiles, kt_begin + p.tiles_per_split);
    const bool do_bias = (p.DB != nullptr) && (mt == 0);

    f32x16 acc[2][2];
    f32x16 accb[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f;
        accb[0][r] = 0.f; accb[1][r] = 0.f;
    }
    const int xrow = p.nchx * 128, yrow = p.nchy * 128;        // bytes per pixel of the images
    const int padpix = S1 ? p.pad_t * p.W + p.pad_l : 0;
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X32 - (long)padpix * (xrow / 2), p.x_bytes + (unsigned)(padpix * xrow));
    const __amdgpu_buffer_rsrc_t rsY = make_rsrc(p.DY32, p.y_bytes);

    // ---- DMA roles: piece 4 wave + j = tile pixels 2 piece, 2 piece + 1 (1 KB); lane -> (pixel, physical granule) ----
    const int pp = lane >> 5, pgn = lane & 31;
    int s_oy[4], s_ox[4], s_n[4], y_voff[4], x_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (4 * wave + j) * 2 + pp;
        const int lg = pgn ^ (((r & 1) << 1) | ((r & 2) << 2));            // the logical granule this lane fetches
        const bool cx_ok = (ci0 >> 5) + (lg >> 3) < p.nchx, cy_ok = (co0 >> 5) + (lg >> 3) < p.nchy;
        const int m = kt_begin * W3_TK + r;
        const int n = fast_div(m, p.mul_howo, p.shr_howo);
        const int rem = m - n * p.HoWo;
        s_n[j] = n;
        s_oy[j] = fast_div(rem, p.mul_wo, p.shr_wo);
        s_ox[j] = rem - s_oy[j] * p.Wo;
        y_voff[j] = cy_ok ? r * yrow + (co0 >> 5) * 128 + lg * 16 : (int)OOB;
        x_voff[j] = cx_ok ? (S1 ? r * xrow : 0) + (ci0 >> 5) * 128 + lg * 16 : (int)OOB;
    }
    auto issue = [&](int kt, int stage) {
        char* dst = smem + stage * W3_STAGE_B + (4 * wave) * 1024;
        const int left = p.Npix - kt * W3_TK;
        const int sy = kt * W3_TK * yrow;
        const int sx = S1 ? (kt * W3_TK + oyoff * p.W + oxoff + padpix) * xrow : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (4 * wave + j) * 2 + pp;
            if (S1) {
                const bool ok = (r < left) & ((unsigned)(s_oy[j] + oyoff) < (unsigned)p.H) & ((unsigned)(s_ox[j] + oxoff) < (unsigned)p.W);
                dma16(rsX, ok ? x_voff[j] : (int)OOB, sx, dst + j * 1024);
            } else {
                const int py = s_oy[j] * p.s + oyoff, px = s_ox[j] * p.s + oxoff;
                const int iy = py >> p.shift, ix = px >> p.shift;
                const bool ok = (r < left) & (x_voff[j] != (int)OOB) & (py >= 0) & (px >= 0) & (iy < p.H) & (ix < p.W);
                const int xo = ((s_n[j] * p.H + iy) * p.W + ix) * xrow + x_voff[j];
                dma16(rsX, ok ? xo : (int)OOB, 0, dst + j * 1024);
            }
            dma16(rsY, (r < left) ? y_voff[j] : (int)OOB, sy, dst + W3_TILE_B + j * 1024);
            s_ox[j] += p.d32_ox;                              // advance this row's pixel by one k-tile (32 pixels)
            const bool c1 = s_ox[j] >= p.Wo;
            s_ox[j] -= c1 ? p.Wo : 0;
            s_oy[j] += p.d32_oy + (c1 ? 1 : 0);
            const bool c2 = s_oy[j] >= p.Ho;
            s_oy[j] -= c2 ? p.Ho : 0;
            if (!S1) s_n[j] += p.d32_n + (c2 ? 1 : 0);
        }
    };

    // ---- transposed fragment reads (see bw_kernel): lane q of a 16-lane group supplies the 8-byte address of pixel q / 4,
    // channels 4 (q % 4) .. +3 of the group's 16 channels and receives the 4 pixels of channel q --------------------------
    const int q = lane & 15;
    const int cgrp = (lane >> 4) & 1;
    const int pq = q >> 2;
    const int swz = ((pq & 1) << 1) | ((pq & 2) << 2);
    const char* pa[2][2];                    // [32-channel block = chunk][hi / lo]
    const char* pb[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ga = ((wrow >> 5) + b) * 8 + h * 4 + cgrp * 2 + ((q & 3) >> 1);
            const int gb = ((wcol >> 5) + b) * 8 + h * 4 + cgrp * 2 + ((q & 3) >> 1);
            pa[b][h] = smem + (half * 8 + pq) * W3_PIX_B + ((ga ^ swz) * 16) + (q & 1) * 8;
            pb[b][h] = smem + W3_TILE_B + (half * 8 + pq) * W3_PIX_B + ((gb ^ swz) * 16) + (q & 1) * 8;
        }
    auto tr_read = [&](const char* a) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    };
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    auto frag = [&](const char* a) -> bf16x8 {
        const s16x4 v0 = tr_read(a), v1 = tr_read(a + 4 * W3_PIX_B);
        const s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    // One k-tile: all 32 transposed reads first, the DMA of the next tile behind them (hipcc puts a vmcnt(0) in front of the
    // first tr read that follows a DMA), then the 24 MFMAs.
    auto ktile = [&](int kt, int stage, bool more) {
        bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];         // [k-step][block]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int o = stage * W3_STAGE_B + ks * 16 * W3_PIX_B;
                ah[ks][b] = frag(pa[b][0] + o); al[ks][b] = frag(pa[b][1] + o);
                bh[ks][b] = frag(pb[b][0] + o); bl[ks][b] = frag(pb[b][1] + o);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(kt + 1, stage ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][mb], bl[ks][nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][mb], bh[ks][nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][mb], bh[ks][nb], acc[mb][nb], 0, 0, 0);
            if (do_bias) {                                     // workgroup-uniform
                const s16x8 one8 = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
                const bf16x8 ones = __builtin_bit_cast(bf16x8, one8);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    accb[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bl[ks][nb], accb[nb], 0, 0, 0);
                    accb[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bh[ks][nb], accb[nb], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    if (kt_begin < kt_end) {
        issue(kt_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int kt = kt_begin;
        for (; kt + 1 < kt_end; kt += 2) {
            ktile(kt, 0, true);
            ktile(kt + 1, 1, kt + 2 < kt_end);
        }
        if (kt < kt_end) ktile(kt, 0, false);
    }

    // ---- epilogue (as bw_kernel) ------------------------------------------------------------------------------------
    const long wsize = (long)p.wrows * p.K;
    float* Cs = reinterpret_cast<float*>(smem);
    if (do_bias && wrow == 0 && half == 0) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int co = co0 + wcol + nb * 32 + l31;
            if (co < p.K) {
                const float v = accb[nb][0];
                if (p.nsplit > 1) p.bias_partial[(long)split * p.K + co] = v;
                else p.DB[co] = (p.beta_b != 0.f) ? p.beta_b * p.DB[co] + v : v;
            }
        }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                Cs[(wrow + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * TN + wcol + nb * 32 + l31] = acc[mb][nb][r];
    __syncthreads();
    float* dst = (p.nsplit > 1) ? p.partial + (long)split * wsize : p.DW;
    const float beta = (p.nsplit > 1) ? 0.f : p.beta;
    const int c = (tid & 31) * 4;
    const int co = co0 + c;
    if (co < p.K) {
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int rl = (tid >> 5) + 8 * it;
            const int ci = ci0 + rl;
            if (ci >= p.C) continue;
            float4 v = *reinterpret_cast<const float4*>(&Cs[rl * TN + c]);
            float4* o = reinterpret_cast<float4*>(dst + ((long)tap * p.C + ci) * p.K + co);
            if (beta != 0.f) {
                const float4 old = *o;
                v.x += beta * old.x; v.y += beta * old.y; v.z += beta * old.z; v.w += beta * old.w;
            }
            *o = v;
        }
    }
}

// out[i] = beta*out[i] + sum_s partial[s][i]; the last block folds the bias-gradient partials
__global__ __launch_bounds__(256) void bw_splitk_sum_kernel(const float* __restrict__ partial, float* __restrict__ out,
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
    if (db != nullptr && blockIdx.x == gridDim.x - 1) {
        for (int co = threadIdx.x; co < K; co += blockDim.x) {
            float v = 0.f;
            for (int s = 0; s < nsplit; ++s) v += bpart[(long)s * K + co];
            db[co] = (beta_b != 0.f) ? beta_b * db[co] + v : v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conversions and filter shadows
template <int TO_BF16>
__global__ __launch_bounds__(256) void cvt_kernel(const void* __restrict__ in, int ldi, void* __restrict__ out, int ldo,
                                                  long rows, int cols) {
    const int c4 = cols >> 2;     // cols % 4 == 0 on this path
    const long total = rows * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        if (TO_BF16) {
            const float4 v = *reinterpret_cast<const float4*>(static_cast<const float*>(in) + r * ldi + c);
            *reinterpret_cast<uint2*>(static_cast<bf16_t*>(out) + r * ldo + c) = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
        } else {
            const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(in) + r * ldi + c);
            *reinterpret_cast<float4*>(static_cast<float*>(out) + r * ldo + c) =
                make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                            __uint_as_float(u.y & 0xffff0000u));
        }
    }
}
template <int TO_BF16>
__global__ __launch_bounds__(256) void cvt_scalar_kernel(const void* __restrict__ in, int ldi, void* __restrict__ out,
                                                         int ldo, long rows, int cols) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        if (TO_BF16) {
            const __bf16 b = (__bf16)static_cast<const float*>(in)[r * ldi + c];
            static_cast<bf16_t*>(out)[r * ldo + c] = __builtin_bit_cast(bf16_t, b);
        } else {
            static_cast<float*>(out)[r * ldo + c] = __uint_as_float((unsigned)static_cast<const bf16_t*>(in)[r * ldi + c] << 16);
        }
    }
}

// dz = dy * act'(y)  /  y = act(x) on bf16 tensors (8 elements = 16 bytes per access; cols % 8 == 0)
template <int BWD>
__global__ __launch_bounds__(256) void act_bf16_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ y,
                                                       int ldy, bf16_t* __restrict__ o, int ldo, long rows, int cols,
                                                       int act, float alpha) {
    const int c8 = cols >> 3;
    const long total = rows * c8;
    const float slope = (act == DPIG_ACT_NONE) ? 1.f : ((act == DPIG_ACT_RELU) ? 0.f : alpha);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(a + r * lda + c), v);
        if (BWD) {
            float m[8];
            unpack8(*reinterpret_cast<const uint4*>(y + r * ldy + c), m);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= (m[e] > 0.f) ? 1.f : slope;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] > 0.f) ? v[e] : (v[e] * slope + 0.f);
        }
        *reinterpret_cast<uint4*>(o + r * ldo + c) = pack8(v);
    }
}

// out[r][0..cols_out) = bf16(in[r][0..cols_in)) followed by zeros: channel padding of a thin input (the 18 pose channels
// -> 32) so that its conv runs on the bf16 matrix-pipe loop; cols_out % 8 == 0
__global__ __launch_bounds__(256) void cvt_pad_kernel(const float* __restrict__ in, int ldi, int cols_in,
                                                      bf16_t* __restrict__ out, int ldo, int cols_out, long rows) {
    const int c8 = cols_out >> 3;
    const long total = rows * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (c + e < cols_in) ? in[r * ldi + c + e] : 0.f;
        *reinterpret_cast<uint4*>(out + r * ldo + c) = pack8(v);
    }
}

// w [taps][C][K] fp32 -> plain [taps][C][K] bf16 and transposed [taps][K][C] bf16 (either may be null): 32 x 32 tiles
// through LDS so that both the read and the transposed write are row-contiguous
// lo_off != 0 (split shadows for DPIG_COMPUTE_BF16X3): the value's second bf16 term, bf16(v - float(bf16(v))), is written
// lo_off elements behind the first one in either layout -- the same two roundings the split k-loop applies in registers.
__device__ __forceinline__ bf16_t bf_hi(float v) { const __bf16 b = (__bf16)v; return __builtin_bit_cast(bf16_t, b); }
__device__ __forceinline__ bf16_t bf_lo(float v) {
    const __bf16 b = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)b);
    return __builtin_bit_cast(bf16_t, l);
}
__global__ __launch_bounds__(256) void shadow_kernel(const float* __restrict__ w, bf16_t* __restrict__ plain,
                                                     bf16_t* __restrict__ trans, int C, int K, long lo_off) {
    __shared__ float t[32][33];
    const int tap = blockIdx.z;
    const int c0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const float* wt = w + (long)tap * C * K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, k = k0 + tx;
        float v = 0.f;
        if (c < C && k < K) {
            v = wt[(long)c * K + k];
            if (plain) {
                plain[(long)tap * C * K + (long)c * K + k] = bf_hi(v);
                if (lo_off) plain[lo_off + (long)tap * C * K + (long)c * K + k] = bf_lo(v);
            }
        }
        t[ty + 8 * i][tx] = v;
    }
    __syncthreads();
    if (trans) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ty + 8 * i, c = c0 + tx;
            if (c < C && k < K) {
                trans[(long)tap * C * K + (long)k * C + c] = bf_hi(t[tx][ty + 8 * i]);
                if (lo_off) trans[lo_off + (long)tap * C * K + (long)k * C + c] = bf_lo(t[tx][ty + 8 * i]);
            }
        }
    }
}

// All filter shadows of one parameter set in ONE launch: table row t = {src offset (floats from `base`), destination
// offset (elements from plain_base / trans_base), taps, C, K, first tile}; block b handles 32 x 32 tile b.
struct ShadowRow { long src, dst, taps, C, K, tile0; };
__global__ __launch_bounds__(256) void shadow_multi_kernel(const float* __restrict__ base, bf16_t* __restrict__ plain_base,
                                                           bf16_t* __restrict__ trans_base,
                                                           const ShadowRow* __restrict__ table, int ntensors, long lo_off) {
    __shared__ float t[32][33];
    const int b = blockIdx.x;
    int lo = 0, hi = ntensors - 1;                      // last row with tile0 <= b (uniform: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].tile0 <= b) lo = mid; else hi = mid - 1;
    }
    const ShadowRow r = table[lo];
    const int C = (int)r.C, K = (int)r.K;
    const int tk = (K + 31) >> 5, tc = (C + 31) >> 5;
    int local = b - (int)r.tile0;
    const int tap = local / (tk * tc);
    local -= tap * tk * tc;
    const int c0 = (local / tk) * 32, k0 = (local % tk) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* wt = base + r.src + (long)tap * C * K;
    bf16_t* plain = plain_base + r.dst + (long)tap * C * K;
    bf16_t* trans = trans_base + r.dst + (long)tap * C * K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, k = k0 + tx;
        float v = 0.f;
        if (c < C && k < K) {
            v = wt[(long)c * K + k];
            plain[(long)c * K + k] = bf_hi(v);
            if (lo_off) plain[lo_off + (long)c * K + k] = bf_lo(v);
        }
        t[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, c = c0 + tx;
        if (c < C && k < K) {
            trans[(long)k * C + c] = bf_hi(t[tx][ty + 8 * i]);
            if (lo_off) trans[lo_off + (long)k * C + c] = bf_lo(t[tx][ty + 8 * i]);
        }
    }
}

// ================================================================================================
// host side
// ================================================================================================
static int gtk() { return TK; }      // k-tile depth of fwd / dgrad

static bool shape_ok(const DpigConvDesc* d) {
    return d->C % 8 == 0 && d->K % 8 == 0 && d->C >= 32 && d->K >= 32 && d->ldx % 8 == 0 && d->ldy % 8 == 0;
}

static int prepare_bg(BGParams& p, int nimg, long filter_elems) {
    p.HrWr = p.Hr * p.Wr;
    find_divisor(p.HrWr, &p.mul_hrwr, &p.shr_hrwr);
    find_divisor(p.Wr, &p.mul_wr, &p.shr_wr);
    const long a_elems = ((long)nimg * p.Hs * p.Ws - 1) * p.lda + p.Cs;
    if (a_elems * 2 >= 0x7fffffffL || filter_elems * 2 >= 0x7fffffffL)
        return fail(DPIG_EINVAL, "tensor exceeds the 2 GiB offset range of one launch");
    p.a_bytes = (unsigned)(a_elems * 2);
    p.b_bytes = (unsigned)(filter_elems * 2);
    const bool al = aligned16(p.A) && aligned16(p.B) && aligned16(p.D) && (!p.D2 || aligned16(p.D2)) &&
                    (!p.bias || aligned16(p.bias)) && (!p.res || aligned16(p.res)) && (!p.res_cls || aligned16(p.res_cls)) &&
                    (!p.mask || aligned16(p.mask)) && (!p.partial || aligned16(p.partial));
    const bool ld = (p.lda % 8 == 0) && (p.ldd % 8 == 0) && (!p.D2 || p.ldd2 % 8 == 0) && (!p.res || p.ldres % 8 == 0) &&
                    (!p.res_cls || p.ldres % 4 == 0) && (!p.mask || p.ldmask % 8 == 0) && (p.Cs % 8 == 0) && (p.Ncols % 8 == 0);
    if (!al || !ld) return fail(DPIG_EALIGN, "bf16 conv: pointers must be 16-byte aligned, channel counts / strides multiples of 8");
    p.mtiles = cdiv(p.M, TM);
    p.ntiles = cdiv(p.Ncols, TN);
    p.cchunks = cdiv(p.Cs, gtk());
    p.ktiles = p.ntaps * p.cchunks;
    return DPIG_OK;
}
static int reduce_blocks(const BGParams& p) {
    const long total8 = (long)p.M * p.Ncols / 8;
    int blocks = cdiv(total8, 256);
    return blocks > 8 * kNumCU ? 8 * kNumCU : blocks;
}
// 2-D patch plan of the halo kernel: returns log2(patch width) (4 or 3) or 0 when the layer should stay on bg_kernel
// (not a 3x3 stride-1 window on a same-size grid, too small to fill the chip without split-K, or patches would be
// mostly padding).  DPIG_BF16_HALO=0 disables it (A/B measurements).
static int halo_plan(const BGParams& p, int nimg, int* tx, int* ty) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("DPIG_BF16_HALO"); enabled = (e && !strcmp(e, "0")) ? 0 : 1; }
    if (!enabled || p.ntaps != 9 || p.tap_nb != 3 || p.sr != 1 || !p.identity_rows || p.replicate) return 0;
    if (p.Hr != p.Hs || p.Wr != p.Ws || p.oys * p.oys != 1 || p.oxs * p.oxs != 1) return 0;
    if (1 + p.oy0 < 0 || 1 + p.oy0 + 2 * p.oys < 0 || 1 + p.oy0 > 2 || 1 + p.oy0 + 2 * p.oys > 2) return 0;
    if (1 + p.ox0 < 0 || 1 + p.ox0 + 2 * p.oxs < 0 || 1 + p.ox0 > 2 || 1 + p.ox0 + 2 * p.oxs > 2) return 0;
    int best = 0;
    double best_u = 0.0;
    for (int twl = 4; twl >= 3; --twl) {
        const int tw = 1 << twl, th = TM / tw;
        const int nx = cdiv(p.Ws, tw), ny = cdiv(p.Hs, th);
        const double u = (double)p.Hs * p.Ws / ((double)nx * tw * ny * th);
        if (u > best_u + 1e-9) { best_u = u; best = twl; *tx = nx; *ty = ny; }
    }
    if (best_u < 0.74) return 0;
    const long tiles = (long)nimg * (*tx) * (*ty) * cdiv(p.Ncols, TN);
    if (tiles < 2 * kNumCU) {
        // fewer patches than resident slots: with DPIG_BF16_HALO_SPLIT=1 the halo kernel splits the reduction over ranges of channel chunks
        // (the plan's nsplit, capped by the chunk count).  Built and measured in round 6 (scripts/bench_halo_split.py,
        // profiles/r06_bf16_small_layer_ab.txt): bit-identical, and NOT faster -- 16 x 8 C512 34.9 -> 34.3 us, 32 x 32 C512 65.6 -> 63.9,
        // 16 x 16 C768 48.4 -> 49.8 -- these launches are bound by their fixed costs, not by L2 -> LDS bytes.  Off by default: the
        // tap-major kernel (with its four-stage ring, bg8d_kernel) keeps these layers.
        static int hsplit = -1;
        if (hsplit < 0) { const char* e = getenv("DPIG_BF16_HALO_SPLIT"); hsplit = (e && !strcmp(e, "1")) ? 1 : 0; }
        if (!hsplit || p.nsplit < 2 || p.cchunks < 2) return 0;
    }
    return best;
}

// Eight-wave forms of bg_kernel (bit 0), bw_kernel (bit 1) and bh_kernel (bit 2) (DPIG_BF16_G8 / dpig_conv_bf16_set_wave8; default 3):
// same tiles, plans and bits.  Measured (profiles/r04_ab_wave8_layers.txt): forward / dgrad -2 % over the 128-tile layers of the three
// graphs (-12..-24 % on the stride-2 dgrads' four-class launch, never worse than +4 %); Market bf16 step +1.0 % (bit 0) and +1.5 %
// (bits 0 + 1), DeepFashion +0.5 %, stage II +0.2 %.  The halo-patch kernel issues 5 pieces per 16 MFMAs already and does not gain (-0.4 %
// on DeepFashion): off by default.
static int g_wave8 = -1;
static int wave8_mode() {
    if (g_wave8 < 0) { const char* e = getenv("DPIG_BF16_G8"); g_wave8 = e ? (atoi(e) & 7) : 3; }
    return g_wave8;
}

static int launch_bg(BGParams& p, int nimg, long filter_elems, hipStream_t st) {
    int rc = prepare_bg(p, nimg, filter_elems);
    if (rc) return rc;
    int tx = 0, ty = 0;
    if (p.stats && (p.nsplit != 1 || !p.identity_rows || p.replicate || !aligned16(p.stats)))
        return fail(DPIG_EINVAL, "bf16 conv fwd with BN statistics needs an un-split plan");
    const int twl = p.stats ? 0 : halo_plan(p, nimg, &tx, &ty);
    {
        p.halo128 = twl != 0;
        const int q = bq_try(p, st);             // large layers: 8-wave 256 x 256 / 512 x 128 tiles (dpig_conv_bf16_q.hip)
        if (q) return q < 0 ? q : DPIG_OK;
    }
    if (twl) {
        p.tiles_x = tx; p.tiles_y = ty;
        p.mtiles = nimg * tx * ty;
        if ((long)p.mtiles * p.ntiles >= 2 * kNumCU || p.nsplit < 2) {
            p.nsplit = 1; p.tiles_per_split = p.cchunks;
        } else {                                               // channel-chunk ranges (tiles_per_split counts CHUNKS for this kernel)
            const int want = p.nsplit < p.cchunks ? p.nsplit : p.cchunks;
            p.tiles_per_split = cdiv(p.cchunks, want);
            p.nsplit = cdiv(p.cchunks, p.tiles_per_split);     // <= the plan's nsplit: the workspace sized for the plan holds the partials
        }
        dim3 hgrid(p.mtiles * p.ntiles, 1, p.nsplit), hblock(256);
        if (wave8_mode() & 4) {
            if (twl == 4) hipLaunchKernelGGL((bh8_kernel<4>), hgrid, dim3(512), 0, st, p);
            else hipLaunchKernelGGL((bh8_kernel<3>), hgrid, dim3(512), 0, st, p);
        } else if (twl == 4) hipLaunchKernelGGL((bh_kernel<4>), hgrid, hblock, 0, st, p);
        else hipLaunchKernelGGL((bh_kernel<3>), hgrid, hblock, 0, st, p);
        rc = check_launch("bh_kernel");
        if (rc == DPIG_OK && p.nsplit > 1) {
            hipLaunchKernelGGL(bg_reduce_kernel, dim3(reduce_blocks(p)), dim3(256), 0, st, p);
            rc = check_launch("bg_reduce_kernel");
        }
        return rc;
    }
    dim3 grid(p.mtiles * p.ntiles, 1, p.nsplit);
    // one round of at most one workgroup per CU (what the split plans give the small layers): the four-stage ring (bg8d_kernel) keeps
    // three k-tiles in flight where the two-stage loop, alone on its CU, waits out every DMA.  DPIG_BF16_DEEP=0: off (A/B measurements).
    static int deep = -1;
    if (deep < 0) { const char* e = getenv("DPIG_BF16_DEEP"); deep = (e && !strcmp(e, "0")) ? 0 : 1; }
    const long wgs = (long)p.mtiles * p.ntiles * p.nsplit;
    if (deep && (wave8_mode() & 1) && !p.stats && wgs <= kNumCU && p.ktiles / p.nsplit >= 4)
        hipLaunchKernelGGL(bg8d_kernel, grid, dim3(512), 0, st, p);
    else if ((wave8_mode() & 1) && !p.stats) hipLaunchKernelGGL(bg8_kernel, grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL(bg_kernel, grid, dim3(256), 0, st, p);
    rc = check_launch("bg_kernel");
    if (rc) return rc;
    if (p.nsplit > 1) {
        hipLaunchKernelGGL(bg_reduce_kernel, dim3(reduce_blocks(p)), dim3(256), 0, st, p);
        rc = check_launch("bg_reduce_kernel");
    }
    return rc;
}
static int launch_bg_multi(BGParams* q, int n, int nimg, long filter_elems, hipStream_t st) {
    BGMulti m = {};
    int max_tiles = 0, max_split = 1, max_red = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = prepare_bg(q[i], nimg, filter_elems);
        if (rc) return rc;
        if (q[i].mtiles * q[i].ntiles > max_tiles) max_tiles = q[i].mtiles * q[i].ntiles;
        if (q[i].nsplit > max_split) max_split = q[i].nsplit;
        if (q[i].nsplit > 1 && reduce_blocks(q[i]) > max_red) max_red = reduce_blocks(q[i]);
        m.q[i] = q[i];
    }
    dim3 grid(max_tiles, n, max_split);
    static int deep = -1;
    if (deep < 0) { const char* e = getenv("DPIG_BF16_DEEP"); deep = (e && !strcmp(e, "0")) ? 0 : 1; }
    long live = 0;                                   // workgroups that do not exit at once
    int min_kt = 1 << 30;
    for (int i = 0; i < n; ++i) {
        live += (long)q[i].mtiles * q[i].ntiles * q[i].nsplit;
        if (q[i].ktiles / q[i].nsplit < min_kt) min_kt = q[i].ktiles / q[i].nsplit;
    }
    if (deep && (wave8_mode() & 1) && live <= kNumCU && min_kt >= 4) hipLaunchKernelGGL(bg8d_multi_kernel, grid, dim3(512), 0, st, m);
    else if (wave8_mode() & 1) hipLaunchKernelGGL(bg8_multi_kernel, grid, dim3(512), 0, st, m);
    else hipLaunchKernelGGL(bg_multi_kernel, grid, dim3(256), 0, st, m);
    int rc = check_launch("bg_multi_kernel");
    if (rc) return rc;
    if (max_red > 0) {
        hipLaunchKernelGGL(bg_reduce_multi_kernel, dim3(max_red, n), dim3(256), 0, st, m);
        rc = check_launch("bg_reduce_multi_kernel");
    }
    return rc;
}

static void plan_s2(const DpigConvDesc* d, int pt, int pl, S2Plan* sp) { plan_dgrad_s2(d, pt, pl, sp, gtk(), kSplitPenalty); }

}  // namespace bfk
}  // namespace dpig

using namespace dpig;
using namespace dpig::bfk;

extern "C" int dpig_conv_bf16_set_wave8(int mode) {
    if (mode < 0 || mode > 7) return fail(DPIG_EINVAL, "wave8 mode out of range");
    g_wave8 = mode;
    return DPIG_OK;
}

extern "C" int dpig_conv2d_bf16_supported(const DpigConvDesc* d, int which) {
    int pt, pl, Ho, Wo;
    if (resolve_desc(d, &pt, &pl, &Ho, &Wo)) return 0;
    (void)which;
    return shape_ok(d) ? 1 : 0;
}

// stride-1 SAME layer whose padded pixel offsets stay inside one buffer descriptor: output pixel m pairs with input pixel m + const
static bool wgrad_is_s1(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo) {
    const long xe = ((long)d->N * d->H * d->W - 1) * d->ldx + d->C;
    return !d->upsample2x && d->stride == 1 && Ho == d->H && Wo == d->W && pt >= 0 && pl >= 0 &&
           (xe * 2 + (long)(pt * d->W + pl) * d->ldx * 2 < 0x7fffffffL);
}

// (The filter gradient keeps all taps on such maps: with beta = 0 -- every first touch of a step -- the padding-only slabs have to be
// written anyway, and the dense launch does that at the same cost as clearing them: measured 11 us dense vs 17 us centre slab + two
// clears on the 1 x 1 level.)
static size_t bf16_workspace_bytes_one(const DpigConvDesc* d, int which) {
    int pt, pl, Ho, Wo;
    if (resolve_desc(d, &pt, &pl, &Ho, &Wo) || !shape_ok(d)) return 0;
    const TapWindow tw = live_taps(d, pt, pl, Ho, Wo);
    if (which == 0) {
        const long M = d->upsample2x ? (long)d->N * d->H * d->W : (long)d->N * Ho * Wo;
        Plan pln = plan_split(cdiv(M, TM) * cdiv(d->K, TN), tw.na * tw.nb * cdiv(d->C, gtk()), d->split_k, gtk(), kSplitPenalty);
        return pln.nsplit > 1 ? (size_t)pln.nsplit * M * d->K * sizeof(float) : 0;
    } else if (which == 1) {
        if (d->upsample2x || d->stride == 1) {
            const long M = (long)d->N * d->H * d->W;
            const int ntaps = d->upsample2x ? 4 : tw.na * tw.nb;
            Plan pln = plan_split(cdiv(M, TM) * cdiv(d->C, TN), ntaps * cdiv(d->K, gtk()), d->split_k, gtk(), kSplitPenalty);
            return pln.nsplit > 1 ? (size_t)pln.nsplit * M * d->C * sizeof(float) : 0;
        }
        S2Plan sp;
        plan_s2(d, pt, pl, &sp);
        return sp.total;
    } else if (which == 2) {
        const long Npix = (long)d->N * Ho * Wo * (d->upsample2x ? 4 : 1);
        const int tiles = (d->upsample2x ? 1 : d->R * d->S) * cdiv(d->C, TM) * cdiv(d->K, TN);
        Plan pln = plan_split(tiles, cdiv(Npix, TK), d->split_k, TK, kSplitPenalty);
        int nsplit = pln.nsplit;
        if (wgrad_is_s1(d, pt, pl, Ho, Wo)) {                    // the large-tile kernel splits deeper (fewer, larger tiles)
            const int v = bwq_choose(d->R * d->S, d->C, d->K, Npix, d->split_k);
            int a, b, s, t;
            if (v) { bwq_plan(d->R * d->S, d->C, d->K, Npix, v, d->split_k, &a, &b, &s, &t); nsplit = s; }
        }
        return nsplit > 1 ? (size_t)nsplit * ((size_t)d->R * d->S * d->C * d->K + d->K) * sizeof(float) : 0;
    }
    return 0;
}

// ---- entry points: one launch, or runs of whole images when a tensor exceeds one launch's 2 GiB range (dpig_conv_plan.h) ----
static int fwd_bf16_one(const DpigConvDesc* d, const uint16_t* x, const uint16_t* w_t, const float* bias,
                        const uint16_t* residual, const float* residual_class, uint16_t* y, uint16_t* y_act, void* ws,
                        size_t ws_bytes, void* stream, float* stats = nullptr);
static int dgrad_bf16_one(const DpigConvDesc* d, const uint16_t* dy, const uint16_t* w, const uint16_t* accum,
                          const uint16_t* mask, uint16_t* dx, void* ws, size_t ws_bytes, void* stream);
static int wgrad_bf16_one(const DpigConvDesc* d, const uint16_t* x, const uint16_t* dy, float* dw, float beta, float* db,
                          float beta_b, void* ws, size_t ws_bytes, void* stream);

extern "C" size_t dpig_conv2d_bf16_workspace_bytes(const DpigConvDesc* d, int which) {
    const int per = images_per_launch(d, 2);
    if (!d || per <= 0 || per >= d->N) return bf16_workspace_bytes_one(d, which);
    DpigConvDesc c = *d;
    c.N = per;
    size_t best = bf16_workspace_bytes_one(&c, which);
    if (d->N % per) {
        c.N = d->N % per;
        const size_t b = bf16_workspace_bytes_one(&c, which);
        if (b > best) best = b;
    }
    return best;
}

extern "C" int dpig_conv2d_fwd_bf16(const DpigConvDesc* d, const uint16_t* x, const uint16_t* w_t, const float* bias,
                                    const uint16_t* residual, const float* residual_class, uint16_t* y, uint16_t* y_act,
                                    void* ws, size_t ws_bytes, void* stream) {
    const int per = images_per_launch(d, 2);
    if (d && per == 0) return fail(DPIG_EINVAL, "one image exceeds the 2 GiB range of a launch");
    if (!d || per >= d->N || !x || !y) return fwd_bf16_one(d, x, w_t, bias, residual, residual_class, y, y_act, ws, ws_bytes, stream);
    long xpix, ypix;
    image_pixels(d, &xpix, &ypix);
    for (int n0 = 0; n0 < d->N; n0 += per) {
        DpigConvDesc c = *d;
        c.N = d->N - n0 < per ? d->N - n0 : per;
        const int rc = fwd_bf16_one(&c, x + (long)n0 * xpix * d->ldx, w_t, bias, residual ? residual + (long)n0 * ypix * d->ldres : nullptr,
                                    residual_class ? residual_class + (long)n0 * 9 * d->ldres : nullptr, y + (long)n0 * ypix * d->ldy,
                                    y_act ? y_act + (long)n0 * ypix * d->ldy2 : nullptr, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    return DPIG_OK;
}

extern "C" int dpig_conv2d_dgrad_bf16(const DpigConvDesc* d, const uint16_t* dy, const uint16_t* w, const uint16_t* accum,
                                      const uint16_t* mask, uint16_t* dx, void* ws, size_t ws_bytes, void* stream) {
    const int per = images_per_launch(d, 2);
    if (d && per == 0) return fail(DPIG_EINVAL, "one image exceeds the 2 GiB range of a launch");
    if (!d || per >= d->N || !dy || !dx) return dgrad_bf16_one(d, dy, w, accum, mask, dx, ws, ws_bytes, stream);
    long xpix, ypix;
    image_pixels(d, &xpix, &ypix);
    for (int n0 = 0; n0 < d->N; n0 += per) {
        DpigConvDesc c = *d;
        c.N = d->N - n0 < per ? d->N - n0 : per;
        const int rc = dgrad_bf16_one(&c, dy + (long)n0 * ypix * d->ldy, w, accum ? accum + (long)n0 * xpix * d->ldres : nullptr,
                                      mask ? mask + (long)n0 * xpix * d->ldmask : nullptr, dx + (long)n0 * xpix * d->ldx, ws, ws_bytes,
                                      stream);
        if (rc) return rc;
    }
    return DPIG_OK;
}

extern "C" int dpig_conv2d_wgrad_bf16(const DpigConvDesc* d, const uint16_t* x, const uint16_t* dy, float* dw, float beta,
                                      float* db, float beta_b, void* ws, size_t ws_bytes, void* stream) {
    const int per = images_per_launch(d, 2);
    if (d && per == 0) return fail(DPIG_EINVAL, "one image exceeds the 2 GiB range of a launch");
    if (!d || per >= d->N || !x || !dy) return wgrad_bf16_one(d, x, dy, dw, beta, db, beta_b, ws, ws_bytes, stream);
    long xpix, ypix;
    image_pixels(d, &xpix, &ypix);
    for (int n0 = 0; n0 < d->N; n0 += per) {     // image runs accumulate in order: same result on every call
        DpigConvDesc c = *d;
        c.N = d->N - n0 < per ? d->N - n0 : per;
        const int rc = wgrad_bf16_one(&c, x + (long)n0 * xpix * d->ldx, dy + (long)n0 * ypix * d->ldy, dw, n0 ? 1.0f : beta, db,
                                      n0 ? 1.0f : beta_b, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    return DPIG_OK;
}

// Forward conv (+ bias) on bf16 tensors that also leaves the batch-norm partial statistics of its output (see
// dpig_conv2d_fwd_stats; merged by dpig_bn_stats_finalize).  Tile count 0: this problem's plan cannot carry them.
extern "C" int dpig_conv2d_bf16_bn_stats_tiles(const DpigConvDesc* d) {
    int pt, pl, Ho, Wo;
    if (resolve_desc(d, &pt, &pl, &Ho, &Wo) || !shape_ok(d)) return 0;
    if (d->upsample2x || d->act != DPIG_ACT_NONE || images_per_launch(d, 2) < d->N) return 0;
    const long M = (long)d->N * Ho * Wo;
    Plan pln = plan_split(cdiv(M, TM) * cdiv(d->K, TN), d->R * d->S * cdiv(d->C, gtk()), d->split_k, gtk(), kSplitPenalty);
    return pln.nsplit == 1 ? (int)cdiv(M, TM) : 0;
}
extern "C" int dpig_conv2d_fwd_bf16_stats(const DpigConvDesc* d, const uint16_t* x, const uint16_t* w_t, const float* bias,
                                          uint16_t* y, float* stats, void* stream) {
    if (!stats) return fail(DPIG_EINVAL, "bf16 conv fwd with BN statistics: null statistics buffer");
    if (dpig_conv2d_bf16_bn_stats_tiles(d) <= 0)
        return fail(DPIG_EINVAL, "bf16 conv fwd with BN statistics: this problem's plan cannot carry them");
    return fwd_bf16_one(d, x, w_t, bias, nullptr, nullptr, y, nullptr, nullptr, 0, stream, stats);
}

static int fwd_bf16_one(const DpigConvDesc* d, const uint16_t* x, const uint16_t* w_t, const float* bias,
                        const uint16_t* residual, const float* residual_class, uint16_t* y, uint16_t* y_act,
                        void* ws, size_t ws_bytes, void* stream, float* stats) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !w_t || !y) return fail(DPIG_EINVAL, "null tensor pointer");
    if (!shape_ok(d)) return fail(DPIG_EINVAL, "bf16 conv needs >= 32 channels in multiples of 8 on both sides");
    if (residual && residual_class) return fail(DPIG_EINVAL, "residual and residual_class are exclusive");
    if (d->upsample2x && (residual || residual_class || y_act)) return fail(DPIG_EINVAL, "residual / y_act unsupported with upsample2x");
    if (residual_class && (d->stride != 1 || d->H < 2 || d->W < 2))
        return fail(DPIG_EINVAL, "res_class needs a stride-1 conv on an image of at least 2x2");
    if (y_act && d->ldy2 < d->K) return fail(DPIG_EINVAL, "ldy2 < K");
    if ((residual || residual_class) && d->ldres < d->K) return fail(DPIG_EINVAL, "ldres < K");
    BGParams p = {};
    p.A = x; p.B = w_t; p.D = y; p.bias = bias; p.res = residual; p.res_cls = residual_class; p.mask = nullptr;
    p.D2 = y_act; p.ldd2 = d->ldy2; p.res_post = d->res_after_act;
    p.partial = static_cast<float*>(ws);
    p.stats = stats;
    p.M = d->upsample2x ? d->N * d->H * d->W : d->N * Ho * Wo;
    p.Hr = d->upsample2x ? d->H : Ho; p.Wr = d->upsample2x ? d->W : Wo;
    p.Hs = d->H; p.Ws = d->W; p.lda = d->ldx; p.Cs = d->C; p.sr = d->stride;
    p.Ncols = d->K;
    if (d->upsample2x) { p.Hd = 2 * d->H; p.Wd = 2 * d->W; p.dr = 2; p.replicate = 1; }
    else { p.Hd = Ho; p.Wd = Wo; p.dr = 1; p.replicate = 0; }
    p.dpy = 0; p.dpx = 0; p.ldd = d->ldy; p.ldres = d->ldres; p.ldmask = 0;
    p.identity_rows = d->upsample2x ? 0 : 1;
    p.act = d->act; p.alpha = d->alpha;
    const TapWindow tw = live_taps(d, pt, pl, Ho, Wo);           // (the whole filter except on maps smaller than it: dpig_conv_plan.h)
    p.ntaps = tw.na * tw.nb;
    p.tap_nb = tw.nb; p.oy0 = -pt + tw.a0; p.oys = 1; p.ox0 = -pl + tw.b0; p.oxs = 1; p.w0 = tw.a0 * d->S + tw.b0; p.wa = d->S; p.wb = 1;
    Plan pln = plan_split(cdiv(p.M, TM) * cdiv(p.Ncols, TN), p.ntaps * cdiv(p.Cs, gtk()), d->split_k, gtk(), kSplitPenalty);
    p.nsplit = pln.nsplit; p.tiles_per_split = pln.tiles_per_split;
    if (p.nsplit > 1 && (!ws || ws_bytes < (size_t)p.nsplit * p.M * p.Ncols * sizeof(float)))
        return fail(DPIG_ENOMEM, "bf16 conv fwd workspace too small: have %zu", ws_bytes);
    return launch_bg(p, d->N, (long)d->R * d->S * d->C * d->K, static_cast<hipStream_t>(stream));
}

static int dgrad_bf16_one(const DpigConvDesc* d, const uint16_t* dy, const uint16_t* w, const uint16_t* accum,
                          const uint16_t* mask, uint16_t* dx, void* ws, size_t ws_bytes, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!dy || !w || !dx) return fail(DPIG_EINVAL, "null tensor pointer");
    if (!shape_ok(d)) return fail(DPIG_EINVAL, "bf16 conv needs >= 32 channels in multiples of 8 on both sides");
    if (accum && d->ldres < d->C) return fail(DPIG_EINVAL, "ldres < C");
    if (mask && d->ldmask < d->C) return fail(DPIG_EINVAL, "ldmask < C");
    hipStream_t st = static_cast<hipStream_t>(stream);
    BGParams p = {};
    p.A = dy; p.B = w; p.D = dx; p.bias = nullptr; p.res = accum; p.mask = mask;
    p.partial = static_cast<float*>(ws);
    p.lda = d->ldy; p.Cs = d->K; p.Ncols = d->C;
    p.Hd = d->H; p.Wd = d->W; p.ldd = d->ldx; p.ldres = d->ldres; p.ldmask = d->ldmask;
    p.act = mask ? d->act : DPIG_ACT_NONE; p.alpha = d->alpha; p.replicate = 0;
    const long felems = (long)d->R * d->S * d->C * d->K;
    if (d->upsample2x) {
        p.M = d->N * d->H * d->W; p.Hr = d->H; p.Wr = d->W;
        p.Hs = 2 * d->H; p.Ws = 2 * d->W; p.sr = 2;
        p.dr = 1; p.identity_rows = 1;
        p.ntaps = 4;
        p.tap_nb = 2; p.oy0 = 0; p.oys = 1; p.ox0 = 0; p.oxs = 1; p.w0 = 0; p.wa = 0; p.wb = 0;
    } else if (d->stride == 1) {
        p.M = d->N * d->H * d->W; p.Hr = d->H; p.Wr = d->W;
        p.Hs = Ho; p.Ws = Wo; p.sr = 1;
        p.dr = 1; p.identity_rows = 1;
        const TapWindow tw = live_taps(d, pt, pl, Ho, Wo);       // (stride 1: the taps live in the forward pass are the ones live here)
        p.ntaps = tw.na * tw.nb;
        p.tap_nb = tw.nb; p.oy0 = pt - tw.a0; p.oys = -1; p.ox0 = pl - tw.b0; p.oxs = -1; p.w0 = tw.a0 * d->S + tw.b0; p.wa = d->S; p.wb = 1;
    } else {
        S2Plan sp;
        plan_s2(d, pt, pl, &sp);
        if (sp.total > 0 && (!ws || ws_bytes < sp.total))
            return fail(DPIG_ENOMEM, "bf16 conv dgrad workspace too small: have %zu", ws_bytes);
        BGParams qs[4];
        for (int i = 0; i < sp.nc; ++i) {
            const DClass& c = sp.cls[i];
            BGParams& q = qs[i];
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
        return launch_bg_multi(qs, sp.nc, d->N, felems, st);
    }
    Plan pln = plan_split(cdiv(p.M, TM) * cdiv(p.Ncols, TN), p.ntaps * cdiv(p.Cs, gtk()), d->split_k, gtk(), kSplitPenalty);
    p.nsplit = pln.nsplit; p.tiles_per_split = pln.tiles_per_split;
    if (p.nsplit > 1 && (!ws || ws_bytes < (size_t)p.nsplit * p.M * p.Ncols * sizeof(float)))
        return fail(DPIG_ENOMEM, "bf16 conv dgrad workspace too small: have %zu", ws_bytes);
    return launch_bg(p, d->N, felems, st);
}

static int wgrad_bf16_one(const DpigConvDesc* d, const uint16_t* x, const uint16_t* dy, float* dw, float beta,
                          float* db, float beta_b, void* ws, size_t ws_bytes, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !dy || !dw) return fail(DPIG_EINVAL, "null tensor pointer");
    if (!shape_ok(d)) return fail(DPIG_EINVAL, "bf16 conv needs >= 32 channels in multiples of 8 on both sides");
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dw) || (ws && !aligned16(ws)))
        return fail(DPIG_EALIGN, "bf16 wgrad: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    BWParams p = {};
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
    const long xe = ((long)d->N * d->H * d->W - 1) * d->ldx + d->C;
    const long ye = ((long)p.Npix - 1) * d->ldy + d->K;
    if (xe * 2 >= 0x7fffffffL || ye * 2 >= 0x7fffffffL)
        return fail(DPIG_EINVAL, "tensor exceeds the 2 GiB offset range of one launch");
    p.x_bytes = (unsigned)(xe * 2); p.y_bytes = (unsigned)(ye * 2);
    find_divisor(p.HoWo, &p.mul_howo, &p.shr_howo);
    find_divisor(p.Wo, &p.mul_wo, &p.shr_wo);
    p.wrows = d->R * d->S * d->C;
    p.cblocks = cdiv(d->C, TM);
    p.ntiles = cdiv(d->K, TN);
    p.ktiles = cdiv(p.Npix, TK);
    const int tiles = p.ntaps * p.cblocks * p.ntiles;
    Plan pln = plan_split(tiles, p.ktiles, d->split_k, TK, kSplitPenalty);
    p.nsplit = pln.nsplit; p.tiles_per_split = pln.tiles_per_split;
    const bool s1 = wgrad_is_s1(d, pt, pl, Ho, Wo);
    const int qv = s1 ? bwq_choose(p.ntaps, d->C, d->K, p.Npix, d->split_k) : 0;      // large-tile kernel (dpig_conv_bf16_wq.hip)?
    if (qv) bwq_plan(p.ntaps, d->C, d->K, p.Npix, qv, d->split_k, &p.q_mtiles, &p.q_ntiles, &p.nsplit, &p.tiles_per_split);
    const long wsize = (long)p.wrows * d->K;
    if (p.nsplit > 1 && (!ws || ws_bytes < (size_t)p.nsplit * (wsize + d->K) * sizeof(float)))
        return fail(DPIG_ENOMEM, "bf16 conv wgrad workspace too small: have %zu", ws_bytes);
    p.DB = db; p.beta_b = beta_b;
    p.bias_partial = p.partial ? p.partial + (long)p.nsplit * wsize : nullptr;
    p.d64_n = TK / p.HoWo;
    p.d64_oy = (TK % p.HoWo) / p.Wo;
    p.d64_ox = (TK % p.HoWo) % p.Wo;
    dim3 grid(tiles, 1, p.nsplit), block(256);
    if (qv) {
        rc = bwq_try(p, qv, st);
        if (rc < 0) return rc;
        rc = DPIG_OK;
    } else {
        if (wave8_mode() & 2) {
            if (s1) hipLaunchKernelGGL((bw8_kernel<true>), grid, dim3(512), 0, st, p);
            else hipLaunchKernelGGL((bw8_kernel<false>), grid, dim3(512), 0, st, p);
        } else if (s1) hipLaunchKernelGGL((bw_kernel<true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((bw_kernel<false>), grid, block, 0, st, p);
        rc = check_launch("bw_kernel");
        if (rc) return rc;
    }
    if (p.nsplit > 1) {
        const long n4 = wsize / 4;
        int blocks = cdiv(n4, 256);
        if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
        hipLaunchKernelGGL(bw_splitk_sum_kernel, dim3(blocks), dim3(256), 0, st, p.partial, dw, n4, p.nsplit, beta,
                           p.bias_partial, db, d->K, beta_b);
        rc = check_launch("bw_splitk_sum_kernel");
    }
    return rc;
}

extern "C" int dpig_cvt_f32_to_bf16(const float* in, int ldi, uint16_t* out, int ldo, int64_t rows, int cols, void* stream) {
    if (!in || !out || rows < 0 || cols <= 0 || ldi < cols || ldo < cols) return fail(DPIG_EINVAL, "cvt: bad arguments");
    if (rows == 0) return DPIG_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = (cols % 4 == 0) && (ldi % 4 == 0) && (ldo % 4 == 0) && aligned16(in) && ((reinterpret_cast<uintptr_t>(out) & 7u) == 0);
    const long total = vec ? rows * (cols / 4) : rows * (long)cols;
    int blocks = cdiv(total, 256);
    if (blocks > 16 * kNumCU) blocks = 16 * kNumCU;
    if (vec) hipLaunchKernelGGL((cvt_kernel<1>), dim3(blocks), dim3(256), 0, st, in, ldi, out, ldo, (long)rows, cols);
    else hipLaunchKernelGGL((cvt_scalar_kernel<1>), dim3(blocks), dim3(256), 0, st, in, ldi, out, ldo, (long)rows, cols);
    return check_launch("cvt_f32_to_bf16");
}

extern "C" int dpig_cvt_bf16_to_f32(const uint16_t* in, int ldi, float* out, int ldo, int64_t rows, int cols, void* stream) {
    if (!in || !out || rows < 0 || cols <= 0 || ldi < cols || ldo < cols) return fail(DPIG_EINVAL, "cvt: bad arguments");
    if (rows == 0) return DPIG_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = (cols % 4 == 0) && (ldi % 4 == 0) && (ldo % 4 == 0) && aligned16(out) && ((reinterpret_cast<uintptr_t>(in) & 7u) == 0);
    const long total = vec ? rows * (cols / 4) : rows * (long)cols;
    int blocks = cdiv(total, 256);
    if (blocks > 16 * kNumCU) blocks = 16 * kNumCU;
    if (vec) hipLaunchKernelGGL((cvt_kernel<0>), dim3(blocks), dim3(256), 0, st, in, ldi, out, ldo, (long)rows, cols);
    else hipLaunchKernelGGL((cvt_scalar_kernel<0>), dim3(blocks), dim3(256), 0, st, in, ldi, out, ldo, (long)rows, cols);
    return check_launch("cvt_bf16_to_f32");
}

extern "C" int dpig_filter_shadow_bf16(const float* w, uint16_t* plain, uint16_t* transposed, int taps, int C, int K,
                                       void* stream) {
    if (!w || taps <= 0 || C <= 0 || K <= 0) return fail(DPIG_EINVAL, "filter shadow: bad arguments");
    if (!plain && !transposed) return DPIG_OK;
    dim3 grid(cdiv(K, 32), cdiv(C, 32), taps);
    hipLaunchKernelGGL(shadow_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), w, plain, transposed, C, K, 0L);
    return check_launch("filter_shadow_bf16");
}
// split32 image of an fp32 activation for the split-bf16 k-loop with both operands by DMA (dpig_conv.hip, PIPE 4):
// out[row][chunk][0..31] = bf16(x), out[row][chunk][32..63] = bf16(x - bf16(x)) for the chunk's 32 channels (zeros past C).
// One thread per 8 channels: two 16-byte loads, two 16-byte stores.
__global__ __launch_bounds__(256) void split32_kernel(const float* __restrict__ x, int ldx, long rows, int C, int nchunk,
                                                      bf16_t* __restrict__ out) {
    const long total = rows * nchunk * 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / (nchunk * 4);
        const int rem = (int)(i - row * (nchunk * 4));
        const int chunk = rem >> 2, g = rem & 3;
        const int c0 = chunk * 32 + g * 8;
        float v[8];
        if (c0 + 8 <= C) {
            const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c0);
            const float4 b = *reinterpret_cast<const float4*>(x + row * ldx + c0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c0 + e < C) ? x[row * ldx + c0 + e] : 0.f;
        }
        unsigned short h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { h[e] = bf_hi(v[e]); l[e] = bf_lo(v[e]); }
        bf16_t* o = out + (row * nchunk + chunk) * 64 + g * 8;
        *reinterpret_cast<uint4*>(o) = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
        *reinterpret_cast<uint4*>(o + 32) = make_uint4(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16), l[4] | ((unsigned)l[5] << 16), l[6] | ((unsigned)l[7] << 16));
    }
}
extern "C" size_t dpig_split32_bytes(int64_t rows, int C) { return (rows > 0 && C > 0) ? (size_t)rows * ((C + 31) / 32) * 128 : 0; }
extern "C" int dpig_split32(const float* x, int ldx, int64_t rows, int C, uint16_t* out, void* stream) {
    if (!x || !out || rows <= 0 || C <= 0 || ldx < C) return fail(DPIG_EINVAL, "split32: bad arguments");
    if ((ldx % 4) || !aligned16(x) || !aligned16(out) || (C % 4)) return fail(DPIG_EINVAL, "split32: operands must be 16-byte addressable");
    const int nchunk = (C + 31) / 32;
    const long total = (long)rows * nchunk * 4;
    long blocks = (total + 255) / 256;
    if (blocks > 16 * 256) blocks = 16 * 256;
    hipLaunchKernelGGL(split32_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, (long)rows, C,
                       nchunk, out);
    return check_launch("split32");
}

// Host side of bw3_kernel for dpig_conv2d_wgrad_x3 (dpig_conv.hip): 1 = launched, 0 = not this layer's case (the caller runs
// the register-path kernel on the fp32 tensors), < 0 = error.  The split-K plan is the fp32 kernel's (k-tile 32, `pen`).
namespace dpig {
int wgrad_x3_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const uint16_t* x32, const uint16_t* dy32, float* dw,
                 float beta, float* db, float beta_b, void* ws, size_t ws_bytes, hipStream_t st, double pen) {
    using namespace bfk;
    if (!x32 || !dy32 || d->C < 32 || d->K <= 32 || d->C % 4 || d->K % 4) return 0;
    if (!aligned16(x32) || !aligned16(dy32) || !aligned16(dw) || (ws && !aligned16(ws))) return 0;
    BW3Params p = {};
    p.X32 = x32; p.DY32 = dy32; p.DW = dw; p.partial = static_cast<float*>(ws);
    p.H = d->H; p.W = d->W; p.C = d->C; p.K = d->K; p.nchx = cdiv(d->C, 32); p.nchy = cdiv(d->K, 32);
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
    const long xb = (long)d->N * d->H * d->W * p.nchx * 128, yb = (long)p.Npix * p.nchy * 128;
    const long padb = (long)(pt > 0 ? pt : 0) * d->W * p.nchx * 128 + (long)(pl > 0 ? pl : 0) * p.nchx * 128;
    if (xb + padb >= 0x7fffffffL || yb >= 0x7fffffffL) return 0;
    p.x_bytes = (unsigned)xb; p.y_bytes = (unsigned)yb;
    find_divisor(p.HoWo, &p.mul_howo, &p.shr_howo);
    find_divisor(p.Wo, &p.mul_wo, &p.shr_wo);
    p.wrows = d->R * d->S * d->C;
    p.cblocks = cdiv(d->C, TM);
    p.ntiles = cdiv(d->K, TN);
    p.ktiles = cdiv(p.Npix, W3_TK);
    const int tiles = p.ntaps * p.cblocks * p.ntiles;
    Plan pln = plan_split(tiles, p.ktiles, d->split_k, 32, pen);
    p.nsplit = pln.nsplit; p.tiles_per_split = pln.tiles_per_split;
    const long wsize = (long)p.wrows * d->K;
    if (p.nsplit > 1 && (!ws || ws_bytes < (size_t)p.nsplit * (wsize + d->K) * sizeof(float)))
        return fail(DPIG_ENOMEM, "conv wgrad workspace too small: have %zu", ws_bytes);
    if (p.nsplit > 1 && (wsize % 4)) return 0;
    p.DB = db; p.beta_b = beta_b;
    p.bias_partial = p.partial ? p.partial + (long)p.nsplit * wsize : nullptr;
    p.d32_n = W3_TK / p.HoWo;
    p.d32_oy = (W3_TK % p.HoWo) / p.Wo;
    p.d32_ox = (W3_TK % p.HoWo) % p.Wo;
    dim3 grid(tiles, 1, p.nsplit), block(256);
    const bool s1 = !d->upsample2x && d->stride == 1 && Ho == d->H && Wo == d->W && pt >= 0 && pl >= 0;
    if (s1) hipLaunchKernelGGL((bw3_kernel<true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((bw3_kernel<false>), grid, block, 0, st, p);
    int rc = check_launch("bw3_kernel");
    if (rc) return rc;
    if (p.nsplit > 1) {
        const long n4 = wsize / 4;
        int blocks = cdiv(n4, 256);
        if (blocks > 8 * kNumCU) blocks = 8 * kNumCU;
        hipLaunchKernelGGL(bw_splitk_sum_kernel, dim3(blocks), dim3(256), 0, st, p.partial, dw, n4, p.nsplit, beta,
                           p.bias_partial, db, d->K, beta_b);
        rc = check_launch("bw_splitk_sum_kernel");
        if (rc) return rc;
    }
    return 1;
}
}  // namespace dpig

// Split shadows for DPIG_COMPUTE_BF16X3 (dpig_conv2d_fwd_x3 / _dgrad_x3): hi planes at `plain_hi` / `trans_hi`, the lo planes
// lo_off ELEMENTS behind each (one allocation per layout, so that one buffer descriptor spans both planes).
extern "C" int dpig_filter_shadow_split(const float* w, uint16_t* plain_hi, uint16_t* trans_hi, int64_t lo_off, int taps, int C,
                                        int K, void* stream) {
    if (!w || taps <= 0 || C <= 0 || K <= 0 || lo_off < (int64_t)taps * C * K) return fail(DPIG_EINVAL, "filter shadow (split): bad arguments");
    if (!plain_hi && !trans_hi) return DPIG_OK;
    dim3 grid(cdiv(K, 32), cdiv(C, 32), taps);
    hipLaunchKernelGGL(shadow_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), w, plain_hi, trans_hi, C, K, (long)lo_off);
    return check_launch("filter_shadow_split");
}
extern "C" int dpig_filter_shadow_split_multi(const float* base, uint16_t* plain_base, uint16_t* trans_base, int64_t lo_off,
                                              const int64_t* table_dev, int ntensors, int total_tiles, void* stream) {
    if (!base || !plain_base || !trans_base || !table_dev || ntensors <= 0 || total_tiles <= 0 || lo_off <= 0)
        return fail(DPIG_EINVAL, "filter shadow (split, multi): bad arguments");
    hipLaunchKernelGGL(shadow_multi_kernel, dim3(total_tiles), dim3(256), 0, static_cast<hipStream_t>(stream), base,
                       plain_base, trans_base, reinterpret_cast<const ShadowRow*>(table_dev), ntensors, (long)lo_off);
    return check_launch("filter_shadow_split_multi");
}

static int act_bf16_args(const void* a, int lda, const void* o, int ldo, int64_t rows, int cols) {
    if (!a || !o || rows < 0 || cols <= 0) return fail(DPIG_EINVAL, "act_bf16: bad arguments");
    if (cols % 8 || lda % 8 || ldo % 8 || !aligned16(a) || !aligned16(o))
        return fail(DPIG_EALIGN, "act_bf16: 16-byte aligned tensors with channel counts / strides in multiples of 8");
    return DPIG_OK;
}
extern "C" int dpig_act_fwd_bf16(const uint16_t* x, int ldx, uint16_t* y, int ldy, int64_t rows, int cols, int act,
                                 float alpha, void* stream) {
    int rc = act_bf16_args(x, ldx, y, ldy, rows, cols);
    if (rc || rows == 0) return rc;
    int blocks = cdiv(rows * (cols / 8), 256);
    if (blocks > 16 * kNumCU) blocks = 16 * kNumCU;
    hipLaunchKernelGGL((act_bf16_kernel<0>), dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx,
                       (const bf16_t*)nullptr, 0, y, ldy, (long)rows, cols, act, alpha);
    return check_launch("act_fwd_bf16");
}
extern "C" int dpig_act_bwd_bf16(const uint16_t* dy, int lddy, const uint16_t* y, int ldy, uint16_t* dz, int lddz,
                                 int64_t rows, int cols, int act, float alpha, void* stream) {
    int rc = act_bf16_args(dy, lddy, dz, lddz, rows, cols);
    if (rc || rows == 0) return rc;
    if (!y || ldy % 8 || !aligned16(y)) return fail(DPIG_EALIGN, "act_bwd_bf16: bad activation tensor");
    int blocks = cdiv(rows * (cols / 8), 256);
    if (blocks > 16 * kNumCU) blocks = 16 * kNumCU;
    hipLaunchKernelGGL((act_bf16_kernel<1>), dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dy, lddy, y,
                       ldy, dz, lddz, (long)rows, cols, act, alpha);
    return check_launch("act_bwd_bf16");
}

// ---- thin layers of 'bf16' mode: the vector-ALU kernels of dpig_thin.hip with their WIDE tensor stored as bf16 ----------
// K == 3 (the generator's image conv, 3x3 s1): x / dx bf16 [.., C], y / dy fp32 [.., 3].
// C == 3 (encoder stem 3x3 s1, critic conv1 5x5 s2): x / dx fp32 image, y / dy bf16 [.., K].
static int thin_kind(const DpigConvDesc* d) { return d->K == 3 ? 1 : (d->C == 3 ? 2 : 0); }

extern "C" int dpig_conv2d_fwd_thin_bf16(const DpigConvDesc* d, const void* x, const float* w, const float* bias, void* y,
                                         void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !w || !y) return fail(DPIG_EINVAL, "null tensor pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int kind = thin_kind(d);
    if (kind == 1) rc = thin_fwd_try(d, pt, pl, x, w, bias, nullptr, static_cast<float*>(y), nullptr, st, true);
    else if (kind == 2) rc = fewc_fwd_try(d, pt, pl, Ho, Wo, static_cast<const float*>(x), w, bias, nullptr, y, nullptr, st, true);
    else rc = 0;
    if (rc == 0) return fail(DPIG_EINVAL, "not a thin layer the vector-ALU kernels accept");
    return rc < 0 ? rc : DPIG_OK;
}

extern "C" int dpig_conv2d_dgrad_thin_bf16(const DpigConvDesc* d, const void* dy, const float* w, void* dx, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!dy || !w || !dx) return fail(DPIG_EINVAL, "null tensor pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int kind = thin_kind(d);
    if (kind == 1) rc = thin_dgrad_try(d, pt, pl, static_cast<const float*>(dy), w, nullptr, nullptr, dx, st, true);
    else if (kind == 2) rc = fewc_dgrad_try(d, pt, pl, Ho, Wo, dy, w, nullptr, nullptr, static_cast<float*>(dx), st, true);
    else rc = 0;
    if (rc == 0) return fail(DPIG_EINVAL, "not a thin layer the vector-ALU kernels accept");
    return rc < 0 ? rc : DPIG_OK;
}

extern "C" int dpig_conv2d_wgrad_thin_bf16(const DpigConvDesc* d, const void* x, const void* dy, float* dw, float beta,
                                           float* db, float beta_b, void* ws, size_t ws_bytes, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !dy || !dw) return fail(DPIG_EINVAL, "null tensor pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int kind = thin_kind(d);
    if (kind == 1) rc = thin_wgrad_try(d, pt, pl, x, static_cast<const float*>(dy), dw, beta, db, beta_b, ws, ws_bytes, st, true);
    else if (kind == 2) rc = fewc_wgrad_try(d, pt, pl, Ho, Wo, static_cast<const float*>(x), dy, dw, beta, db, beta_b, ws, ws_bytes, st, true);
    else rc = 0;
    if (rc == 0) return fail(DPIG_EINVAL, "not a thin layer the vector-ALU kernels accept");
    return rc < 0 ? rc : DPIG_OK;
}

extern "C" int dpig_cvt_f32_to_bf16_pad(const float* in, int ldi, int cols_in, uint16_t* out, int ldo, int cols_out,
                                        int64_t rows, void* stream) {
    if (!in || !out || rows < 0 || cols_in <= 0 || cols_out < cols_in || ldi < cols_in || ldo < cols_out)
        return fail(DPIG_EINVAL, "cvt_pad: bad arguments");
    if (cols_out % 8 || ldo % 8 || !aligned16(out)) return fail(DPIG_EALIGN, "cvt_pad: output rows must be 16-byte vectors");
    if (rows == 0) return DPIG_OK;
    int blocks = cdiv(rows * (cols_out / 8), 256);
    if (blocks > 16 * kNumCU) blocks = 16 * kNumCU;
    hipLaunchKernelGGL(cvt_pad_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), in, ldi, cols_in, out,
                       ldo, cols_out, (long)rows);
    return check_launch("cvt_f32_to_bf16_pad");
}

extern "C" int dpig_filter_shadow_bf16_multi(const float* base, uint16_t* plain_base, uint16_t* trans_base,
                                             const int64_t* table_dev, int ntensors, int total_tiles, void* stream) {
    if (!base || !plain_base || !trans_base || !table_dev || ntensors <= 0 || total_tiles <= 0)
        return fail(DPIG_EINVAL, "filter shadow (multi): bad arguments");
    hipLaunchKernelGGL(shadow_multi_kernel, dim3(total_tiles), dim3(256), 0, static_cast<hipStream_t>(stream), base,
                       plain_base, trans_base, reinterpret_cast<const ShadowRow*>(table_dev), ntensors, 0L);
    return check_launch("filter_shadow_bf16_multi");
}
