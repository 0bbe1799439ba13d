// bf16-STORAGE gather-GEMM, large-tile form: 8 waves per workgroup, ONE workgroup per CU, block tile 256 x 256 x 64
// (bq_kernel<2, 4>) or 512 x 128 x 64 (bq_kernel<4, 2>, for 128-column layers), every wave a 128 x 64 sub-tile =
// 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16.  Same problem abstraction as bg_kernel (dpig_conv_bf16.hip): rows =
// output pixels, k = (filter tap, 64-channel chunk), A gathered per tap with the halo as out-of-range DMA lanes, B = the
// filter shadow; forward 3x3 / 5x5 / 1x1 and the stride-1 dgrad run on it (tflib/ops/conv2d.py:106-112, models.py:396-573).
//
// Why: at 128 x 128 a wave must issue 8 LDS-DMA pieces per 16 MFMAs and the k-loop is bound by their issue cost and
// landing latency (profiles/r02_bf16_kloop_timeline_knockout.md).  Here a wave issues 8 (10) pieces per 32 MFMAs and
// nothing in the loop ever waits for the DMA queue to drain:
//  * two LDS slots (k-tiles t, t+1) of [A tile | B tile], rows of 128 B = 64 bf16 of k, lane-linear DMA images with the
//    bank swizzle on the SOURCE side (slot s of row r holds chunk s ^ ((r >> 1) & 7)), fragments by ds_read_b128;
//  * a k-tile is TWO phases of 16 MFMAs (a 64 x 64 half of the wave's accumulators) each:
//        PA  read B[nb0], B[nb1] (8) + A[mh0] (8)   stage UA1(t+1)                  MFMA (mh0, nb0), (mh0, nb1)
//        PB  read A[mh1] (8)                        stage UA0, UB0, UB1 of (t+2)    MFMA (mh1, nb0), (mh1, nb1)
//    a staging unit = the rows the waves read in ONE phase: UA0 = rows 0..63 of every wave row's 128, UA1 = rows
//    64..127, UB0 / UB1 = columns 0..31 / 32..63 of every wave column's 64;
//  * a phase is   { ds_reads ; DMA issue ; s_waitcnt vmcnt(N) ; s_waitcnt lgkmcnt(0) ; s_barrier ; 16 MFMAs at
//    s_setprio 1 ; s_barrier }   with RAW barriers (no fence: an LDS-DMA is a pending LDS write, __syncthreads() would
//    drain it) and the two wave groups (waves 0-3 / 4-7 = the two waves of each SIMD) staggered by one barrier, so that
//    one wave of every SIMD is in its MFMA cluster while its partner reads fragments and issues DMA ("ping-pong").
//    Measured on this kernel (scripts/ubench/knockout_q.sh): each barrier interval costs ~120-140 cycles on top of its
//    MFMA cluster whatever the cluster's length, so the clusters are 16 MFMAs (512 cycles), not 8; removing the barriers
//    altogether is SLOWER (the groups stop being complementary); the counted waits cost nothing;
//  * hazards (p = phase count, b = barrier count; group 0 runs L(p) in [b(2p-2), b(2p-1)] and M(p) in [b(2p-1), b(2p)],
//    group 1 one barrier later):
//      RAW  a unit is read one phase AFTER the phase whose counted vmcnt retires it: every wave waits for ITS pieces
//           before the first barrier of phase r, readers start behind that barrier in phase r + 1.  N = 2 (NA + NB)
//           pieces = the units issued after the one being retired (UA1(t) is retired in PA(t) behind {UA0, UB0, UB1}
//           (t+1) and UA1(t+1); {UA0, UB0, UB1}(t+1) in PB(t) behind UA1(t+1) and the three units of t+2);
//      WAR  a unit is re-staged ONE phase after its last read: the reading wave's lgkmcnt(0) sits before the first
//           barrier of its phase, and both groups' issuing phases lie behind that barrier (a phase's DMA issue follows
//           the other group's first barrier of the previous phase);
//    past the last k-tile the same DMA instructions are issued with out-of-range offsets (zero fill of a free unit), so
//    the counts stay uniform; the queue is drained once, before the epilogue reuses the LDS.
// Epilogue: the MFMAs are fed filter-fragment first, so the accumulators hold D^T and each wave transposes its own 128 x 64
// sub-tile through 8.5 KB of LDS of its own (no workgroup barrier) into the fused row-contiguous epilogue of the family
// (bias, activation, residual before / after, second output, dgrad mask); the rare flag combinations (class-indexed
// residual, 2 x 2 replication) go through the generic per-element epilogue behind workgroup barriers.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "dpig_bf16_common.h"
#include "dpig_conv_plan.h"

#ifdef DPIG_TRACE   // dev aid (scripts/ubench/trace_q.py; never in the shipped build): s_memtime segment sums per wave
__device__ unsigned long long dpig_bq_prof[256 * 8 * 16];
extern "C" int dpig_debug_bq_prof_read(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dpig_bq_prof), sizeof(unsigned long long) * n);
}
#define QT_DECL unsigned long long qt_last = 0, qt_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long qt_begin = __builtin_amdgcn_s_memtime();
#define QT_START() do { qt_last = __builtin_amdgcn_s_memtime(); } while (0)
#define QT_SEG(i) do { const unsigned long long qt_now = __builtin_amdgcn_s_memtime(); qt_sum[i] += qt_now - qt_last; qt_last = qt_now; } while (0)
#else
#define QT_DECL
#define QT_START() do { } while (0)
#define QT_SEG(i) do { } while (0)
#endif

namespace dpig {
namespace bfk {

template <int WM, int WN>
struct QGeom {
    static constexpr int BM = WM * 128, BN = WN * 64;
    static constexpr int ASLOT = 128 * ROWB, BSLOT = 64 * ROWB;     // one slot of one wave row / wave column: 16 KB / 8 KB
    static constexpr int B_OFF = WM * 2 * ASLOT;                     // B region behind the A region
    static constexpr int SMEM = B_OFF + WN * 2 * BSLOT;
    static constexpr int NA = WM;              // DMA pieces per wave per A unit (WM * 64 rows / 8 rows / 8 waves)
    static constexpr int NB = WN / 2;          // per B unit
    static constexpr int VMC = 2 * (NA + NB);  // pieces left in flight by the per-phase wait
    static constexpr int LDCQ = BN + 4;        // fp32 staging stride of the epilogue
    static constexpr int RP = WM * 32;         // staged rows per epilogue pass
    static_assert(WM * WN == 8 && (WN == 2 || WN == 4), "8 waves");
    static_assert(RP * LDCQ * 4 <= SMEM && SMEM <= 163840, "LDS plan");
};

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// (the round-3 knock-out builds -- one ingredient of the loop compiled out at a time -- are recorded in profiles/r03_q_knockout.md;
// their #ifdefs have been removed from the loops)
template <int N>
__device__ __forceinline__ void loop_wait_vm() {
    wait_vm<N>();
}

typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ void dma16l(__amdgpu_buffer_rsrc_t rs, int voff, int soff, lds_char* lds_dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_dst, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ void q_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void loop_barrier() {
    q_barrier();
}

// ---- wave-private epilogue ---------------------------------------------------------------------------------------------
// The k-loop feeds the filter fragment as the MFMA's FIRST operand, so an accumulator block holds D^T: lane (l31, half)
// owns output PIXEL l31 of its 32-pixel block and the channels 8q + 4*half + 0..3 (q = 0..3) of its 32-channel block --
// four runs of 4 consecutive fp32 = four 16-byte LDS stores per block instead of sixteen 4-byte ones.  Each wave
// transposes its own 32 pixels x 64 channels (8.5 KB of LDS of its own; the LDS executes one wave's instructions in order,
// so no barrier and no cross-wave wait), then lane -> (pixel = 8 i + lane / 8, 8 consecutive channels = lane % 8) applies
// bias / residual / activation (mask) / second output with 16-byte accesses: 8 lanes cover one pixel's 128 contiguous
// bytes, an instruction covers 8 full lines.  (Measured alternatives: staging the whole tile as fp32 rows behind
// workgroup barriers with 4-byte LDS stores cost 13 % of a 256 x 256 x 2304 tile; a register-only epilogue with
// v_permlane32_swap + per-pixel 16-byte stores is address-coalescer bound: 32 segments per instruction.)
constexpr int WEP_ROW = 64 * 4 + 16;              // bytes per staged pixel row (+16: the 8 lanes of a store group hit 8 bank groups)
constexpr int WEP_BYTES = 32 * WEP_ROW;           // per wave
// `patch_w` > 0 (bhq_kernel): the wave's 128 rows are an 8-row x 16-column patch of one image; row_base is the pixel index of
// its first pixel and patch_w the image width, so row r is pixel row_base + (r >> 4) * patch_w + (r & 15).
template <bool HAS_RES, bool RES_POST, bool HAS_MASK, bool HAS_D2>
__device__ __forceinline__ void q_epilogue_wave(const BGParams& p, f32x16 (&acc)[4][2], lds_char* W, int row_base, int cb0,
                                                int lane, float slope, int patch_w = 0) {
    const int l31 = lane & 31, half = lane >> 5;
    const int c8 = (lane & 7) * 8, prow = lane >> 3;
    const int col = cb0 + c8;
    const bool cok = col < p.Ncols;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    if (p.bias && cok) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(lds_f4*)(W + l31 * WEP_ROW + (nb * 32 + 8 * q + 4 * half) * 4) =
                    f32x4{acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = 8 * i + prow;
            const f32x4 v0 = *(const lds_f4*)(W + px * WEP_ROW + c8 * 4);
            const f32x4 v1 = *(const lds_f4*)(W + px * WEP_ROW + c8 * 4 + 16);
            const int rr = mb * 32 + px;
            const long r0 = patch_w ? (long)row_base + (rr >> 4) * patch_w + (rr & 15) : (long)row_base + rr;
            if (!cok || r0 >= p.M) continue;
            float v[8] = {v0[0] + bv[0], v0[1] + bv[1], v0[2] + bv[2], v0[3] + bv[3], v1[0] + bv[4], v1[1] + bv[5], v1[2] + bv[6], v1[3] + bv[7]};
            float rv[8];
            if (HAS_RES) unpack8(*reinterpret_cast<const uint4*>(p.res + r0 * p.ldres + col), rv);
            if (HAS_RES && !RES_POST) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            if (HAS_MASK) {
                float mv[8];
                unpack8(*reinterpret_cast<const uint4*>(p.mask + r0 * p.ldmask + col), mv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= (mv[e] > 0.f) ? 1.f : slope;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (v[e] > 0.f) ? v[e] : (v[e] * slope + 0.f);
            }
            if (HAS_D2) {
                const uint4 o2 = pack8(v);
                *reinterpret_cast<uint4*>(p.D2 + r0 * p.ldd2 + col) = o2;
                if (HAS_RES && RES_POST) unpack8(o2, v);             // the sum is formed from the STORED (rounded) activation
            }
            if (HAS_RES && RES_POST) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            *reinterpret_cast<uint4*>(p.D + r0 * p.ldd + col) = pack8(v);
        }
    }
}

template <int WM, int WN>
__global__ __launch_bounds__(512, 2) void bq_kernel(const BGParams p) {
    using G = QGeom<WM, WN>;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM];        // the ONLY LDS object
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int grp = wave >> 2;                                         // waves w and w + 4 share a SIMD
    const int l31 = lane & 31, half = lane >> 5;

    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * G::BM, n0 = nt * G::BN;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B, p.b_bytes);

    // ---- LDS plan: A region = [wave row][slot][128 rows x 128 B], B region = [wave column][slot][64 rows x 128 B]: the
    // slot is ONE address bit (G::ASLOT / G::BSLOT), so the k-loop has a single body and toggles its addresses by XOR, and
    // every fragment offset is a 16-bit immediate of 8 address registers.
    // ---- DMA roles.  A unit h, piece j of this wave: rows h*64 + 8*wave .. +7 of wave row j; B unit h, piece j: columns
    // h*32 + 8*(wave & 3) .. +7 of wave column (wave >> 2) + 2j.  lane -> (row = lane >> 3, 16-byte slot = lane & 7).
    const int prow = lane >> 3;
    int a_base[2][G::NA], a_yx[2][G::NA], a_voff[2][G::NA], b_voff[2][G::NB];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < G::NA; ++j) {
            const int trow = j * 128 + h * 64 + 8 * wave + prow;
            const int chunk = (lane & 7) ^ ((trow >> 1) & 7);
            const int m = m0 + trow;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int n = fast_div(mm, p.mul_hrwr, p.shr_hrwr);
            const int rem = mm - n * p.HrWr;
            const int r = fast_div(rem, p.mul_wr, p.shr_wr);
            const int c = rem - r * p.Wr;
            const int iy0 = ok ? r * p.sr : -16384;                   // a row beyond M fails every bounds test
            a_yx[h][j] = (int)(((unsigned)iy0 << 16) | ((unsigned)(c * p.sr) & 0xffffu));
            a_base[h][j] = ((((n * p.Hs + r * p.sr) * p.Ws + c * p.sr) * p.lda) + chunk * 8) * 2;
        }
#pragma unroll
        for (int j = 0; j < G::NB; ++j) {
            const int tcol = ((wave >> 2) + 2 * j) * 64 + h * 32 + 8 * (wave & 3) + prow;
            const int chunk = (lane & 7) ^ ((tcol >> 1) & 7);
            const int nn = n0 + tcol;
            b_voff[h][j] = (nn < p.Ncols) ? (nn * p.Cs + chunk * 8) * 2 : (int)OOB;
        }
    }
    // ---- staging cursor (scalar state, advanced incrementally): the k-tile whose units are being issued ----------
    const int kt_begin = blockIdx.z * p.tiles_per_split;
    const int nkt = min(p.ktiles, kt_begin + p.tiles_per_split) - kt_begin;
    int st_left = nkt, st_tb, st_oy, st_ox, st_kA, st_kB;           // st_kA / st_kB: scalar byte offsets of the DMA
    {
        const int tap = kt_begin / p.cchunks;
        const int c0 = (kt_begin - tap * p.cchunks) * TK;
        const int ta = tap / p.tap_nb;
        st_tb = tap - ta * p.tap_nb;
        st_oy = p.oy0 + ta * p.oys;
        st_ox = p.ox0 + st_tb * p.oxs;
        st_kA = c0 * 2;
        st_kB = ((p.w0 + ta * p.wa + st_tb * p.wb) * p.Ncols * p.Cs + c0) * 2;
    }
    const int cs2 = p.Cs * 2;
    int st_cleft = cs2 - st_kA;                                       // bytes of k left in this tap's channel run
    auto enter_tap = [&]() {
        const int shift = ((st_oy * p.Ws + st_ox) * p.lda) * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < G::NA; ++j) {
                const int iy = (a_yx[h][j] >> 16) + st_oy;
                const int ix = (int)(short)(a_yx[h][j] & 0xffff) + st_ox;
                const bool ok = ((unsigned)iy < (unsigned)p.Hs) & ((unsigned)ix < (unsigned)p.Ws);
                a_voff[h][j] = ok ? a_base[h][j] + shift : (int)OOB;
            }
    };
    int st_dead = 0;                                                   // 0, or OOB once no k-tile is left: OR-ed into every offset
    auto advance = [&]() {                                             // cursor -> next k-tile (all branches uniform)
        --st_left;
        st_kA += TK * 2;
        st_kB += TK * 2;
        st_cleft -= TK * 2;
        if (st_left <= 0) {
            st_dead = (int)OOB;                                        // every further piece is a zero fill of a free unit
        } else if (st_cleft <= 0) {                                    // next tap
            st_kA = 0;
            st_cleft = cs2;
            const int nc2 = p.Ncols * cs2;
            if (++st_tb == p.tap_nb) {
                st_tb = 0;
                st_oy += p.oys;
                st_ox = p.ox0;
                st_kB += (p.wa - (p.tap_nb - 1) * p.wb) * nc2 - cs2;
            } else {
                st_ox += p.oxs;
                st_kB += p.wb * nc2 - cs2;
            }
            enter_tap();
        }
    };
    // LDS destinations of this wave's pieces: one scalar base per region, the slot toggled by XOR, the rest immediates
    lds_char* const L = (lds_char*)smem;                               // LDS-address-space view of the one LDS object
    int dA = (8 * wave) * ROWB;                                        // byte offsets in L of the CURRENT slot
    int dB = G::B_OFF + (wave >> 2) * (2 * G::BSLOT) + (8 * (wave & 3)) * ROWB;
    int sA = G::ASLOT, sB = G::BSLOT;                                  // current slot -> other slot (sign flips per k-tile)
    auto issueA = [&](int other, auto H) {                             // `other`: literal 0 / 1 = current / other slot
        constexpr int h = decltype(H)::value;
        const int d = (other ? dA + sA : dA) + h * 64 * ROWB;
#pragma unroll
        for (int j = 0; j < G::NA; ++j) dma16l(rsA, a_voff[h][j] | st_dead, st_kA, L + (d + j * 2 * G::ASLOT));
    };
    auto issueB = [&](int other, auto H) {
        constexpr int h = decltype(H)::value;
        const int d = (other ? dB + sB : dB) + h * 32 * ROWB;
#pragma unroll
        for (int j = 0; j < G::NB; ++j) dma16l(rsB, b_voff[h][j] | st_dead, st_kB, L + (d + j * 4 * G::BSLOT));
    };

    // ---- fragment addresses: A row = l31 (+ mb*32) of wave row wr, B column = l31 (+ nb*32) of wave column wc, 8 consecutive
    // k = chunk 2 ks + half at slot chunk ^ ((row >> 1) & 7) ------------------------------------------------------------
    const int fsw = (l31 >> 1) & 7;
    int fa[4], fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int so = ((2 * ks + half) ^ fsw) * 16;
        fa[ks] = wr * (2 * G::ASLOT) + l31 * ROWB + so;
        fb[ks] = G::B_OFF + wc * (2 * G::BSLOT) + l31 * ROWB + so;
    }
    auto lds16 = [&](int off) -> bf16x8 { return *(const __attribute__((address_space(3))) bf16x8*)(L + off); };

    bf16x8 fA[2][4], fB0[4], fB1[4];
    auto rdA = [&](auto MH) {
        constexpr int mh = decltype(MH)::value;
#pragma unroll
        for (int mbi = 0; mbi < 2; ++mbi)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fA[mbi][ks] = lds16(fa[ks] + (mh * 64 + mbi * 32) * ROWB);
    };
    auto rdB = [&](auto NBK, bf16x8 (&f)[4]) {
        constexpr int nb = decltype(NBK)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[ks] = lds16(fb[ks] + nb * 32 * ROWB);
    };
    auto mma = [&](auto MH, auto NBK, const bf16x8 (&fbv)[4]) {
        constexpr int mh = decltype(MH)::value, nb = decltype(NBK)::value;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mbi = 0; mbi < 2; ++mbi)
                acc[2 * mh + mbi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbv[ks], fA[mbi][ks], acc[2 * mh + mbi][nb], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

#define LRD(x) x
#define LDMA(x) x
    QT_DECL
    // ---- prologue: k-tile 0 complete + {UA0, UB0, UB1} of k-tile 1 in flight ------------------------------------------
    enter_tap();
    issueA(0, I0{});
    issueB(0, I0{});
    issueB(0, I1{});
    issueA(0, I1{});
    advance();
    issueA(1, I0{});
    issueB(1, I0{});
    issueB(1, I1{});
    wait_vm<G::NA + 2 * G::NB>();
    q_barrier();
    if (grp == 1) q_barrier();                       // stagger: this group runs one barrier behind
    for (int t = 0; t < nkt; ++t) {
        // PA: rows mh0 x all 64 columns
        QT_START();
        LRD(rdB(I0{}, fB0));
        LRD(rdB(I1{}, fB1));
        __builtin_amdgcn_sched_barrier(0);
        LRD(rdA(I0{}));
        __builtin_amdgcn_sched_barrier(0);
        LDMA(issueA(1, I1{}));
        QT_SEG(0);                                                   // issue of 16 reads + NA pieces
        loop_wait_vm<G::VMC>();
        QT_SEG(1);                                                   // counted vmcnt
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // fragments home BEFORE the barrier (WAR rule above)
        QT_SEG(2);                                                   // fragments home
        loop_barrier();
        QT_SEG(3);                                                   // first barrier
        mma(I0{}, I0{}, fB0);
        mma(I0{}, I1{}, fB1);
        QT_SEG(4);                                                   // 16 MFMAs issued
        loop_barrier();
        QT_SEG(5);                                                   // second barrier
        // PB: rows mh1
        LRD(rdA(I1{}));
        __builtin_amdgcn_sched_barrier(0);
        advance();
        LDMA(issueA(0, I0{}));
        LDMA(issueB(0, I0{}));
        LDMA(issueB(0, I1{}));
        QT_SEG(6);                                                   // issue of 8 reads + cursor + NA + 2 NB pieces
        loop_wait_vm<G::VMC>();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        QT_SEG(7);                                                   // vmcnt + fragments home
        loop_barrier();
        QT_SEG(3);
        mma(I1{}, I0{}, fB0);
        mma(I1{}, I1{}, fB1);
        QT_SEG(4);
        loop_barrier();
        QT_SEG(5);
        // the other slot becomes the current one
        dA += sA;
        dB += sB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa[ks] += sA; fb[ks] += sB; }
        sA = -sA;
        sB = -sB;
    }
#undef LRD
#undef LDMA
    if (grp == 0) q_barrier();
    wait_vm<0>();                                    // zero fills issued past the last k-tile: drain before reusing the LDS
    __syncthreads();
#ifdef DPIG_TRACE
    const unsigned long long qt_loop_end = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue: four passes, pass mb stages rows wr*128 + mb*32 .. +31 of every wave row ---------------------------
    const bool lean = p.identity_rows && !p.res_cls && !p.replicate && p.nsplit == 1;
    const bool hr = p.res != nullptr, hm = p.mask != nullptr, h2 = p.D2 != nullptr, rpost = p.res_post != 0;
    const int kind = !lean ? 0 : ((!hr && !hm && !h2) ? 1 : ((hr && !rpost && !hm && !h2) ? 2 : ((!hr && hm && !h2) ? 3 : ((hr && rpost && !hm && h2) ? 4 : ((hr && !rpost && hm && !h2) ? 5 : 0)))));
    const float slope = (p.act == DPIG_ACT_NONE) ? 1.f : ((p.act == DPIG_ACT_RELU) ? 0.f : p.alpha);
    if (kind) {                                                       // (workgroup-uniform) the flag combinations the models produce
        const int row_base = m0 + wr * 128, cb0 = n0 + wc * 64;
        lds_char* W = (lds_char*)smem + wave * WEP_BYTES;             // this wave's own staging rows
        switch (kind) {
            case 1: q_epilogue_wave<false, false, false, false>(p, acc, W, row_base, cb0, lane, slope); break;   // bias + activation
            case 2: q_epilogue_wave<true, false, false, false>(p, acc, W, row_base, cb0, lane, slope); break;    // + residual before the activation
            case 3: q_epilogue_wave<false, false, true, false>(p, acc, W, row_base, cb0, lane, slope); break;    // dgrad * activation mask
            case 4: q_epilogue_wave<true, true, false, true>(p, acc, W, row_base, cb0, lane, slope); break;      // res-block tail
            default: q_epilogue_wave<true, false, true, false>(p, acc, W, row_base, cb0, lane, slope); break;    // (dgrad + accum) * mask
        }
    } else
    {
        // every other combination (class-indexed residual, 2 x 2 replication, split-K partials, ...): four passes through LDS,
        // pass mb stages the mb-th 32-pixel block of every wave as fp32 rows, then epi8 on 8 consecutive columns per thread
        float* Cs = reinterpret_cast<float*>(smem);
        constexpr int TPR = G::BN / 8, RPS = 512 / TPR;               // threads per staged row, rows per sweep
        const int c = (tid % TPR) * 8;
        const int col = n0 + c;
        const int rl0 = tid / TPR;
        const bool cok = col < p.Ncols;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
        if (p.bias && cok) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
            const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            if (mb) __syncthreads();
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q)                           // (transposed accumulators: 4 consecutive channels per run)
                    *reinterpret_cast<float4*>(&Cs[(wr * 32 + l31) * G::LDCQ + wc * 64 + nb * 32 + 8 * q + 4 * half]) =
                        make_float4(acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]);
            __syncthreads();
            if (!cok) continue;
#pragma unroll 2
            for (int it = 0; it < G::RP / RPS; ++it) {
                const int rl = rl0 + RPS * it;
                const int row = m0 + (rl >> 5) * 128 + mb * 32 + (rl & 31);
                if (row >= p.M) continue;
                const float4 v0 = *reinterpret_cast<const float4*>(&Cs[rl * G::LDCQ + c]);
                const float4 v1 = *reinterpret_cast<const float4*>(&Cs[rl * G::LDCQ + c + 4]);
                if (p.nsplit > 1) {
                    float* pp = p.partial + ((long)blockIdx.z * p.M + row) * p.Ncols + col;
                    *reinterpret_cast<float4*>(pp) = v0;
                    *reinterpret_cast<float4*>(pp + 4) = v1;
                    continue;
                }
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                epi8(p, row, col, v, bv);
            }
        }
    }
#ifdef DPIG_TRACE
    if ((int)blockIdx.x < 256 && lane == 0) {
        unsigned long long* o = dpig_bq_prof + ((int)blockIdx.x * 8 + wave) * 16;
        for (int i = 0; i < 8; ++i) o[i] = qt_sum[i];
        o[8] = (unsigned long long)nkt;
        o[9] = qt_loop_end - qt_begin;                               // prologue + k-loop
        o[10] = __builtin_amdgcn_s_memtime() - qt_loop_end;          // epilogue
    }
#endif
}

// ================================================================================================
// bhq_kernel: bq_kernel<2, 4> with bh_kernel's halo staging of the A operand (3 x 3 stride-1 windows: forward, and the stride-1 dgrad
// = the same window with flipped taps).  The 128 rows of a wave row are an 8 x 16 patch of one image (a workgroup = a 16 x 16 region);
// the patch WITH ITS HALO (10 x 18 pixels x 64 channels = 23 KB) is staged once per channel chunk and the nine taps read shifted
// windows of it, so the k order is (chunk, tap) and only the 32-KB filter tile streams per k-tile: a wave issues 1 + 4 DMA pieces per
// k-tile instead of 8 (the knock-out builds put the DMA stream at 21 % of the kernel), and the input is fetched once, not nine times.
// LDS: A = [wave row][chunk parity][184 rows x 128 B] (92 KB), B as bq_kernel (64 KB).  The halo of chunk c + 1 is fetched during
// chunk c, one piece per wave per k-tile of taps 0..5: piece tap * 8 + wave (< 46).  The per-phase wait is "everything but the four
// filter pieces just issued", so it does not care whether a halo piece was issued in the phase before.  Swizzle as bh_kernel: pixel (hy, hx) keeps its
// 16-byte chunk c at slot c ^ ((hx >> 1) & 7).
// ================================================================================================
struct HQ {
    static constexpr int P = 18, NPIX = 10 * 18, HROWS = 184, HALO_B = HROWS * ROWB, NPIECE = 23;
    static constexpr int BSLOT = 64 * ROWB, B_OFF = 2 * 2 * HALO_B, SMEM = B_OFF + 4 * 2 * BSLOT;
    static_assert(SMEM <= 163840 && 8 * WEP_BYTES <= SMEM, "LDS plan");
};
template <bool SLACK>
__global__ __launch_bounds__(512, 2) void bhq_kernel(const BGParams p) {
    __shared__ __attribute__((aligned(16))) char smem[HQ::SMEM];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int grp = wave >> 2;
    const int l31 = lane & 31, half = lane >> 5;

    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int n0 = nt * 256;
    // the images of the batch are one tall stack of N * Hs pixel rows; a tile is 16 of them (two wave rows of 8, each inside ONE image
    // because Hs % 8 == 0) x 16 columns: tiles need not align with images, only the halo's validity is image-local
    const int tyi = mt / p.tiles_x;
    const int g0 = tyi * 16, x0 = (mt - tyi * p.tiles_x) * 16;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B, p.b_bytes);
    lds_char* const L = (lds_char*)smem;

    // ---- halo DMA roles: in tap t < 6 this wave fetches piece id = 8 t + wave (< 46) = (wave row j, piece q): halo pixels 8 q .. 8 q + 7
    int h_voff[6], h_dst[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int id = (8 * t + wave) % 46;
        const int j = id / HQ::NPIECE, q = id - j * HQ::NPIECE;
        const int hp = 8 * q + (lane >> 3);
        const int hy = hp / HQ::P, hx = hp - hy * HQ::P;
        const int gj = g0 + 8 * j, yj = gj % p.Hs;               // this wave row's first pixel row: global, and inside its image
        const int y = yj - 1 + hy, x = x0 - 1 + hx;
        const bool ok = (hp < HQ::NPIX) & ((unsigned)y < (unsigned)p.Hs) & ((unsigned)x < (unsigned)p.Ws);
        const int g = (lane & 7) ^ ((hx >> 1) & 7);
        h_voff[t] = ok ? (((((gj - 1 + hy) * p.Ws) + x) * p.lda) + g * 8) * 2 : (int)OOB;
        h_dst[t] = (j * 2) * HQ::HALO_B + q * 8 * ROWB;
    }
    // ---- filter DMA roles (as bq_kernel<2, 4>): unit h, piece j: columns h*32 + 8*(wave & 3) .. +7 of wave column (wave >> 2) + 2j
    const int prow = lane >> 3;
    int b_voff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tcol = ((wave >> 2) + 2 * j) * 64 + h * 32 + 8 * (wave & 3) + prow;
            const int chunk = (lane & 7) ^ ((tcol >> 1) & 7);
            const int nn = n0 + tcol;
            b_voff[h][j] = (nn < p.Ncols) ? (nn * p.Cs + chunk * 8) * 2 : (int)OOB;
        }
    // ---- cursors.  k order (chunk, tap).  Staging cursor (filter tiles, two k-tiles ahead of the MFMAs) and compute cursor.
    const int nkt = p.ktiles;
    const int nch = p.cchunks;
    const int cs2 = p.Cs * 2, nc2 = p.Ncols * cs2;
    int st_left = nkt, st_ta = 0, st_tb = 0;
    int st_kB = (p.w0 * p.Ncols * p.Cs) * 2;                           // tap (0, 0) of chunk 0
    int st_dead = 0;
    auto advance = [&]() {
        --st_left;
        if (st_left <= 0) {
            st_dead = (int)OOB;
        } else if (++st_tb == 3) {
            st_tb = 0;
            if (++st_ta == 3) {                                        // first tap of the next chunk
                st_ta = 0;
                st_kB += TK * 2 - (2 * p.wa + 2 * p.wb) * nc2;
            } else {
                st_kB += (p.wa - 2 * p.wb) * nc2;
            }
        } else {
            st_kB += p.wb * nc2;
        }
    };
    int dB = HQ::B_OFF + (wave >> 2) * (2 * HQ::BSLOT) + (8 * (wave & 3)) * ROWB;
    int sB = HQ::BSLOT;
    auto issueB = [&](int other, auto H) {
        constexpr int h = decltype(H)::value;
        const int d = (other ? dB + sB : dB) + h * 32 * ROWB;
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16l(rsB, b_voff[h][j] | st_dead, st_kB, L + (d + j * 4 * HQ::BSLOT));
    };
    auto issueH = [&](int t, int chunk) {                              // piece of tap slot t (a literal) for the halo of `chunk`
        if (8 * t + wave >= 2 * HQ::NPIECE) return;                    // (wave-uniform) 46 pieces: taps 0..4 all waves, tap 5 waves 0..5
        const int dead = chunk < nch ? 0 : (int)OOB;
        dma16l(rsA, h_voff[t] | dead, chunk * (TK * 2), L + (h_dst[t] + (chunk & 1) * HQ::HALO_B));
    };

    // ---- fragment addresses ---------------------------------------------------------------------------------------------
    const int fsw = (l31 >> 1) & 7;
    int fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = HQ::B_OFF + wc * (2 * HQ::BSLOT) + l31 * ROWB + (((2 * ks + half) ^ fsw) * 16);
    const int f_tx = l31 & 15, f_tyl = l31 >> 4;
    int fa[4];
    auto set_fa = [&](int ta, int tb, int chunk) {                     // A fragment addresses of tap (ta, tb) of `chunk`'s halo
        const int dyy = 1 + p.oy0 + ta * p.oys, dxx = 1 + p.ox0 + tb * p.oxs;
        const int hx = f_tx + dxx;
        const int sw = (hx >> 1) & 7;
        const int base = (wr * 2 + (chunk & 1)) * HQ::HALO_B + ((dyy + f_tyl) * HQ::P + hx) * ROWB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[ks] = base + (((2 * ks + half) ^ sw) << 4);
    };
    auto lds16 = [&](int off) -> bf16x8 { return *(const __attribute__((address_space(3))) bf16x8*)(L + off); };
    bf16x8 fA[2][4], fB0[4], fB1[4];
    auto rdA = [&](auto MH) {
        constexpr int mh = decltype(MH)::value;
#pragma unroll
        for (int mbi = 0; mbi < 2; ++mbi)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fA[mbi][ks] = lds16(fa[ks] + (mh * 4 + mbi * 2) * HQ::P * ROWB);
    };
    auto rdB = [&](auto NBK, bf16x8 (&f)[4]) {
        constexpr int nb = decltype(NBK)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[ks] = lds16(fb[ks] + nb * 32 * ROWB);
    };
    auto mma = [&](auto MH, auto NBK, const bf16x8 (&fbv)[4]) {
        constexpr int mh = decltype(MH)::value, nb = decltype(NBK)::value;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mbi = 0; mbi < 2; ++mbi)
                acc[2 * mh + mbi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbv[ks], fA[mbi][ks], acc[2 * mh + mbi][nb], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- prologue: the halo of chunk 0 (46 pieces over the 8 waves), filter tiles 0 and 1 ----------------------------
#pragma unroll
    for (int t = 0; t < 6; ++t) issueH(t, 0);
    issueB(0, I0{});
    issueB(0, I1{});
    advance();
    issueB(1, I0{});
    issueB(1, I1{});
    wait_vm<4>();                                    // halo 0 + filter tile 0 home (tile 1 may be in flight)
    q_barrier();
    if (grp == 1) q_barrier();                       // stagger: this group runs one barrier behind
    for (int c = 0; c < nch; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            set_fa(tap / 3, tap % 3, c);
            // PA: rows mh0 x all 64 columns
            rdB(I0{}, fB0);
            rdB(I1{}, fB1);
            __builtin_amdgcn_sched_barrier(0);
            rdA(I0{});
            __builtin_amdgcn_sched_barrier(0);
            if (tap < 6) issueH(tap, c + 1);         // next chunk's halo: one piece per wave in taps 0..5
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            q_barrier();
            mma(I0{}, I0{}, fB0);
            mma(I0{}, I1{}, fB1);
            q_barrier();
            // PB: rows mh1
            rdA(I1{});
            __builtin_amdgcn_sched_barrier(0);
            advance();
            issueB(0, I0{});                         // filter tile t + 2 into the slot tile t was read from
            issueB(0, I1{});
            // filter tile t + 1 must be home; this k-tile's halo piece (issued in PA, needed only by the next chunk) may stay in flight
            if (SLACK && tap < 5) wait_vm<5>();
            else if (SLACK && tap == 5 && wave < 6) wait_vm<5>();
            else wait_vm<4>();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            q_barrier();
            mma(I1{}, I0{}, fB0);
            mma(I1{}, I1{}, fB1);
            q_barrier();
            // next k-tile: the other filter slot
            dB += sB;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb[ks] += sB;
            sB = -sB;
        }
    }
    if (grp == 0) q_barrier();
    wait_vm<0>();
    __syncthreads();

    // ---- epilogue: the wave-private transposing epilogue of bq_kernel; rows are patch pixels -----------------------------------
    const bool hr = p.res != nullptr, hm = p.mask != nullptr, h2 = p.D2 != nullptr, rpost = p.res_post != 0;
    const int kind = (!hr && !hm && !h2) ? 1 : ((hr && !rpost && !hm && !h2) ? 2 : ((!hr && hm && !h2) ? 3 : ((hr && rpost && !hm && h2) ? 4 : 5)));
    const float slope = (p.act == DPIG_ACT_NONE) ? 1.f : ((p.act == DPIG_ACT_RELU) ? 0.f : p.alpha);
    const int row_base = (g0 + 8 * wr) * p.Ws + x0, cb0 = n0 + wc * 64;
    lds_char* W = L + wave * WEP_BYTES;
    switch (kind) {
        case 1: q_epilogue_wave<false, false, false, false>(p, acc, W, row_base, cb0, lane, slope, p.Ws); break;
        case 2: q_epilogue_wave<true, false, false, false>(p, acc, W, row_base, cb0, lane, slope, p.Ws); break;
        case 3: q_epilogue_wave<false, false, true, false>(p, acc, W, row_base, cb0, lane, slope, p.Ws); break;
        case 4: q_epilogue_wave<true, true, false, true>(p, acc, W, row_base, cb0, lane, slope, p.Ws); break;
        default: q_epilogue_wave<true, false, true, false>(p, acc, W, row_base, cb0, lane, slope, p.Ws); break;
    }
}

// ================================================================================================
// bhq32_kernel: the halo-staged schedule at BK = 32 for the 128-column layers (C -> 128 at full resolution: the encoder stems / residual
// blocks of models.py:396-400, 420-426) that bq_kernel<4, 2> serves without halo staging.
// 8 waves as 4 (8 x 16-pixel patches) x 2 (64 channels): a 32 x 16-pixel patch x 128 channels per workgroup; LDS rows of 64 B; per
// 32-channel chunk the 10 x 18 halo of every wave row at a pitch of 20 pixels (13 pieces of 16 pixels, two chunk slots = 104 KB), 16-byte
// slot s of pixel hx holds granule s ^ ((hx >> 2) & 3); filter tiles [128 columns][32 k] = 8 KB in FOUR slots (prefetch distance 3), one
// piece per wave per k-tile; a k-tile is one phase {12 fragment reads; filter piece of tile t + 3, halo piece of the next chunk in taps
// 0..6; counted vmcnt {3, 4, 5, 5, 5, 5, 5, 4, 3}[tap]; barrier; 16 MFMAs; barrier}, groups staggered by one barrier.  The wave tile
// and the D^T accumulators are bhq_kernel's, so the epilogue is the same.  scripts/ubench/bhq32_probe.hip is the stand-alone form
// (self-check against a direct convolution); its index math is also emulated on the host (scripts/ubench/emulate_bhq32.py).
// Measured (round 4, profiles/r04_bhq32_first_run.txt): 8 x 256 x 256 x 128 -> 128 forward 914 -> 1021 TFLOP/s, dgrad 676 -> 843 against
// the 128 x 128 halo kernel on the same box; DPIG_BF16_QH32=0 is the A/B switch.
// ================================================================================================
struct H32 {
    static constexpr int RB = 64, HP = 20, NPX = 10 * HP, NPIECE = 13, WR_B = NPIECE * 16 * RB, HSLOT = 4 * WR_B;
    static constexpr int B_OFF = 2 * HSLOT, BSL = 128 * RB, PAD_OFF = B_OFF + 4 * BSL, SMEM = PAD_OFF + 1024;
    static_assert(SMEM <= 163840 && 8 * WEP_BYTES <= SMEM, "LDS plan");
};
__global__ __launch_bounds__(512, 2) void bhq32_kernel(const BGParams p) {
    __shared__ __attribute__((aligned(16))) char smem[H32::SMEM];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int grp = wave >> 2;
    const int l31 = lane & 31, half = lane >> 5;

    const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int n0 = nt * 128;
    const int tyi = mt / p.tiles_x;                              // (tall-stack tiling as in bhq_kernel: 32 pixel rows = four wave rows of 8)
    const int g0 = tyi * 32, x0 = (mt - tyi * p.tiles_x) * 16;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B, p.b_bytes);
    lds_char* const L = (lds_char*)smem;

    // halo DMA roles: in tap slot t <= 6 this wave fetches piece id = 8 t + wave (ids >= 52 are dead: out-of-range source, scratch
    // destination, so that every wave issues the same number of pieces); lane -> (pixel slot 16 q + lane / 4, 16-byte slot lane % 4)
    int h_voff[7], h_dst[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        const int id = 8 * t + wave;
        const bool live = id < 4 * H32::NPIECE;
        const int j = live ? id / H32::NPIECE : 0, q = live ? id - j * H32::NPIECE : 0;
        const int hp = 16 * q + (lane >> 2);
        const int hy = hp / H32::HP, hx = hp - hy * H32::HP;
        const int gj = g0 + 8 * j, yj = gj % p.Hs;
        const int y = yj - 1 + hy, x = x0 - 1 + hx;
        const bool ok = live & (hp < H32::NPX) & (hx < 18) & ((unsigned)y < (unsigned)p.Hs) & ((unsigned)x < (unsigned)p.Ws);
        const int g = (lane & 3) ^ ((hx >> 2) & 3);
        h_voff[t] = ok ? (((((gj - 1 + hy) * p.Ws) + x) * p.lda) + g * 8) * 2 : (int)OOB;
        h_dst[t] = live ? (j * H32::WR_B + q * 1024) : -1;
    }
    int b_voff;                                              // filter DMA role: columns 16 wave .. 16 wave + 15 of the 128
    {
        const int n = 16 * wave + (lane >> 2);
        const int g = (lane & 3) ^ ((n >> 2) & 3);
        b_voff = (n0 + n < p.Ncols) ? ((n0 + n) * p.Cs + g * 8) * 2 : (int)OOB;
    }
    const int nch = p.Cs >> 5, nkt = 9 * nch;
    const int tapB = p.Ncols * p.Cs * 2;                     // bytes between two filter slabs
    auto issueB = [&](int t) {                               // filter k-tile t (chunk t / 9, tap t % 9) into slot t % 4
        const int c = t / 9, tap = t - 9 * c;
        const int ta = tap / 3, tb = tap - 3 * ta;
        const int dead = t < nkt ? 0 : (int)OOB;
        dma16l(rsB, b_voff | dead, (p.w0 + ta * p.wa + tb * p.wb) * tapB + c * 64, L + (H32::B_OFF + (t & 3) * H32::BSL + wave * 1024));
    };
    auto issueH = [&](int t, int chunk) {                    // halo piece of tap slot t (a literal) for `chunk`
        const int dead = chunk < nch ? 0 : (int)OOB;
        const int dst = h_dst[t] >= 0 ? (chunk & 1) * H32::HSLOT + h_dst[t] : H32::PAD_OFF;
        dma16l(rsA, h_voff[t] | dead, chunk * 64, L + dst);
    };

    const int f_tx = l31 & 15, f_tyl = l31 >> 4;
    int fb[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int n = wc * 64 + 32 * nb + l31;
            fb[nb][ks] = H32::B_OFF + n * H32::RB + (((2 * ks + half) ^ ((n >> 2) & 3)) << 4);
        }
    auto lds16 = [&](int off) -> bf16x8 { return *(const __attribute__((address_space(3))) bf16x8*)(L + off); };
    bf16x8 fA[4][2], fB[2][2];
    auto rdA = [&](int ta, int tb, int chunk) {              // tap (ta, tb) literals; halo coordinates = image coordinates + 1
        const int dyy = 1 + p.oy0 + ta * p.oys, dxx = 1 + p.ox0 + tb * p.oxs;
        const int hx = f_tx + dxx;
        const int sw = (hx >> 2) & 3;
        const int base = (chunk & 1) * H32::HSLOT + wr * H32::WR_B + ((dyy + f_tyl) * H32::HP + hx) * H32::RB;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fA[mb][ks] = lds16(base + (2 * mb) * H32::HP * H32::RB + (((2 * ks + half) ^ sw) << 4));
    };
    auto rdB = [&](int slot) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fB[nb][ks] = lds16(fb[nb][ks] + slot * H32::BSL);
    };
    auto mma = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fB[nb][ks], fA[mb][ks], acc[mb][nb], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // prologue: halo of chunk 0 (7 pieces per wave, dead ones included), filter tiles 0, 1, 2
#pragma unroll
    for (int t = 0; t < 7; ++t) issueH(t, 0);
    issueB(0);
    issueB(1);
    issueB(2);
    wait_vm<2>();                                    // halo 0 + filter tile 0 home
    q_barrier();
    if (grp == 1) q_barrier();                       // stagger: this group runs one barrier behind
    int t = 0;
    for (int c = 0; c < nch; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++t) {
            rdB(t & 3);
            __builtin_amdgcn_sched_barrier(0);
            rdA(tap / 3, tap % 3, c);
            __builtin_amdgcn_sched_barrier(0);
            issueB(t + 3);                           // into the slot tile t - 1 was read from (both groups are past those reads)
            if (tap < 7) issueH(tap, c + 1);
            __builtin_amdgcn_sched_barrier(0);
            // filter tile t + 1 must be home: the pieces younger than it (issue order per k-tile: filter, halo)
            if (tap == 0 || tap == 8) wait_vm<3>();
            else if (tap == 1 || tap == 7) wait_vm<4>();
            else wait_vm<5>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            q_barrier();
            mma();
            q_barrier();
        }
    }
    if (grp == 0) q_barrier();
    wait_vm<0>();
    __syncthreads();

    // epilogue: bhq_kernel's (the wave tile and the accumulator layout are the same)
    const bool hr = p.res != nullptr, hm = p.mask != nullptr, h2 = p.D2 != nullptr, rpost = p.res_post != 0;
    const int kind = (!hr && !hm && !h2) ? 1 : ((hr && !rpost && !hm && !h2) ? 2 : ((!hr && hm && !h2) ? 3 : ((hr && rpost && !hm && h2) ? 4 : 5)));
    const float slope = (p.act == DPIG_ACT_NONE) ? 1.f : ((p.act == DPIG_ACT_RELU) ? 0.f : p.alpha);
    const int row_base = (g0 + 8 * wr) * p.Ws + x0, cb0 = n0 + wc * 64;
    lds_char* W = L + wave * WEP_BYTES;
    switch (kind) {
        case 1: q_epilogue_wave<false, false, false, false>(p, acc, W, row_base, cb0, lane, slope, p.Ws); break;
        case 2: q_epilogue_wave<true, false, false, false>(p, acc, W, row_base, cb0, lane, slope, p.Ws); break;
        case 3: q_epilogue_wave<false, false, true, false>(p, acc, W, row_base, cb0, lane, slope, p.Ws); break;
        case 4: q_epilogue_wave<true, true, false, true>(p, acc, W, row_base, cb0, lane, slope, p.Ws); break;
        default: q_epilogue_wave<true, false, true, false>(p, acc, W, row_base, cb0, lane, slope, p.Ws); break;
    }
}

// ================================================================================================
// host side
// ================================================================================================
static int g_q_mode = -1;      // 0 off, 1 automatic, 2 whenever the layer is legal for the kernel (tests)
static int g_q_variant = 0;    // 0 automatic, 1 = 256 x 256, 2 = 512 x 128 (measurements)
static double g_q_mineff = 1.04;     // required advantage over the 128 x 128 family's fill (DPIG_BF16_Q_MINEFF)

static void q_init() {
    if (g_q_mode >= 0) return;
    const char* e = getenv("DPIG_BF16_Q");
    g_q_mode = e ? atoi(e) : 1;
    const char* v = getenv("DPIG_BF16_Q_VARIANT");
    g_q_variant = v ? atoi(v) : 0;
    const char* m = getenv("DPIG_BF16_Q_MINEFF");
    if (m) g_q_mineff = atof(m);
}

static int g_q_halo = []() { const char* e = getenv("DPIG_BF16_QH"); return e ? atoi(e) : 1; }();     // halo-staged variant (A/B switch)
static int g_q_nohalo_bonus = []() { const char* e = getenv("DPIG_BF16_Q_NOHALO"); return e ? atoi(e) : 1; }();   // (A/B switch of the hint below)
// 3 x 3 stride-1 window (forward / flipped-tap dgrad) on whole 16 x 16 regions with one of the five fused epilogues
static bool bhq_eligible(const BGParams& p) {
    if (p.ntaps != 9 || p.tap_nb != 3 || p.sr != 1 || !p.identity_rows || p.replicate || p.res_cls || p.stats) return false;
    if (p.Hr != p.Hs || p.Wr != p.Ws || p.oys * p.oys != 1 || p.oxs * p.oxs != 1) return false;
    if (1 + p.oy0 < 0 || 1 + p.oy0 + 2 * p.oys < 0 || 1 + p.oy0 > 2 || 1 + p.oy0 + 2 * p.oys > 2) return false;
    if (1 + p.ox0 < 0 || 1 + p.ox0 + 2 * p.oxs < 0 || 1 + p.ox0 > 2 || 1 + p.ox0 + 2 * p.oxs > 2) return false;
    if ((p.Hs & 7) || (p.Ws & 15) || (p.Cs % TK) || p.M % (p.Hs * p.Ws) || ((p.M / p.Ws) & 15)) return false;   // 8-row wave patches inside images; 16-row tiles over the stack
    const bool hr = p.res != nullptr, hm = p.mask != nullptr, h2 = p.D2 != nullptr, rpost = p.res_post != 0;
    const bool known = (!hr && !hm && !h2) || (hr && !rpost && !hm && !h2) || (!hr && hm && !h2) || (hr && rpost && !hm && h2) || (hr && !rpost && hm && !h2);
    return known;
}

static int g_q_halo32 = []() { const char* e = getenv("DPIG_BF16_QH32"); return e ? atoi(e) : 1; }();   // halo-staged 512 x 128 variant (A/B switch)
static bool bhq32_eligible(const BGParams& p) { return bhq_eligible(p) && !((p.M / p.Ws) & 31); }

// Fraction of the launched MFMA work that is real when the tiles of bm x bn run `slots` at a time in whole rounds.
static double q_eff(long M, long N, int bm, int bn, int slots) {
    const long tiles = (long)cdiv(M, bm) * cdiv(N, bn);
    const long rounds = (tiles + slots - 1) / slots;
    return (double)M * (double)N / ((double)rounds * slots * bm * bn);
}

// 1 = launched on the large-tile kernel, 0 = not this layer's case (the caller continues with the 128 x 128 kernels), < 0 = error
int bq_try(BGParams& p, hipStream_t st) {
    q_init();
    if (!g_q_mode) return 0;
    if (p.stats || (p.Cs % TK) || p.Hs >= 16384 || p.Ws >= 16384) return 0;
    const int ktiles = p.ntaps * (p.Cs / TK);
    if (ktiles < 2) return 0;
    // Selection (automatic mode), from the A/B of scripts/bench_conv_bf16q.py (profiles/r03_conv_bf16_tile_ab.txt): a full
    // CU runs the 256 x 256 kernel ~1.15x and the 512 x 128 kernel ~1.04x as fast as two 128 x 128 workgroups; what decides
    // is how well each tile grid fills whole rounds of the chip (256 slots here, 512 there).
    const bool halo = g_q_halo && bhq_eligible(p);       // (the halo-staged 256 x 256 kernel: another ~5 %)
    const bool halo32 = g_q_halo32 && bhq32_eligible(p);   // (the halo-staged 512 x 128 kernel: ~1.12x the 128 x 128 halo kernel)
    const double e1 = q_eff(p.M, p.Ncols, 256, 256, kNumCU) * (halo ? 1.21 : 1.15);
    const double e2 = q_eff(p.M, p.Ncols, 512, 128, kNumCU) * (halo32 ? 1.12 : 1.04);
    int variant = g_q_variant ? g_q_variant : (e2 > e1 ? 2 : 1);
    if (g_q_mode == 1) {
        if (p.nsplit > 1) return 0;                      // the split-K plan of the 128 x 128 family wins on small layers
        // the 128 x 128 family's tap-major kernel (layers its halo-patch kernel does not take: 12 x 12 / 6 x 6 maps, stride 2, 5 x 5) runs at
        // ~0.7 of the halo-patch kernel's rate (profiles/r03_conv_bf16_tile_ab.txt: 24 x 24 C256 736 vs 1148 TFLOP/s on the 256 x 256 tile)
        const double e0 = q_eff(p.M, p.Ncols, 128, 128, 2 * kNumCU) * ((p.halo128 || !g_q_nohalo_bonus) ? 1.0 : 0.72);
        if ((variant == 1 ? e1 : e2) < e0 * g_q_mineff) return 0;
        if (variant == 2 && p.Ncols <= 128 && q_eff(p.M, p.Ncols, 512, 128, kNumCU) < 0.95) return 0;   // 128-column layers: only on full rounds
    }
    const int bm = variant == 1 ? 256 : 512, bn = variant == 1 ? 256 : 128;
    BGParams q = p;
    q.mtiles = cdiv(p.M, bm);
    q.ntiles = cdiv(p.Ncols, bn);
    q.cchunks = p.Cs / TK;
    q.ktiles = ktiles;
    q.nsplit = 1;
    q.tiles_per_split = ktiles;
    dim3 grid(q.mtiles * q.ntiles, 1, 1), block(512);
    if (variant == 1 && halo) {
        q.tiles_x = p.Ws / 16; q.tiles_y = (p.M / p.Ws) / 16;             // tiles over the stack of all images' pixel rows
        q.mtiles = q.tiles_x * q.tiles_y;
        dim3 hgrid(q.mtiles * q.ntiles, 1, 1);
        if (g_q_halo == 2) hipLaunchKernelGGL(bhq_kernel<true>, hgrid, block, 0, st, q);
        else hipLaunchKernelGGL(bhq_kernel<false>, hgrid, block, 0, st, q);
        const int rch = check_launch("bhq_kernel");
        return rch ? rch : 1;
    }
    if (variant == 2 && halo32) {
        q.tiles_x = p.Ws / 16; q.tiles_y = (p.M / p.Ws) / 32;
        q.mtiles = q.tiles_x * q.tiles_y;
        dim3 hgrid(q.mtiles * q.ntiles, 1, 1);
        hipLaunchKernelGGL(bhq32_kernel, hgrid, block, 0, st, q);
        const int rch = check_launch("bhq32_kernel");
        return rch ? rch : 1;
    }
    if (variant == 1) hipLaunchKernelGGL((bq_kernel<2, 4>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((bq_kernel<4, 2>), grid, block, 0, st, q);
    const int rc = check_launch("bq_kernel");
    return rc ? rc : 1;
}

}  // namespace bfk
}  // namespace dpig

// Large-tile kernel selection (0 = never, 1 = automatic [default; DPIG_BF16_Q], 2 = whenever legal) and tile variant
// (0 = automatic, 1 = 256 x 256, 2 = 512 x 128): for tests and measurements.
extern "C" int dpig_conv_bf16_set_large_tile(int mode, int variant) {
    if (mode < 0 || mode > 2 || variant < 0 || variant > 2) return dpig::fail(DPIG_EINVAL, "large-tile mode / variant out of range");
    dpig::bfk::q_init();
    dpig::bfk::g_q_mode = mode;
    dpig::bfk::g_q_variant = variant;
    return DPIG_OK;
}
