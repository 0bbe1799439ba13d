// Winograd F(4x4, 3x3) convolution on the fp32 matrix pipe (gfx950): the large-map companion of dpig_conv_wino.hip's F(2x2, 3x3).
//
// A 4 x 4 output tile of a 3 x 3 stride-1 SAME conv comes from a 6 x 6 input patch with 36 multiplies per (input channel, output
// channel) instead of 144:  Y = A^T [ (G g G^T) o (B^T d B) ] A  with the matrices of Lavin & Gray (interpolation points 0, +-1, +-2,
// inf) -- 4x fewer matrix-pipe FLOPs than the direct kernels, 1.78x fewer than F(2x2, 3x3).  The price is numerical (the transforms
// multiply by up to 8 / 5 / 1/24: errors of a few 1e-6 of the largest activation instead of a few 1e-7; scripts/f43_numerics.py holds
// every golden activation of the full-width model to 6.2e-6 with this form on all maps whose sides are multiples of 4, bar 1e-4) and
// structural: 36 position GEMMs  M_p[tile, k] = sum_c V_p[tile, c] U_p[c, k]  need 36 accumulator blocks, so a workgroup owns 32 tiles
// (512 output pixels, a block of BW x 32 / BW tiles, BW = 4, 2 or 3, on the stack of all images' tile rows) x 64 output channels:
//   * eight waves = 4 position groups (the 3 x 3 sub-squares of the 6 x 6 position grid) x 2 channel halves; a wave holds nine
//     32 x 32 accumulator blocks of v_mfma_f32_32x32x2_f32 (144 registers), two waves per SIMD, one workgroup per CU (persistent);
//   * no two waves share a filter fragment (one tile block per workgroup), so the transformed filter U never touches LDS: the image
//     dpig_wino4_filter_transform leaves is in FRAGMENT order ([column block][chunk][position][channel half][lane][4 floats]) and a
//     wave's fragment of (position, chunk) is ONE coalesced 1-KB buffer load straight into the MFMA's A registers, re-issued for chunk
//     c + 1 the moment position p of chunk c has been multiplied (a full chunk, ~4600 cycles, of latency cover; 36 registers);
//   * the block's raw pixels of a chunk of 8 input channels ((4 BW + 2) x (128 / BW + 2) x 32 bytes) are gathered once by LDS-DMA two
//     chunks ahead; B^T d B runs in two wave-private passes over LDS: pass A, thread = (tile, channel quad, patch column), transforms
//     its column of 6 rows and writes a scratch row; pass B, thread = (tile, channel quad, transform row), reads that row's 6 columns,
//     transforms them and writes six 16-byte rows of V.  (One thread per full 6 x 6 patch would hold 36 float4; one thread per
//     transform row would read 4 patch rows = 24 float4 from LDS instead of 6 + 6.)  Six of every eight lanes work in both passes;
//   * the output transform A^T M A is linear in the positions: each wave transforms its own 3 x 3 sub-square (register arithmetic
//     with wave-uniform coefficients) into a partial 4 x 4 tile; the four partials meet in LDS one output row at a time (4 x 32 KB
//     staging) on their way into the family's fused epilogue (bias, activation, residual, second output, dgrad's (. + accum) act').
// dgrad is the same kernel on the image of the rotated / transposed filter, as in the F(2x2, 3x3) family.  The filter gradient stays on
// F(3x3, 2x2) / the direct kernels.
//
// LDS (154 KB): raw[2] (21 KB each) | V[2] (36 positions x (32 rows x 32 B + 16), 36.6 KB each) | scratch (8 waves x 4.9 KB) | zero slot;
// the staging images of the output transform reuse the front of it.  V's position planes are 1040 bytes apart so that the six
// transform rows a 16-lane store group covers (6 planes = 6240 bytes = 24 banks apart) fall on disjoint banks; the scratch strides
// (96 bytes per row, 624 per item) keep both passes conflict-free (16-lane groups: two items x six lanes).
#include "dpig_wino_common.h"

namespace dpig {
namespace wino4 {

using wino::WParams;
using wino::lds_char;
using wino::lds_void;
using wino::OOB;
using wino::fast_div;
using wino::make_rsrc;
using wino::wait_vm;
using wino::epi4;

constexpr int KB = 64;                        // output channels per workgroup
constexpr int CH = 8;                         // reduction channels per chunk
constexpr int TB = 32;                        // 4 x 4 output tiles per workgroup
constexpr int PS = TB * CH * 4 + 16;          // bytes between position planes of V
constexpr int VB = 36 * PS;                   // one V buffer
constexpr int RAW_PIECES = 21, RAWB = RAW_PIECES * 1024;
constexpr int T_XS = 96, T_IT = 624, T_WAVE = 8 * T_IT;
constexpr int RAW_OFF = 0, V_OFF = 2 * RAWB, T_OFF = V_OFF + 2 * VB, ZERO_OFF = T_OFF + 8 * T_WAVE, SMEM = ZERO_OFF + 64;
constexpr int EP_ROW = KB * 4 + 16;           // staging: bytes per (partial, pixel column, tile) row of 64 channels (+16: conflict-free stores)
constexpr int UPOS = 2048;                    // bytes of one (chunk, position) of the filter image: 64 channels x 8
constexpr int UCHUNK = 36 * UPOS;
static_assert(16 * TB * EP_ROW <= ZERO_OFF && SMEM <= 163840 && (V_OFF % 16) == 0 && (T_OFF % 16) == 0, "LDS plan");

typedef __attribute__((address_space(3))) f32x4 lds_f4;
typedef const __attribute__((address_space(3))) f32x4 lds_cf4;

// B^T of F(4, 3) applied along one axis:  [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1],
// in three parts of two results each (the k-loop spreads them over three of its steps); every result leaves through
// `out(index, value)` as soon as it exists (the callers store it: no second set of six registers)
// (written on register pairs; the compiler's post-RA peephole unpacks most of them -- v_pk_fma_f32 -> 2 x v_fma_f32 -- in the shadow of
// the MFMAs.  Forced packed through inline asm (48 instead of 84 vector instructions per chunk) the k-loop takes the SAME 5890 cycles:
// fp32 vector arithmetic costs matrix-pipe time by its lane-operations, not its instruction count -- the fp32 MFMA and packed fp32
// vector peaks are the same 157 TFLOP/s.  scripts/ko_wino4.sh: the transforms' arithmetic is 970 of the loop's 1100 cycles over the
// bare MFMAs, their LDS reads and stores 60, filter loads + raw gather 75)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 fma4(f32x4 a, float b, f32x4 c) {
    const f32x2_t bb = {b, b};
    const f32x2_t lo = __builtin_elementwise_fma(f32x2_t{a[0], a[1]}, bb, f32x2_t{c[0], c[1]});
    const f32x2_t hi = __builtin_elementwise_fma(f32x2_t{a[2], a[3]}, bb, f32x2_t{c[2], c[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 c) {
    const f32x2_t lo = f32x2_t{a[0], a[1]} + f32x2_t{c[0], c[1]}, hi = f32x2_t{a[2], a[3]} + f32x2_t{c[2], c[3]};
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 c) {
    const f32x2_t lo = f32x2_t{a[0], a[1]} - f32x2_t{c[0], c[1]}, hi = f32x2_t{a[2], a[3]} - f32x2_t{c[2], c[3]};
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
template <int PART, class F>
__device__ __forceinline__ void bt6(const f32x4 (&d)[6], F&& out) {
    if (PART == 0) {
        out(0, fma4(d[0], 4.f, fma4(d[2], -5.f, d[4])));
        out(5, fma4(d[1], 4.f, fma4(d[3], -5.f, d[5])));
    } else if (PART == 1) {
        const f32x4 a = fma4(d[2], -4.f, d[4]);
        const f32x4 b = fma4(d[1], -4.f, d[3]);
        out(1, add4(a, b));
        out(2, sub4(a, b));
    } else {
        const f32x4 c = sub4(d[4], d[2]);
        const f32x4 e = sub4(d[3], d[1]);
        out(3, fma4(e, 2.f, c));
        out(4, fma4(e, -2.f, c));
    }
}

// A^T of F(4, 3): [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].  The output transform of one position group (rows
// 3 PA .. 3 PA + 2, columns 3 PB .. 3 PB + 2 of the 6 x 6 grid) for output row Y, on PAIRS of accumulator elements (v_pk_* instructions)
// with the coefficients as compile-time integers: zeros vanish, +-1 are adds (the generic form, 84 multiply-adds per element with
// wave-uniform coefficients, was 1344 vector instructions per wave and most of the phase's ~19 k cycles).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ATI[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
template <int K>
__device__ __forceinline__ void mac2(f32x2& v, bool& have, f32x2 a) {
    if constexpr (K != 0) {
        if (!have) {
            v = (K == 1) ? a : (float)K * a;
            have = true;
        } else if constexpr (K == 1) v += a;
        else if constexpr (K == -1) v -= a;
        else v = __builtin_elementwise_fma(a, f32x2{(float)K, (float)K}, v);
    }
}
template <int K0, int K1, int K2>
__device__ __forceinline__ f32x2 lin3(f32x2 a0, f32x2 a1, f32x2 a2) {
    f32x2 v = {0.f, 0.f};
    bool have = false;
    mac2<K0>(v, have, a0);
    mac2<K1>(v, have, a1);
    mac2<K2>(v, have, a2);
    return v;
}
// partial of output row Y, columns 0..3, for accumulator elements 4 g .. 4 g + 3 of the wave's nine positions: P[x] (16 bytes along channels)
template <int PA, int PB, int Y>
__device__ __forceinline__ void out_row_group(const f32x16 (&acc)[9], int g, f32x4 (&P)[4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e0 = 4 * g + 2 * h;
        f32x2 r[3];
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
            r[jj] = lin3<ATI[Y][3 * PA], ATI[Y][3 * PA + 1], ATI[Y][3 * PA + 2]>(f32x2{acc[jj][e0], acc[jj][e0 + 1]}, f32x2{acc[3 + jj][e0], acc[3 + jj][e0 + 1]},
                                                                            f32x2{acc[6 + jj][e0], acc[6 + jj][e0 + 1]});
        const f32x2 p0 = lin3<ATI[0][3 * PB], ATI[0][3 * PB + 1], ATI[0][3 * PB + 2]>(r[0], r[1], r[2]);
        const f32x2 p1 = lin3<ATI[1][3 * PB], ATI[1][3 * PB + 1], ATI[1][3 * PB + 2]>(r[0], r[1], r[2]);
        const f32x2 p2 = lin3<ATI[2][3 * PB], ATI[2][3 * PB + 1], ATI[2][3 * PB + 2]>(r[0], r[1], r[2]);
        const f32x2 p3 = lin3<ATI[3][3 * PB], ATI[3][3 * PB + 1], ATI[3][3 * PB + 2]>(r[0], r[1], r[2]);
        P[0][2 * h] = p0[0]; P[0][2 * h + 1] = p0[1];
        P[1][2 * h] = p1[0]; P[1][2 * h + 1] = p1[1];
        P[2][2 * h] = p2[0]; P[2][2 * h + 1] = p2[1];
        P[3][2 * h] = p3[0]; P[3][2 * h + 1] = p3[1];
    }
}

// workgroup barrier that orders LDS traffic only (__syncthreads() also waits for the global stores in flight -- on this target vmcnt
// counts stores -- which put a store round trip on every output row: the staging is all these barriers protect)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (KO: knock-out bits for timing experiments -- DPIG_WINO4_KO; results are wrong: 1 no transforms, 2 no filter loads, 4 no raw gather, 8 no MFMAs)
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// a zero the compiler cannot hoist (hoisted out of the persistent loop, the zero vector was spilled and reloaded per output row)
__device__ __forceinline__ f32x4 fresh_zero4() {
    float z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return f32x4{z, z, z, z};
}

template <int BW, int KO = 0>
__device__ __forceinline__ void wino4_body(const WParams& p, lds_char* const L, const int vb, const int wave) {
    // (BW = 3 -- maps with three, six, nine .. tile columns, 12 x 12 say -- uses 30 of the 32 tile slots: the MFMA rows of slots 30, 31 and
    // of tile rows past the end of the stack (a partial last block) hold whatever the transform made of out-of-range pixels, and the
    // output phase leaves them out)
    constexpr int BH = TB / BW, TBV = BW * BH, RW = 4 * BW + 2, RH = 4 * BH + 2, RPIX = RW * RH, PIECES = (2 * RPIX + 63) / 64;
    static_assert(PIECES <= RAW_PIECES && PIECES > 16, "raw gather plan");
    // (the thread index passes through an empty asm: everything derived from it is recomputed per item, a few dozen instructions.  Hoisted
    // out of the persistent loop those values were spilled across the k-loop -- the register file is full -- and every item began with
    // five scratch reloads, each behind a wait for ALL memory operations in flight, the previous item's output stores included:
    // 3-4 k cycles per item)
    int lane = lane_id();
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;                 // (the wave's index is a scalar made once per kernel: no register holds threadIdx.x)
    const int l31 = lane & 31, half = lane >> 5;
    const int bid = xcd_remap(vb, p.mtiles * p.ntiles * p.nsplit);
    const int sp = bid / (p.mtiles * p.ntiles), tile = bid - sp * (p.mtiles * p.ntiles);
    // (both orders by integer arithmetic: as a boolean the choice was kept in a vector register -- the scalar file is full -- spilled, and
    // reloaded at the top of every item behind a wait for the previous item's stores)
    const int nt_f = tile / p.mtiles, mt_f = tile - nt_f * p.mtiles, mt_x = tile / p.ntiles, nt_x = tile - mt_x * p.ntiles;
    const int nt = nt_f + p.xmajor * (nt_x - nt_f), mt = mt_f + p.xmajor * (mt_x - mt_f);
    const int n0 = nt * KB;
    const int cb = sp * p.cps, ce = min(cb + p.cps, p.nch);
    const int bcols = p.TW / BW;
    const int brow = mt / bcols, bcol = mt - brow * bcols;
    const int R0 = BH * brow, C0 = BW * bcol;                         // first stacked tile row / tile column of the block
    auto stamp = [&](int slot) {
        if (p.trace && tid == 0) {
            p.trace[(long)vb * 8 + slot] = __builtin_amdgcn_s_memtime();
            if (slot == 0) p.trace[(long)vb * 8 + 5] = __builtin_amdgcn_s_memrealtime();
            if (slot == 4) p.trace[(long)vb * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rsU = make_rsrc(p.U, p.u_bytes);
    if (tid < 4) *(lds_f4*)(L + ZERO_OFF + tid * 16) = fresh_zero4();

    // ---- raw-gather role: piece ids wave, wave + 8, wave + 16 (< PIECES); lane slot s = 64 id + lane = (raw pixel s / 2, 16-byte half s & 1)
    int g_voff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int s = (wave + 8 * k) * 64 + lane;
        const int idx = s >> 1, hq = s & 1;
        const int row = idx / RW, colr = idx - row * RW;
        const int g = 4 * R0 - 1 + row, x = 4 * C0 - 1 + colr;       // stacked pixel row, column
        const bool ok = (idx < RPIX) & ((unsigned)g < (unsigned)(p.N * p.H)) & ((unsigned)x < (unsigned)p.W);
        g_voff[k] = ok ? ((g * p.W + x) * p.ldx + hq * 4) * 4 : (int)OOB;
    }
    // (a chunk index past the workgroup's range -- the loop's look-ahead at its end -- re-reads the last chunk: the data is staged and
    // never multiplied; no per-load select on the vector ALU)
    auto dmaRaw = [&](int chunk, int slot) {
        const int so = min(chunk, ce - 1) * (CH * 4);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < 2 || wave + 16 < PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(L + RAW_OFF + slot * RAWB + (wave + 8 * k) * 1024), 16,
                                                         g_voff[k], so, 0, 0);
    };
    // ---- input-transform role: item = (tile tl = 4 wave + lane / 16, channel quad q = (lane / 8) & 1), sub-index j = lane % 8 (< 6 works):
    // pass A: patch column j; pass B: transform row xi = j
    const int j = min(lane & 7, 5), it = lane >> 3;                   // (lanes 6 and 7 of an item repeat lane 5: no EXEC games, counted waits)
    const int tl = 4 * wave + (it >> 1), q = it & 1;
    const int tlv = min(tl, TBV - 1);                                 // (unused slots repeat the last tile's patch: reads stay inside the raw slot)
    const int bty = tlv / BW, btx = tlv - bty * BW;
    int a_mid, a_top[2], a_bot[2];                                    // LDS byte addresses of patch rows 1..4 (slot 0), row 0 and row 5 (per slot; the zero slot where the row is padding)
    {
        const int R = R0 + bty;
        const int n = fast_div(R, p.mul_th, p.shr_th);
        const int ty = R - n * (p.H >> 2);
        const int base = ((4 * bty) * RW + 4 * btx + j) * 32 + q * 16;
        a_mid = RAW_OFF + base;
        const bool top_ok = ty > 0, bot_ok = 4 * ty + 4 < p.H;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            a_top[s] = top_ok ? RAW_OFF + s * RAWB + base : ZERO_OFF + q * 16;
            a_bot[s] = bot_ok ? RAW_OFF + s * RAWB + base + 5 * RW * 32 : ZERO_OFF + q * 16;
        }
    }
    const int t_base = T_OFF + wave * T_WAVE + it * T_IT;
    const int t_wr = t_base + j * 16;                                  // pass A: + xi * T_XS
    const int t_rd = t_base + j * T_XS;                                // pass B: + column * 16
    const int v_wr = V_OFF + j * PS + tl * (CH * 4) + ((q ^ ((tl >> 3) & 1)) << 4);   // pass B: + 6 nu * PS (+ buffer): plane of position (xi, nu) = 6 nu + xi
    f32x4 d[6];
    auto readA = [&](int slot) {
        d[0] = *(lds_cf4*)(L + a_top[slot]);
#pragma unroll
        for (int r = 1; r < 5; ++r) d[r] = *(lds_cf4*)(L + a_mid + slot * RAWB + r * RW * 32);
        d[5] = *(lds_cf4*)(L + a_bot[slot]);
    };
    auto passA = [&](auto PART) {                                       // column transform -> scratch rows
        if (KO & 16) {                               // (knock-out: the stores without the arithmetic)
            constexpr int P = decltype(PART)::value;
            *(lds_f4*)(L + t_wr + (P == 0 ? 0 : P == 1 ? 1 : 3) * T_XS) = d[0];
            *(lds_f4*)(L + t_wr + (P == 0 ? 5 : P == 1 ? 2 : 4) * T_XS) = d[1];
            return;
        }
        bt6<decltype(PART)::value>(d, [&](int xi, f32x4 v) { *(lds_f4*)(L + t_wr + xi * T_XS) = v; });
    };
    auto readB = [&]() {
#pragma unroll
        for (int s = 0; s < 6; ++s) d[s] = *(lds_cf4*)(L + t_rd + s * 16);
    };
    auto passB = [&](int buf, auto PART) {                              // row transform -> V rows
        if (KO & 16) {
            constexpr int P = decltype(PART)::value;
            *(lds_f4*)(L + v_wr + buf * VB + (6 * (P == 0 ? 0 : P == 1 ? 1 : 3)) * PS) = d[0];
            *(lds_f4*)(L + v_wr + buf * VB + (6 * (P == 0 ? 5 : P == 1 ? 2 : 4)) * PS) = d[1];
            return;
        }
        bt6<decltype(PART)::value>(d, [&](int nu, f32x4 v) { *(lds_f4*)(L + v_wr + buf * VB + (6 * nu) * PS) = v; });
    };
    // ---- MFMA role: wave = (position group pg = (a, b): rows 3 a .. 3 a + 2, columns 3 b .. 3 b + 2 of the 6 x 6 grid; channel half wc) ------
    const int pg = wave >> 1, wc = wave & 1;
    const int pa = pg >> 1, pb = pg & 1;
    const int pos0 = 18 * pa + 3 * pb;                                 // the wave's first position; its pp-th: pos0 + 6 (pp / 3) + pp % 3
    const int fv = V_OFF + (18 * pb + 3 * pa) * PS + l31 * (CH * 4) + ((half ^ ((l31 >> 3) & 1)) << 4);   // (V planes: 6 nu + xi)
    const int u_voff = pos0 * UPOS + wc * 1024 + lane * 16;
    const int u_base = (nt * p.nch) * UCHUNK;
    f32x4 u[9];
    auto loadU = [&](int chunk, int pp) {
        const int so = u_base + min(chunk, ce - 1) * UCHUNK;
        u[pp] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, u_voff + (6 * (pp / 3) + pp % 3) * UPOS, so, 0));
    };
    f32x16 acc[9];
#pragma unroll
    for (int pp = 0; pp < 9; ++pp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pp][e] = 0.f;

    // ---- prologue: raw pixels of chunks 0 and 1 gathered, the filter fragments of chunk 0 loaded, chunk 0 transformed -----------------------
    if (KO & 128) stamp(1);
    // Only the first raw gather is waited for before chunk 0 is transformed; the filter fragments (72 KB per workgroup, most of the
    // prologue's bytes: their ISSUE alone, 120 one-KB loads through the CU's one address pipe, took ~3 k cycles in front of that wait)
    // go out between the passes of the transform, whose LDS round trips would otherwise idle.
    dmaRaw(cb, 0);
    dmaRaw(cb + 1, 1);
    if (KO & 128) stamp(2);
    wait_vm<2>();                                    // (2 or 3 pieces of chunk 1's gather are younger than chunk 0's)
    if (KO & 128) stamp(3);
    __syncthreads();
    if (!(KO & 32)) stamp(7);
    readA(0);
    loadU(cb, 0); loadU(cb, 1); loadU(cb, 2);
    passA(std::integral_constant<int, 0>{});
    passA(std::integral_constant<int, 1>{});
    loadU(cb, 3); loadU(cb, 4); loadU(cb, 5);
    passA(std::integral_constant<int, 2>{});
    readB();
    loadU(cb, 6); loadU(cb, 7); loadU(cb, 8);
    passB(0, std::integral_constant<int, 0>{});
    passB(0, std::integral_constant<int, 1>{});
    passB(0, std::integral_constant<int, 2>{});
    wait_vm<9>();                                    // the second gather is home (the fragment loads may still fly: the MFMAs wait for theirs)
    __syncthreads();                                 // V slot 0 complete, raw slots 0 (free) and 1 (filled) visible
    if (!(KO & 32) && !(KO & 128)) stamp(1);
    if (KO & 128) stamp(4);
    // chunk c = nine steps (the wave's positions): {V fragment of the next position, 4 MFMAs, the filter fragment of this position for
    // chunk c + 1, a slice of the staging work}.  Slices: step 0 the raw gather of chunk c + 2 (into the raw slot chunk c was transformed
    // from), 1 pass A's LDS reads of chunk c + 1, 2-4 its transform + scratch stores (two rows per step), 5 pass B's reads, 6-8 its
    // transform + V stores.
    auto body = [&](int c, auto PAR) {
        constexpr int buf = decltype(PAR)::value, oth = buf ^ 1;
        lds_char* const Vb = L + buf * VB + fv;
        f32x4 fb[2];
        fb[0] = *(lds_cf4*)(Vb);
#pragma unroll
        for (int pp = 0; pp < 9; ++pp) {
            // Order within a step (each boundary pinned): the NEXT position's V fragment read, the staging slice, this position's four
            // MFMAs, the filter fragment's reload (+ the raw gather).  Measured per chunk (scripts/ko_wino4.sh): slice after the MFMAs
            // 6200 cycles, slice before them 5815, "one MFMA, four slice instructions" enforced 6070, priority flips around the MFMAs
            // +130, fragment read after the slice +90; left to itself the scheduler issues the fragment read after the MFMAs and its
            // latency, behind the slice's slow 16-byte stores, lands on the next step.
            if (pp < 8) fb[(pp + 1) & 1] = *(lds_cf4*)(Vb + (6 * ((pp + 1) % 3) + (pp + 1) / 3) * PS);
            __builtin_amdgcn_sched_barrier(0);
            if (!(KO & 1)) {
                if (pp == 1) readA(oth);
                if (pp == 2) passA(std::integral_constant<int, 0>{});
                if (pp == 3) passA(std::integral_constant<int, 1>{});
                if (pp == 4) passA(std::integral_constant<int, 2>{});
                if (pp == 5) readB();
                if (pp == 6) passB(oth, std::integral_constant<int, 0>{});
                if (pp == 7) passB(oth, std::integral_constant<int, 1>{});
                if (pp == 8) passB(oth, std::integral_constant<int, 2>{});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                if (!(KO & 8)) acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[pp][s4], fb[pp & 1][s4], acc[pp], 0, 0, 0);
            if (pp == 0 && !(KO & 4)) dmaRaw(c + 2, buf);
            if (!(KO & 2)) loadU(c + 1, pp);
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vm<9>();                                // the raw pieces of chunk c + 2 are home (only the nine filter fragments are younger)
        __syncthreads();                             // + V slot `oth` written, every wave done reading V slot `buf` and raw slot `oth`
    };
    for (int c = cb; c < ce; c += 2) {
        body(c, std::integral_constant<int, 0>{});
        if (c + 1 < ce) body(c + 1, std::integral_constant<int, 1>{});
    }
    wait_vm<0>();
    if (!(KO & 32) && !(KO & 128)) stamp(2);
    // ---- output transform, one output row y of the 4 x 4 tiles at a time.  A lane holds element (tile l31, channel 32 wc + 8 g + 4 half + e)
    // of its nine positions M[3 a + i][3 b + jj]; its partial of Y[y][x] is  sum_i AT[y][3 a + i] sum_jj AT[x][3 b + jj] M[i][jj].
    // Staging: [partial pg][x][tile][64 channels]; the epilogue threads (16-byte channel group cg = tid % 16, tile tid / 16) add the four
    // partials in fixed order and run ONE of three workgroup-uniform paths on the row's four pixels.
    auto out_row = [&](auto Yc) {
        constexpr int y = decltype(Yc)::value;
        // (the row's lane roles and addresses are made here, from the thread index through an empty asm: kept across the rows they were
        // spilled, and every reload waited for the previous row's output stores -- vmcnt counts stores)
        int lane2 = lane_id();
        asm volatile("" : "+v"(lane2));
        const int tid2 = wave * 64 + lane2;
        // (and the accumulators: two rows of a position group can share sub-expressions -- rows 1 and 3 of the groups that hold grid rows
        // 0-2 are identical -- and values kept from one row for another do not fit either)
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]));
        const int cg = tid2 & 15, tloc = tid2 >> 4, col = n0 + 4 * cg;
        const int ebty = tloc / BW, ebtx = tloc - ebty * BW;
        const bool live = (tloc < TBV) & (4 * (R0 + ebty) < p.N * p.H);         // a tile slot in use, on a tile row of the stack
        const long pix = live ? (long)(4 * (R0 + ebty) + y) * p.W + 4 * (C0 + ebtx) : 0;
        if ((KO & 32) && y == 1) stamp(1);
        auto stage = [&](auto PAc, auto PBc) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 P[4];
                out_row_group<decltype(PAc)::value, decltype(PBc)::value, decltype(Yc)::value>(acc, g, P);
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    *(lds_f4*)(L + ((pg * 4 + x) * TB + l31) * EP_ROW + (32 * wc + 8 * g + 4 * half) * 4) = P[x];
                __builtin_amdgcn_sched_barrier(0);   // (one channel group at a time)
            }
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
        if (pg == 0) stage(I0{}, I0{});
        else if (pg == 1) stage(I0{}, I1{});
        else if (pg == 2) stage(I1{}, I0{});
        else stage(I1{}, I1{});
        // the general epilogue's operands (residual / accumulate tensor, activation mask) of this row's four pixels are asked for HERE,
        // between the staging stores and the barrier: loaded where they are used, behind this row's own LDS reads, the masked dgrad ran
        // 14 % behind the plain one (a memory round trip per row, plus the previous row's output stores: one counter); asked for before
        // the transform they do not fit beside the accumulators (the allocator spilled all eight)
        const bool general = p.nsplit == 1 && (p.res || p.mask || p.D2);
        f32x4 rv[4], mv[4];
        if (general && live) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {            // (left unset where the launch has no such operand: never read then)
                if (p.res) rv[x] = *reinterpret_cast<const f32x4*>(p.res + (pix + x) * p.ldres + col);
                if (p.mask) mv[x] = *reinterpret_cast<const f32x4*>(p.mask + (pix + x) * p.ldmask + col);
            }
        }
        if ((KO & 32) && y == 1) stamp(2);
        lds_barrier();
        if ((KO & 32) && y == 1) stamp(3);
        f32x4 v[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const lds_char* const s = L + (x * TB + tloc) * EP_ROW + cg * 16;
            v[x] = (*(lds_cf4*)(s) + *(lds_cf4*)(s + 4 * TB * EP_ROW)) + (*(lds_cf4*)(s + 8 * TB * EP_ROW) + *(lds_cf4*)(s + 12 * TB * EP_ROW));
            if (x == 1) __builtin_amdgcn_sched_barrier(0);      // (eight reads in flight, not sixteen: beside the accumulators and the general path's operands 64 registers of read data spill)
        }
        f32x4 bv = fresh_zero4();
        if (p.bias && p.nsplit == 1) bv = *reinterpret_cast<const f32x4*>(p.bias + col);
        if (!live) {
            // (nothing to store: an unused tile slot, or a tile row past the end of the stack)
        } else if (p.nsplit > 1) {
            float* const base = p.partial + (long)sp * p.N * p.H * p.W * p.Kout + col;
#pragma unroll
            for (int x = 0; x < 4; ++x) *reinterpret_cast<f32x4*>(base + (pix + x) * p.Kout) = v[x];
        } else if (!p.res && !p.mask && !p.D2) {
            float* const base = p.D + col;
            auto plain = [&](auto ACT) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    f32x4 o = v[x] + bv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = act_apply(o[e], decltype(ACT)::value, p.alpha);
                    *reinterpret_cast<f32x4*>(base + (pix + x) * p.ldd) = o;
                }
            };
            if (p.act == DPIG_ACT_RELU) plain(std::integral_constant<int, DPIG_ACT_RELU>{});
            else if (p.act == DPIG_ACT_LRELU) plain(std::integral_constant<int, DPIG_ACT_LRELU>{});
            else plain(std::integral_constant<int, DPIG_ACT_NONE>{});
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) {            // (epi4 of dpig_wino_common.h on the operands loaded above)
                f32x4 o = v[x] + bv;
                if (p.res && !p.res_post) o += rv[x];
                if (p.mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= act_grad(mv[x][e], p.act, p.alpha);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = act_apply(o[e], p.act, p.alpha);
                }
                if (p.D2) *reinterpret_cast<f32x4*>(p.D2 + (pix + x) * p.ldd2 + col) = o;
                if (p.res && p.res_post) o += rv[x];
                *reinterpret_cast<f32x4*>(p.D + (pix + x) * p.ldd + col) = o;
            }
        }
        if ((KO & 32) && y == 1) stamp(7);
        if (y < 3) lds_barrier();                    // the row's staging reads are done before the next row's stores
        if (!(KO & 32) && !(KO & 128) && y == 0) stamp(3);
        if ((KO & 32) && y == 1) stamp(4);
    };
    out_row(std::integral_constant<int, 0>{});
    out_row(std::integral_constant<int, 1>{});
    out_row(std::integral_constant<int, 2>{});
    out_row(std::integral_constant<int, 3>{});
    if (!(KO & 32) && !(KO & 128)) stamp(4);
}

// Persistent launch (as wino_block_kernel): one workgroup per CU, or per item when there are fewer, walking the items.
template <int BW, int KO = 0>
__global__ __launch_bounds__(512, 2) void wino4_kernel(const WParams p) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    lds_char* const L = (lds_char*)smem;
    const int total = p.mtiles * p.ntiles * p.nsplit;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
        wino4_body<BW, KO>(p, L, vb, wave);
        lds_barrier();                                // the last row's staging reads are done before the next item's gather lands
    }
}

// ---- filter transform: U = G g G^T, G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1], of every (input
// channel, output channel) pair, written in the kernel's fragment order.  One workgroup per (64-channel column block, 8-channel chunk):
// its 512 transforms are staged in LDS in image order and leave as one contiguous 72-KB run.  `DGRAD`: the transposed conv's filter
// g'[r][s][k][c] = w[2 - r][2 - s][c][k].  w is HWIO [3][3][C][K].
__device__ __forceinline__ void g6(float g0, float g1, float g2, float (&o)[6]) {
    const float s = g0 + g2;
    o[0] = 0.25f * g0;
    o[1] = (-1.f / 6.f) * (s + g1);
    o[2] = (-1.f / 6.f) * (s - g1);
    const float e = (1.f / 24.f) * g0 + (1.f / 6.f) * g2;
    o[3] = e + (1.f / 12.f) * g1;
    o[4] = e - (1.f / 12.f) * g1;
    o[5] = g2;
}
template <bool DGRAD>
__device__ __forceinline__ void filter4_body(const float* __restrict__ w, float* __restrict__ U, int C, int K, int blk, float* img) {
    const int cin = DGRAD ? K : C;
    const int nch = cin / CH;
    const int kb = blk / nch, chunk = blk - kb * nch;
    const int tid = threadIdx.x;
#pragma unroll
    for (int jx = 0; jx < 2; ++jx) {
        // (m, cc) = (output channel within the block, reduction channel within the chunk); the lanes run along w's fastest axis
        const int m = DGRAD ? (tid >> 3) + 32 * jx : (tid & 63);
        const int cc = DGRAD ? (tid & 7) : (tid >> 6) + 4 * jx;
        const int ko = kb * 64 + m, ci = chunk * CH + cc;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                g[r][s] = DGRAD ? w[(((2 - r) * 3 + (2 - s)) * (long)C + ko) * K + ci] : w[((r * 3 + s) * (long)C + ci) * K + ko];
        float tt[3][6];                                // [filter column s][transform row xi]
#pragma unroll
        for (int s = 0; s < 3; ++s) g6(g[0][s], g[1][s], g[2][s], tt[s]);
        // image order within (chunk, position): [channel half m / 32][k half cc / 4][channel m % 32][cc % 4]
        float* const o = img + (m >> 5) * 256 + (cc >> 2) * 128 + (m & 31) * 4 + (cc & 3);
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
            float uu[6];
            g6(tt[0][xi], tt[1][xi], tt[2][xi], uu);
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) o[(xi * 6 + nu) * 512] = uu[nu];
        }
    }
    __syncthreads();
    f32x4* const dst = reinterpret_cast<f32x4*>(U + (long)blk * (36 * 512));
    const f32x4* const src = reinterpret_cast<const f32x4*>(img);
#pragma unroll
    for (int i = 0; i < 18; ++i) dst[tid + 256 * i] = src[tid + 256 * i];
}
template <bool DGRAD>
__global__ __launch_bounds__(256) void wino4_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int C, int K) {
    __shared__ __attribute__((aligned(16))) float img[36 * 512];
    filter4_body<DGRAD>(w, U, C, K, xcd_remap(blockIdx.x, gridDim.x), img);
}
// every filter of a parameter set in one launch (as wino_filter_jobs_kernel; the same job table and block numbering)
__global__ __launch_bounds__(256) void wino4_filter_jobs_kernel(const DpigWinoFilterJob* __restrict__ jobs, int njobs, int total) {
    __shared__ __attribute__((aligned(16))) float img[36 * 512];
    const int lb = xcd_remap(blockIdx.x, 2 * total);
    const int dir = lb >= total;
    const int b = lb - dir * total;
    int cnt = 0;
    for (int i0 = 0; i0 < njobs; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        cnt += __syncthreads_count(i < njobs && jobs[i].first_block <= b);
    }
    const DpigWinoFilterJob j = jobs[cnt - 1];
    if (dir) {
        if (j.u_dgrad) filter4_body<true>(j.w, j.u_dgrad, j.C, j.K, b - j.first_block, img);
    } else {
        if (j.u_fwd) filter4_body<false>(j.w, j.u_fwd, j.C, j.K, b - j.first_block, img);
    }
}

static int g_mode = -1;            // 0 never, 1 where the cost model says it pays (default), 2 wherever legal (tests)
static void init_mode() {
    if (g_mode >= 0) return;
    const char* e = getenv("DPIG_WINO4");
    g_mode = e ? atoi(e) : 1;
}

// block width (tiles) the layer's tile grid can be cut with: BW x 32 / BW tiles on the stack of all images' tile rows; 0: none
static int block_width(const DpigConvDesc* d) {
    const int tw = d->W / 4;
    const long rows = (long)d->N * (d->H / 4);
    if (tw % 4 == 0 && rows % 8 == 0) return 4;
    if (tw % 2 == 0 && rows % 16 == 0) return 2;
    if (tw % 3 == 0) return 3;                 // 3 x 10 tiles (30 of the 32 slots), a partial last block where the stack's rows are no multiple of 10
    if (tw % 4 == 0) return 4;                 // the even widths with a partial last block
    if (tw % 2 == 0) return 2;
    return 0;
}
// row blocks x column blocks of the tile grid
static long tile_blocks(const DpigConvDesc* d) {
    const int bw = block_width(d);
    if (!bw) return 0;
    const long rows = (long)d->N * (d->H / 4);
    return cdiv(rows, TB / bw) * ((d->W / 4) / bw);
}
// geometry both entry points share: 3 x 3, stride 1, SAME, sides multiples of 4 with a block form, 16-byte addressable channel vectors
static bool shape_ok(const DpigConvDesc* d, int cin, int kout, int ld_in, int ld_out) {
    if (d->R != 3 || d->S != 3 || d->stride != 1 || d->upsample2x || d->res_class || d->split_k > 1) return false;
    if ((d->H & 3) || (d->W & 3) || d->H < 4 || d->W < 4) return false;
    if (!block_width(d)) return false;
    if (cin % CH || kout % KB || (ld_in & 3) || (ld_out & 3)) return false;
    if (d->pad_t >= 0 && d->pad_t != 1) return false;
    if (d->pad_l >= 0 && d->pad_l != 1) return false;
    const long lim = 0x7f000000L;
    if ((long)d->N * d->H * d->W * ld_in * 4 >= lim || (long)d->N * d->H * d->W * ld_out * 4 >= lim) return false;
    if ((long)36 * cin * kout * 4 >= lim) return false;
    return true;
}
// One workgroup per CU, whole rounds of 256: a workgroup's life is ~5900 cycles per 8-channel chunk (72 MFMAs per SIMD = 4608 of them)
// + ~26 k cycles of prologue (first gather + transform, ~7.5 k) and output transform / epilogue (~18 k) -- scripts/trace_wino4.py.
// Split plans as in the F(2x2, 3x3) family (partials summed by wino_reduce_kernel).
constexpr double CHUNK_CYCLES = 5900.0, FIXED_CYCLES = 26000.0;
struct FPlan { int nsplit, cps; double cycles; };
static FPlan fwd_plan(const DpigConvDesc* d, int cin, int kout) {
    const long wgs1 = tile_blocks(d) * (kout / KB);
    const int nch = cin / CH;
    FPlan best = {1, nch, 0.0};
    static const int force = getenv("DPIG_WINO_SPLIT") ? atoi(getenv("DPIG_WINO_SPLIT")) : 0;       // (A/B switch: 1 = never split)
    for (int s = 1; s <= 16; ++s) {
        const int cps = cdiv(nch, s);
        if (s > 1 && (cps < 6 || cdiv(nch, cps) != s || force == 1)) continue;
        const long rounds = (wgs1 * s + kNumCU - 1) / kNumCU;
        double cyc = (double)rounds * ((double)cps * CHUNK_CYCLES + FIXED_CYCLES);
        if (s > 1) cyc += 8000.0 + 2.0 * s * (double)d->N * d->H * d->W * kout * 4.0 / 1740.0;     // 4 TB/s = 1740 B per cycle
        if (best.cycles == 0.0 || cyc < best.cycles * 0.97) best = {s, cps, cyc};
    }
    return best;
}
// Does this form beat what the layer would otherwise run on (the F(2x2, 3x3) plan, or the direct family where that one does not pay)?
static bool pays(const DpigConvDesc* d, int cin, int kout, int ld_in, int ld_out) {
    init_mode();
    if (g_mode == 0) return false;
    if (g_mode == 2) return true;
    return fwd_plan(d, cin, kout).cycles < 0.97 * wino::alt_cycles(d, cin, kout, ld_in, ld_out);
}

static int launch(const DpigConvDesc* d, const float* in, const float* U, const float* bias, const float* res, const float* mask,
                  float* out, float* out2, int cin, int kout, int ld_in, int ld_out, int act, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!aligned16(in) || !aligned16(U) || !aligned16(out) || (bias && !aligned16(bias)) || (res && (!aligned16(res) || (d->ldres & 3))) ||
        (mask && (!aligned16(mask) || (d->ldmask & 3))) || (out2 && (!aligned16(out2) || (d->ldy2 & 3))))
        return fail(DPIG_EINVAL, "winograd F(4x4) conv: operands must be 16-byte addressable");
    WParams p = {};
    p.X = in; p.U = U; p.D = out; p.D2 = out2; p.bias = bias; p.res = res; p.mask = mask;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = cin; p.Kout = kout;
    p.ldx = ld_in; p.ldd = ld_out; p.ldres = d->ldres; p.ldmask = d->ldmask; p.ldd2 = d->ldy2;
    p.TW = d->W / 4; p.THW = (d->H / 4) * p.TW; p.T = d->N * p.THW;
    p.nch = cin / CH;
    p.mtiles = (int)tile_blocks(d); p.ntiles = kout / KB;
    p.act = act; p.alpha = d->alpha; p.res_post = d->res_after_act;
    p.x_bytes = (unsigned)((long)d->N * d->H * d->W * ld_in * 4);
    p.u_bytes = (unsigned)((long)36 * cin * kout * 4);
    find_divisor(p.THW, &p.mul_thw, &p.shr_thw);
    find_divisor(p.TW, &p.mul_tw, &p.shr_tw);
    find_divisor(d->H / 4, &p.mul_th, &p.shr_th);
    p.trace = wino::trace_buffer();
    const FPlan pl = fwd_plan(d, cin, kout);
    p.nsplit = pl.nsplit; p.cps = pl.cps;
    {   // workgroup order by HBM bytes (as the F(2x2, 3x3) launch)
        const double xb = (double)d->N * d->H * d->W * cin * 4.0, ub = 36.0 * cin * kout * 4.0;
        const double filter_major = xb * (p.ntiles < kNumXCD ? p.ntiles : kNumXCD) + ub;
        const double act_major = xb + ub * (p.mtiles < kNumXCD ? p.mtiles : kNumXCD);
        static const char* pin = getenv("DPIG_WINO_XMAJOR");
        p.xmajor = pin ? (atoi(pin) != 0) : (act_major < filter_major);
    }
    if (p.nsplit > 1) {
        const size_t need = (size_t)p.nsplit * d->N * d->H * d->W * kout * sizeof(float);
        if (!ws || ws_bytes < need || !aligned16(ws)) return fail(DPIG_ENOMEM, "winograd F(4x4) conv workspace too small: have %zu, need %zu", ws_bytes, need);
        p.partial = static_cast<float*>(ws);
    }
    const unsigned items = (unsigned)(p.mtiles * p.ntiles * p.nsplit);
    static const char* pers = getenv("DPIG_WINO_PERSIST");          // 0: one workgroup per item (measurements)
    const dim3 pgrid((pers && atoi(pers) == 0) || items <= (unsigned)kNumCU ? items : (unsigned)kNumCU);
    static const int ko = getenv("DPIG_WINO4_KO") ? atoi(getenv("DPIG_WINO4_KO")) : 0;
    if (block_width(d) == 2) hipLaunchKernelGGL(wino4_kernel<2>, pgrid, dim3(512), 0, st, p);
    else if (block_width(d) == 3) hipLaunchKernelGGL(wino4_kernel<3>, pgrid, dim3(512), 0, st, p);
#ifdef DPIG_WINO4_KNOCKOUT
    else if (ko == 1) hipLaunchKernelGGL((wino4_kernel<4, 1>), pgrid, dim3(512), 0, st, p);
    else if (ko == 2) hipLaunchKernelGGL((wino4_kernel<4, 2>), pgrid, dim3(512), 0, st, p);
    else if (ko == 4) hipLaunchKernelGGL((wino4_kernel<4, 4>), pgrid, dim3(512), 0, st, p);
    else if (ko == 7) hipLaunchKernelGGL((wino4_kernel<4, 7>), pgrid, dim3(512), 0, st, p);
    else if (ko == 8) hipLaunchKernelGGL((wino4_kernel<4, 8>), pgrid, dim3(512), 0, st, p);
    else if (ko == 3) hipLaunchKernelGGL((wino4_kernel<4, 3>), pgrid, dim3(512), 0, st, p);
    else if (ko == 16) hipLaunchKernelGGL((wino4_kernel<4, 16>), pgrid, dim3(512), 0, st, p);
    else if (ko == 32) hipLaunchKernelGGL((wino4_kernel<4, 32>), pgrid, dim3(512), 0, st, p);
    else if (ko == 192) hipLaunchKernelGGL((wino4_kernel<4, 192>), pgrid, dim3(512), 0, st, p);
#endif
    else hipLaunchKernelGGL(wino4_kernel<4>, pgrid, dim3(512), 0, st, p);
    (void)ko;
    const int rc = check_launch("wino4_kernel");
    if (rc || p.nsplit == 1) return rc;
    return wino::launch_reduce(p, st);
}

}  // namespace wino4
}  // namespace dpig

using namespace dpig;

// Elements (floats) of one F(4x4, 3x3) image of a [3][3][C][K] filter; 0 when the shape has no such form.
extern "C" size_t dpig_wino4_filter_elems(int C, int K) {
    if (C <= 0 || K <= 0 || C % 64 || K % 64) return 0;
    return (size_t)36 * C * K;
}

// u_fwd / u_dgrad (either may be null): images for dpig_conv2d_fwd_wino4 / dpig_conv2d_dgrad_wino4 of the HWIO filter w.
extern "C" int dpig_wino4_filter_transform(const float* w, int C, int K, float* u_fwd, float* u_dgrad, void* stream) {
    if (!w || !dpig_wino4_filter_elems(C, K)) return fail(DPIG_EINVAL, "winograd F(4x4) filter transform: C and K must be positive multiples of 64");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int blocks = (C / 64) * (K / wino4::CH);                // = (K / 64) * (C / 8): the same count in both directions
    if (u_fwd) hipLaunchKernelGGL(wino4::wino4_filter_kernel<false>, dim3(blocks), dim3(256), 0, st, w, u_fwd, C, K);
    if (u_dgrad) hipLaunchKernelGGL(wino4::wino4_filter_kernel<true>, dim3(blocks), dim3(256), 0, st, w, u_dgrad, C, K);
    return check_launch("wino4_filter_kernel");
}

// A whole parameter set in one launch: the job table of dpig_wino_filter_jobs_plan (same block numbering), u_fwd / u_dgrad holding
// dpig_wino4_filter_elems floats each.
extern "C" int dpig_wino4_filter_transform_jobs(const DpigWinoFilterJob* jobs_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_blocks <= 0) return fail(DPIG_EINVAL, "winograd F(4x4) filter jobs: empty or unplanned job list");
    hipLaunchKernelGGL(wino4::wino4_filter_jobs_kernel, dim3(2 * (unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       jobs_dev, njobs, total_blocks);
    return check_launch("wino4_filter_jobs_kernel");
}

// 1: dpig_conv2d_fwd_wino4 (which = 0) / dpig_conv2d_dgrad_wino4 (which = 1) accepts this descriptor AND is expected to beat the
// F(2x2, 3x3) kernel (DPIG_WINO4=2 / dpig_conv_wino4_set_mode(2): wherever legal); 0 otherwise.  No device work.
extern "C" int dpig_conv2d_wino4_eligible(const DpigConvDesc* d, int which) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->compute != DPIG_COMPUTE_F32) return 0;
    const bool dg = which == 1;
    const int cin = dg ? d->K : d->C, kout = dg ? d->C : d->K;
    const int ld_in = dg ? d->ldy : d->ldx, ld_out = dg ? d->ldx : d->ldy;
    if (d->C % 64 || d->K % 64) return 0;                       // (one transformed image pair serves both directions)
    if (!wino4::shape_ok(d, cin, kout, ld_in, ld_out)) return 0;
    return wino4::pays(d, cin, kout, ld_in, ld_out) ? 1 : 0;
}

extern "C" int dpig_conv_wino4_set_mode(int mode) {
    if (mode < 0 || mode > 2) return fail(DPIG_EINVAL, "winograd F(4x4) mode out of range");
    wino4::g_mode = mode;
    return DPIG_OK;
}

extern "C" int dpig_conv_wino4_get_mode(void) {
    wino4::init_mode();
    return wino4::g_mode;
}

// Workspace of dpig_conv2d_fwd_wino4 (which = 0) / dpig_conv2d_dgrad_wino4 (which = 1): the partial outputs of a split plan, 0 for most layers.
extern "C" size_t dpig_conv2d_wino4_workspace_bytes(const DpigConvDesc* d, int which) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0) return 0;
    const bool dg = which == 1;
    const int cin = dg ? d->K : d->C, kout = dg ? d->C : d->K;
    if (d->C % 64 || d->K % 64 || !wino4::shape_ok(d, cin, kout, dg ? d->ldy : d->ldx, dg ? d->ldx : d->ldy)) return 0;
    const wino4::FPlan pl = wino4::fwd_plan(d, cin, kout);
    return pl.nsplit > 1 ? (size_t)pl.nsplit * d->N * d->H * d->W * kout * sizeof(float) : 0;
}

// y = act(conv3x3_SAME(x, w) + bias + residual)  (or act(..) + residual with res_after_act, y_act receiving the activation) through the
// F(4x4, 3x3) image u_fwd of dpig_wino4_filter_transform.  Same descriptor and epilogue semantics as dpig_conv2d_fwd.
extern "C" int dpig_conv2d_fwd_wino4(const DpigConvDesc* d, const float* x, const float* u_fwd, const float* bias, const float* residual,
                                     float* y, float* y_act, void* ws, size_t ws_bytes, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !u_fwd || !y) return fail(DPIG_EINVAL, "null tensor pointer");
    if (!wino4::shape_ok(d, d->C, d->K, d->ldx, d->ldy) || d->C % 64) return fail(DPIG_EINVAL, "winograd F(4x4) conv: unsupported shape");
    if (residual && d->ldres < d->K) return fail(DPIG_EINVAL, "ldres < K");
    if (y_act && d->ldy2 < d->K) return fail(DPIG_EINVAL, "ldy2 < K");
    return wino4::launch(d, x, u_fwd, bias, residual, nullptr, y, y_act, d->C, d->K, d->ldx, d->ldy, d->act, ws, ws_bytes,
                         static_cast<hipStream_t>(stream));
}

// dx = (conv_backward_data(dy, w) + accum) * act'(mask) through u_dgrad.  Same semantics as dpig_conv2d_dgrad.
extern "C" int dpig_conv2d_dgrad_wino4(const DpigConvDesc* d, const float* dy, const float* u_dgrad, const float* accum, const float* mask,
                                       float* dx, void* ws, size_t ws_bytes, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!dy || !u_dgrad || !dx) return fail(DPIG_EINVAL, "null tensor pointer");
    if (!wino4::shape_ok(d, d->K, d->C, d->ldy, d->ldx) || d->K % 64) return fail(DPIG_EINVAL, "winograd F(4x4) conv: unsupported shape");
    if (accum && d->ldres < d->C) return fail(DPIG_EINVAL, "ldres < C");
    if (mask && d->ldmask < d->C) return fail(DPIG_EINVAL, "ldmask < C");
    DpigConvDesc e = *d;
    e.res_after_act = 0; e.ldy2 = 0;
    return wino4::launch(&e, dy, u_dgrad, nullptr, accum, mask, dx, nullptr, d->K, d->C, d->ldy, d->ldx, mask ? d->act : DPIG_ACT_NONE,
                         ws, ws_bytes, static_cast<hipStream_t>(stream));
}
