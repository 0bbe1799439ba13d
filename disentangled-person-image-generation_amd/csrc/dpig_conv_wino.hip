// Winograd F(2x2, 3x3) convolution on the fp32 matrix pipe (gfx950), fp32 in / fp32 products / fp32 accumulate.
//
// The exact direct-conv family (dpig_conv.hip) sits at 0.77 of the fp32 MFMA peak (the pipe is ~97 % busy at the clock the part
// sustains), so the only lever left on the BASELINE metric in its own arithmetic type is fewer executed multiplies.  For a 3 x 3
// stride-1 SAME conv (89 % of the step's FLOPs: models.py:396-400, 425-427, 458-460, 534-535, 564-565 of the reference) Winograd's
// minimal filtering computes every 2 x 2 output tile from a 4 x 4 input patch with 16 multiplies per (input channel, output channel)
// instead of 36:
//        Y = A^T [ (G g G^T) o (B^T d B) ] A        (o = element-wise product, summed over input channels)
// i.e. 16 independent GEMMs  M_p[tile, k] = sum_c V_p[tile, c] U_p[c, k],  p = (xi, nu) in 4 x 4, on 1/4 of the rows: 2.25x fewer
// MFMA FLOPs.  Everything is fused in ONE kernel so that neither the 4x-expanded input transform V nor the 4x-expanded product M
// ever exists in HBM:
//   * a workgroup owns 64 tiles (256 output pixels) x 64 output channels and ALL 16 positions: eight waves = 2 position halves x 2 tile
//     halves x 2 channel halves, each holding 8 accumulator blocks of v_mfma_f32_32x32x2_f32 (32 output channels x 32 tiles per
//     position) = 128 accumulator registers -> two waves per SIMD (the two position halves of one block: one stages while the other
//     feeds the matrix pipe), one workgroup per CU.  (The first form, four waves with all 16 positions = 256 accumulators each and one
//     wave per SIMD, ran the k-loop at ~70 % of the pipe: nothing hides a lone wave's own issue stalls; scripts/trace_wino.py.)
//   * the reduction runs over input channels in chunks of 8: per chunk a thread loads the 2 x 4 input pixels its row of the row
//     transform needs (buffer loads with out-of-range offsets as the zero padding), applies B^T . B in registers and writes four
//     16-byte rows of V_p into LDS; the transformed filter U_p arrives by LDS-DMA from a precomputed image (dpig_wino_filter_transform:
//     once per optimizer step, like the bf16 filter shadows) that already HAS the LDS layout, bank swizzle included;
//   * per chunk a wave issues 32 MFMAs (2048 cycles of pipe time; 4096 per SIMD) against 16 ds_read_b128, 8 global loads, 4 DMA pieces,
//     ~32 VALU and 4 ds_write_b128: the loop is matrix-bound with a wide margin;
//   * the output transform A^T . A is per-lane register arithmetic on each wave's eight positions (a lane holds the same (tile, channel)
//     element of all of them); the two position halves' partial 2 x 2 pixels meet in LDS on their way into the family's fused
//     row-contiguous epilogue (bias, activation, residual before / after the activation with the second output, dgrad's
//     (. + accum) * act'(mask)).
// dgrad of a 3 x 3 stride-1 SAME conv is the same conv with the filter rotated by 180 degrees and its channel roles swapped: the
// same kernel on a second transformed image (U' from w[2-r][2-s][c][k] read as [k][c]).  The filter gradient has its own kernel further
// down, wino_wgrad_kernel = F(3x3, 2x2) with the tiles as reduction axis (dpig_conv2d_wgrad_wino; own eligibility + cost model).
//
// LDS image of one chunk of one operand (V or U): [position 16][row 64][8 floats]; row = tile (V) or output channel (U); the two
// 16-byte halves of a row are swapped when (row >> 3) & 1 so that the 16-lane groups of ds_read_b128 cover all 64 banks.  Within a
// chunk the channel order is free as long as both operands agree: MFMA k-step s multiplies channels {s, 4 + s}.
//
// Accuracy: products and sums are fp32; the transforms add +-1 / +-1/2 combinations of 4 inputs / 3 filter taps, so results differ
// from the direct kernel by a few fp32 ulps of the largest intermediate (measured against the fp64 oracle in tests/test_wino_gpu.py).
#include "dpig_wino_common.h"

namespace dpig {
namespace wino {

constexpr int TB = 64;                        // 2 x 2 output tiles per workgroup
constexpr int KB = 64;                        // output channels per workgroup
constexpr int CH = 8;                         // reduction channels per chunk
constexpr int ROWB = CH * 4;                  // bytes of one (position, row)
constexpr int PLANE = 64 * ROWB;              // one position: 2 KB
constexpr int OPB = 16 * PLANE;               // one operand, one chunk: 32 KB
constexpr int EP_ROW = KB * 4 + 16;           // epilogue staging: bytes per pixel row (+16: conflict-free 16-byte stores)
constexpr int SMEM = 2 * 256 * EP_ROW;        // the loop's V[2] | U[2] (128 KB) or the epilogue's two staging images (136 KB)
static_assert(4 * OPB <= SMEM && SMEM <= 163840, "LDS plan");
typedef const __attribute__((address_space(3))) f32x4 lds_cf4_t;
// A workgroup's fused epilogue: thread = (16-byte channel group cg = tid % 16, staged rows tid / 16 + 32 k2 of the four 2 x 2 sub-pixel
// images), i.e. the four pixels of two tiles; pix0[k2] = the tile's top-left output pixel (ok[k2]: the tile exists).  All 16 LDS reads
// go first (one wait instead of eight read -> wait -> store round trips), then ONE of three workgroup-uniform paths: the partial slab
// of a split plan, the plain bias + activation (most launches; the activation a compile-time constant), or the family's general
// form.  (With the reads inside the per-pixel loop and every activation / residual / mask test inside it too, this phase took 6.4 k
// cycles of a C = 128 workgroup's 92 k -- scripts/trace_wino.py.)
__device__ __forceinline__ void wino_epilogue(const WParams& p, const lds_char* const L, int tid, int n0, int sp, const long (&pix0)[2],
                                              const bool (&ok)[2]) {
    const int cg = tid & 15, col = n0 + 4 * cg;
    f32x4 v[8];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            const int row = ij * 64 + (tid >> 4) + 32 * k2;
            v[4 * k2 + ij] = *(lds_cf4_t*)(L + row * EP_ROW + cg * 16) + *(lds_cf4_t*)(L + 256 * EP_ROW + row * EP_ROW + cg * 16);
        }
    if (p.nsplit > 1) {
        float* const base = p.partial + (long)sp * p.N * p.H * p.W * p.Kout + col;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            if (!ok[k2]) continue;
#pragma unroll
            for (int ij = 0; ij < 4; ++ij)
                *reinterpret_cast<f32x4*>(base + (pix0[k2] + (ij >> 1) * p.W + (ij & 1)) * p.Kout) = v[4 * k2 + ij];
        }
        return;
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + col);
    if (!p.res && !p.mask && !p.D2) {
        float* const base = p.D + col;
        auto plain = [&](auto ACT) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                if (!ok[k2]) continue;
#pragma unroll
                for (int ij = 0; ij < 4; ++ij) {
                    f32x4 o = v[4 * k2 + ij] + bv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = act_apply(o[e], decltype(ACT)::value, p.alpha);
                    *reinterpret_cast<f32x4*>(base + (pix0[k2] + (ij >> 1) * p.W + (ij & 1)) * p.ldd) = o;
                }
            }
        };
        if (p.act == DPIG_ACT_RELU) plain(std::integral_constant<int, DPIG_ACT_RELU>{});
        else if (p.act == DPIG_ACT_LRELU) plain(std::integral_constant<int, DPIG_ACT_LRELU>{});
        else plain(std::integral_constant<int, DPIG_ACT_NONE>{});
        return;
    }
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        if (!ok[k2]) continue;
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) epi4(p, pix0[k2] + (ij >> 1) * p.W + (ij & 1), col, v[4 * k2 + ij], bv);
    }
}

__global__ __launch_bounds__(512, 2) void wino_kernel(const WParams p) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    lds_char* const L = (lds_char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    // tile order (xcd_remap hands every XCD one contiguous range of it).  Filter-major: the row blocks of one 64-channel column block
    // are consecutive, an XCD streams ONE slice of the transformed filter through its L2 and the column blocks' XCDs each fetch the
    // activation (min(8, K / 64) times in all).  Activation-major (p.xmajor): the column blocks of one row block are consecutive, the
    // activation is fetched once and every XCD streams the whole filter image.  The host picks the order that moves fewer bytes.
    const int bid = xcd_remap(blockIdx.x, p.mtiles * p.ntiles * p.nsplit);
    const int sp = bid / (p.mtiles * p.ntiles), tile = bid - sp * (p.mtiles * p.ntiles);
    const int nt = p.xmajor ? tile % p.ntiles : tile / p.mtiles, mt = p.xmajor ? tile / p.ntiles : tile - nt * p.mtiles;
    const int t0 = mt * TB, n0 = nt * KB;
    const int cb = sp * p.cps, ce = min(cb + p.cps, p.nch);          // this workgroup's chunks (input-channel range of a split plan)
    auto stamp = [&](int slot) {
        if (p.trace && tid == 0) p.trace[(long)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rsU = make_rsrc(p.U, p.u_bytes);

    // ---- input-transform role: tile tl = 32 (wave & 1) + lane / 2, channel quad q = lane & 1, transform row xi = wave / 2 ---------------
    // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: row xi of B^T d is  A + sgn * C  of two patch rows (A, C):
    //   xi 0: (0, 2) -   xi 1: (1, 2) +   xi 2: (2, 1) -   xi 3: (1, 3) -        -- one instruction stream for all eight waves
    const int tl = 32 * (wave & 1) + (lane >> 1), q = lane & 1, xi = wave >> 1;
    const float sgn = xi == 1 ? 1.f : -1.f;
    int voff[2][4];
    {
        const int t = t0 + tl;
        const bool tok = t < p.T;
        const int tt = tok ? t : 0;
        const int n = fast_div(tt, p.mul_thw, p.shr_thw);
        const int rem = tt - n * p.THW;
        const int ty = fast_div(rem, p.mul_tw, p.shr_tw);
        const int tx = rem - ty * p.TW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int prow = i == 0 ? (xi == 0 ? 0 : (xi == 2 ? 2 : 1)) : (xi == 2 ? 1 : (xi == 3 ? 3 : 2));
            const int iy = 2 * ty - 1 + prow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ix = 2 * tx - 1 + j;
                const bool ok = tok & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                voff[i][j] = ok ? ((((n * p.H + iy) * p.W + ix) * p.ldx) + q * 4) * 4 : (int)OOB;
            }
        }
    }
    const int v_wr = (4 * xi) * PLANE + tl * ROWB + ((q ^ ((tl >> 3) & 1)) << 4);   // this thread's 16 bytes of V rows 4 xi .. 4 xi + 3
    // TWO register sets of patch rows: the loads of chunk c + 3 are issued while chunk c is multiplied (set = chunk parity), 1.6 chunks
    // (~8000 cycles) ahead of their transform.  A patch pixel's 128-byte line holds 4 chunks, so every fourth chunk's loads go to HBM;
    // with one set (issued ~2500 cycles ahead) that stall cost ~590 of a chunk's 5050 cycles (profiles/r05_wino_knockout.txt).
    f32x4 d[2][2][4];
    auto loadVrow = [&](int chunk, int i, int set) {                 // patch row slot i of `chunk` into set `set` (literals)
        const int so = chunk < ce ? chunk * ROWB : 0;
        const int dead = chunk < ce ? 0 : (int)OOB;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            d[set][i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, voff[i][j] | dead, so, 0));
    };
    f32x4 r[4];
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    auto rowV = [&](int set) {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = sgn * d[set][1][j] + d[set][0][j];
    };
    auto colsV = [&](int buf, int pair) {                            // positions 4 xi + 2 pair, + 1
        lds_char* const base = L + buf * OPB + v_wr;
        if (pair == 0) {
            *(lds_f4*)(base + 0 * PLANE) = r[0] - r[2];
            *(lds_f4*)(base + 1 * PLANE) = r[1] + r[2];
        } else {
            *(lds_f4*)(base + 2 * PLANE) = r[2] - r[1];
            *(lds_f4*)(base + 3 * PLANE) = r[1] - r[3];
        }
    };
    // ---- filter DMA role: the chunk's 32-KB image is 32 pieces of 1 KB; wave w moves pieces 4 w .. 4 w + 3 ---------------------------
    const int u_base = (nt * p.nch) * OPB;                           // byte offset of this column block's first chunk
    auto dmaU = [&](int chunk, int buf) {
        const int dead = chunk < ce ? 0 : (int)OOB;
        const int so = u_base + (chunk < ce ? chunk : 0) * OPB + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (lds_void*)(L + 2 * OPB + buf * OPB + wave * 4096 + i * 1024), 16,
                                                     (lane * 16 + i * 1024) | dead, so, 0, 0);
    };

    // ---- MFMA role: wave = (position half ph, tile half wr, channel half wc): positions 8 ph .. 8 ph + 7 of a 32-tile x 32-channel block;
    // waves w and w + 4 share a SIMD (the two position halves of one block).  Fragment = 16 bytes of row l31, slot half ^ swizzle ------
    const int ph = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
    const int f_row_v = 32 * wr + l31, f_row_u = 32 * wc + l31;
    const int fv = (8 * ph) * PLANE + f_row_v * ROWB + ((half ^ ((f_row_v >> 3) & 1)) << 4);
    const int fu = 2 * OPB + (8 * ph) * PLANE + f_row_u * ROWB + ((half ^ ((f_row_u >> 3) & 1)) << 4);
    f32x16 acc[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pp][e] = 0.f;
    typedef const __attribute__((address_space(3))) f32x4 lds_cf4;

    // ---- prologue: chunk 0 staged, the patch rows of chunks 1 and 2 in flight ----------------------------------------------------------
    dmaU(cb, 0);
    loadVrow(cb, 0, 0);
    loadVrow(cb, 1, 0);
    loadVrow(cb + 1, 0, 1);
    loadVrow(cb + 1, 1, 1);
    rowV(0);
    colsV(0, 0);
    colsV(0, 1);
    loadVrow(cb + 2, 0, 0);
    loadVrow(cb + 2, 1, 0);
    wait_vm<16>();                                   // the filter pieces of chunk 0 are older than the 16 loads of chunks 1 and 2
    __syncthreads();
    stamp(1);
    // One chunk = 8 steps (this wave's positions), each {fragments of the NEXT position, 4 MFMAs on this position's accumulator block
    // (256 cycles of pipe time), a slice of the staging work}; the SIMD's other wave (the other position half) fills the matrix pipe
    // while this one stages.  Slices: step 0 the filter DMA of chunk c + 1 (4 pieces), 1 the row transform of chunk c + 1's patch rows
    // (register set (c + 1) & 1), 2-3 its column transform + four 16-byte LDS stores, 4-5 the patch loads of chunk c + 3 into the set
    // just consumed.  The body is instantiated for both parities of c (static register sets and LDS slots).
    auto body = [&](int c, auto PAR) {
        constexpr int buf = decltype(PAR)::value, oth = buf ^ 1;
        lds_char* const Vb = L + buf * OPB + fv;
        lds_char* const Ub = L + buf * OPB + fu;
        f32x4 fa[2], fb[2];
        fa[0] = *(lds_cf4*)(Ub);
        fb[0] = *(lds_cf4*)(Vb);
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            if (pp < 7) {
                fa[(pp + 1) & 1] = *(lds_cf4*)(Ub + (pp + 1) * PLANE);
                fb[(pp + 1) & 1] = *(lds_cf4*)(Vb + (pp + 1) * PLANE);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[pp & 1][s4], fb[pp & 1][s4], acc[pp], 0, 0, 0);
            if (pp == 0) dmaU(c + 1, oth);           // (that slot was last read in iteration c - 1; every wave is past its barrier)
            if (pp == 1) rowV(oth);
            if (pp == 2) colsV(oth, 0);
            if (pp == 3) colsV(oth, 1);
            if (pp == 4) loadVrow(c + 3, 0, oth);
            if (pp == 5) loadVrow(c + 3, 1, oth);
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vm<8>();                                // filter pieces of chunk c + 1 home (only the 8 loads of chunk c + 3 are younger)
        __syncthreads();                             // + this wave's V rows written, every wave done reading slot `buf`
    };
    for (int c = cb; c < ce; c += 2) {           // (the parity of the body is the chunk's position in the range, not its index)
        body(c, std::integral_constant<int, 0>{});
        if (c + 1 < ce) body(c + 1, std::integral_constant<int, 1>{});
    }
    wait_vm<0>();
    __syncthreads();
    stamp(2);

    // ---- output transform: Y = A^T M A (A^T = [1 1 1 0; 0 1 -1 -1]).  A lane holds element (tile l31 of half wr, channel 8 g + 4 half + e)
    // of ITS eight positions = transform rows xi = 2 ph, 2 ph + 1: the column transform s[xi][j] is local, the row transform
    // y[0][j] = s0 + s1 + s2, y[1][j] = s1 - s2 - s3 splits into the two position halves' partial sums
    //     ph 0: (s0 + s1, s1)          ph 1: (s2, -s2 - s3)
    // which go to two staging images [ph][pixel (i, j) of the 2 x 2][tile 0..63][64 channels]; the epilogue adds them.
    const float c0 = ph ? 0.f : 1.f, c1 = ph ? -1.f : 0.f, c2 = ph ? -1.f : 1.f;
    const int strow = 32 * wr + l31;
    lds_char* const ST = L + ph * (256 * EP_ROW);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 y[2][2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rr = 4 * g + e;
            float sx[2][2];
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2) {
                sx[x2][0] = acc[4 * x2 + 0][rr] + acc[4 * x2 + 1][rr] + acc[4 * x2 + 2][rr];
                sx[x2][1] = acc[4 * x2 + 1][rr] - acc[4 * x2 + 2][rr] - acc[4 * x2 + 3][rr];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                y[0][j][e] = sx[0][j] + c0 * sx[1][j];
                y[1][j][e] = c1 * sx[0][j] + c2 * sx[1][j];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *(lds_f4*)(ST + ((2 * i + j) * 64 + strow) * EP_ROW + (32 * wc + 8 * g + 4 * half) * 4) = y[i][j];
    }
    __syncthreads();
    stamp(3);
    // ---- fused epilogue: a thread's staged rows are the 4 pixels of tiles t0 + tid / 16 and + 32 ----------------------------------------
    long pix0[2];
    bool ok[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        const int t = t0 + (tid >> 4) + 32 * k2;
        ok[k2] = t < p.T;
        const int tt = ok[k2] ? t : 0;
        const int n = fast_div(tt, p.mul_thw, p.shr_thw);
        const int rem = tt - n * p.THW;
        const int ty = fast_div(rem, p.mul_tw, p.shr_tw);
        const int tx = rem - ty * p.TW;
        pix0[k2] = ((long)n * p.H + 2 * ty) * p.W + 2 * tx;
    }
    wino_epilogue(p, L, tid, n0, sp, pix0, ok);
    stamp(4);
}

// ================================================================================================
// wino_block_kernel: the same kernel for layers whose tiles can be cut into blocks of 4 (wide) x 16 (tall) on the STACK of all images'
// tile rows (W % 8 == 0, N * H % 32 == 0: every big layer of the graphs).  What changes is how the input patches arrive.  In
// wino_kernel every thread fetches its own 2 x 4 patch pixels from L2 (32 bytes of a 128-byte line per lane pair): each pixel of the
// workgroup's region is requested ~6 times (4 overlapping patches x the row pairs of the 4 transform rows) as 32 distinct lines per wave
// instruction -- the texture path, not the matrix pipe, paced the loop (knock-out builds, profiles/r05_wino_knockout.txt: 5030 cycles
// per chunk with the loads, 4430 without, 4096 = the MFMAs).  Here the block's 34 x 10 raw pixels of a chunk (340 x 32 bytes) are
// gathered ONCE by LDS-DMA (11 pieces per workgroup instead of 64 load instructions), two chunks ahead, and the transform threads read
// their patch pixels from LDS.  Rows that belong to a neighbouring image in the stack (a tile row at an image border needs zero
// padding where the stack holds the other image's pixels) are read from a 32-byte zero slot instead -- an address select per thread,
// made once; out-of-image columns and rows beyond the stack are out-of-range DMA lanes (zeros).  Output pixels of the block need no
// division either: the stack's pixel row of tile row R is 2 R + i.
// ================================================================================================
constexpr int RAW_PIECES = 11, RAWB = RAW_PIECES * 1024;             // 340 pixels x 32 B in 11 DMA pieces
constexpr int RAW_OFF = 4 * OPB, ZERO_OFF = RAW_OFF + 2 * RAWB, SMEM_BLOCK = ZERO_OFF + 64;
static_assert(SMEM_BLOCK <= 163840 && SMEM <= SMEM_BLOCK, "LDS plan");

// one workgroup's work item vb (a virtual block index: what the block index is to a launch with one workgroup per item)
template <bool SLICE_FIRST>
__device__ __forceinline__ void wino_block_body(const WParams& p, lds_char* const L, const int vb) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int bid = xcd_remap(vb, p.mtiles * p.ntiles * p.nsplit);
    const int sp = bid / (p.mtiles * p.ntiles), tile = bid - sp * (p.mtiles * p.ntiles);
    const int nt = p.xmajor ? tile % p.ntiles : tile / p.mtiles, mt = p.xmajor ? tile / p.ntiles : tile - nt * p.mtiles;
    const int n0 = nt * KB;
    const int cb = sp * p.cps, ce = min(cb + p.cps, p.nch);
    const int bcols = p.TW >> 2;
    const int brow = mt / bcols, bcol = mt - brow * bcols;
    const int R0 = 16 * brow, C0 = 4 * bcol;                         // first stacked tile row / tile column of the block
    auto stamp = [&](int slot) {
        if (p.trace && tid == 0) {
            p.trace[(long)vb * 8 + slot] = __builtin_amdgcn_s_memtime();
            // (s_memtime counts per XCD: the item's start and end also on the 100-MHz clock all XCDs share, for ramp / tail analysis)
            if (slot == 0) p.trace[(long)vb * 8 + 5] = __builtin_amdgcn_s_memrealtime();
            if (slot == 4) p.trace[(long)vb * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rsU = make_rsrc(p.U, p.u_bytes);
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    typedef const __attribute__((address_space(3))) f32x4 lds_cf4;
    if (tid < 4) *(lds_f4*)(L + ZERO_OFF + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- raw-gather role: piece ids wave and wave + 8 (< 11); lane slot s = 64 id + lane = (raw pixel s / 2, 16-byte half s & 1) -------
    int g_voff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int s = (wave + 8 * k) * 64 + lane;
        const int idx = s >> 1, hq = s & 1;
        const int row = idx / 10, colr = idx - row * 10;
        const int g = 2 * R0 - 1 + row, x = 2 * C0 - 1 + colr;       // stacked pixel row, column
        const bool ok = (wave + 8 * k < RAW_PIECES) & (idx < 340) & ((unsigned)g < (unsigned)(p.N * p.H)) & ((unsigned)x < (unsigned)p.W);
        g_voff[k] = ok ? ((g * p.W + x) * p.ldx + hq * 4) * 4 : (int)OOB;
    }
    auto dmaRaw = [&](int chunk, int slot) {
        const int dead = chunk < ce ? 0 : (int)OOB;
        const int so = chunk < ce ? chunk * ROWB : 0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(L + RAW_OFF + slot * RAWB + wave * 1024), 16, g_voff[0] | dead, so, 0, 0);
        if (wave + 8 < RAW_PIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_void*)(L + RAW_OFF + slot * RAWB + (wave + 8) * 1024), 16, g_voff[1] | dead, so, 0, 0);
    };
    // ---- input-transform role: tile tl = 32 (wave & 1) + lane / 2 = (block row tl / 4, block column tl % 4), quad q, transform row xi ---
    const int tl = 32 * (wave & 1) + (lane >> 1), q = lane & 1, xi = wave >> 1;
    const float sgn = xi == 1 ? 1.f : -1.f;
    const int bty = tl >> 2, btx = tl & 3;
    int roff[2][4];                                                   // LDS byte offsets (within a raw slot, or the zero slot) of the 2 x 4 patch pixels
    {
        const int R = R0 + bty;
        const int n = fast_div(R, p.mul_th, p.shr_th);
        const int ty = R - n * (p.H >> 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int prow = i == 0 ? (xi == 0 ? 0 : (xi == 2 ? 2 : 1)) : (xi == 2 ? 1 : (xi == 3 ? 3 : 2));
            const int y = 2 * ty - 1 + prow;
            const bool ok = (unsigned)y < (unsigned)p.H;             // else: the stack holds the neighbouring image there -> zero padding
#pragma unroll
            for (int j = 0; j < 4; ++j)
                roff[i][j] = ok ? ((2 * bty + prow) * 10 + 2 * btx + j) * 32 + q * 16 : -1;
        }
    }
    const int v_wr = (4 * xi) * PLANE + tl * ROWB + ((q ^ ((tl >> 3) & 1)) << 4);
    f32x4 d[2][4], r[4];
    auto readRaw = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                d[i][j] = *(lds_cf4*)(L + (roff[i][j] >= 0 ? RAW_OFF + slot * RAWB + roff[i][j] : ZERO_OFF + q * 16));
    };
    auto rowV = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = sgn * d[1][j] + d[0][j];
    };
    auto colsV = [&](int buf, int pair) {
        lds_char* const base = L + buf * OPB + v_wr;
        if (pair == 0) {
            *(lds_f4*)(base + 0 * PLANE) = r[0] - r[2];
            *(lds_f4*)(base + 1 * PLANE) = r[1] + r[2];
        } else {
            *(lds_f4*)(base + 2 * PLANE) = r[2] - r[1];
            *(lds_f4*)(base + 3 * PLANE) = r[1] - r[3];
        }
    };
    const int u_base = (nt * p.nch) * OPB;
    auto dmaU = [&](int chunk, int buf) {
        const int dead = chunk < ce ? 0 : (int)OOB;
        const int so = u_base + (chunk < ce ? chunk : 0) * OPB + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsU, (lds_void*)(L + 2 * OPB + buf * OPB + wave * 4096 + i * 1024), 16,
                                                     (lane * 16 + i * 1024) | dead, so, 0, 0);
    };
    // ---- MFMA role (as wino_kernel) ---------------------------------------------------------------------------------------------------
    const int ph = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
    const int f_row_v = 32 * wr + l31, f_row_u = 32 * wc + l31;
    const int fv = (8 * ph) * PLANE + f_row_v * ROWB + ((half ^ ((f_row_v >> 3) & 1)) << 4);
    const int fu = 2 * OPB + (8 * ph) * PLANE + f_row_u * ROWB + ((half ^ ((f_row_u >> 3) & 1)) << 4);
    f32x16 acc[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pp][e] = 0.f;

    // ---- prologue: filter chunk 0 and the raw pixels of chunks 0 and 1 gathered; chunk 0 transformed ------------------------------------
    dmaU(cb, 0);
    dmaRaw(cb, 0);
    dmaRaw(cb + 1, 1);
    wait_vm<0>();
    __syncthreads();
    readRaw(0);
    rowV();
    colsV(0, 0);
    colsV(0, 1);
    __syncthreads();                                 // V slot 0 complete, raw slot 0 free
    stamp(1);
    // chunk c: step 0 {filter DMA of chunk c + 1, raw gather of chunk c + 2 into the raw slot chunk c was transformed from}, step 1 the
    // eight 16-byte LDS reads of chunk c + 1's patch pixels, 2 the row transform, 3-4 the column transform + four LDS stores.
    auto body = [&](int c, auto PAR) {
        constexpr int buf = decltype(PAR)::value, oth = buf ^ 1;
        lds_char* const Vb = L + buf * OPB + fv;
        lds_char* const Ub = L + buf * OPB + fu;
        f32x4 fa[2], fb[2];
        fa[0] = *(lds_cf4*)(Ub);
        fb[0] = *(lds_cf4*)(Vb);
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            if (pp < 7) {
                fa[(pp + 1) & 1] = *(lds_cf4*)(Ub + (pp + 1) * PLANE);
                fb[(pp + 1) & 1] = *(lds_cf4*)(Vb + (pp + 1) * PLANE);
            }
            if (SLICE_FIRST) {
                __builtin_amdgcn_sched_barrier(0);
                if (pp == 1) readRaw(oth);
                if (pp == 2) rowV();
                if (pp == 3) colsV(oth, 0);
                if (pp == 4) colsV(oth, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[pp & 1][s4], fb[pp & 1][s4], acc[pp], 0, 0, 0);
            if (pp == 0) { dmaU(c + 1, oth); dmaRaw(c + 2, buf); }
            if (!SLICE_FIRST) {
                if (pp == 1) readRaw(oth);
                if (pp == 2) rowV();
                if (pp == 3) colsV(oth, 0);
                if (pp == 4) colsV(oth, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_vm<0>();                                // this wave's DMA pieces (issued seven steps ago) are home
        __syncthreads();
    };
    for (int c = cb; c < ce; c += 2) {
        body(c, std::integral_constant<int, 0>{});
        if (c + 1 < ce) body(c + 1, std::integral_constant<int, 1>{});
    }
    stamp(2);

    // ---- output transform + staging (as wino_kernel) -----------------------------------------------------------------------------------
    const float c0 = ph ? 0.f : 1.f, c1 = ph ? -1.f : 0.f, c2 = ph ? -1.f : 1.f;
    const int strow = 32 * wr + l31;
    lds_char* const ST = L + ph * (256 * EP_ROW);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 y[2][2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rr = 4 * g + e;
            float sx[2][2];
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2) {
                sx[x2][0] = acc[4 * x2 + 0][rr] + acc[4 * x2 + 1][rr] + acc[4 * x2 + 2][rr];
                sx[x2][1] = acc[4 * x2 + 1][rr] - acc[4 * x2 + 2][rr] - acc[4 * x2 + 3][rr];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                y[0][j][e] = sx[0][j] + c0 * sx[1][j];
                y[1][j][e] = c1 * sx[0][j] + c2 * sx[1][j];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *(lds_f4*)(ST + ((2 * i + j) * 64 + strow) * EP_ROW + (32 * wc + 8 * g + 4 * half) * 4) = y[i][j];
    }
    __syncthreads();
    stamp(3);
    // ---- fused epilogue: the stack's pixel of tile (block row, block column), sub-pixel (i, j) is (2 R + i) W + 2 (C0 + btx) + j ---------
    long pix0[2];
    const bool ok[2] = {true, true};
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        const int tloc = (tid >> 4) + 32 * k2;
        pix0[k2] = (long)(2 * (R0 + (tloc >> 2))) * p.W + 2 * (C0 + (tloc & 3));
    }
    wino_epilogue(p, L, tid, n0, sp, pix0, ok);
    stamp(4);
}

// Persistent launch: the host starts one workgroup per CU (or one per item when there are fewer) and every workgroup walks the items
// vb = blockIdx.x, + gridDim.x, ... (gridDim.x a multiple of the XCD count keeps an item's XCD what xcd_remap assumes).  With one
// 136-KB workgroup per CU nothing else can be resident while the dispatcher retires a workgroup and starts the next: that gap
// (~6 us per round) was the difference between 4 rounds x 87.7 k cycles = 146 us and the 178 us a C = 128 layer took.
template <bool SLICE_FIRST>
__global__ __launch_bounds__(512, 2) void wino_block_kernel(const WParams p) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BLOCK];
    lds_char* const L = (lds_char*)smem;
    const int total = p.mtiles * p.ntiles * p.nsplit;
    for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
        wino_block_body<SLICE_FIRST>(p, L, vb);
        __syncthreads();                              // the epilogue's staging reads are done before the next item's DMA lands
    }
}

// ================================================================================================
// Filter gradient: F(3x3, 2x2).  dW[r][s][c][k] = sum over tiles of sum_{i,j in 2x2} d[r + i][s + j][c] dy[i][j][k]  (d = the tile's 4 x 4
// input patch) is, per tile, a 3 x 3-output correlation with a 2 x 2 "filter" dy: minimal filtering gives
//        dW = A'^T [ sum_tiles (B'^T d B') o (G' dy G'^T) ] A'
//   B'^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1]   G' = [1 0; 1/2 1/2; 1/2 -1/2; 0 1]   A'^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1]
// i.e. again 16 position GEMMs  dU_p[c, k] = sum_t V'_p[t, c] Y'_p[t, k]  with the TILES as the reduction axis: 16 multiplies per
// (tile, c, k) instead of 36.  A workgroup owns a 64 x 64 block of (c, k), all 16 positions (eight waves: 2 position halves x 2 x 2
// 32 x 32 sub-blocks, 128 accumulator registers each) and a contiguous range of tiles, which it walks in chunks of 8: per chunk a wave
// takes one tile, a thread loads one row of its input patch (4 x 16 bytes; the second row its transform row combines comes from a
// neighbouring lane by ds_bpermute) and the tile's 2 x 2 dy pixels (4 x 16 bytes), transforms both in registers and writes 16-byte
// rows [position][tile][channel] into LDS; the MFMA fragments are 4-byte LDS reads (the reduction index of v_mfma_f32_32x32x2_f32 runs
// across lanes).  At the end each wave applies A'^T . A' to its own position half in registers (nine 32 x 32 tap blocks), the workgroup's two position halves meet in LDS, and the block leaves ONE of S partial filter gradients
// [3][3][C][K]; wino_wgrad_reduce_kernel sums them in fixed order
// (deterministic) into dw (+ beta dw).  The bias gradient db[k] = sum over pixels of dy rides along: the 2 x 2 dy pixels of the tiles
// partition the map, so the workgroups of channel block 0 add up what their transform threads load anyway (the lanes of transform row 0 deliver)
// and leave one [K] partial per split; the reduce kernel sums those in split order too.
// ================================================================================================
struct WGParams {
    const float* X;       // forward input, NHWC, channel stride ldx
    const float* DY;      // output gradient, NHWC, channel stride ldy
    float* part;          // [nsplit][3][3][C][K]
    float* bias_part;     // [nsplit][K] after the filter slabs, or null: no bias gradient asked
    int N, H, W, C, K, ldx, ldy;
    int T, THW, TW;
    int nchunks, cps;     // 8-tile chunks in the batch, chunks per split
    int cblocks, kblocks, nsplit;
    unsigned x_bytes, y_bytes;
    unsigned mul_thw, shr_thw, mul_tw, shr_tw;
};

template <bool SLICE_FIRST, int KO = 0, bool XCH = true>      // XCH: the x row transform's partner row through LDS instead of ds_bpermute; KO (builds with -DDPIG_WINO4_KNOCKOUT only; results are wrong): 1 no x arithmetic, 2 no dy arithmetic, 4 no ds_bpermute, 8 no per-chunk address arithmetic
__global__ __launch_bounds__(512, 2) void wino_wgrad_kernel(const WGParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * OPB + 8 * 4096];      // + a 4-KB exchange area per wave (row transform of x)
    lds_char* const L = (lds_char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // block order: the (c, k) blocks of one tile range are consecutive, so the workgroups of an XCD read one range of x / dy
    const int bid = xcd_remap(blockIdx.x, p.cblocks * p.kblocks * p.nsplit);
    const int nb = p.cblocks * p.kblocks;
    const int sp = bid / nb, blk = bid - sp * nb;
    const int cb = blk / p.kblocks, kb = blk - cb * p.kblocks;
    const int ch0 = sp * p.cps, ch1 = min(ch0 + p.cps, p.nchunks);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rsY = make_rsrc(p.DY, p.y_bytes);

    // ---- transform roles: tile slot tl = wave, channel quad cq = lane % 16, transform row xi = lane / 16 ------------------------------
    // The loop is short of L1 request bandwidth, not of arithmetic (knock-out builds: without the global loads it runs 302 instead of
    // 238 TFLOP/s effective on dec3, without a quarter of the LDS fragment reads not a bit faster), so the requests are what is
    // economised: a thread loads ONE patch row (its own transform row's first operand, 4 pixels) and takes the second operand of
    // B'^T from the lane that loaded it (ds_bpermute: 16 LDS-crossbar moves, no memory) -- 4 patch loads instead of 8 -- and the
    // wave's four transform rows ask for the SAME 2 x 2 dy pixels of the wave's one tile, which the texture unit serves as 2 lines
    // per instruction instead of 8.  Per workgroup and chunk: 64 load instructions / 320 line requests instead of 96 / 768.
    const int cq = lane & 15, xi = lane >> 4, tl = wave;
    // B'^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1]: row xi = (own patch row xi) + sgn (patch row 2 for xi < 2, else patch row 1)
    const float sgn = xi == 1 ? 1.f : -1.f;
    const int peer = (((xi < 2) ? 2 : 1) * 16 + cq) * 4;            // ds_bpermute address of the lane that holds the other row
    // G' row xi of the 2 x 2 dy = ga y0 + gb y1:  (1, 0), (1/2, 1/2), (1/2, -1/2), (0, 1)
    const float ga = xi == 0 ? 1.f : (xi == 3 ? 0.f : 0.5f), gb = xi == 0 ? 0.f : (xi == 1 ? 0.5f : (xi == 2 ? -0.5f : 1.f));
    const int xcol = (cb * 64 + cq * 4) * 4, ycol = (kb * 64 + cq * 4) * 4;      // byte offsets of this thread's channel quads
    // Per-chunk addressing: the wave's tile is wave-uniform (scalar divisions), a lane adds its patch row and channel quad; 24-bit
    // multiplies (full rate).  The stack's pixel row of tile row (n, ty) is n H + 2 ty.
    const int ldx4 = p.ldx * 4, ldy4 = p.ldy * 4;
    const int rsx = p.W * ldx4, rsy = p.W * ldy4;                   // bytes per pixel row
    // Addresses = (scalar part of the wave's tile, per chunk, in the loads' scalar offset) + (a lane's patch row and channel quad, made
    // once).  The x descriptor starts one pixel row and one pixel BEFORE the tensor so that the scalar part (tile's pixel row 2 ty - 1,
    // column 2 tx - 1) is never negative; nothing is read there: a lane whose patch row or column lies outside the image gets an
    // out-of-range vector offset (zero fill).  Per chunk that is three vector selects instead of ~30 vector instructions of address
    // arithmetic -- knock-out builds put 4.5 % of the kernel on them
    // (scripts/ko_wgrad.sh; every vector instruction of this loop costs matrix-pipe time).
    const __amdgpu_buffer_rsrc_t rsX1 = make_rsrc(reinterpret_cast<const char*>(p.X) - (rsx + ldx4), p.x_bytes + (unsigned)(rsx + ldx4));
    const int vx_lane = xi * rsx + xcol;                             // + j ldx4 in the scalar offset
    int sx, sy;                                                      // scalar parts of the current tile (x: row r2 - 1 / column 2 tx - 1 folded into the descriptor)
    int xv;                                                          // this lane's vector offset for the x loads, or out of range
    bool x_left, x_right, y_ok;                                      // wave-uniform: patch column 0 / 3 inside the image, tile inside the range
    auto tile_offsets = [&](int chunk) {                             // the tile 8 chunk + tl
        const int t = chunk * 8 + tl;
        const bool tok = (chunk < ch1) & (t < p.T);
        const int tt = tok ? t : 0;
        const int n = fast_div(tt, p.mul_thw, p.shr_thw);
        const int rem = tt - n * p.THW;
        const int ty = fast_div(rem, p.mul_tw, p.shr_tw);
        const int tx = rem - ty * p.TW;
        const int r2 = n * p.H + 2 * ty;
        sx = r2 * rsx + 2 * tx * ldx4;
        sy = r2 * rsy + 2 * tx * ldy4;
        const bool dead = !tok | ((ty == 0) & (xi == 0)) | ((2 * ty + 2 == p.H) & (xi == 3));
        xv = dead ? (int)OOB : vx_lane;
        x_left = tok & (tx > 0);
        x_right = tok & (tx < p.TW - 1);
        y_ok = tok;
    };
    f32x4 d[4], y[2][2];
    auto loadX = [&]() {                             // (the edge columns by an out-of-range vector offset too: no load is ever conditional)
        d[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX1, x_left ? xv : (int)OOB, sx, 0));
        d[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX1, xv, sx + ldx4, 0));
        d[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX1, xv, sx + 2 * ldx4, 0));
        d[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX1, x_right ? xv : (int)OOB, sx + 3 * ldx4, 0));
    };
    auto loadY = [&]() {
        const int yv = y_ok ? ycol : (int)OOB;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) y[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, yv, sy + i * rsy + j * ldy4, 0));
    };
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    const int wr_off = (4 * xi) * PLANE + tl * 256 + cq * 16;        // [position 4 xi + nu][tile tl][channel quad cq]
    f32x4 r[4], ry[2];
    const bool sum_bias = p.bias_part != nullptr && cb == 0;         // (workgroup-uniform; the lanes of transform row 0 deliver the sums)
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto rowX = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float own = d[j][e];               // (a named scalar: bit-casting the vector element in place made clang move element 0 four times)
                const float o = (KO & 4) ? own : __int_as_float(__builtin_amdgcn_ds_bpermute(peer, __float_as_int(own)));
                r[j][e] = (KO & 1) ? o : sgn * o + own;
            }
    };
    // the same through the wave's exchange area: four 16-byte stores of the own row, four 16-byte reads of the partner's (the LDS runs a
    // wave's operations in order: no wait between them) instead of sixteen ds_bpermute_b32 -- knock-out builds put 6 % of the kernel on
    // those sixteen (scripts/ko_wgrad.sh)
    const int xch_wr = 4 * OPB + wave * 4096 + lane * 16, xch_rd = 4 * OPB + wave * 4096 + (((xi < 2) ? 2 : 1) * 16 + cq) * 16;
    typedef const __attribute__((address_space(3))) f32x4 lds_cf4x;
    auto rowX2 = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(lds_f4*)(L + xch_wr + j * 1024) = d[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 o = *(lds_cf4x*)(L + xch_rd + j * 1024);
            r[j] = sgn * o + d[j];
        }
    };
    auto colsX = [&](int buf, int pair) {
        lds_char* const base = L + buf * OPB + wr_off;
        if (KO & 1) {
            *(lds_f4*)(base + (2 * pair) * PLANE) = r[2 * pair];
            *(lds_f4*)(base + (2 * pair + 1) * PLANE) = r[2 * pair + 1];
        } else if (pair == 0) {
            *(lds_f4*)(base + 0 * PLANE) = r[0] - r[2];
            *(lds_f4*)(base + 1 * PLANE) = r[1] + r[2];
        } else {
            *(lds_f4*)(base + 2 * PLANE) = r[2] - r[1];
            *(lds_f4*)(base + 3 * PLANE) = r[3] - r[1];
        }
    };
    auto rowY = [&]() {
        if (KO & 2) { ry[0] = y[0][0]; ry[1] = y[1][1]; return; }
#pragma unroll
        for (int j = 0; j < 2; ++j) ry[j] = ga * y[0][j] + gb * y[1][j];
        if (sum_bias) bsum += (y[0][0] + y[0][1]) + (y[1][0] + y[1][1]);     // (tiles past the range loaded zeros)
    };
    auto colsY = [&](int buf, int pair) {
        lds_char* const base = L + 2 * OPB + buf * OPB + wr_off;
        if (KO & 2) {
            *(lds_f4*)(base + (2 * pair) * PLANE) = ry[0];
            *(lds_f4*)(base + (2 * pair + 1) * PLANE) = ry[1];
        } else if (pair == 0) {
            *(lds_f4*)(base + 0 * PLANE) = ry[0];
            *(lds_f4*)(base + 1 * PLANE) = 0.5f * (ry[0] + ry[1]);
        } else {
            *(lds_f4*)(base + 2 * PLANE) = 0.5f * (ry[0] - ry[1]);
            *(lds_f4*)(base + 3 * PLANE) = ry[1];
        }
    };

    // ---- MFMA role: wave = (position half ph, channel half cr of the 64 c, half kc of the 64 k); A = V'[tile][c], B = Y'[tile][k] ------
    const int ph = wave >> 2, cr = (wave >> 1) & 1, kc = wave & 1;
    const int fa_off = (8 * ph) * PLANE + half * 256 + (32 * cr + l31) * 4;
    const int fb_off = 2 * OPB + (8 * ph) * PLANE + half * 256 + (32 * kc + l31) * 4;
    f32x16 acc[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[pp][e] = 0.f;
    typedef const __attribute__((address_space(3))) float lds_cf;

    // ---- prologue: chunk ch0 staged, chunk ch0 + 1 in registers ------------------------------------------------------------------------
    tile_offsets(ch0);
    loadX(); loadY();
    if (XCH) rowX2(); else rowX();
    colsX(0, 0); colsX(0, 1);
    rowY(); colsY(0, 0); colsY(0, 1);
    tile_offsets(ch0 + 1);
    loadX(); loadY();
    __syncthreads();
    for (int c = ch0; c < ch1; ++c) {
        const int buf = (c - ch0) & 1;
        lds_char* const Ab = L + buf * OPB + fa_off;
        lds_char* const Bb = L + buf * OPB + fb_off;
        float fa[2][4], fb[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            fa[0][s] = *(lds_cf*)(Ab + s * 512);
            fb[0][s] = *(lds_cf*)(Bb + s * 512);
        }
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            if (pp < 7) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    fa[(pp + 1) & 1][s] = *(lds_cf*)(Ab + (pp + 1) * PLANE + s * 512);
                    fb[(pp + 1) & 1][s] = *(lds_cf*)(Bb + (pp + 1) * PLANE + s * 512);
                }
            }
            auto slice = [&]() {
                if (pp == 0) { if (!(KO & 8)) tile_offsets(c + 2); if (XCH) rowX2(); else rowX(); rowY(); }
                if (pp == 1) { loadX(); loadY(); }
                if (pp == 2) colsX(buf ^ 1, 0);
                if (pp == 3) colsX(buf ^ 1, 1);
                if (pp == 4) colsY(buf ^ 1, 0);
                if (pp == 5) colsY(buf ^ 1, 1);
            };
            if (SLICE_FIRST) {                       // (the order the F(4x4, 3x3) kernel measured 6 % faster: fragment reads, slice, MFMAs)
                __builtin_amdgcn_sched_barrier(0);
                slice();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[pp & 1][s], fb[pp & 1][s], acc[pp], 0, 0, 0);
            // staging slices.  The row transforms of chunk c + 1 come FIRST: they free the load registers, so the loads of chunk c + 2 can
            // follow at once and have seven steps (~5000 cycles) to land before the next chunk's step 0 needs them -- issued at the END
            // of the chunk (two steps ahead of their use) an L2 hit's latency was exposed every chunk: a knock-out build without the
            // loads ran 293 instead of 219 TFLOP/s on dec3 (build/ko/run_wg.sh; an L2 prefetch four chunks ahead changed nothing:
            // the lines were L2-resident already, the other (c, k) blocks of the XCD read them too).
            if (!SLICE_FIRST) slice();
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- output transform A'^T dU A' on this wave's position half (transform rows xi = 2 ph, 2 ph + 1), straight to its partial slab.
    // Column transform (over nu) per row: s0 = M0 + M1 + M2, s1 = M1 - M2, s2 = M1 + M2 + M3; row transform w0 = S0 + S1 + S2,
    // w1 = S1 - S2, w2 = S1 + S2 + S3 splits into  ph 0: (S0 + S1, S1, S1)   ph 1: (S2, -S2, S2 + S3).
    if (p.bias_part != nullptr && cb == 0) {         // (workgroup-uniform) bias partial: 8 tile slots x 16 channel quads -> 64 channels
        typedef __attribute__((address_space(3))) float lds_f;
        if (xi == 0) *(lds_f4*)(L + tl * 256 + cq * 16) = bsum;       // (every lane summed; one transform row's lanes deliver)
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += *(lds_f*)(L + i * 256 + tid * 4);
            p.bias_part[(long)sp * p.K + kb * 64 + tid] = t;
        }
    }
    // The two position halves of a (c, k) block sit in DIFFERENT waves of this workgroup (ph 0: transform rows 0, 1; ph 1: rows 2, 3) and
    // each holds a partial value of all nine taps.  They meet here, through the LDS the loop no longer needs, so that the workgroup
    // leaves ONE partial gradient per split instead of two (round 5 wrote 2 S slabs: twice the stores and twice the bytes the reduce
    // pass reads).  Two rounds of 8 accumulator elements: in round 0 the ph 1 waves hand theirs over and the ph 0 waves add + store, in
    // round 1 the roles swap (balanced stores).  own + partner is one fp32 addition, commutative: the bits do not depend on the round.
    float* const slab = p.part + (long)sp * 9 * p.C * p.K;
    const int kcol = kb * 64 + 32 * kc + l31;
    typedef __attribute__((address_space(3))) float lds_f;
    lds_f* const XL = (lds_f*)L + ((wave & 3) * 72) * 64 + lane;             // [wave pair][8 elements x 9 taps][lane]
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
        const bool sender = (ph != rnd);
        float wv[8][9];
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
            const int e = 8 * rnd + e8;
            float sx[2][3];
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2) {
                const float m0 = acc[4 * x2 + 0][e], m1 = acc[4 * x2 + 1][e], m2 = acc[4 * x2 + 2][e], m3 = acc[4 * x2 + 3][e];
                sx[x2][0] = m0 + m1 + m2;
                sx[x2][1] = m1 - m2;
                sx[x2][2] = m1 + m2 + m3;
            }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const float a = sx[0][s3], b = sx[1][s3];
                wv[e8][0 * 3 + s3] = ph ? a : a + b;
                wv[e8][1 * 3 + s3] = ph ? -a : b;
                wv[e8][2 * 3 + s3] = ph ? a + b : b;
            }
        }
        __syncthreads();                 // the LDS is free: the loop's last reads / the bias rows / the previous round's readers are done
        if (sender) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8)
#pragma unroll
                for (int t = 0; t < 9; ++t) XL[(e8 * 9 + t) * 64] = wv[e8][t];
        }
        __syncthreads();
        if (!sender) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int e = 8 * rnd + e8;
                const int crow = cb * 64 + 32 * cr + 8 * (e >> 2) + 4 * half + (e & 3);
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    slab[((long)t * p.C + crow) * p.K + kcol] = wv[e8][t] + XL[(e8 * 9 + t) * 64];
            }
        }
    }
}

// dw = beta dw + sum of the nparts partial gradients (fixed order: deterministic)
// (+ the bias gradient: db = beta_b db + sum of the nparts partial [K] rows, when db is given)
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long n4, int nparts,
                                                                 float beta, const float* __restrict__ bias_part, float* __restrict__ db,
                                                                 int K, float beta_b) {
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
    f32x4* o4 = reinterpret_cast<f32x4*>(dw);
    const long total = n4 + (db ? K : 0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (i < n4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int s = 0; s < nparts; ++s) v += p4[(long)s * n4 + i];
            if (beta != 0.f) v += beta * o4[i];
            o4[i] = v;
        } else {
            const int k = (int)(i - n4);
            float t = 0.f;
            for (int s = 0; s < nparts; ++s) t += bias_part[(long)s * K + k];
            if (beta_b != 0.f) t += beta_b * db[k];
            db[k] = t;
        }
    }
}

// Second pass of a split plan: sum the nsplit partial outputs in split order (deterministic) and run the fused epilogue.
__global__ __launch_bounds__(256) void wino_reduce_kernel(const WParams p) {
    const int k4 = p.Kout >> 2;
    const long total = (long)p.N * p.H * p.W * k4, slab = (long)p.N * p.H * p.W * p.Kout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / k4;
        const int col = (int)(i - pix * k4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.nsplit; ++s) v += *reinterpret_cast<const f32x4*>(p.partial + s * slab + pix * p.Kout + col);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + col);
        epi4(p, pix, col, v, bv);
    }
}

// ---- filter transform: U = G g G^T (G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]) of every (input channel, output channel) pair, written
// in the kernel's LDS image order.  `dgrad`: the transposed conv's filter g'[r][s][k][c] = w[2 - r][2 - s][c][k] (input channels = the
// forward conv's output channels).  w is HWIO [3][3][C][K].
// One workgroup per (64-channel column block kb, 8-channel chunk): the 512 transforms are staged in LDS in image order and leave as one
// contiguous 32-KB run of 16-byte stores.  Reads follow the filter's fastest axis (forward: output channels, dgrad: input channels).
template <bool DGRAD>
__device__ __forceinline__ void filter_body(const float* __restrict__ w, float* __restrict__ U, int C, int K, int blk, float* img) {
    const int cin = DGRAD ? K : C;
    const int nch = cin / CH;
    const int kb = blk / nch, chunk = blk - kb * nch;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        // (m, cc) = (output channel within the block, reduction channel within the chunk); the lanes run along w's fastest axis
        const int m = DGRAD ? (tid >> 3) + 32 * j : (tid & 63);
        const int cc = DGRAD ? (tid & 7) : (tid >> 6) + 4 * j;
        const int ko = kb * 64 + m, ci = chunk * CH + cc;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s)
                g[r][s] = DGRAD ? w[(((2 - r) * 3 + (2 - s)) * (long)C + ko) * K + ci] : w[((r * 3 + s) * (long)C + ci) * K + ko];
        float t[4][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = g[0][s];
            t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
            t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
            t[3][s] = g[2][s];
        }
        float* const o = img + m * 8 + (((cc >> 2) ^ ((m >> 3) & 1)) << 2) + (cc & 3);
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            o[(xi * 4 + 0) * 512] = t[xi][0];
            o[(xi * 4 + 1) * 512] = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]);
            o[(xi * 4 + 2) * 512] = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]);
            o[(xi * 4 + 3) * 512] = t[xi][2];
        }
    }
    __syncthreads();
    f32x4* const dst = reinterpret_cast<f32x4*>(U + (long)blk * (16 * 512));
    const f32x4* const src = reinterpret_cast<const f32x4*>(img);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[tid + 256 * i] = src[tid + 256 * i];
}
template <bool DGRAD>
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int C, int K) {
    __shared__ __attribute__((aligned(16))) float img[16 * 512];
    filter_body<DGRAD>(w, U, C, K, xcd_remap(blockIdx.x, gridDim.x), img);
}
// Every filter of a parameter set in ONE launch (a model has ~90 of them; one launch pair each was 1.7 ms of a 46-ms step, all launch
// latency): blocks [0, total) write the forward images, [total, 2 total) the dgrad images; a block finds its job by bisection over
// the jobs' first_block.
__global__ __launch_bounds__(256) void wino_filter_jobs_kernel(const DpigWinoFilterJob* __restrict__ jobs, int njobs, int total) {
    __shared__ __attribute__((aligned(16))) float img[16 * 512];
    // (consecutive blocks of the dgrad direction read neighbouring 32-byte pieces of the same filter rows: on ONE XCD they share the
    // lines through its L2 -- dealt round-robin over the XCDs every piece was its own HBM fetch, 2.4 GB read for 0.45 GB of filters)
    const int lb = xcd_remap(blockIdx.x, 2 * total);
    const int dir = lb >= total;
    const int b = lb - dir * total;
    // job = the last one whose first_block <= b: every thread tests a few jobs and the workgroup counts the hits -- ONE round of
    // independent loads (a bisection's chain of eight dependent loads was most of this kernel's time: 618 us for the Market model)
    int cnt = 0;
    for (int i0 = 0; i0 < njobs; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        cnt += __syncthreads_count(i < njobs && jobs[i].first_block <= b);
    }
    const DpigWinoFilterJob j = jobs[cnt - 1];                    // (first_block of job 0 is 0: cnt >= 1)
    if (dir) {
        if (j.u_dgrad) filter_body<true>(j.w, j.u_dgrad, j.C, j.K, b - j.first_block, img);
    } else {
        if (j.u_fwd) filter_body<false>(j.w, j.u_fwd, j.C, j.K, b - j.first_block, img);
    }
}

int launch_reduce(const WParams& p, hipStream_t st) {
    const long total = (long)p.N * p.H * p.W * (p.Kout / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    return check_launch("wino_reduce_kernel");
}

static unsigned long long* g_trace = nullptr;
unsigned long long* trace_buffer() { return g_trace; }
static int g_mode = -1;            // 0 never, 1 where the cost model says it pays (default), 2 wherever legal (tests)
static void init_mode() {
    if (g_mode >= 0) return;
    const char* e = getenv("DPIG_WINO");
    g_mode = e ? atoi(e) : 1;
}

// geometry both entry points share: 3 x 3, stride 1, SAME, even image, 16-byte addressable channel vectors
static bool shape_ok(const DpigConvDesc* d, int cin, int kout, int ld_in, int ld_out) {
    if (d->R != 3 || d->S != 3 || d->stride != 1 || d->upsample2x || d->res_class || d->split_k > 1) return false;
    if ((d->H & 1) || (d->W & 1) || d->H < 2 || d->W < 2) return false;
    if (cin % CH || kout % KB || (ld_in & 3) || (ld_out & 3)) return false;
    if (d->pad_t >= 0 && d->pad_t != 1) return false;
    if (d->pad_l >= 0 && d->pad_l != 1) return false;
    const long lim = 0x7f000000L;
    if ((long)d->N * d->H * d->W * ld_in * 4 >= lim || (long)d->N * d->H * d->W * ld_out * 4 >= lim) return false;
    if ((long)16 * cin * kout * 4 >= lim) return false;
    return true;
}
// Does the fused Winograd kernel beat the direct kernel on this layer?  One workgroup per CU, whole rounds of 256: a workgroup's life is
// ~5050 cycles per 8-channel chunk (64 MFMAs per SIMD = 4096 of them) + ~16 k cycles of prologue / output transform / epilogue
// (scripts/trace_wino.py); the direct family delivers ~115 TFLOP/s = 50 k FLOP per cycle at the same 2.3 GHz on layers that fill the chip
// (profiles/r05_wino_layers.txt: the model's choice agrees with the measured faster kernel on all 14 Market layer shapes).
// Split plan: with fewer workgroups than CUs (or a fractional last round) the reduction over input channels is cut into `nsplit` ranges,
// each workgroup leaves a pre-epilogue partial output and wino_reduce_kernel sums them in split order and runs the epilogue
// (deterministic).  Cost in cycles at 2.3 GHz: rounds x (chunks per split x 4800 + 16 k) + the partials' write + read at ~4 TB/s.
struct FPlan { int nsplit, cps; double cycles; };
static FPlan fwd_plan(const DpigConvDesc* d, int cin, int kout) {
    const long T = (long)d->N * (d->H / 2) * (d->W / 2);
    const long wgs1 = (long)cdiv(T, TB) * (kout / KB);
    const int nch = cin / CH;
    FPlan best = {1, nch, 0.0};
    static const int force = getenv("DPIG_WINO_SPLIT") ? atoi(getenv("DPIG_WINO_SPLIT")) : 0;       // (A/B switch: 1 = never split)
    for (int s = 1; s <= 16; ++s) {
        const int cps = cdiv(nch, s);
        if (s > 1 && (cps < 6 || cdiv(nch, cps) != s || force == 1)) continue;
        const long rounds = (wgs1 * s + kNumCU - 1) / kNumCU;
        double cyc = (double)rounds * ((double)cps * 4800.0 + 16000.0);
        if (s > 1) cyc += 8000.0 + 2.0 * s * (double)d->N * d->H * d->W * kout * 4.0 / 1740.0;     // 4 TB/s = 1740 B per cycle
        if (best.cycles == 0.0 || cyc < best.cycles * 0.97) best = {s, cps, cyc};
    }
    return best;
}
static double direct_cycles_of(const DpigConvDesc* d, int cin, int kout) {
    // the direct family: ~115 TFLOP/s = 50 k FLOP per cycle on layers with >= 512 of its 128 x 128 tiles, down to ~65 % of that on the
    // smallest maps (8 x 4 C640: 75 TFLOP/s measured; its split-K plans pay partial-sum passes too)
    const long dtiles = (long)cdiv((long)d->N * d->H * d->W, 128) * cdiv(kout, 128);
    const double drate = 50000.0 * (dtiles >= 512 ? 1.0 : 0.65 + 0.35 * (double)dtiles / 512.0);
    return 2.0 * d->N * d->H * d->W * 9.0 * cin * kout / drate;
}
static bool pays(const DpigConvDesc* d, int cin, int kout) {
    init_mode();
    if (g_mode == 0) return false;
    if (g_mode == 2) return true;
    return fwd_plan(d, cin, kout).cycles < 0.95 * direct_cycles_of(d, cin, kout);
}
// what the layer costs WITHOUT the F(4x4, 3x3) form, in this file's cost model: the F(2x2, 3x3) plan where this family would take it
// (mode and shape permitting), the direct family otherwise (dpig_conv_wino4.hip's cost model compares against it)
double alt_cycles(const DpigConvDesc* d, int cin, int kout, int ld_in, int ld_out) {
    init_mode();
    const double direct = direct_cycles_of(d, cin, kout);
    if (g_mode == 0 || !shape_ok(d, cin, kout, ld_in, ld_out)) return direct;
    const double f2 = fwd_plan(d, cin, kout).cycles;
    if (g_mode == 2) return f2;
    return f2 < 0.95 * direct ? f2 : direct;
}

static int launch(const DpigConvDesc* d, const float* in, const float* U, const float* bias, const float* res, const float* mask,
                  float* out, float* out2, int cin, int kout, int ld_in, int ld_out, int act, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!aligned16(in) || !aligned16(U) || !aligned16(out) || (bias && !aligned16(bias)) || (res && (!aligned16(res) || (d->ldres & 3))) ||
        (mask && (!aligned16(mask) || (d->ldmask & 3))) || (out2 && (!aligned16(out2) || (d->ldy2 & 3))))
        return fail(DPIG_EINVAL, "winograd conv: operands must be 16-byte addressable");
    WParams p = {};
    p.X = in; p.U = U; p.D = out; p.D2 = out2; p.bias = bias; p.res = res; p.mask = mask;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = cin; p.Kout = kout;
    p.ldx = ld_in; p.ldd = ld_out; p.ldres = d->ldres; p.ldmask = d->ldmask; p.ldd2 = d->ldy2;
    p.TW = d->W / 2; p.THW = (d->H / 2) * p.TW; p.T = d->N * p.THW;
    p.nch = cin / CH;
    p.mtiles = cdiv(p.T, TB); p.ntiles = kout / KB;
    p.act = act; p.alpha = d->alpha; p.res_post = d->res_after_act;
    p.x_bytes = (unsigned)((long)d->N * d->H * d->W * ld_in * 4);
    p.u_bytes = (unsigned)((long)16 * cin * kout * 4);
    find_divisor(p.THW, &p.mul_thw, &p.shr_thw);
    find_divisor(p.TW, &p.mul_tw, &p.shr_tw);
    find_divisor(d->H / 2, &p.mul_th, &p.shr_th);
    p.trace = g_trace;
    const FPlan pl = fwd_plan(d, cin, kout);
    p.nsplit = pl.nsplit; p.cps = pl.cps;
    {   // workgroup order by HBM bytes: every XCD has its own L2, so whichever operand the XCDs do NOT partition is fetched by each
        // of the XCDs that work on the launch (DPIG_WINO_XMAJOR=0 / 1 pins the order: measurements)
        const double xb = (double)d->N * d->H * d->W * cin * 4.0, ub = 16.0 * cin * kout * 4.0;
        const double filter_major = xb * (p.ntiles < kNumXCD ? p.ntiles : kNumXCD) + ub;
        const double act_major = xb + ub * (p.mtiles < kNumXCD ? p.mtiles : kNumXCD);
        static const char* pin = getenv("DPIG_WINO_XMAJOR");
        p.xmajor = pin ? (atoi(pin) != 0) : (act_major < filter_major);
    }
    if (p.nsplit > 1) {
        const size_t need = (size_t)p.nsplit * d->N * d->H * d->W * kout * sizeof(float);
        if (!ws || ws_bytes < need || !aligned16(ws)) return fail(DPIG_ENOMEM, "winograd conv workspace too small: have %zu, need %zu", ws_bytes, need);
        p.partial = static_cast<float*>(ws);
    }
    const dim3 grid(p.mtiles * p.ntiles * p.nsplit);
    // blocks of 4 x 16 tiles on the stack of all images' tile rows: the raw-gather form (DPIG_WINO_BLOCK=0: A/B switch)
    static const bool block_on = !(getenv("DPIG_WINO_BLOCK") && atoi(getenv("DPIG_WINO_BLOCK")) == 0);
    int rc;
    if (block_on && (p.TW & 3) == 0 && ((d->N * (d->H / 2)) & 15) == 0) {
        static const char* pers = getenv("DPIG_WINO_PERSIST");      // 0: one workgroup per item (measurements)
        const unsigned items = grid.x;
        const dim3 pgrid((pers && atoi(pers) == 0) || items <= (unsigned)kNumCU ? items : (unsigned)kNumCU);
        static const bool slice_first = getenv("DPIG_WINO_ORDER") && atoi(getenv("DPIG_WINO_ORDER")) == 1;      // (A/B switch)
        if (slice_first) hipLaunchKernelGGL(wino_block_kernel<true>, pgrid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL(wino_block_kernel<false>, pgrid, dim3(512), 0, st, p);
        rc = check_launch("wino_block_kernel");
    } else {
        hipLaunchKernelGGL(wino_kernel, grid, dim3(512), 0, st, p);
        rc = check_launch("wino_kernel");
    }
    if (rc || p.nsplit == 1) return rc;
    return launch_reduce(p, st);
}

// ---- filter-gradient plan: splits over the tile axis so that (C / 64)(K / 64) S workgroups fill whole rounds of the chip -----------
struct WGPlan { int nsplit, cps, nchunks; };
static WGPlan wgrad_plan(const DpigConvDesc* d) {
    WGPlan pl;
    const long T = (long)d->N * (d->H / 2) * (d->W / 2);
    pl.nchunks = cdiv(T, 8);
    const int blocks = (d->C / 64) * (d->K / 64);
    int best = 1;
    double best_score = -1.0;
    for (int s = 1; s <= 256; ++s) {
        if (pl.nchunks / s < 16 && s > 1) break;                   // at least 16 chunks (128 tiles) per workgroup
        const long wgs = (long)blocks * s;
        const long rounds = (wgs + kNumCU - 1) / kNumCU;
        const double cps = (double)cdiv(pl.nchunks, s);
        const double cost = rounds * (cps * 5800.0 + 30000.0) + 2.0 * s * 600.0;      // + the partial slabs' round trip
        const double score = 1.0 / cost;
        if (score > best_score) { best_score = score; best = s; }
    }
    pl.nsplit = best;
    pl.cps = cdiv(pl.nchunks, best);
    pl.nsplit = cdiv(pl.nchunks, pl.cps);
    return pl;
}
static bool wgrad_shape_ok(const DpigConvDesc* d) {
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0) return false;   // (an empty batch made wgrad_plan divide by zero: tests/test_host_sweep.py)
    if (d->R != 3 || d->S != 3 || d->stride != 1 || d->upsample2x) return false;
    if ((d->H & 1) || (d->W & 1) || d->H < 2 || d->W < 2 || d->C % 64 || d->K % 64 || (d->ldx & 3) || (d->ldy & 3)) return false;
    if (d->pad_t >= 0 && d->pad_t != 1) return false;
    if (d->pad_l >= 0 && d->pad_l != 1) return false;
    const long lim = 0x7f000000L;
    // wino_wgrad_kernel forms its byte offsets with 24-bit multiplies (__mul24): a row pitch or a row count of 2^23 or more would wrap silently
    const long m24 = 1L << 23;
    if ((long)d->W * d->ldx * 4 >= m24 || (long)d->W * d->ldy * 4 >= m24 || (long)d->N * d->H >= m24) return false;
    return (long)d->N * d->H * d->W * d->ldx * 4 < lim && (long)d->N * d->H * d->W * d->ldy * 4 < lim;
}
static bool wgrad_pays(const DpigConvDesc* d) {
    init_mode();
    if (g_mode == 0) return false;
    if (g_mode == 2) return true;
    const WGPlan pl = wgrad_plan(d);
    const long wgs = (long)(d->C / 64) * (d->K / 64) * pl.nsplit;
    const long rounds = (wgs + kNumCU - 1) / kNumCU;
    // ~5800 cycles per 8-tile chunk (4-byte fragment reads) + ~30 k fixed, + the partial slabs' write / read and the reduction launch
    const double wino_cycles = rounds * ((double)pl.cps * 5800.0 + 30000.0) + 2.0 * pl.nsplit * 600.0 + 15000.0;
    // (the direct filter gradient reaches ~115 TFLOP/s = 50 k FLOP per cycle on layers with long pixel reductions, ~0.7 of that below
    // ~1000 pixels: 8 x 4 C768 runs 75 against this kernel's 88, 8 x 4 C640 88 against 64 -- profiles/r05_wino_layers.txt)
    const double rate = 50000.0 * ((long)d->N * d->H * d->W < 1024 ? 0.7 : 1.0);
    const double direct_cycles = 2.0 * d->N * d->H * d->W * 9.0 * d->C * d->K / rate;
    return wino_cycles < 0.95 * direct_cycles;
}

}  // namespace wino
}  // namespace dpig

using namespace dpig;

// 1: dpig_conv2d_wgrad_wino accepts the descriptor and is expected to beat dpig_conv2d_wgrad (mode switch as dpig_conv2d_wino_eligible).
extern "C" int dpig_conv2d_wgrad_wino_eligible(const DpigConvDesc* d) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->compute != DPIG_COMPUTE_F32) return 0;
    if (!wino::wgrad_shape_ok(d)) return 0;
    return wino::wgrad_pays(d) ? 1 : 0;
}
extern "C" size_t dpig_conv2d_wgrad_wino_workspace_bytes(const DpigConvDesc* d) {
    if (!d || !wino::wgrad_shape_ok(d)) return 0;
    const wino::WGPlan pl = wino::wgrad_plan(d);
    return ((size_t)pl.nsplit * 9 * d->C * d->K + (size_t)pl.nsplit * d->K) * sizeof(float);
}
// dw[3][3][C][K] = beta dw + conv_backward_filter(x, dy) by F(3x3, 2x2) minimal filtering (fp32 tensors, products and sums; deterministic);
// with db ([K], may be null) the same two launches also leave the bias gradient db = beta_b db + sum over pixels of dy, as
// dpig_conv2d_wgrad does.
extern "C" int dpig_conv2d_wgrad_wino(const DpigConvDesc* d, const float* x, const float* dy, float* dw, float beta, float* db, float beta_b,
                                      void* ws, size_t ws_bytes, void* stream) {
    int pt, pl_, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl_, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !dy || !dw) return fail(DPIG_EINVAL, "null tensor pointer");
    if (!wino::wgrad_shape_ok(d)) return fail(DPIG_EINVAL, "winograd wgrad: unsupported shape");
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dw) || !aligned16(ws)) return fail(DPIG_EINVAL, "winograd wgrad: operands must be 16-byte aligned");
    const wino::WGPlan pl = wino::wgrad_plan(d);
    const size_t slabs = (size_t)pl.nsplit * 9 * d->C * d->K;
    const size_t need = (slabs + (size_t)pl.nsplit * d->K) * sizeof(float);
    if (!ws || ws_bytes < need) return fail(DPIG_ENOMEM, "winograd wgrad workspace too small: have %zu, need %zu", ws_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    wino::WGParams p = {};
    p.X = x; p.DY = dy; p.part = static_cast<float*>(ws);
    p.bias_part = db ? p.part + slabs : nullptr;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.K = d->K; p.ldx = d->ldx; p.ldy = d->ldy;
    p.TW = d->W / 2; p.THW = (d->H / 2) * p.TW; p.T = d->N * p.THW;
    p.nchunks = pl.nchunks; p.cps = pl.cps; p.nsplit = pl.nsplit;
    p.cblocks = d->C / 64; p.kblocks = d->K / 64;
    p.x_bytes = (unsigned)((long)d->N * d->H * d->W * d->ldx * 4);
    p.y_bytes = (unsigned)((long)d->N * d->H * d->W * d->ldy * 4);
    find_divisor(p.THW, &p.mul_thw, &p.shr_thw);
    find_divisor(p.TW, &p.mul_tw, &p.shr_tw);
    static const bool slice_first = getenv("DPIG_WINO_WG_ORDER") && atoi(getenv("DPIG_WINO_WG_ORDER")) == 1;      // (A/B switch)
    static const bool no_xch = getenv("DPIG_WINO_WG_XCH") && atoi(getenv("DPIG_WINO_WG_XCH")) == 0;       // (A/B switch: the ds_bpermute form)
#ifdef DPIG_WINO4_KNOCKOUT
    static const int wg_ko = getenv("DPIG_WINO_WG_KO") ? atoi(getenv("DPIG_WINO_WG_KO")) : 0;
    if (wg_ko == 1) hipLaunchKernelGGL((wino::wino_wgrad_kernel<false, 1>), dim3(p.cblocks * p.kblocks * p.nsplit), dim3(512), 0, st, p);
    else if (wg_ko == 2) hipLaunchKernelGGL((wino::wino_wgrad_kernel<false, 2>), dim3(p.cblocks * p.kblocks * p.nsplit), dim3(512), 0, st, p);
    else if (wg_ko == 3) hipLaunchKernelGGL((wino::wino_wgrad_kernel<false, 3>), dim3(p.cblocks * p.kblocks * p.nsplit), dim3(512), 0, st, p);
    else if (wg_ko == 7) hipLaunchKernelGGL((wino::wino_wgrad_kernel<false, 7>), dim3(p.cblocks * p.kblocks * p.nsplit), dim3(512), 0, st, p);
    else if (wg_ko == 8) hipLaunchKernelGGL((wino::wino_wgrad_kernel<false, 8>), dim3(p.cblocks * p.kblocks * p.nsplit), dim3(512), 0, st, p);
    else if (wg_ko == 15) hipLaunchKernelGGL((wino::wino_wgrad_kernel<false, 15>), dim3(p.cblocks * p.kblocks * p.nsplit), dim3(512), 0, st, p);
    else
#endif
    if (no_xch) hipLaunchKernelGGL((wino::wino_wgrad_kernel<false, 0, false>), dim3(p.cblocks * p.kblocks * p.nsplit), dim3(512), 0, st, p);
    else if (slice_first) hipLaunchKernelGGL(wino::wino_wgrad_kernel<true>, dim3(p.cblocks * p.kblocks * p.nsplit), dim3(512), 0, st, p);
    else hipLaunchKernelGGL(wino::wino_wgrad_kernel<false>, dim3(p.cblocks * p.kblocks * p.nsplit), dim3(512), 0, st, p);
    rc = check_launch("wino_wgrad_kernel");
    if (rc) return rc;
    const long n4 = (long)9 * d->C * d->K / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(wino::wino_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.part, dw, n4, pl.nsplit, beta,
                       p.bias_part, db, d->K, beta_b);
    return check_launch("wino_wgrad_reduce_kernel");
}

// Elements (floats) of one transformed filter image of a [3][3][C][K] filter; 0 when the shape has no Winograd form.
extern "C" size_t dpig_wino_filter_elems(int C, int K) {
    if (C <= 0 || K <= 0 || C % 64 || K % 64) return 0;
    return (size_t)16 * C * K;
}

// u_fwd / u_dgrad (either may be null): transformed images for dpig_conv2d_fwd_wino / dpig_conv2d_dgrad_wino of the HWIO filter w.
extern "C" int dpig_wino_filter_transform(const float* w, int C, int K, float* u_fwd, float* u_dgrad, void* stream) {
    if (!w || !dpig_wino_filter_elems(C, K)) return fail(DPIG_EINVAL, "winograd filter transform: C and K must be positive multiples of 64");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int blocks = (C / 64) * (K / wino::CH);                 // = (K / 64) * (C / 8): the same count in both directions
    if (u_fwd) hipLaunchKernelGGL(wino::wino_filter_kernel<false>, dim3(blocks), dim3(256), 0, st, w, u_fwd, C, K);
    if (u_dgrad) hipLaunchKernelGGL(wino::wino_filter_kernel<true>, dim3(blocks), dim3(256), 0, st, w, u_dgrad, C, K);
    return check_launch("wino_filter_kernel");
}

// The same for a whole parameter set in one launch.  dpig_wino_filter_jobs_plan (host) fills first_block of every job and returns the
// block total (0: a job without a Winograd form, or a null filter); the planned array, copied to device memory, feeds
// dpig_wino_filter_transform_jobs.
extern "C" int dpig_wino_filter_jobs_plan(DpigWinoFilterJob* jobs, int njobs) {
    if (!jobs || njobs <= 0) return 0;
    long total = 0;
    for (int i = 0; i < njobs; ++i) {
        if (!jobs[i].w || !dpig_wino_filter_elems(jobs[i].C, jobs[i].K)) return 0;
        jobs[i].first_block = (int)total;
        jobs[i].reserved = 0;
        total += (long)(jobs[i].C / 64) * (jobs[i].K / wino::CH);
        if (total > 0x3fffffffL) return 0;
    }
    return (int)total;
}
extern "C" int dpig_wino_filter_transform_jobs(const DpigWinoFilterJob* jobs_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_blocks <= 0) return fail(DPIG_EINVAL, "winograd filter jobs: empty or unplanned job list");
    hipLaunchKernelGGL(wino::wino_filter_jobs_kernel, dim3(2 * (unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       jobs_dev, njobs, total_blocks);
    return check_launch("wino_filter_jobs_kernel");
}

// 1: dpig_conv2d_fwd_wino (which = 0) / dpig_conv2d_dgrad_wino (which = 1) accepts this descriptor AND is expected to beat the
// direct kernel (DPIG_WINO=2 / dpig_conv_wino_set_mode(2): wherever legal); 0 otherwise.  No device work.
extern "C" int dpig_conv2d_wino_eligible(const DpigConvDesc* d, int which) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->compute != DPIG_COMPUTE_F32) return 0;
    const bool dg = which == 1;
    const int cin = dg ? d->K : d->C, kout = dg ? d->C : d->K;
    const int ld_in = dg ? d->ldy : d->ldx, ld_out = dg ? d->ldx : d->ldy;
    if (d->C % 64 || d->K % 64) return 0;                       // (one transformed image serves both directions)
    if (!wino::shape_ok(d, cin, kout, ld_in, ld_out)) return 0;
    return wino::pays(d, cin, kout) ? 1 : 0;
}

// dev aid (scripts/trace_wino.py; not in include/dpig_hip.h): every subsequent Winograd launch writes 8 s_memtime stamps per workgroup
// (start, prologue done, k-loop done, output transform staged, end) to `buf` (device memory, >= 64 bytes x workgroups); null turns it off.
extern "C" int dpig_debug_wino_trace(unsigned long long* buf) {
    wino::g_trace = buf;
    return DPIG_OK;
}

extern "C" int dpig_conv_wino_set_mode(int mode) {
    if (mode < 0 || mode > 2) return fail(DPIG_EINVAL, "winograd mode out of range");
    wino::g_mode = mode;
    return DPIG_OK;
}

extern "C" int dpig_conv_wino_get_mode(void) {
    wino::init_mode();
    return wino::g_mode;
}

// y = act(conv3x3_SAME(x, w) + bias + residual)  (or act(..) + residual with res_after_act, y_act receiving the activation) through
// the transformed filter image u_fwd of dpig_wino_filter_transform.  Same descriptor and epilogue semantics as dpig_conv2d_fwd.
// Workspace of dpig_conv2d_fwd_wino (which = 0) / dpig_conv2d_dgrad_wino (which = 1): the partial outputs of a split plan, 0 for most layers.
extern "C" size_t dpig_conv2d_wino_workspace_bytes(const DpigConvDesc* d, int which) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0) return 0;
    const bool dg = which == 1;
    const int cin = dg ? d->K : d->C, kout = dg ? d->C : d->K;
    if (d->C % 64 || d->K % 64 || !wino::shape_ok(d, cin, kout, dg ? d->ldy : d->ldx, dg ? d->ldx : d->ldy)) return 0;
    const wino::FPlan pl = wino::fwd_plan(d, cin, kout);
    return pl.nsplit > 1 ? (size_t)pl.nsplit * d->N * d->H * d->W * kout * sizeof(float) : 0;
}

extern "C" int dpig_conv2d_fwd_wino(const DpigConvDesc* d, const float* x, const float* u_fwd, const float* bias, const float* residual,
                                    float* y, float* y_act, void* ws, size_t ws_bytes, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !u_fwd || !y) return fail(DPIG_EINVAL, "null tensor pointer");
    if (!wino::shape_ok(d, d->C, d->K, d->ldx, d->ldy) || d->C % 64) return fail(DPIG_EINVAL, "winograd conv: unsupported shape");
    if (residual && d->ldres < d->K) return fail(DPIG_EINVAL, "ldres < K");
    if (y_act && d->ldy2 < d->K) return fail(DPIG_EINVAL, "ldy2 < K");
    return wino::launch(d, x, u_fwd, bias, residual, nullptr, y, y_act, d->C, d->K, d->ldx, d->ldy, d->act, ws, ws_bytes,
                        static_cast<hipStream_t>(stream));
}

// dx = (conv_backward_data(dy, w) + accum) * act'(mask) through u_dgrad.  Same semantics as dpig_conv2d_dgrad.
extern "C" int dpig_conv2d_dgrad_wino(const DpigConvDesc* d, const float* dy, const float* u_dgrad, const float* accum, const float* mask,
                                      float* dx, void* ws, size_t ws_bytes, void* stream) {
    int pt, pl, Ho, Wo;
    int rc = resolve_desc(d, &pt, &pl, &Ho, &Wo);
    if (rc) return rc;
    if (!dy || !u_dgrad || !dx) return fail(DPIG_EINVAL, "null tensor pointer");
    if (!wino::shape_ok(d, d->K, d->C, d->ldy, d->ldx) || d->K % 64) return fail(DPIG_EINVAL, "winograd conv: unsupported shape");
    if (accum && d->ldres < d->C) return fail(DPIG_EINVAL, "ldres < C");
    if (mask && d->ldmask < d->C) return fail(DPIG_EINVAL, "ldmask < C");
    DpigConvDesc e = *d;
    e.res_after_act = 0; e.ldy2 = 0;
    return wino::launch(&e, dy, u_dgrad, nullptr, accum, mask, dx, nullptr, d->K, d->C, d->ldy, d->ldx, mask ? d->act : DPIG_ACT_NONE,
                        ws, ws_bytes, static_cast<hipStream_t>(stream));
}
