// Host-side launch planning shared by the fp32 (dpig_conv.hip) and bf16-storage (dpig_conv_bf16.hip) conv families:
// descriptor validation, TF 'SAME' geometry, magic-number division, the split-K plan and the stride-2 dgrad parity
// classes.  Both families use 128 x 128 block tiles.
#pragma once
#include "dpig_common.h"

namespace dpig {

constexpr int kTileM = 128, kTileN = 128;
constexpr int MAX_TAPS = 25;

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// (mul, shr) such that n / d == umulhi(n, mul) >> shr for every 0 <= n < 2^31; d == 1 -> mul = 0
static inline void find_divisor(int d, unsigned* mul, unsigned* shr) {
    if (d == 1) { *mul = 0; *shr = 0; return; }
    unsigned lg = 0;
    while ((1u << lg) < (unsigned)d) ++lg;                 // ceil(log2 d)
    const unsigned p = 31 + lg;
    *mul = (unsigned)(((1ull << p) + (unsigned)d - 1) / (unsigned)d);
    *shr = p - 32;
}

static inline int resolve_desc(const DpigConvDesc* d, int* pt, int* pl, int* Ho, int* Wo) {
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

// Filter taps that touch at least one REAL input pixel for some output position: a contiguous window [a0, a0 + na) x [b0, b0 + nb) of
// the R x S filter.  Every tap outside it multiplies padding only -- on the 1 x 1 maps at the bottom of the DeepFashion ROI tower
// (models.py:420-431 at trainer_256.py:40-41: repeat_num 7) eight of a 3 x 3 filter's nine taps, on its 2 x 2 -> 1 x 1 stride-2 conv
// five -- so the kernels' tap loops run over the window only (the affine tap family of the parameter blocks addresses a sub-rectangle
// of the filter directly); results are identical.  Forward: tap row a is live iff 0 <= oy s - pt + a < H for some output row oy.
struct TapWindow { int a0, na, b0, nb; };
static inline void live_range(int in, int out, int k, int s, int pad, int* lo, int* n) {
    int first = -1, last = -1;
    for (int a = 0; a < k; ++a) {
        bool live = false;
        for (int o = 0; o < out && !live; ++o) live = (o * s - pad + a >= 0) && (o * s - pad + a < in);
        if (live) { if (first < 0) first = a; last = a; }
    }
    if (first < 0) { first = 0; last = 0; }
    *lo = first; *n = last - first + 1;
}
static inline TapWindow live_taps(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo) {
    TapWindow t;
    live_range(d->H, Ho, d->R, d->stride, pt, &t.a0, &t.na);
    live_range(d->W, Wo, d->S, d->stride, pl, &t.b0, &t.nb);
    return t;
}

// Split-K plan.  The grid is tiles x splits workgroups of which 2 per CU are resident (512 slots):
// pick the split count that best fills whole "rounds" of 512 slots, charged with the HBM round trip
// of the fp32 partial sums (~pen*s/K relative to the MFMA time of a K-deep reduction: pen = 120 for the fp32
// matrix pipe, several times that for the bf16 pipe; kdepth = reduction depth of one k-tile).
static inline int choose_split(int tiles, int ktiles, int forced, int kdepth = 32, double pen = 120.0) {
    if (forced > 0) return forced < ktiles ? forced : (ktiles > 0 ? ktiles : 1);
    if (ktiles <= 0) return 1;
    const int slots = 2 * kNumCU;
    const double kred = (double)kdepth * ktiles;
    double best = -1.0;
    int best_s = 1;
    // (a handful of tiles with a very deep reduction -- the fully connected layers, 20480 -> 128 on 16 rows -- is one k-tile chain per
    // workgroup: latency-bound, so it may be cut into one piece per CU; the partial sums of so few tiles are a few MB)
    const int cap = tiles <= 8 ? kNumCU : 64;
    const int smax = ktiles / 2 > 0 ? (ktiles / 2 < cap ? ktiles / 2 : cap) : 1;
    for (int s = 1; s <= smax; ++s) {
        const int tps = cdiv(ktiles, s);
        const int sr = cdiv(ktiles, tps);                  // effective split count
        if (sr != s) continue;
        const long blocks = (long)tiles * s;
        const long rounds = (blocks + slots - 1) / slots;
        double eff = (double)blocks / (double)(rounds * slots);
        if (rounds == 1 && blocks <= kNumCU) eff = 0.75 * (double)blocks / kNumCU;   // lone block per CU
        const double score = eff / (s > 1 ? 1.0 + pen * s / kred : 1.0);
        if (score > best * 1.02) { best = score; best_s = s; }
    }
    return best_s;
}

struct Plan { int nsplit, tiles_per_split; };
static inline Plan plan_split(int tiles, int ktiles, int forced, int kdepth = 32, double pen = 120.0) {
    Plan pl;
    pl.nsplit = choose_split(tiles, ktiles, forced, kdepth, pen);
    pl.tiles_per_split = cdiv(ktiles > 0 ? ktiles : 1, pl.nsplit);
    pl.nsplit = cdiv(ktiles > 0 ? ktiles : 1, pl.tiles_per_split);   // drop empty splits
    return pl;
}


// dgrad stride-2 parity class geometry
struct DClass { int py, px, Hr, Wr, ntaps, nky, nkx, ky0, kx0, oy0, ox0; };
static inline int build_dgrad_classes(const DpigConvDesc* d, int pt, int pl, DClass* cls) {
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
// split plan of the parity classes when they share one launch: every class is planned against the TOTAL tile
// count (that is what fills the machine); partial slabs are laid out back to back in the workspace
struct S2Plan { int nc; DClass cls[4]; Plan pl[4]; long M[4]; size_t off[4]; size_t total; };
static inline void plan_dgrad_s2(const DpigConvDesc* d, int pt, int pl, S2Plan* sp, int bk = 32, double pen = 120.0) {
    sp->nc = build_dgrad_classes(d, pt, pl, sp->cls);
    const int ntile_n = cdiv(d->C, d->C <= 32 ? 32 : kTileN);
    int total_tiles = 0;
    for (int i = 0; i < sp->nc; ++i) {
        sp->M[i] = (long)d->N * sp->cls[i].Hr * sp->cls[i].Wr;
        total_tiles += cdiv(sp->M[i], kTileM) * ntile_n;
    }
    sp->total = 0;
    for (int i = 0; i < sp->nc; ++i) {
        sp->pl[i] = plan_split(total_tiles, sp->cls[i].ntaps * cdiv(d->K, bk), d->split_k, bk, pen);
        sp->off[i] = sp->total;
        if (sp->pl[i].nsplit > 1) sp->total += (size_t)sp->pl[i].nsplit * sp->M[i] * d->C * sizeof(float);
    }
}

// ---- batches beyond one launch's address range ---------------------------------------------------------------------------
// A launch addresses every tensor through ONE buffer descriptor and 32-bit offsets, so it must see less than 2 GiB of each
// operand.  Larger batches are cut into runs of whole images: independent launches for forward / dgrad, accumulated in image
// order (beta = 1 after the first run: deterministic) for wgrad.  Returns the images per launch (>= N: one launch), or 0 when
// a single image is already too large.  `es` = bytes per activation element (4, or 2 for the bf16-storage entry points).
static inline int images_per_launch(const DpigConvDesc* d, long es) {
    if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->R <= 0 || d->S <= 0 || (d->stride != 1 && d->stride != 2)) return d ? d->N : 1;
    int ho, wo;
    dpig_same_pad(d->H, d->R, d->stride, &ho, nullptr);
    dpig_same_pad(d->W, d->S, d->stride, &wo, nullptr);
    const long xpix = (long)d->H * d->W, ypix = d->upsample2x ? 4 * xpix : (long)ho * wo;
    long ldx = d->ldx, ldy = d->ldy;            // aux tensors ride on either pixel set: residual / accum, mask, second output
    if (d->ldres > ldx) ldx = d->ldres;
    if (d->ldmask > ldx) ldx = d->ldmask;
    if (d->ldres > ldy) ldy = d->ldres;
    if (d->ldy2 > ldy) ldy = d->ldy2;
    if (ldx <= 0 || ldy <= 0) return d->N;
    const long per = (xpix * ldx > ypix * ldy ? xpix * ldx : ypix * ldy) * es;
    const long lim = 0x7f000000L;               // 2 GiB minus room for the halo offsets folded into the descriptors
    const long n = lim / per;
    return n <= 0 ? 0 : (n >= d->N ? d->N : (int)n);
}
// per-image element strides of the tensors of one conv (x-side pixels, y-side pixels)
static inline void image_pixels(const DpigConvDesc* d, long* xpix, long* ypix) {
    int ho, wo;
    dpig_same_pad(d->H, d->R, d->stride, &ho, nullptr);
    dpig_same_pad(d->W, d->S, d->stride, &wo, nullptr);
    *xpix = (long)d->H * d->W;
    *ypix = d->upsample2x ? 4 * *xpix : (long)ho * wo;
}

}  // namespace dpig
