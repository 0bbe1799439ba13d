"""Host-side builders of the model's geometric inputs from OpenPose keypoints -- the numpy-only part of the reference's
offline converter (datasets/convert_market.py), restated so that inputs for the hot path (pose mask `mask_r6`, the 7 body-part
boxes `part_bbox` and their visibility `part_vis`) can be produced from keypoints without the converter:

    valid_peaks      datasets/convert_market.py:339-376, utils.py:459-490, datasets/convert_DF.py:302-338  `_get_valid_peaks` (three variants)
    part_bbox7       datasets/convert_market.py:578-638  `get_part_bbox7`    (7 region proposals + visibility)
    part_bbox37      datasets/convert_DF.py:522-655      `get_part_bbox`     (DeepFashion: 37 proposals, whole-body rule, lifted nose)
    pose_mask_raster datasets/convert_market.py:229-276  `_getPoseMask`      (radius-4 discs along the 23 limbs)
    pose_mask        ... + :277-283 the 5x5 morphological closing (skimage `dilation` then `erosion`)

Plain numpy on the host: input preparation, not the hot path (no device work, nothing here is timed).  Pinned against outputs
of the reference's own functions (tests/golden/prep_reference.npz, made by tests/golden/make_prep_golden.py which executes the
reference's function text) -- except the closing: skimage is not in this image, so `close5` follows skimage's documented
behaviour (neighbours outside the image are ignored) and that step is unpinned.

`peaks`: the converter's structure -- a list of 18 entries, each `[]` (keypoint missing) or `[(x, y, score, id)]`.
"""
import numpy as np

# limbs of _getPoseMask (1-based MSCOCO keypoint ids, convert_market.py:243-245)
LIMBS = ((2, 3), (2, 6), (3, 4), (4, 5), (6, 7), (7, 8), (2, 9), (9, 10), (10, 11), (2, 12), (12, 13), (13, 14), (2, 1), (1, 15),
         (15, 17), (1, 16), (16, 18), (2, 17), (2, 18), (9, 12), (12, 6), (9, 3), (17, 18))
# keypoint groups of get_part_bbox7 (0-based, convert_market.py:588-594): head/shoulders, torso, legs, L arm, R arm, L leg, R leg
PARTS7 = ((0, 1, 2, 5, 14, 15, 16, 17), (2, 3, 4, 5, 6, 7, 8, 11), (8, 9, 10, 11, 12, 13), (5, 6, 7), (2, 3, 4), (11, 12, 13), (8, 9, 10))


def peaks_from_array(kp):
    """[18, 3] (x, y, present) -> the converter's list-of-lists structure."""
    return [[(float(x), float(y), 1.0, i)] if p else [] for i, (x, y, p) in enumerate(np.asarray(kp, dtype=np.float64))]


def valid_peaks(all_peaks, subsets, variant="market"):
    """`_get_valid_peaks`: keep, per keypoint, the candidate that belongs to the person (row of `subsets`) with the highest total score
    (`subset[-2]`; only scores above -1 count, the first maximum wins); of several matching candidates the LAST one.  The reference
    holds three variants of this function that differ in what comes back:
        "market"  datasets/convert_market.py:339-376  the selection; with no person: ALL candidates, untouched (":367 Avoid to return None")
        "utils"   utils.py:459-490                    the selection; with no person: None
        "df"      datasets/convert_DF.py:302-338      computes the selection and returns ALL candidates, untouched (:333); no person: None
    and None whenever anything raises (a bare except), e.g. a 1-D `subsets` of scalars."""
    if variant not in ("market", "utils", "df"):
        raise ValueError("variant must be 'market', 'utils' or 'df'")
    subsets = np.asarray(subsets)
    if subsets.size == 0:                                  # OpenPose found nobody (an empty array of any rank): `subsets.tolist()` is [],
        return all_peaks if variant == "market" else None  # no loop iteration runs -- "market" hands back all candidates (:367)
    if subsets.ndim != 2:
        return None                                        # (`subset[-2]` of a scalar raises in the reference)
    scores = subsets[:, -2].tolist() if subsets.shape[0] else []
    best = -1
    best_score = -1
    for i, sc in enumerate(scores):
        if sc > best_score:
            best, best_score = i, sc
    if best < 0:
        return all_peaks if variant == "market" else None
    if variant == "df":
        return all_peaks
    ids = subsets[best, :18].tolist()
    out = []
    for cands in all_peaks:
        keep = None
        for p in cands:
            if p[-1] in ids:
                keep = p
        out.append([keep] if keep is not None and len(keep) > 0 else [])
    return out


def part_bbox7(peaks, radius=7, img_H=128, img_W=64, radius_single=10):
    """convert_market.py:578-638: per part the tight box of its visible keypoints grown by `radius` (10 px when a single keypoint
    is visible), clipped to [0, H-1] x [0, W-1]; an invisible part gets the sentinel [0, 0, 1, 1] and visibility 0.
    Returns (bbox [7, 4] as (y1, x1, y2, x2), vis [7]) -- float64 like the keypoints they come from (the converter casts to int64
    when it writes the record, :570-571)."""
    bbox = np.zeros((len(PARTS7), 4), dtype=np.float64)
    vis = np.zeros(len(PARTS7), dtype=np.int64)
    for k, part in enumerate(PARTS7):
        pts = [peaks[i][0] for i in part if len(peaks[i])]
        if not pts:
            bbox[k] = (0, 0, 1, 1)
            continue
        xs = np.array([p[0] for p in pts], dtype=np.float64)
        ys = np.array([p[1] for p in pts], dtype=np.float64)
        r = radius if len(pts) > 1 else radius_single
        bbox[k] = (max(0, ys.min() - r), max(0, xs.min() - r), min(img_H - 1, ys.max() + r), min(img_W - 1, xs.max() + r))
        vis[k] = 1
    return bbox, vis


# DeepFashion converter (datasets/convert_DF.py:522-570): 37 keypoint groups -- the 7 of PARTS7 (arm groups as that file lists them),
# torso corners, 8 limb segments, the whole body, the 18 single keypoints, the right and the left side
PARTS37 = (PARTS7[:3] + ((5, 6, 7), (2, 3, 4), (11, 12, 13), (8, 9, 10)) +
           ((2, 5, 8, 11), (5, 6), (6, 7), (2, 3), (3, 4), (11, 12), (12, 13), (8, 9), (9, 10), tuple(range(18))) +
           tuple((i,) for i in range(18)) + ((2, 3, 4, 8, 9, 10), (5, 6, 7, 11, 12, 13)))


def part_bbox37(peaks, img_H=256, img_W=256):
    """datasets/convert_DF.py:522-655 `get_part_bbox`: the DeepFashion records' 37 region proposals (trainer_256.py:34-41 feeds the first 7
    to the appearance encoder).  A person counts as a whole body when both lower legs (groups 13 and 15: left / right knee-ankle) have a
    visible keypoint: margins (10, 20 for a single keypoint) then, (20, 40) for upper-body crops; the nose is lifted by 10 / 25 pixels
    (not above the top edge) before the boxes are taken, so head boxes include the hair.  Otherwise the rules of `part_bbox7`.
    Returns (bbox [37, 4] as (y1, x1, y2, x2), vis [37])."""
    vis = np.array([1 if any(len(peaks[i]) for i in part) else 0 for part in PARTS37], dtype=np.int64)
    whole = bool(vis[13] and vis[15])
    r, r_single, lift = (10, 20, 10) if whole else (20, 40, 25)
    bbox = np.zeros((len(PARTS37), 4), dtype=np.float64)
    for k, part in enumerate(PARTS37):
        xs, ys = [], []
        for i in part:
            if len(peaks[i]):
                x, y = peaks[i][0][0], peaks[i][0][1]
                if i == 0:
                    y = max(0, y - lift)
                xs.append(x)
                ys.append(y)
        if not xs:
            bbox[k] = (0, 0, 1, 1)
            continue
        m = r if len(xs) > 1 else r_single
        bbox[k] = (max(0, min(ys) - m), max(0, min(xs) - m), min(img_H - 1, max(ys) + m), min(img_W - 1, max(xs) + m))
    return bbox, vis


def _disc(radius):
    o = np.arange(-radius, radius + 1)
    ii, jj = np.meshgrid(o, o, indexing="ij")
    keep = np.sqrt((ii ** 2 + jj ** 2).astype(np.float64)) <= radius
    return ii[keep], jj[keep]


def pose_mask_raster(peaks, height, width, radius=4):
    """convert_market.py:229-276: for every limb with both ends visible, a radius-`radius` disc at each end and at the
    int(distance / radius) - 1 interior points of the segment (coordinates truncated to integers like `int()`), clipped to the
    image: the body mask before the morphological closing.  [height, width] float64 in {0, 1}."""
    di, dj = _disc(radius)
    centres = []
    for a, b in LIMBS:
        p0, p1 = peaks[a - 1], peaks[b - 1]
        if not len(p0) or not len(p1):
            continue
        r0, c0, r1, c1 = p0[0][1], p0[0][0], p1[0][1], p1[0][0]
        centres += [(r0, c0), (r1, c1)]
        n = int(np.sqrt((r0 - r1) ** 2 + (c0 - c1) ** 2) / radius)
        for i in range(1, n):                            # (empty unless n > 1)
            centres.append((r0 + (r1 - r0) * i / n, c0 + (c1 - c0) * i / n))
    dense = np.zeros((height, width), dtype=np.float64)
    for r, c in centres:
        rr, cc = int(r) + di, int(c) + dj
        ok = (rr >= 0) & (rr < height) & (cc >= 0) & (cc < width)
        dense[rr[ok], cc[ok]] = 1.0
    return dense


def close5(mask, size=5):
    """skimage.morphology dilation then erosion with square(5) (convert_market.py:281-282); neighbours outside the image are
    ignored by both (skimage's border rule).  UNPINNED: skimage is absent from this image."""
    def sweep(a, fn, fill):
        h = size // 2
        p = np.pad(a, h, mode="constant", constant_values=fill)
        out = a.copy()
        for i in range(size):
            for j in range(size):
                out = fn(out, p[i:i + a.shape[0], j:j + a.shape[1]])
        return out
    m = np.asarray(mask, dtype=np.float64)
    return sweep(sweep(m, np.maximum, -np.inf), np.minimum, np.inf)


def pose_mask(peaks, height, width, radius=4):
    """`_getPoseMask` in full: the rasterised limbs, closed with a 5x5 square."""
    return close5(pose_mask_raster(peaks, height, width, radius))


def model_inputs_from_keypoints(kp, img_H=128, img_W=64):
    """One person's geometric inputs as the records hold them: kp [18, 3] (x, y, present) ->
    dict(mask_r6 [H, W, 1] float32 in {0, 1}, part_bbox [7, 4] int64 (y1, x1, y2, x2), part_vis [7] int64)
    (the record fields `pose_mask_r6`, `part_bbox`, `part_vis` read at trainer.py:553-560).  The record key says r6 but the converter
    fills it with `_getPoseMask(..., radius=7)` (convert_market.py:480, 499, 555-556): radius 7 here too."""
    peaks = peaks_from_array(kp)
    bbox, vis = part_bbox7(peaks, img_H=img_H, img_W=img_W)
    return {"mask_r6": pose_mask(peaks, img_H, img_W, radius=7).astype(np.float32)[..., None], "part_bbox": bbox.astype(np.int64), "part_vis": vis}
