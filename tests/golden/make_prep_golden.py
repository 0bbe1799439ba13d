"""Generate tests/golden/prep_reference.npz -- body masks, part boxes and peak selection produced BY THE REFERENCE'S OWN CODE.

The converter that writes the TFRecords (datasets/convert_market.py) needs TensorFlow / skimage / python 2 as a module, but
five of its functions (and `get_part_bbox` of datasets/convert_DF.py:522-655, the DeepFashion records' 37 region proposals) are plain
python + numpy:

    _get_valid_peaks   :339-376   pick the best-scoring person's keypoints out of OpenPose's candidates
    get_part_bbox7     :578-638   the 7 body-part boxes + visibility the Fg encoder crops (models.py:405-415)
    _getPoseMask       :229-283   the body mask `mask_r6` (discs along 23 limbs, then a 5x5 closing)
    _getSparseKeypoint :286-305, _sparse2dense :330-337   (called by _getPoseMask)

Like make_pose_golden.py, this script parses the reference file at run time, compiles ONLY those function definitions from the
reference's text (nothing of it is written anywhere) and calls them under python 3.  Shims: `np.float` -> builtin float,
`xrange` -> range; the file's `from __future__ import division` is python 3's division.  skimage is absent, so the LAST two
statements before `_getPoseMask`'s return (dilation / erosion with square(5)) are cut from the parsed function: the fixture pins
the rasterisation; the closing stays unpinned (disentangled-person-image-generation_amd/dataprep.py::close5 says so).

Keypoint coordinates are integral floats (what OpenPose's peak lists hold), so int() truncation and the box arithmetic are exact.

    python tests/golden/make_prep_golden.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/datasets/convert_market.py"
WANTED = ("_get_valid_peaks", "get_part_bbox7", "_getPoseMask", "_getSparseKeypoint", "_sparse2dense")


class _NumpyWithFloatAlias(object):
    float = float

    def __getattr__(self, name):
        return getattr(np, name)


def reference_functions(REF=REF, WANTED=WANTED):
    tree = ast.parse(open(REF).read(), REF)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(d.name for d in defs) == sorted(WANTED), [d.name for d in defs]
    for d in defs:
        if d.name == "_getPoseMask":                # drop `dense = dilation(...)`, `dense = erosion(...)`; keep `return dense`
            cut = [s for s in d.body if not (isinstance(s, ast.Assign) and isinstance(s.value, ast.Call) and
                                            getattr(s.value.func, "id", None) in ("dilation", "erosion"))]
            assert len(d.body) - len(cut) == 2 and isinstance(cut[-1], ast.Return)
            d.body = cut
    ns = {"np": _NumpyWithFloatAlias(), "xrange": range}
    exec(compile(ast.Module(body=defs, type_ignores=[]), REF, "exec"), ns)
    return {n: ns[n] for n in WANTED}


def keypoint_cases():
    """[n, 18, 3] (x, y, present): random people on the 128x64 Market canvas + the corner cases of the box rules."""
    rng = np.random.RandomState(20250927)
    n = 24
    kp = np.zeros((n, 18, 3), dtype=np.float64)
    kp[:, :, 0] = rng.randint(0, 64, size=(n, 18))
    kp[:, :, 1] = rng.randint(0, 128, size=(n, 18))
    kp[:, :, 2] = rng.uniform(size=(n, 18)) < 0.8
    kp[0, :, 2] = 1                                   # everything visible
    kp[1, :, 2] = 0                                   # nothing visible: 7 sentinels, empty mask
    kp[2, :, 2] = 0; kp[2, 9, :] = (5, 120, 1)        # one keypoint only: single-keypoint parts (radius 10), clipped at the border
    kp[3, :, 2] = 1; kp[3, [8, 9, 10, 11, 12, 13], 2] = 0    # no legs: parts 3, 6, 7 invisible
    kp[4, :, 2] = 1; kp[4, :, 0] = 0; kp[4, :, 1] = 0        # all keypoints in the top-left corner
    kp[5, :, 2] = 1; kp[5, :, 0] = 63; kp[5, :, 1] = 127     # ... bottom-right corner
    kp[6, :, 2] = 1; kp[6, 1, :2] = (32, 10); kp[6, 8, :2] = (30, 100)   # a long limb (neck - hip): many interior discs
    return kp


REF_DF = "/root/reference/datasets/convert_DF.py"


def reference_df_part_bbox():
    """`get_part_bbox` of the DeepFashion converter (datasets/convert_DF.py:522-655): 37 region proposals; its image dumps sit behind
    `idx is not None` and are never reached."""
    tree = ast.parse(open(REF_DF).read(), REF_DF)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_part_bbox"]
    assert len(defs) == 1
    ns = {"np": _NumpyWithFloatAlias(), "xrange": range}
    exec(compile(ast.Module(body=defs, type_ignores=[]), REF_DF, "exec"), ns)
    return ns["get_part_bbox"]


def reference_valid_peaks(path):
    """The `_get_valid_peaks` of another file of the reference (utils.py:459-490, datasets/convert_DF.py:302-338: same selection,
    different return conventions)."""
    import sys
    tree = ast.parse(open(path).read(), path)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_get_valid_peaks"]
    assert len(defs) == 1
    ns = {"np": np, "sys": sys}
    exec(compile(ast.Module(body=defs, type_ignores=[]), path, "exec"), ns)
    return ns["_get_valid_peaks"]


def df_keypoint_cases():
    """[n, 18, 3] on the 256x256 DeepFashion canvas: whole bodies, upper-body crops (no lower legs), sparse detections, corner cases."""
    rng = np.random.RandomState(4242)
    n = 32
    kp = np.zeros((n, 18, 3), dtype=np.float64)
    kp[:, :, 0] = rng.randint(0, 256, size=(n, 18))
    kp[:, :, 1] = rng.randint(0, 256, size=(n, 18))
    kp[:, :, 2] = rng.uniform(size=(n, 18)) < 0.85
    kp[8:16, [9, 10, 12, 13], 2] = 0                      # upper-body crops: knees and ankles missing -> the wide margins
    kp[16:24, :, 2] = rng.uniform(size=(8, 18)) < 0.3     # sparse
    kp[0, :, 2] = 1
    kp[1, :, 2] = 0                                       # nothing visible: 37 sentinels
    kp[2, :, 2] = 1; kp[2, 0, :2] = (100, 4)              # nose near the top edge: the lift is clipped at 0
    kp[3, :, 2] = 0; kp[3, 0, :] = (128, 200, 1)          # the nose alone: single-keypoint margin 40, lifted by 25
    kp[4, :, 2] = 1; kp[4, :, 0] = 255; kp[4, :, 1] = 255
    return kp


def to_peaks(kp):
    return [[(float(x), float(y), 1.0, i)] if p else [] for i, (x, y, p) in enumerate(kp)]


def valid_peak_cases():
    """OpenPose output for images with several people: per keypoint a list of candidates (x, y, score, id) and a `subset`
    table [people, 20] (18 candidate ids, total score, count)."""
    rng = np.random.RandomState(77)
    cases = []
    for people in (1, 2, 3):
        all_peaks, nid = [], 0
        for k in range(18):
            c = []
            for _ in range(rng.randint(0, people + 1)):
                c.append((float(rng.randint(0, 64)), float(rng.randint(0, 128)), float(rng.uniform()), nid))
                nid += 1
            all_peaks.append(c)
        subsets = -np.ones((people, 20))
        for pi in range(people):
            for k in range(18):
                if all_peaks[k] and rng.uniform() < 0.8:
                    subsets[pi, k] = all_peaks[k][rng.randint(len(all_peaks[k]))][3]
            subsets[pi, 18] = rng.uniform(1, 30)
            subsets[pi, 19] = (subsets[pi, :18] >= 0).sum()
        cases.append((all_peaks, subsets))
    return cases


def encode_peaks(peaks):
    """list of 18 [] / [(x, y, score, id)] -> [18, 5] with a presence flag."""
    out = np.zeros((18, 5), dtype=np.float64)
    for k, p in enumerate(peaks):
        if len(p):
            out[k, :4] = p[0]
            out[k, 4] = 1
    return out


def main():
    f = reference_functions()
    kp = keypoint_cases()
    bbox, vis, masks, masks7 = [], [], [], []
    for person in kp:
        peaks = to_peaks(person)
        b, v = f["get_part_bbox7"](peaks)
        bbox.append(np.array(b, dtype=np.float64))
        vis.append(np.array(v, dtype=np.int64))
        m = f["_getPoseMask"](peaks, 128, 64, radius=4, mode="Solid")
        m = np.asarray(m, dtype=np.float64).reshape(128, 64)
        assert set(np.unique(m).tolist()) <= {0.0, 1.0}
        masks.append(m.astype(np.uint8))
        # radius 7: what the converter stores under the record key `pose_mask_r6_*` (convert_market.py:480, 499, 555-556), the mask the
        # Fg/Bg encoder consumes (trainer.py:581)
        m7 = np.asarray(f["_getPoseMask"](peaks, 128, 64, radius=7, mode="Solid"), dtype=np.float64).reshape(128, 64)
        assert set(np.unique(m7).tolist()) <= {0.0, 1.0}
        masks7.append(m7.astype(np.uint8))
    fix = {"keypoints": kp, "part_bbox": np.stack(bbox), "part_vis": np.stack(vis), "mask_raster_bits": np.packbits(np.stack(masks), axis=-1),
           "mask_raster_r7_bits": np.packbits(np.stack(masks7), axis=-1)}
    get_part_bbox = reference_df_part_bbox()
    dkp = df_keypoint_cases()
    dbox, dvis = [], []
    for person in dkp:
        b, v = get_part_bbox(to_peaks(person))
        dbox.append(np.array(b, dtype=np.float64))
        dvis.append(np.array(v, dtype=np.int64))
    # the DeepFashion converter's own copy of the mask builder (datasets/convert_DF.py:197-247), on its 256 x 256 canvas
    fdf = reference_functions(REF_DF, ("_getPoseMask", "_getSparseKeypoint", "_sparse2dense"))
    dmasks = []
    for person in dkp[:8]:
        m = np.asarray(fdf["_getPoseMask"](to_peaks(person), 256, 256, radius=4, mode="Solid"), dtype=np.float64).reshape(256, 256)
        assert set(np.unique(m).tolist()) <= {0.0, 1.0}
        dmasks.append(m.astype(np.uint8))
    fix["df_mask_raster_bits"] = np.packbits(np.stack(dmasks), axis=-1)
    fix["df_keypoints"] = dkp
    fix["df_part_bbox"] = np.stack(dbox)
    fix["df_part_vis"] = np.stack(dvis)
    variants = {"market": f["_get_valid_peaks"], "utils": reference_valid_peaks("/root/reference/utils.py"),
                "df": reference_valid_peaks(REF_DF)}
    cases = valid_peak_cases()
    empty_scores = (cases[1][0], cases[1][1].copy())
    empty_scores[1][:, 18] = -3.0                      # two people, neither with a score above -1: "no person"
    cases += [empty_scores, (cases[2][0], np.zeros((0, 20)))]
    for i, (all_peaks, subsets) in enumerate(cases):
        flat = np.array([list(p) + [k] for k, c in enumerate(all_peaks) for p in c], dtype=np.float64).reshape(-1, 5)
        fix["vp%d_candidates" % i] = flat              # (x, y, score, id, keypoint)
        fix["vp%d_subsets" % i] = subsets
        for v, fn in variants.items():                 # what each of the reference's three variants returns: 0 None, 1 a selection, 2 all_peaks itself
            got = fn(all_peaks, subsets)
            kind = 0 if got is None else (2 if got is all_peaks else 1)
            fix["vp%d_%s_kind" % (i, v)] = np.array(kind)
            if kind == 1:
                fix["vp%d_%s_selected" % (i, v)] = encode_peaks(got)
        if int(fix["vp%d_market_kind" % i]) == 1:
            fix["vp%d_selected" % i] = fix["vp%d_market_selected" % i]
    path = os.path.join(HERE, "prep_reference.npz")
    np.savez_compressed(path, **fix)
    print("wrote", path, {k: v.shape for k, v in fix.items()})


if __name__ == "__main__":
    main()
