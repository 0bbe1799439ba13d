"""Input builders (`dataprep`: body mask, part boxes, peak selection) against outputs of the REFERENCE'S OWN functions
(tests/golden/prep_reference.npz, produced by tests/golden/make_prep_golden.py executing datasets/convert_market.py:229-376,
578-638 and datasets/convert_DF.py:522-655 from the reference's text), and the synthetic generators against the conventions those outputs show."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIX = np.load(os.path.join(ROOT, "tests", "golden", "prep_reference.npz"))


def test_part_boxes_and_visibility_equal_the_reference_builder():
    from dpig_amd import dataprep
    kp = FIX["keypoints"]
    for i in range(kp.shape[0]):
        bbox, vis = dataprep.part_bbox7(dataprep.peaks_from_array(kp[i]))
        assert np.array_equal(bbox, FIX["part_bbox"][i]), i
        assert np.array_equal(vis, FIX["part_vis"][i]), i
    # the conventions the hot path relies on (models.py:405-415): boxes inside the image, sentinel <=> invisible
    b, v = FIX["part_bbox"], FIX["part_vis"]
    assert np.all(b[v == 0] == np.array([0, 0, 1, 1]))
    assert np.all(b[..., 0] >= 0) and np.all(b[..., 2] <= 127) and np.all(b[..., 1] >= 0) and np.all(b[..., 3] <= 63)
    assert np.all(b[v == 1][:, 2] >= b[v == 1][:, 0]) and np.all(b[v == 1][:, 3] >= b[v == 1][:, 1])


def test_deepfashion_part_boxes_equal_the_reference_builder():
    """datasets/convert_DF.py:522-655 `get_part_bbox` (37 proposals per person; trainer_256.py:34-41 uses the first 7): whole-body rule,
    lifted nose, single-keypoint margins -- `dataprep.part_bbox37` against the reference function's own outputs."""
    from dpig_amd import dataprep
    kp = FIX["df_keypoints"]
    whole = 0
    for i in range(kp.shape[0]):
        bbox, vis = dataprep.part_bbox37(dataprep.peaks_from_array(kp[i]))
        assert np.array_equal(bbox, FIX["df_part_bbox"][i]), i
        assert np.array_equal(vis, FIX["df_part_vis"][i]), i
        whole += int(vis[13] and vis[15])
    assert 0 < whole < kp.shape[0]                              # both margin regimes are in the fixture
    b, v = FIX["df_part_bbox"], FIX["df_part_vis"]
    assert b.shape[1:] == (37, 4) and np.all(b[v == 0] == np.array([0, 0, 1, 1]))
    assert np.all(b >= 0) and np.all(b[..., 2] <= 255) and np.all(b[..., 3] <= 255)


def test_body_mask_rasterisation_equals_the_reference_builder():
    from dpig_amd import dataprep
    kp = FIX["keypoints"]
    ref = np.unpackbits(FIX["mask_raster_bits"], axis=-1)[..., :64].astype(np.float64)
    assert ref.sum() > 0 and ref[1].sum() == 0           # (person 1 has no keypoints)
    for i in range(kp.shape[0]):
        got = dataprep.pose_mask_raster(dataprep.peaks_from_array(kp[i]), 128, 64, radius=4)
        assert np.array_equal(got, ref[i]), i
    # radius 7 = the record field `pose_mask_r6_*` (convert_market.py:480, 555-556), what model_inputs_from_keypoints builds
    ref7 = np.unpackbits(FIX["mask_raster_r7_bits"], axis=-1)[..., :64].astype(np.float64)
    assert ref7.sum() > ref.sum()
    for i in range(kp.shape[0]):
        assert np.array_equal(dataprep.pose_mask_raster(dataprep.peaks_from_array(kp[i]), 128, 64, radius=7), ref7[i]), i
    m = dataprep.model_inputs_from_keypoints(kp[0])["mask_r6"][..., 0]
    assert np.array_equal(m, dataprep.close5(ref7[0]).astype(np.float32))
    # the DeepFashion converter's copy of the builder (datasets/convert_DF.py:197-247) on its 256 x 256 canvas
    dkp = FIX["df_keypoints"]
    dref = np.unpackbits(FIX["df_mask_raster_bits"], axis=-1)[..., :256].astype(np.float64)
    assert dref.shape == (8, 256, 256) and dref[1].sum() == 0 and dref[0].sum() > 0
    for i in range(dref.shape[0]):
        assert np.array_equal(dataprep.pose_mask_raster(dataprep.peaks_from_array(dkp[i]), 256, 256, radius=4), dref[i]), i


def test_valid_peak_selection_equals_the_reference_builder():
    """`_get_valid_peaks` exists three times in the reference (datasets/convert_market.py:339-376, utils.py:459-490,
    datasets/convert_DF.py:302-338) with the same selection and different return conventions (no person: all candidates / None; the
    DeepFashion copy returns all candidates even after selecting): every variant against its own function's output, for 1 - 3 people,
    for people whose scores are all below -1 and for an empty person table."""
    from dpig_amd import dataprep
    n = 0
    while "vp%d_candidates" % n in FIX.files:
        n += 1
    assert n == 5
    kinds = set()
    for i in range(n):
        cand, subsets = FIX["vp%d_candidates" % i], FIX["vp%d_subsets" % i]
        all_peaks = [[tuple(r[:3]) + (int(r[3]),) for r in cand if int(r[4]) == k] for k in range(18)]
        for v in ("market", "utils", "df"):
            kind = int(FIX["vp%d_%s_kind" % (i, v)])
            kinds.add((v, kind))
            got = dataprep.valid_peaks(all_peaks, subsets, v)
            if kind == 0:
                assert got is None, (i, v)
            elif kind == 2:
                assert got is all_peaks, (i, v)
            else:
                enc = np.zeros((18, 5))
                for k, p in enumerate(got):
                    if len(p):
                        enc[k, :4], enc[k, 4] = p[0], 1
                assert np.array_equal(enc, FIX["vp%d_%s_selected" % (i, v)]), (i, v)
    assert kinds == {("market", 1), ("market", 2), ("utils", 1), ("utils", 0), ("df", 2), ("df", 0)}


def test_closing_is_scipy_grey_closing_with_ignored_borders():
    """The one step the fixture cannot pin (skimage absent): cross-checked against scipy.ndimage's dilation / erosion with the
    border treated as 'no neighbour' (pad with the identity of max / min)."""
    import scipy.ndimage as ndi
    from dpig_amd import dataprep
    ref = np.unpackbits(FIX["mask_raster_bits"], axis=-1)[..., :64].astype(np.float64)
    for i in (0, 2, 4, 5, 6):
        d = ndi.grey_dilation(ref[i], size=(5, 5), mode="constant", cval=0.0)
        e = ndi.grey_erosion(d, size=(5, 5), mode="constant", cval=1.0)
        got = dataprep.close5(ref[i])
        assert np.array_equal(got, e), i
        assert np.all(got >= ref[i])                     # closing is extensive


def test_synthetic_batches_follow_the_builder_conventions():
    from dpig_amd import synthetic
    for batch in (synthetic.make_batch(8, seed=3), synthetic.make_batch_from_keypoints(8, seed=3)):
        b, v = batch["part_bbox"], batch["part_vis"]
        assert b.shape == (8, 7, 4) and v.shape == (8, 7)
        assert np.all(b[v == 0] == np.array([0, 0, 1, 1]))
        assert np.all(b[..., 0] >= 0) and np.all(b[..., 2] <= 127) and np.all(b[..., 1] >= 0) and np.all(b[..., 3] <= 63)
        live = b[v == 1]
        assert np.all(live[:, 2] > live[:, 0]) and np.all(live[:, 3] > live[:, 1])
        assert set(np.unique(batch["mask_r6"]).tolist()) <= {0.0, 1.0} and set(np.unique(batch["pose"]).tolist()) <= {-1.0, 1.0}
    kb = synthetic.make_batch_from_keypoints(8, seed=3)
    assert (kb["part_vis"][3, [2, 5, 6]] == 0).all()     # the leg-less figure
    frac = kb["mask_r6"].mean()
    assert 0.1 < frac < 0.6
    # every visible keypoint lies inside the body mask and inside the box of each part it belongs to
    from dpig_amd import dataprep
    for bi in range(8):
        for k in range(18):
            x, y, p = kb["keypoints"][bi, k]
            if not p:
                continue
            for part, members in enumerate(dataprep.PARTS7):
                if k in members:
                    y1, x1, y2, x2 = kb["part_bbox"][bi, part]
                    assert y1 <= y <= y2 and x1 <= x <= x2


def test_no_person_found_is_an_empty_subsets_array_of_any_rank():
    """OpenPose returns an empty 1-D array when it finds nobody: the Market converter's `_get_valid_peaks` runs no loop iteration and
    hands back every candidate (convert_market.py:367 "Avoid to return None"); the utils / DeepFashion variants return None."""
    from dpig_amd import dataprep
    peaks = [[(3.0, 4.0, 0.9, 0)], []] + [[] for _ in range(16)]
    for empty in (np.zeros((0,)), np.zeros((0, 20)), []):
        assert dataprep.valid_peaks(peaks, empty, "market") is peaks
        assert dataprep.valid_peaks(peaks, empty, "utils") is None
        assert dataprep.valid_peaks(peaks, empty, "df") is None
    assert dataprep.valid_peaks(peaks, np.float64(3.0), "market") is None      # a scalar: `subset[-2]` raises in the reference
