"""Generate tests/golden/pose_reference.npz -- pose target maps produced BY THE REFERENCE'S OWN CODE.

Most of the reference needs TensorFlow 1.4 / python 2 and cannot run here, but its pose rasterisers are plain
python + numpy:

    utils.py  py_poseInflate        the function tester.py:399-400 leaves the graph for (radius-4 disc around every
                                    visible keypoint, from normalised or pixel (row, col, visibility) triplets)
    utils.py  _getSparseKeypoint / _getSparsePose / _sparse2dense
                                    the dataset converter's 'Solid' pose channels (datasets/convert_market.py uses
                                    the same three functions to write `pose_peaks` dense maps)

`import utils` fails on `import tensorflow`, so this script parses /root/reference/utils.py at run time, compiles
ONLY those four function definitions from the reference's text (nothing of it is written anywhere), and calls them
under python 3.  The single compatibility shim: the name `np.float` (removed from numpy 1.24+) resolves to the
builtin `float` it always aliased.  Inputs are seeded; inputs and the reference's outputs (bit-packed, the maps only
hold -1 / +1) go into the fixture, so the tests need neither the reference nor this script.

    python tests/golden/make_pose_golden.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/utils.py"
WANTED = ("py_poseInflate", "_getSparseKeypoint", "_getSparsePose", "_sparse2dense")


class _NumpyWithFloatAlias(object):
    """`np` as the reference's python-2-era numpy spelled it: np.float is the builtin float."""
    float = float

    def __getattr__(self, name):
        return getattr(np, name)


def reference_functions():
    tree = ast.parse(open(REF).read(), REF)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(d.name for d in defs) == sorted(WANTED), [d.name for d in defs]
    ns = {"np": _NumpyWithFloatAlias()}
    exec(compile(ast.Module(body=defs, type_ignores=[]), REF, "exec"), ns)
    return {n: ns[n] for n in WANTED}


def cases():
    """name -> (rcv [B,K,3] float64, is_normalized, H, W)."""
    rng = np.random.RandomState(20240607)
    out = {}
    # pixel coordinates as the records hold them (integers inside the image), corners and borders included
    B, K, H, W = 3, 18, 128, 64
    r = rng.randint(0, H, size=(B, K)).astype(np.float64)
    c = rng.randint(0, W, size=(B, K)).astype(np.float64)
    v = (rng.uniform(size=(B, K)) < 0.8).astype(np.float64)
    r[0, :6] = [0, 0, H - 1, H - 1, 2, H - 3]
    c[0, :6] = [0, W - 1, 0, W - 1, 3, W - 2]
    v[0, :6] = 1
    out["pixel_128x64"] = (np.stack([r, c, v], -1), False, H, W)
    # normalised coordinates as the pose decoder emits them: fractional, some outside [-1,1] (clamped to the image)
    r = rng.uniform(-1.15, 1.15, size=(B, K))
    c = rng.uniform(-1.15, 1.15, size=(B, K))
    v = (rng.uniform(size=(B, K)) < 0.8).astype(np.float64)
    r[1, :4] = [-1.0, 1.0, -1.0, 0.999]
    c[1, :4] = [-1.0, 1.0, 1.0, -0.999]
    v[1, :4] = 1
    out["normalized_128x64"] = (np.stack([r, c, v], -1), True, H, W)
    # DeepFashion geometry
    B, H, W = 2, 256, 256
    r = rng.uniform(-1.05, 1.05, size=(B, K))
    c = rng.uniform(-1.05, 1.05, size=(B, K))
    v = (rng.uniform(size=(B, K)) < 0.8).astype(np.float64)
    out["normalized_256x256"] = (np.stack([r, c, v], -1), True, H, W)
    return out


def main():
    f = reference_functions()
    fix = {}
    all_cases = {k: (v[0].astype(np.float32).astype(np.float64),) + v[1:] for k, v in cases().items()}   # fp32-exact inputs
    for name, (rcv, norm, H, W) in all_cases.items():
        dense = f["py_poseInflate"](rcv.copy(), is_normalized=norm, radius=4, img_H=H, img_W=W)
        assert dense.shape == (rcv.shape[0], H, W, rcv.shape[1]) and set(np.unique(dense)) <= {-1.0, 1.0}
        fix[name + "/rcv"] = rcv
        fix[name + "/meta"] = np.array([int(norm), H, W])
        fix[name + "/bits"] = np.packbits(dense > 0)
        print("%-20s %s  %d pixels set" % (name, dense.shape, int((dense > 0).sum())))
    # the converter's Solid channels for the pixel case: peaks[k] = [] or [(col, row)] (x first, utils.py:432-433)
    rcv, _, H, W = all_cases["pixel_128x64"]
    maps = []
    for b in range(rcv.shape[0]):
        peaks = [[(int(cc), int(rr))] if vv else [] for rr, cc, vv in rcv[b]]
        ind, val, shape = f["_getSparsePose"](peaks, H, W, rcv.shape[1], radius=4, mode='Solid')
        maps.append(f["_sparse2dense"](ind, val, shape))
    solid = np.stack(maps)
    assert set(np.unique(solid)) <= {0.0, 1.0}
    fix["pixel_128x64/solid_bits"] = np.packbits(solid > 0)
    path = os.path.join(HERE, "pose_reference.npz")
    np.savez_compressed(path, **fix)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
