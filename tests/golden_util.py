"""Loading of the committed golden vectors (tests/golden/*.npz, made by oracle/gen_golden.py from
the real libmmd)."""
import glob
import os

import numpy as np

from simple_mmd_renderer_amd.synth import FlatModel

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixture_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "g*.npz"))
                  if "checksums" not in p)


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    sdef = z["sdef"] if z["sdef"].shape[0] else None
    m = FlatModel(z["positions"], z["normals"], z["uvs"], z["skin_type"], z["bone_ids"],
                  z["bone_weights"], z["bone_pos"], z["bone_parent"], z["morph_type"], z["morph_off"],
                  z["morph_index"], z["morph_value"], sdef)
    exp = {k: z[k] for k in ("rates", "palette", "expect_pos", "expect_nrm", "expect_v32",
                             "norm_type", "norm_ids", "norm_w")}
    exp["normalize"] = bool(int(z["normalize"]))
    return m, exp


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(got, want, what=""):
    g, w = bits(got), bits(want)
    if not np.array_equal(g, w):
        bad = np.argwhere(g != w)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {bad.shape[0]} of {g.size} values differ bitwise; first at {i}: "
                             f"got {np.asarray(got)[i]!r} want {np.asarray(want)[i]!r}")


def assert_bits_equal_or_both_nan(got, want, what=""):
    """Bit-exact except where BOTH are NaN: the reference's CCD-IK produces NaN for degenerate chains (0/0 when a
    link coincides with its target), and a NaN's sign / payload bits differ between x86 and the GPU."""
    g, w = bits(got), bits(want)
    both_nan = np.isnan(np.asarray(got, np.float32)) & np.isnan(np.asarray(want, np.float32))
    diff = (g != w) & ~both_nan
    if diff.any():
        bad = np.argwhere(diff)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {bad.shape[0]} of {g.size} values differ bitwise; first at {i}: "
                             f"got {np.asarray(got)[i]!r} want {np.asarray(want)[i]!r}")
