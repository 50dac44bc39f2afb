import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_NAMES = ["small_64x48_n300", "small_80x64_n120_tile8", "cull_96x80_n400", "c1_256x256_n2000",
                "tile2_40x32_n80", "pose_70x50_n250", "dense_48x48_n1500", "wide_64x64_n400", "tiny_48x48_n600",
                "defaults_64x64_n800", "needle_160x160_n110", "trainedlike_128x128_n3000",
                # at most three visible Gaussians: the reference's BLAS sums in other orders there (GSX_FLAG_SMALL_BATCH / _ONE_VISIBLE)
                "fewvisible_48x48_n9", "three_48x48_n3", "onevisible_48x48_n7", "single_48x48_n1"]
# stage 1 of the reference at the benchmark sizes C2 / C3 (oracle/capture_golden.py: STAGE1_FIXTURES)
STAGE1_NAMES = ["stage1_c2_1080p_n100000", "stage1_c3_1080p_n1000000"]


def pytest_addoption(parser):
    parser.addoption("--reverse", action="store_true", help="run the collected tests in reverse order "
                     "(shakes out tests that only pass because an earlier one initialised something)")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--reverse"):
        items.reverse()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real AMD GPU (MI355X); run with -m gpu")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def oracle_camera(g):
    """cpu_ref.Camera from the camera constants the reference itself computed (fixture)."""
    from oracle import cpu_ref

    return cpu_ref.Camera(g["world2view"], g["full_proj_transform"], g["tan_fovX"][0], g["tan_fovY"][0],
                          g["f_x"][0], g["f_y"][0], int(g["width"]), int(g["height"]))


def golden_preprocessed(g):
    """The reference's own PreprocessedScene arrays as a cpu_ref.Preprocessed."""
    from oracle import cpu_ref

    return cpu_ref.Preprocessed(
        g["pre_points"], g["pre_colors"], g["pre_covariance_2d"], g["pre_depths"],
        g["pre_inverse_covariance_2d"], g["pre_radius"], g["pre_points_xy"], g["pre_min_x"], g["pre_min_y"],
        g["pre_max_x"], g["pre_max_y"], g["pre_sigmoid_opacity"], g["order"])


@pytest.fixture(params=GOLDEN_NAMES)
def golden(request):
    return load_golden(request.param)


from oracle.golden_check import (FUZZ_TILE_FIXTURE_NAMES, STAGE1_FIELDS, TILE_FIXTURE_NAMES,  # noqa: E402,F401
                                 compare_stage1_with_reference, compare_tiles_with_reference, fuzz_tiles_cases,
                                 rows_in_reference_order, sha256, stage1_scene, tile_lists, tiles_scene)


def tie_mask(depths_sorted) -> np.ndarray:
    """Sorted positions that belong to a run of at least two EQUAL depths (bit patterns)."""
    d = np.ascontiguousarray(depths_sorted, np.float32).reshape(-1).view(np.uint32)
    eq = d[1:] == d[:-1]
    return np.concatenate([eq, [False]]) | np.concatenate([[False], eq])


def assert_same_order_outside_ties(order, ref_order, depths_sorted) -> int:
    """The permutation equals the reference's except inside runs of equal depths, where the reference's unstable
    ``torch.argsort`` (splat/gaussian_scene.py:117) leaves the order to its sort library and this build takes the
    original index; inside a run both hold the same Gaussians.  Returns the number of positions that differ."""
    order, ref_order = np.asarray(order, np.int64), np.asarray(ref_order, np.int64)
    assert order.shape == ref_order.shape
    tied = tie_mask(depths_sorted)
    assert np.array_equal(order[~tied], ref_order[~tied]), "depth permutation differs from the reference's argsort"
    d = np.ascontiguousarray(depths_sorted, np.float32).reshape(-1).view(np.uint32)
    run = np.concatenate([[0], np.cumsum(d[1:] != d[:-1])])
    key = lambda o: np.lexsort((o, run))            # noqa: E731  (members of every run, ascending)
    assert np.array_equal(order[key(order)], ref_order[key(ref_order)]), "a run of equal depths holds other Gaussians"
    return int(np.count_nonzero(order != ref_order))


def rows_by_index(a, order, n):
    """Depth-ordered rows put back at their original Gaussian index (other rows zero)."""
    a = np.ascontiguousarray(a)
    full = np.zeros((n,) + a.shape[1:], a.dtype)
    full[np.asarray(order, np.int64)] = a
    return full
