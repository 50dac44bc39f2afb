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
                "defaults_64x64_n800"]


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
