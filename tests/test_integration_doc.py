"""INTEGRATION.md is code a maintainer pastes: its fenced Python is compiled, section B's ctypes stub is held against
the header's structs (CPU) and EXECUTED on a reference-made fixture with canary words behind its stats struct (``-m gpu``).

Round 5's document declared a 64-byte ``GsxFrameStats`` while the library wrote 72 bytes: nothing ran the document.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden
from intro_to_gaussian_splatting_amd import _ffi

DOC = os.path.join(ROOT, "INTEGRATION.md")


def _python_blocks():
    return re.findall(r"```python\n(.*?)```", open(DOC).read(), flags=re.S)


def _section_b_namespace():
    """Section B's stub, executed as written except for the library's path."""
    block = [b for b in _python_blocks() if "def gsx_render_image" in b]
    assert len(block) == 1
    src = block[0].replace("/path/to/intro_to_gaussian_splatting_amd/libgsx.so", _ffi.LIB_PATH)
    assert _ffi.LIB_PATH in src
    ns = {}
    exec(compile(src, "INTEGRATION.md#B", "exec"), ns)
    return ns


def test_every_python_block_of_the_document_compiles():
    blocks = _python_blocks()
    assert len(blocks) >= 3
    for k, b in enumerate(blocks):
        compile(b, "INTEGRATION.md block %d" % k, "exec")


def test_documented_structs_are_the_headers():
    """The stub's GsxFrameStats is field for field (name, offset, size) the binding's own mirror of include/gsx.h (whose
    layout tests/test_cabi.py pins), and the sizes the prose quotes are the real ones."""
    ns = _section_b_namespace()
    doc, mine = ns["GsxFrameStats"], _ffi.GsxFrameStats
    layout = lambda c: [(n, getattr(c, n).offset, getattr(c, n).size) for n, _ in c._fields_]  # noqa: E731
    assert layout(doc) == layout(mine) and ctypes.sizeof(doc) == ctypes.sizeof(mine) == 72
    text = open(DOC).read()
    assert "`GsxParams`, %d bytes in ABI %d" % (ctypes.sizeof(_ffi.GsxParams), _ffi.load().gsx_version()) in text
    # the C loops: whoever may set GSX_FLAG_PLAIN_FOOTPRINTS reads n_redo, and the version is checked
    c_blocks = re.findall(r"```c\n(.*?)```", text, flags=re.S)
    viewer = [b for b in c_blocks if "GSX_FLAG_HINTS_VALID" in b]
    assert len(viewer) == 1 and "st.n_redo" in viewer[0] and "gsx_version() != GSX_VERSION" in viewer[0]
    # DESIGN.md quotes the same version
    assert "`GSX_VERSION` %d" % _ffi.load().gsx_version() in open(os.path.join(ROOT, "DESIGN.md")).read()


class _Legacy64(ctypes.Structure):
    """What round 5's document declared ("ABI 300: 64 bytes") and a maintainer may have pasted."""
    _fields_ = [("n_visible", ctypes.c_int64), ("n_instances", ctypes.c_int64), ("n_tiles", ctypes.c_int64),
                ("reserved", ctypes.c_int64), ("stage_ms", ctypes.c_float * 6), ("n_kept", ctypes.c_int64)]


@pytest.mark.gpu
@pytest.mark.parametrize("legacy", [False, True])
def test_section_b_stub_runs_and_does_not_overrun_its_stats(legacy):
    """The document's ``gsx_render_image`` (replacing ``ext.render_image`` at splat/gaussian_scene.py:263-285) on the
    reference's own stage-1 arrays of a golden fixture: the reference's image, and not a byte written behind the stats
    struct -- the one the document declares (72 bytes, of which a NULL-params call writes 64) and the 64-byte one of
    round 5's document."""
    import torch

    assert torch.cuda.is_available(), "needs a GPU"
    ns = _section_b_namespace()
    struct = _Legacy64 if legacy else ns["GsxFrameStats"]
    backing = []

    def canaried():
        buf = (ctypes.c_ubyte * 256)(*([0xAB] * 256))
        backing.append(buf)
        return struct.from_buffer(buf)

    ns["GsxFrameStats"] = canaried
    g = load_golden("c1_256x256_n2000")
    dev = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).to("cuda:0")  # noqa: E731
    w, h = int(g["width"]), int(g["height"])
    img = ns["gsx_render_image"](h, w, int(g["tile"]), dev("pre_points"), dev("pre_colors"), dev("pre_inverse_covariance_2d"),
                                 dev("pre_min_x"), dev("pre_max_x"), dev("pre_min_y"), dev("pre_max_y"),
                                 dev("pre_sigmoid_opacity"))
    torch.cuda.synchronize()
    assert tuple(img.shape) == (w, h, 3)
    assert float(np.abs(img.cpu().numpy() - g["image"]).max()) <= 1e-5
    assert len(backing) >= 1
    for buf in backing:
        st = struct.from_buffer(buf)
        assert st.n_visible == g["pre_points"].shape[0] and st.n_instances > 0 and st.n_tiles == 15 * 15
        assert bytes(buf[64:]) == b"\xab" * (256 - 64), "the call wrote behind the first 64 bytes of GsxFrameStats"


@pytest.mark.gpu
def test_stats_size_72_delivers_n_redo_and_64_does_not():
    """GsxParams.stats_size through the whole-path entry: a needle scene has tiles with reference-order records
    (n_redo > 0); told 72 bytes the call reports them, told 64 it leaves byte 64.. alone."""
    import tempfile

    import torch

    from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians
    from intro_to_gaussian_splatting_amd.synthetic import write_colmap_text

    assert torch.cuda.is_available(), "needs a GPU"
    g = load_golden("needle_160x160_n110")
    sc = {k: g[k] for k in ("points", "colors_0_255", "scales", "quaternions", "opacity", "qvec", "tvec", "fx", "fy", "cx",
                            "cy", "width", "height")}
    with tempfile.TemporaryDirectory() as tmp:
        write_colmap_text(tmp, sc)
        scene = GaussianScene(tmp, Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"],
                                                         sc["opacity"], device="cuda:0"))
    lib = _ffi.load()
    gs, cam = scene.gaussians, scene.images[1].gsx_camera()
    n, w, h = gs.points.shape[0], int(g["width"]), int(g["height"])
    out = torch.empty((w, h, 3), dtype=torch.float32, device="cuda:0")
    ws = torch.empty(lib.gsx_workspace_bytes(n, w, h, 16, 64 * n + 4096), dtype=torch.uint8, device="cuda:0")
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    seen = {}
    for size in (72, 64):
        prm = _ffi.default_params()
        prm.stats_size = size
        buf = (ctypes.c_ubyte * 128)(*([0xAB] * 128))
        st = _ffi.GsxFrameStats.from_buffer(buf)
        rc = lib.gsx_render_forward(ctypes.byref(cam), ptr(gs.points), ptr(gs.scales), ptr(gs.quaternions), ptr(gs.opacity),
                                    ptr(gs.colors), n, 16, ptr(out), ctypes.byref(prm), ctypes.byref(st), ptr(ws),
                                    ws.numel(), None)
        _ffi.check(rc)
        torch.cuda.synchronize()
        assert float(np.abs(out.cpu().numpy() - g["image"]).max()) <= 1e-5
        assert bytes(buf[size:]) == b"\xab" * (128 - size)
        seen[size] = st.n_redo if size == 72 else None
    assert seen[72] > 0
