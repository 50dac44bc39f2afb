"""Host-side Python of the package: camera constants, COLMAP reader, containers, generator,
strip plan.  CPU only; nothing here renders."""
import os
import struct

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from intro_to_gaussian_splatting_amd import GaussianScene, Gaussians, colmap, strips
from intro_to_gaussian_splatting_amd.image import GaussianImage
from intro_to_gaussian_splatting_amd.synthetic import make_scene, write_colmap_text


def _camera_from_fixture(g):
    cam = colmap.Camera(1, "PINHOLE", int(g["width"]), int(g["height"]),
                        np.array([float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"])]))
    img = colmap.Image(1, np.asarray(g["qvec"]), np.asarray(g["tvec"]), 1, "x.jpg")
    return GaussianImage(cam, img, device="cpu")


def test_camera_constants_bit_equal_to_reference(golden):
    g = golden
    im = _camera_from_fixture(g)
    assert np.array_equal(im.world2view.numpy(), g["world2view"])
    assert np.array_equal(im.projection_matrix.numpy(), g["projection_matrix"])
    assert np.array_equal(im.full_proj_transform.numpy(), g["full_proj_transform"])
    assert np.array_equal(im.tan_fovX.numpy(), g["tan_fovX"]) and np.array_equal(im.tan_fovY.numpy(), g["tan_fovY"])
    assert np.array_equal(im.f_x.numpy(), g["f_x"]) and np.array_equal(im.f_y.numpy(), g["f_y"])
    c = im.gsx_camera()
    assert np.array_equal(np.array(c.world2view[:], np.float32), g["world2view"].reshape(-1))
    assert np.array_equal(np.array(c.full_proj[:], np.float32), g["full_proj_transform"].reshape(-1))
    assert (c.width, c.height) == (int(g["width"]), int(g["height"]))
    assert np.float32(c.tan_fovx) == g["tan_fovX"][0] and np.float32(c.fx) == g["f_x"][0]


def test_colmap_text_and_binary_readers_agree(tmp_path):
    sc = make_scene(10, 320, 200, seed=1)
    txt = tmp_path / "txt"
    write_colmap_text(str(txt), sc, image_id=7, name="frame.jpg")
    cams, imgs = colmap.read_camera_file(str(txt)), colmap.read_image_file(str(txt))
    assert cams[1].model == "PINHOLE" and (cams[1].width, cams[1].height) == (320, 200)
    assert np.allclose(cams[1].params, [240.0, 240.0, 160.0, 100.0])
    assert imgs[7].name == "frame.jpg" and imgs[7].camera_id == 1
    assert np.array_equal(imgs[7].qvec, sc["qvec"]) and np.array_equal(imgs[7].tvec, sc["tvec"])
    # the same model in COLMAP's binary format (cameras.bin / images.bin take precedence)
    b = tmp_path / "bin"
    os.makedirs(b)
    with open(b / "cameras.bin", "wb") as f:
        f.write(struct.pack("<Q", 1))
        f.write(struct.pack("<iiQQ", 1, 1, 320, 200))
        f.write(struct.pack("<dddd", 240.0, 240.0, 160.0, 100.0))
    with open(b / "images.bin", "wb") as f:
        f.write(struct.pack("<Q", 1))
        f.write(struct.pack("<idddddddi", 7, *sc["qvec"], *sc["tvec"], 1))
        f.write(b"frame.jpg\x00")
        f.write(struct.pack("<Q", 2))
        f.write(struct.pack("<ddqddq", 1.0, 2.0, -1, 3.0, 4.0, 5))
    cams_b, imgs_b = colmap.read_camera_file(str(b)), colmap.read_image_file(str(b))
    assert cams_b[1].model == cams[1].model and np.array_equal(cams_b[1].params, cams[1].params)
    assert imgs_b[7].name == "frame.jpg" and np.array_equal(imgs_b[7].qvec, imgs[7].qvec)
    with pytest.raises(ValueError):
        colmap.read_camera_file(str(tmp_path / "nothing"))


def test_gaussians_defaults_follow_the_reference():
    pts = torch.rand(5, 3)
    rgb = torch.tensor([[0.0, 128.0, 255.0]] * 5)
    g = Gaussians(pts, rgb, device="cpu")
    assert g.points.dtype == torch.float32 and g.colors.shape == (5, 3)
    assert torch.equal(g.colors[0], torch.tensor([0.0, 0.5, 255.0 / 256.0]))
    assert torch.all(g.scales == 0.001) and g.scales.shape == (5, 3)
    assert torch.equal(g.quaternions, torch.tensor([[1.0, 0, 0, 0]] * 5))
    assert g.opacity.shape == (5, 1)
    assert abs(torch.sigmoid(g.opacity[0, 0]).item() - 0.9999) < 1e-6
    assert len(g) == 5


def test_scene_construction_and_cpu_tensors_are_rejected(tmp_path):
    sc = make_scene(50, 64, 48, seed=2)
    write_colmap_text(str(tmp_path), sc)
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"],
                              device="cpu")
    scene = GaussianScene(str(tmp_path), g)
    assert list(scene.images) == [1] and scene.gaussians is g
    assert scene.images[1].width.item() == 64.0 and scene.images[1].name == "synthetic.jpg"
    # the product path has no CPU fallback: CPU-resident Gaussians must raise, not render
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        scene.render_image(1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        scene.preprocess(1)


def test_synthetic_generator_is_frozen():
    a, b = make_scene(1000, 256, 256, seed=0), make_scene(1000, 256, 256, seed=0)
    for k in a:
        assert np.array_equal(a[k], b[k])
    g = load_golden("c1_256x256_n2000")
    c = make_scene(2000, 256, 256, seed=0)
    for k in ("points", "scales", "quaternions", "opacity", "colors_0_255"):
        assert np.array_equal(c[k], g[k]), k   # the committed fixture came from this generator
    d = make_scene(1000, 256, 256, seed=0, behind_fraction=0.3)
    assert np.array_equal(d["scales"][:, 0] > 0, np.ones(1000, bool))


def test_strip_plan_covers_every_tile_once():
    for n_tiles in (0, 1, 2, 7, 67, 119, 239):
        for world in (1, 2, 3, 4, 8):
            per, plan = strips.strip_plan(n_tiles, world)
            assert len(plan) == world
            covered = [t for a, b in plan for t in range(a, b)]
            assert covered == list(range(n_tiles))
            assert all(b - a <= per for a, b in plan)
            assert all(a == min(r * per, n_tiles) for r, (a, b) in enumerate(plan))
    assert strips.tiles_along(1920, 16) == 119 and strips.tiles_along(1080, 16) == 67
    assert strips.tiles_along(16, 16) == 0 and strips.tiles_along(17, 16) == 1
    assert strips.tiles_along(3840, 16) == 239 and strips.tiles_along(2160, 16) == 134


def test_trained_ply_round_trip_and_point_cloud(tmp_path):
    """ply.py (build extension): trained-3DGS layout <-> arrays, and the xyz+rgb point-cloud flavour."""
    from intro_to_gaussian_splatting_amd import ply

    rs = np.random.RandomState(3)
    n, deg = 37, 2
    k = (deg + 1) ** 2
    pts = rs.normal(size=(n, 3)).astype(np.float32)
    sh = rs.normal(size=(n, k, 3)).astype(np.float32)
    scales = np.exp(rs.normal(-3.0, 0.5, size=(n, 3))).astype(np.float32)
    quats = rs.normal(size=(n, 4)).astype(np.float32)
    op = rs.normal(size=(n, 1)).astype(np.float32)
    path = str(tmp_path / "trained.ply")
    ply.save_trained(path, pts, sh, scales, quats, op)
    d = ply.load_gaussians(path)
    assert int(d["sh_degree"]) == deg and d["sh"].shape == (n, k, 3)
    assert np.array_equal(d["points"], pts) and np.array_equal(d["sh"], sh)
    assert np.array_equal(d["quaternions"], quats) and np.array_equal(d["opacity"], op)
    assert np.allclose(d["scales"], scales, rtol=1e-6)          # stored as log, returned linear
    g = Gaussians.from_ply(path, device="cpu")
    assert g.sh is not None and g.sh_degree == deg and tuple(g.sh.shape) == (n, k, 3)
    assert torch.allclose(g.scales, torch.from_numpy(scales), rtol=1e-6)
    # channel-major f_rest ordering of the published format: f_rest_0 is coefficient 1 of RED
    v = ply.read_vertices(path)
    assert np.array_equal(v["f_rest_0"], sh[:, 1, 0]) and np.array_equal(v["f_rest_%d" % (k - 1)], sh[:, 1, 1])
    # point cloud flavour (what the reference's storePly writes): x y z nx ny nz red green blue
    pc = str(tmp_path / "cloud.ply")
    rec = np.zeros(5, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
                             ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["red"], rec["blue"] = np.arange(5), 255, 128
    with open(pc, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 5\n")
        for nm in ("x", "y", "z", "nx", "ny", "nz"):
            f.write(b"property float %s\n" % nm.encode())
        for nm in ("red", "green", "blue"):
            f.write(b"property uchar %s\n" % nm.encode())
        f.write(b"end_header\n")
        rec.tofile(f)
    g2 = Gaussians.from_ply(pc, device="cpu")
    assert g2.sh is None and torch.all(g2.colors[:, 0] == 255.0 / 256.0) and torch.all(g2.colors[:, 2] == 0.5)
    assert torch.all(g2.scales == 0.001)


def test_sh_oracle_degree0_reproduces_rgb():
    from oracle import cpu_ref

    rs = np.random.RandomState(0)
    rgb = rs.uniform(0, 1, size=(50, 3))
    pts = rs.normal(size=(50, 3))
    sh0 = ((rgb - 0.5) / 0.28209479177387814)[:, None, :]
    out = cpu_ref.sh_to_rgb(pts, sh0, 0, [0.1, 0.2, 0.3])
    assert np.allclose(out, rgb, atol=1e-6)
    # higher bands integrate to zero over the sphere: the mean colour over many directions is the DC term
    sh = np.concatenate([sh0[:1].repeat(20000, 0), rs.normal(size=(20000, 15, 3)) * 0 + rs.normal(size=(1, 15, 3))], 1)
    dirs = rs.normal(size=(20000, 3))
    cols = cpu_ref.sh_to_rgb(dirs * 5.0, sh * np.array([1.0] + [0.05] * 15)[None, :, None], 3, [0, 0, 0])
    assert np.allclose(cols.mean(0), rgb[0], atol=5e-3)


def test_constructor_defaults_equal_the_references_bit_for_bit():
    """defaults_64x64_n800 was captured without overwriting anything the reference's Gaussians
    constructor sets (gaussians.py:19-33): our constructor must produce the same tensors."""
    from conftest import ROOT, load_golden

    g = load_golden("defaults_64x64_n800")
    ours = Gaussians(torch.from_numpy(g["points"]), torch.from_numpy(g["colors_0_255"]), device="cpu")
    assert np.array_equal(ours.scales.numpy(), g["scales"])
    assert np.array_equal(ours.quaternions.numpy(), g["quaternions"])
    assert np.array_equal(ours.opacity.numpy(), g["opacity"])
    assert np.array_equal(ours.colors.numpy(), g["colors"])


def test_colmap_readers_equal_the_references_parse_of_a_committed_model():
    """tests/golden/colmap_model/{bin,txt}: a 4-camera / 4-image sparse model (PINHOLE, SIMPLE_PINHOLE, OPENCV,
    SIMPLE_RADIAL; images with 0..5 keypoints, a name with a directory) written by oracle/capture_golden.py;
    tests/golden/colmap_model.npz: what the REFERENCE's readers (splat/read_colmap.py:87-239) returned for
    those very files.  Our readers must return the same records, both flavours."""
    from conftest import GOLDEN_DIR

    g = np.load(os.path.join(GOLDEN_DIR, "colmap_model.npz"))
    for fl in ("bin", "txt"):
        model = os.path.join(GOLDEN_DIR, "colmap_model", fl)
        cams, imgs = colmap.read_camera_file(model), colmap.read_image_file(model)
        assert sorted(cams) == list(g[fl + "_camera_ids"]) and sorted(imgs) == list(g[fl + "_image_ids"])
        for cid, c in cams.items():
            assert c.id == cid and c.model == str(g["%s_cam%d_model" % (fl, cid)])
            assert [c.width, c.height] == list(g["%s_cam%d_size" % (fl, cid)])
            assert np.array_equal(c.params, g["%s_cam%d_params" % (fl, cid)])
        for iid, im in imgs.items():
            assert im.id == iid and im.name == str(g["%s_img%d_name" % (fl, iid)])
            assert im.camera_id == int(g["%s_img%d_camera_id" % (fl, iid)])
            assert np.array_equal(im.qvec, g["%s_img%d_qvec" % (fl, iid)])
            assert np.array_equal(im.tvec, g["%s_img%d_tvec" % (fl, iid)])
            assert np.array_equal(np.asarray(im.xys).reshape(-1, 2), g["%s_img%d_xys" % (fl, iid)])
            assert np.array_equal(im.point3D_ids, g["%s_img%d_point3D_ids" % (fl, iid)])
    # the binary flavour wins when both are present, as in the reference (splat/utils.py:269-290)
    assert colmap.read_camera_file(os.path.join(GOLDEN_DIR, "colmap_model", "bin"))[7].model == "OPENCV"


def test_unused_camera_matrices_equal_the_references(golden):
    """intrinsic_matrix, extrinsic_matrix, projection and camera_center (splat/image.py:32-39, 66-70): carried
    for users of the reference's attribute surface, bit-equal to what the reference computed."""
    im = _camera_from_fixture(golden)
    assert np.array_equal(im.intrinsic_matrix.numpy(), golden["intrinsic_matrix"])
    assert np.array_equal(im.extrinsic_matrix.numpy(), golden["extrinsic_matrix"])
    assert np.allclose(im.projection.numpy(), golden["projection"], rtol=1e-6, atol=1e-6)
    assert np.allclose(im.camera_center.numpy(), golden["camera_center"], rtol=1e-5, atol=1e-6)


def test_bench_gpus_n_never_reports_fewer_gpus_than_asked(tmp_path):
    """`python bench.py --gpus N` without a launcher starts N fresh ranks itself or exits non-zero (round-2 verdict:
    it used to run world = 1 and print "n_gpus": 1).  This container has no GPU, so both forms must refuse -- with
    no JSON line on stdout -- and a WORLD_SIZE that contradicts --gpus must be refused too."""
    import subprocess
    import sys

    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout
    assert "refusing" in r.stderr or "needs a GPU" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "1"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "WORLD_SIZE=2" in r.stderr


def test_device_buffer_caches_are_bounded():
    """The workspace keeps one scratch buffer per (device, stream) and a scene one hints buffer per view: both forget
    their least recently used entry beyond a limit (a viewer that creates streams, a sweep over hundreds of cameras
    or a strip plan that moves with every rebalance must not grow device memory for the life of the process)."""
    from intro_to_gaussian_splatting_amd.gaussian_scene import _HINT_VIEWS, _Lru, _Workspace

    lru = _Lru(3)
    for k in "abc":
        lru.store(k, k.upper())
    assert lru.lookup("a") == "A"            # refreshed: "b" is now the oldest
    lru.store("d", "D")
    assert list(lru) == ["c", "a", "d"] and lru.lookup("b") is None
    lru.store("a", "A2")                     # an entry that is replaced (a larger buffer) keeps one slot
    assert len(lru) == 3 and lru.lookup("a") == "A2"
    assert _Workspace().buffers.limit == 8 and _HINT_VIEWS == 64


def test_gpu_count_comes_from_sysfs_not_from_hip(tmp_path, monkeypatch):
    """bench.py --gpus N counts the GPUs its children would see from the KFD topology in sysfs (nodes with SIMDs), narrowed
    by the *_VISIBLE_DEVICES variables -- the launching process itself never opens the GPU (round-3 verdict, weak #9:
    torch.cuda.device_count() in the parent may initialise HIP before the ranks are started)."""
    import importlib
    import sys

    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    for i, simd in enumerate([0, 0, 256, 256, 256, 256]):          # two CPU nodes, four GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpus(str(tmp_path)) == 4
    assert bench.visible_gpus(str(tmp_path / "absent")) == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus(str(tmp_path)) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpus(str(tmp_path)) == 1                  # HIP's "0,2" is then applied to ONE device: index 2 is out
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus(str(tmp_path)) == 0
    # and the launcher's source has no device query left in the parent
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def launch_ranks"):src.index("def main()")]
    assert "torch.cuda" not in body and "HSA_ENABLE_IPC_MODE_LEGACY\"" not in body


def test_orbit_poses_and_trained_like_generator():
    """synthetic.orbit_poses: rigid poses 1 degree apart around a pivot in front of the base camera, the middle one IS the
    base pose; synthetic.make_trained_like_scene: deterministic, needle footprints to 50:1, bimodal opacity, SH."""
    from intro_to_gaussian_splatting_amd.synthetic import (TREEHILL_QVEC, TREEHILL_TVEC, _rotation, make_trained_like_scene,
                                                           orbit_poses)

    poses = orbit_poses(9)
    q_mid, t_mid = poses[4]
    assert np.allclose(q_mid, np.asarray(TREEHILL_QVEC) / np.linalg.norm(TREEHILL_QVEC), atol=1e-9)
    assert np.allclose(t_mid, TREEHILL_TVEC, atol=1e-12)
    pivot_cam = np.array([0.0, 0.0, 6.0])
    pivot_world = _rotation(TREEHILL_QVEC).T @ (pivot_cam - np.asarray(TREEHILL_TVEC))
    for k, (q, t) in enumerate(poses):
        R = _rotation(q)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1.0) < 1e-12
        assert np.allclose(R @ pivot_world + t, pivot_cam, atol=1e-9)              # the pivot stays where it is in every view
        if k:
            Rp = _rotation(poses[k - 1][0])
            angle = np.degrees(np.arccos(np.clip((np.trace(R @ Rp.T) - 1.0) / 2.0, -1.0, 1.0)))
            assert abs(angle - 1.0) < 1e-6
    a, b = make_trained_like_scene(4000, 640, 360), make_trained_like_scene(4000, 640, 360)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a["sh"].shape == (4000, 16, 3) and a["sh"].dtype == np.float32
    ratio = a["scales"].max(axis=1) / a["scales"].min(axis=1)
    assert ratio.max() > 30.0 and ratio.max() <= 50.0 * 1.001 and np.median(ratio) > 3.0
    op = a["opacity"][:, 0]
    assert 0.35 < (op < 0.0).mean() < 0.65 and (op > 2.0).mean() > 0.3 and (op < -2.0).mean() > 0.3


def test_visible_rows_flag_names_the_class_or_says_clear():
    """_ffi.visible_rows_flag (GSX_FLAG_SMALL_BATCH / _ONE_VISIBLE, include/gsx.h): 0 = the call assumed the right row
    class, a flag = issue it again with that flag, -1 = it carried a flag and four or more Gaussians are visible: issue
    it again with NEITHER (round 5 returned 0 there: "nothing to do")."""
    from intro_to_gaussian_splatting_amd import _ffi

    one, few = _ffi.GSX_FLAG_ONE_VISIBLE, _ffi.GSX_FLAG_SMALL_BATCH
    f = _ffi.visible_rows_flag
    assert f(9, 9, 0) == 0 and f(9, 4, 0) == 0 and f(3, 3, 0) == 0 and f(1, 1, 0) == 0
    assert f(9, 2, 0) == few and f(9, 3, 0) == few and f(9, 1, 0) == one and f(3, 1, 0) == one and f(3, 2, 0) == 0
    assert f(9, 2, few) == 0 and f(9, 1, one) == 0 and f(9, 1, few) == one and f(9, 3, one) == few
    assert f(9, 4, few) == -1 and f(9, 9, one) == -1
    assert f(9, 0, 0) == 0 and f(9, 0, few) == 0                       # nothing visible: nothing was multiplied
    assert _ffi.with_rows_flag(few | 1024, -1) == 1024 and _ffi.with_rows_flag(few | 2, one) == (one | 2)
    assert _ffi.with_rows_flag(0, few) == few


def test_spatially_ordered_is_a_permutation_that_puts_neighbours_together():
    """Gaussians.spatially_ordered() (opt-in): every parameter array -- SH coefficients included -- is
    the same permutation of the original's, ``original_index`` says which; rows that follow each other are close in space;
    ordering an ordered container is the identity; the source is untouched."""
    import torch

    from intro_to_gaussian_splatting_amd import Gaussians
    from intro_to_gaussian_splatting_amd.synthetic import make_trained_like_scene

    sc = make_trained_like_scene(4000, 320, 240, seed=2)
    g = Gaussians.from_arrays(sc["points"], sc["colors_0_255"], sc["scales"], sc["quaternions"], sc["opacity"], device="cpu")
    g.sh, g.sh_degree = torch.from_numpy(sc["sh"]), int(sc["sh_degree"])
    before = g.points.clone()
    o = g.spatially_ordered()
    assert g.original_index is None and torch.equal(g.points, before)
    idx = o.original_index
    assert idx.dtype == torch.int32 and sorted(idx.tolist()) == list(range(4000))
    assert torch.equal(o.row_of_index[idx.long()].long(), torch.arange(4000))                    # the inverse permutation
    bb = o.block_bounds
    assert tuple(bb.shape) == (16, 8) and bool((bb[:, 0:3] <= bb[:, 4:7]).all())
    blk = o.points[256:512]
    assert torch.equal(bb[1, 0:3], blk.amin(dim=0)) and torch.equal(bb[1, 4:7], blk.amax(dim=0))
    assert float(bb[1, 3]) == float(o.scales[256:512].abs().max())
    for name in ("points", "colors", "scales", "quaternions", "opacity", "sh"):
        assert torch.equal(getattr(o, name), getattr(g, name)[idx.long()]), name
    assert o.sh_degree == g.sh_degree and o.spatially_ordered() is o
    step = lambda p: float((p[1:] - p[:-1]).norm(dim=1).mean())  # noqa: E731
    assert step(o.points) < 0.25 * step(g.points)
    empty = Gaussians(torch.zeros((0, 3)), torch.zeros((0, 3)), device="cpu").spatially_ordered()
    assert empty.original_index.numel() == 0


def test_block_bounds_follow_points_and_scales():
    """Gaussians.current_block_bounds(): the boxes a strip's projection trusts (GsxParams.block_bounds) are recomputed -- in
    place, at the address a captured frame holds -- when ``points`` / ``scales`` were modified in place or replaced since."""
    import torch

    from intro_to_gaussian_splatting_amd import Gaussians

    g = Gaussians(torch.rand(1000, 3), torch.rand(1000, 3) * 255, device="cpu").spatially_ordered()
    first = g.current_block_bounds()
    before, addr = first.clone(), first.data_ptr()
    assert g.current_block_bounds() is first and torch.equal(first, before)          # nothing changed: nothing recomputed
    g.points[5] += 100.0
    after = g.current_block_bounds()
    assert after.data_ptr() == addr and not torch.equal(after, before) and float(after[0, 4:7].max()) > 50.0
    g.scales = g.scales * 2
    assert float(g.current_block_bounds()[0, 3]) == float(g.scales[:256].abs().max()) and g.block_bounds.data_ptr() == addr
    plain = Gaussians(torch.rand(10, 3), torch.rand(10, 3), device="cpu")
    assert plain.current_block_bounds() is None
