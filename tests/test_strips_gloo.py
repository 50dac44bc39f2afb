"""The N>1 path on CPU: two gloo ranks run the package's strip partition + frame gather
(intro_to_gaussian_splatting_amd/strips.py) with the oracle standing in for the per-strip
renderer, and the gathered frame must equal the single-process frame bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_preprocessed, load_golden
from intro_to_gaussian_splatting_amd import strips
from oracle import c_oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_strip_renderer(pre, width, height, tile, layout):
    def render(window, out, origin):
        img, _, _ = c_oracle.render(pre, width, height, tile, nthreads=1, window=window)   # (W,H,3)
        if layout == "hw3":
            img = np.ascontiguousarray(img.transpose(1, 0, 2))
        lead0 = origin[0] if layout == "wh3" else origin[1]
        out.copy_(torch.from_numpy(img[lead0:lead0 + out.shape[0]]))
    return render


def _worker(rank, world, port, name, layout, all_ranks, result_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden(name)
        w, h, t = int(g["width"]), int(g["height"]), int(g["tile"])
        fn = _oracle_strip_renderer(golden_preprocessed(g), w, h, t, layout)
        frame = strips.render_sharded(fn, w, h, t, layout, torch.device("cpu"), all_ranks=all_ranks)
        if frame is not None:
            np.save(os.path.join(result_dir, "frame_%d.npy" % rank), frame.numpy())
        else:
            assert rank != 0 and not all_ranks
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,layout,all_ranks", [
    ("c1_256x256_n2000", "wh3", False),
    ("c1_256x256_n2000", "hw3", True),
    ("cull_96x80_n400", "wh3", True),
    ("small_64x48_n300", "hw3", False),
])
def test_two_rank_strips_equal_single_frame(tmp_path, name, layout, all_ranks):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, layout, all_ranks, str(tmp_path)), nprocs=world, join=True)
    g = load_golden(name)
    ref = g["image"] if layout == "wh3" else g["image"].transpose(1, 0, 2)
    full, _, _ = c_oracle.render(golden_preprocessed(g), int(g["width"]), int(g["height"]), int(g["tile"]))
    full = full if layout == "wh3" else full.transpose(1, 0, 2)
    ranks = [0, 1] if all_ranks else [0]
    for r in ranks:
        frame = np.load(tmp_path / ("frame_%d.npy" % r))
        assert frame.shape == ref.shape
        assert np.array_equal(frame, full)                     # strips == one-process render, bit for bit
        assert np.max(np.abs(frame - ref)) <= 1e-6             # and both match the reference's image
    if not all_ranks:
        assert not (tmp_path / "frame_1.npy").exists()


def test_single_process_path_without_process_group():
    g = load_golden("small_64x48_n300")
    w, h, t = int(g["width"]), int(g["height"]), int(g["tile"])
    fn = _oracle_strip_renderer(golden_preprocessed(g), w, h, t, "wh3")
    frame = strips.render_sharded(fn, w, h, t, "wh3", torch.device("cpu"))
    full, _, _ = c_oracle.render(golden_preprocessed(g), w, h, t)
    assert np.array_equal(frame.numpy(), full)
    cache = {}
    a = strips.render_sharded(fn, w, h, t, "wh3", torch.device("cpu"), cache=cache)
    b = strips.render_sharded(fn, w, h, t, "wh3", torch.device("cpu"), cache=cache)
    assert a.data_ptr() == b.data_ptr() and np.array_equal(b.numpy(), full)   # buffers are reused


# ---- GSX_SEM_STD_3DGS: every tile of the frame, partial edge tiles -> the last strip overhangs

def _std_strip_renderer(sc, cam, colors, tile, layout):
    def render(window, out, origin):
        img, _, _, _ = c_oracle.render_std3dgs(sc["points"], colors, sc["scales"], sc["quaternions"], sc["opacity"],
                                               cam, tile=tile, nthreads=1, window=window)          # (H,W,3)
        if layout == "wh3":
            img = np.ascontiguousarray(img.transpose(1, 0, 2))
        lead0 = origin[0] if layout == "wh3" else origin[1]
        rows = img[lead0:lead0 + out.shape[0]]
        out.zero_()                                            # rows of the strip past the frame edge
        out[:rows.shape[0]].copy_(torch.from_numpy(np.ascontiguousarray(rows)))
    return render


def _std_inputs():
    from intro_to_gaussian_splatting_amd.synthetic import make_scene
    from oracle import cpu_ref

    w, h, tile = 100, 70, 16                                   # 7 x 5 tiles, both axes with a partial tile
    sc = make_scene(400, w, h, seed=11)
    cam = cpu_ref.build_camera(sc["qvec"], sc["tvec"], sc["fx"], sc["fy"], w, h)
    colors = (sc["colors_0_255"] / np.float32(256.0)).astype(np.float32)
    return sc, cam, colors, w, h, tile


def _std_worker(rank, world, port, layout, result_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, cam, colors, w, h, tile = _std_inputs()
        fn = _std_strip_renderer(sc, cam, colors, tile, layout)
        frame = strips.render_sharded(fn, w, h, tile, layout, torch.device("cpu"), semantics="std_3dgs")
        if frame is not None:
            np.save(os.path.join(result_dir, "frame_%d.npy" % rank), frame.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout,world", [("hw3", 2), ("wh3", 3)])
def test_std3dgs_strips_with_partial_edge_tiles(tmp_path, layout, world):
    mp.spawn(_std_worker, args=(world, _free_port(), layout, str(tmp_path)), nprocs=world, join=True)
    sc, cam, colors, w, h, tile = _std_inputs()
    full, _, _, _ = c_oracle.render_std3dgs(sc["points"], colors, sc["scales"], sc["quaternions"], sc["opacity"],
                                            cam, tile=tile, nthreads=1)
    full = full if layout == "hw3" else full.transpose(1, 0, 2)
    frame = np.load(tmp_path / "frame_0.npy")
    assert frame.shape == full.shape and np.array_equal(frame, full)


# ---- StripPipeline (frames in flight on a GPU; one frame at a time on the CPU devices of this test)

def _pipeline_worker(rank, world, port, name, layout, result_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden(name)
        w, h, t = int(g["width"]), int(g["height"]), int(g["tile"])
        pipe = strips.StripPipeline(_oracle_strip_renderer(golden_preprocessed(g), w, h, t, layout), w, h, t, layout,
                                    torch.device("cpu"), depth=3)
        frames = [pipe.submit() for _ in range(4)]             # more frames than buffers
        if rank == 0:
            assert all(f is not None for f in frames)
            np.save(os.path.join(result_dir, "pipe_%d.npy" % rank), frames[-1].numpy())
        else:
            assert all(f is None for f in frames)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,layout,world", [("pose_70x50_n250", "wh3", 2), ("small_64x48_n300", "hw3", 3)])
def test_strip_pipeline_equals_single_frame(tmp_path, name, layout, world):
    mp.spawn(_pipeline_worker, args=(world, _free_port(), name, layout, str(tmp_path)), nprocs=world, join=True)
    g = load_golden(name)
    full, _, _ = c_oracle.render(golden_preprocessed(g), int(g["width"]), int(g["height"]), int(g["tile"]))
    full = full if layout == "wh3" else full.transpose(1, 0, 2)
    assert np.array_equal(np.load(tmp_path / "pipe_0.npy"), full)


# ---- balanced (unequal) strips: point-to-point gather straight into place

def test_balanced_plan_minimises_the_largest_strip():
    plan = strips.balanced_plan([1, 1, 1, 1, 10, 1, 1, 1], 3)
    assert plan == [(0, 4), (4, 5), (5, 8)]
    assert strips.balanced_plan([5.0] * 8, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert strips.balanced_plan([3, 0, 0], 4) == [(0, 1), (1, 3), (3, 3), (3, 3)] or \
        strips.balanced_plan([3, 0, 0], 4)[0] == (0, 3)
    for cost, world in (([7, 2, 9, 4, 4, 1, 8, 3, 3, 6], 4), ([0, 0, 0, 5], 2), ([1], 3), ([2, 2], 5)):
        plan = strips.balanced_plan(cost, world)
        assert len(plan) == world and plan[0][0] == 0 and plan[-1][1] == len(cost)
        assert all(a[1] == b[0] for a, b in zip(plan, plan[1:])) and all(t0 <= t1 for t0, t1 in plan)
        best = max(sum(cost[a:b]) for a, b in plan)
        # no contiguous partition into `world` parts does better (brute force over the cut positions)
        import itertools
        n = len(cost)
        others = []
        for cuts in itertools.combinations_with_replacement(range(n + 1), world - 1):
            edges = (0,) + cuts + (n,)
            others.append(max(sum(cost[a:b]) for a, b in zip(edges, edges[1:])))
        assert best <= min(others) + 1e-9
    # peer_extra: every row costs a rank behind rank 0 that much more (the unhidden part of sending it): rank 0 takes more
    assert strips.balanced_plan([5.0] * 8, 2, peer_extra=5.0) == [(0, 6), (6, 8)]
    for cost, world, extra in (([7, 2, 9, 4, 4, 1, 8, 3, 3, 6], 4, 3.0), ([1, 1, 1, 1, 1, 1], 2, 1.0), ([4] * 8, 3, 2.0)):
        import itertools
        value = lambda pl: max(sum(cost[a:b]) + (extra * (b - a) if i else 0.0) for i, (a, b) in enumerate(pl))  # noqa: E731
        plan = strips.balanced_plan(cost, world, peer_extra=extra)
        n = len(cost)
        best = min(value(list(zip((0,) + cuts + (n,), cuts + (n,))))
                   for cuts in itertools.combinations_with_replacement(range(n + 1), world - 1))
        assert len(plan) == world and value(plan) <= best + 1e-9
    counts = torch.arange(12, dtype=torch.int32).reshape(4, 3)          # x-major: 4 tile columns x 3 tile rows
    assert strips.tile_row_costs(counts, 4, 3, lead_is_x=True, per_tile=0.0) == [3.0, 12.0, 21.0, 30.0]
    assert strips.tile_row_costs(counts, 3, 4, lead_is_x=False, per_tile=1.0) == [22.0, 26.0, 30.0]


def _planned_worker(rank, world, port, name, layout, plan, all_ranks, pipeline, result_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden(name)
        w, h, t = int(g["width"]), int(g["height"]), int(g["tile"])
        fn = _oracle_strip_renderer(golden_preprocessed(g), w, h, t, layout)
        if pipeline:
            pipe = strips.StripPipeline(fn, w, h, t, layout, torch.device("cpu"), depth=2, plan=plan)
            frame = [pipe.submit() for _ in range(3)][-1]
        else:
            frame = strips.render_sharded(fn, w, h, t, layout, torch.device("cpu"), all_ranks=all_ranks, plan=plan)
        if frame is not None:
            np.save(os.path.join(result_dir, "frame_%d.npy" % rank), frame.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,layout,plan,all_ranks,pipeline", [
    ("c1_256x256_n2000", "wh3", [(0, 3), (3, 4), (4, 15)], False, False),
    ("c1_256x256_n2000", "hw3", [(0, 11), (11, 15), (15, 15)], True, False),      # a rank without rows
    ("pose_70x50_n250", "wh3", [(0, 1), (1, 4)], False, True),
    # world size 8, what BASELINE config 5 runs: a balanced plan over the 15 tile rows of the C1 stand-in with ONE
    # rank that gets no rows (its strip is empty: it still takes part in the gather), plain and pipelined
    ("c1_256x256_n2000", "wh3", "balanced8", False, False),
    ("c1_256x256_n2000", "hw3", "balanced8", False, True),
])
def test_unequal_strips_equal_single_frame(tmp_path, name, layout, plan, all_ranks, pipeline):
    if plan == "balanced8":
        # what strips.balanced_plan gives for row costs with a heavy middle: 8 ranks, 15 tile rows, the last rank idle
        plan = strips.balanced_plan([1, 1, 1, 2, 9, 9, 9, 9, 9, 9, 2, 1, 1, 1, 1], 8)
        assert len(plan) == 8 and plan[0][0] == 0 and plan[-1][1] == 15
        if all(b > a for a, b in plan):      # (should the partition ever use all eight: idle the last rank by hand)
            plan = plan[:6] + [(plan[6][0], 15), (15, 15)]
        assert any(a == b for a, b in plan) and len({b - a for a, b in plan}) > 2
    world = len(plan)
    mp.spawn(_planned_worker, args=(world, _free_port(), name, layout, plan, all_ranks, pipeline, str(tmp_path)),
             nprocs=world, join=True)
    g = load_golden(name)
    full, _, _ = c_oracle.render(golden_preprocessed(g), int(g["width"]), int(g["height"]), int(g["tile"]))
    full = full if layout == "wh3" else full.transpose(1, 0, 2)
    for r in (range(world) if all_ranks else [0]):
        assert np.array_equal(np.load(tmp_path / ("frame_%d.npy" % r)), full)


def test_equal_strips_at_world_size_eight(tmp_path):
    """The default N > 1 path of bench.py (equal strips, ONE dist.gather) at the world size of BASELINE config 5:
    15 tile rows over 8 ranks = 2 rows each, the last rank gets one, and the never-rendered last tile row stays zero."""
    world, name, layout = 8, "c1_256x256_n2000", "wh3"
    mp.spawn(_worker, args=(world, _free_port(), name, layout, False, str(tmp_path)), nprocs=world, join=True)
    g = load_golden(name)
    full, _, _ = c_oracle.render(golden_preprocessed(g), int(g["width"]), int(g["height"]), int(g["tile"]))
    assert np.array_equal(np.load(tmp_path / "frame_0.npy"), full)
    assert not any((tmp_path / ("frame_%d.npy" % r)).exists() for r in range(1, world))


def test_strip_plans_are_validated():
    """A plan that overlaps, runs backwards or leaves the frame would silently put pixels in the wrong rows (or make
    rank 0's receive slices alias each other): render_sharded and StripPipeline refuse it."""
    g = load_golden("small_64x48_n300")
    w, h, t = int(g["width"]), int(g["height"]), int(g["tile"])
    fn = _oracle_strip_renderer(golden_preprocessed(g), w, h, t, "wh3")
    n_lead = strips.tiles_along(w, t)
    assert strips.check_plan([(0, n_lead)], n_lead, 1) == [(0, n_lead)]
    for bad, world in (([(0, 2), (1, 3)], 2), ([(2, 3), (0, 2)], 2), ([(0, n_lead + 1)], 1), ([(2, 1)], 1), ([(0, 1)], 2),
                       ([(-1, 1)], 1)):
        with pytest.raises(ValueError):
            strips.check_plan(bad, n_lead, world)
    with pytest.raises(ValueError):
        strips.render_sharded(fn, w, h, t, "wh3", torch.device("cpu"), plan=[(0, n_lead + 2)])
    with pytest.raises(ValueError):
        strips.StripPipeline(fn, w, h, t, "wh3", torch.device("cpu"), plan=[(1, 0)])
    # one process, a plan that does not start at tile row 0: the strip lands in ITS rows of the frame (round-2 advisor)
    full, _, _ = c_oracle.render(golden_preprocessed(g), w, h, t)
    pipe = strips.StripPipeline(fn, w, h, t, "wh3", torch.device("cpu"), plan=[(1, n_lead)])
    frame = pipe.submit().numpy()
    assert np.array_equal(frame[t:], full[t:]) and not frame[:t].any()
    part = strips.render_sharded(fn, w, h, t, "wh3", torch.device("cpu"), plan=[(1, n_lead)]).numpy()
    assert np.array_equal(part[t:], full[t:]) and not part[:t].any()


# ---- render_overlapped: the gather of a frame overlapped with its compositing (sub-strips sent as they finish)

def _overlap_worker(rank, world, port, name, layout, plan, parts, result_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden(name)
        w, h, t = int(g["width"]), int(g["height"]), int(g["tile"])
        base = _oracle_strip_renderer(golden_preprocessed(g), w, h, t, layout)
        seen = []

        def fn(window, out, origin, bounds):
            # the CPU stand-in renders the strip whole and returns None ("complete on return"); what is checked here is
            # the protocol: the bounds cut THIS rank's strip, ascending, into at most `parts` runs
            lo, hi = (window[0], window[1]) if layout == "wh3" else (window[2], window[3])
            assert bounds[0] == lo and bounds[-1] == hi and len(bounds) == parts + 1
            assert all(a <= b for a, b in zip(bounds, bounds[1:]))
            seen.append(tuple(bounds))
            base(window, out, origin)
            return None

        cache = {}
        for _ in range(2):                  # twice: the cached frame / strip buffers are reused
            frame = strips.render_overlapped(fn, w, h, t, layout, torch.device("cpu"), parts=parts, plan=plan, cache=cache)
        assert (frame is not None) == (rank == 0)
        if frame is not None:
            np.save(os.path.join(result_dir, "frame_%d.npy" % rank), frame.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,layout,plan,parts", [
    ("c1_256x256_n2000", "wh3", None, 4),                                  # 2 ranks, equal strips of 8 and 7 tile rows
    ("c1_256x256_n2000", "hw3", [(0, 9), (9, 10), (10, 15)], 4),           # a one-row strip: three of its parts are empty
    ("pose_70x50_n250", "wh3", [(0, 1), (1, 4)], 2),
    ("c1_256x256_n2000", "wh3", "balanced8", 3),                           # world 8, one rank without rows
    ("c1_256x256_n2000", "hw3", "balanced8", 1),                           # parts = 1: one message per peer
])
def test_overlapped_gather_equals_single_frame(tmp_path, name, layout, plan, parts):
    """strips.render_overlapped (round 4): every rank cuts its strip into sub-strips, peers send sub-strip j while j + 1
    is composited, rank 0 posts receive group j = sub-strip j of every peer before it renders its own strip.  On CPU
    ranks the renderer is the oracle and every part is ready at once -- the message protocol (K groups on rank 0, up
    to K sends per peer, empty parts skipped on both sides) is what runs -- and the frame must equal the
    one-process render bit for bit."""
    if plan == "balanced8":
        plan = strips.balanced_plan([1, 1, 1, 2, 9, 9, 9, 9, 9, 9, 2, 1, 1, 1, 1], 8)
        if all(b > a for a, b in plan):
            plan = plan[:6] + [(plan[6][0], 15), (15, 15)]
    world = 2 if plan is None else len(plan)
    mp.spawn(_overlap_worker, args=(world, _free_port(), name, layout, plan, parts, str(tmp_path)), nprocs=world, join=True)
    g = load_golden(name)
    full, _, _ = c_oracle.render(golden_preprocessed(g), int(g["width"]), int(g["height"]), int(g["tile"]))
    full = full if layout == "wh3" else full.transpose(1, 0, 2)
    assert np.array_equal(np.load(tmp_path / "frame_0.npy"), full)
    assert not any((tmp_path / ("frame_%d.npy" % r)).exists() for r in range(1, world))


def test_substrip_bounds():
    assert strips.substrip_bounds(0, 8, 4) == [0, 2, 4, 6, 8]
    assert strips.substrip_bounds(3, 10, 4) == [3, 5, 7, 9, 10]
    assert strips.substrip_bounds(9, 10, 4) == [9, 10, 10, 10, 10]          # fewer rows than parts: trailing parts empty
    assert strips.substrip_bounds(5, 5, 3) == [5, 5, 5, 5]
    assert strips.substrip_bounds(0, 30, 1) == [0, 30]
    g = load_golden("small_64x48_n300")
    w, h, t = int(g["width"]), int(g["height"]), int(g["tile"])
    base = _oracle_strip_renderer(golden_preprocessed(g), w, h, t, "wh3")
    frame = strips.render_overlapped(lambda win, out, org, b: base(win, out, org), w, h, t, "wh3", torch.device("cpu"))
    full, _, _ = c_oracle.render(golden_preprocessed(g), w, h, t)
    assert np.array_equal(frame.numpy(), full)                              # one process: no group needed
