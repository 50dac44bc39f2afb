"""Per-stage HIP-event times of one rank's strip for 1, 2, 4, 8 ranks (equal strips, or PLAN=balanced; the slowest of
three ranks probed), on ONE GPU:   python tools/attic/strip_probe.py [workload]
What a rank of an N-GPU run would spend per frame before the gather (timing mode: separate launches)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from intro_to_gaussian_splatting_amd import _ffi, strips
if os.environ.get("GSX_USE_TEST_LIB"):      # measurement knobs (GSX_*) from the environment
    _ffi.use_test_library()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
arrays, scene = bench.build_scene(wl, "cuda")
n, w, h, _ = bench.WORKLOADS[wl]
ntx, nty = strips.tiles_along(w, 16), strips.tiles_along(h, 16)
counts = torch.zeros(ntx * nty, dtype=torch.int32, device="cuda")
scene.render_image_hip(1, tile_counts=counts)
for world in [int(v) for v in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    plan = strips.balanced_plan(strips.tile_row_costs(counts, ntx, nty, lead_is_x=True), world)
    if os.environ.get("PLAN", "equal") == "equal":       # bench.py's default; PLAN=balanced: --balance
        plan = strips.strip_plan(ntx, world)[1]
    worst = None
    for r in (0, world // 2, world - 1):
        win = (plan[r][0], plan[r][1], 0, nty)
        for _ in range(3):
            scene.render_image_hip(1, tile_window=win)
        acc = {}
        reps = 10
        for _ in range(reps):
            st = {}
            scene.render_image_hip(1, tile_window=win, stats=st, timing=True)
            for k, v in st["stage_ms"].items():
                acc[k] = acc.get(k, 0.0) + v / reps
        if worst is None or acc["total"] > worst[1]["total"]:
            worst = (r, acc)
    print(wl, "world", world, "slowest of ranks probed:", worst[0], {k: round(v, 3) for k, v in worst[1].items()})
