import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
sc, scene = bench.build_scene(sys.argv[1] if len(sys.argv) > 1 else "c3", "cuda")
for split in (True, False, True, False):
    for _ in range(20):
        scene.render_image_hip(1, split_long_tiles=split)
    acc = []
    for _ in range(150):
        st = {}
        scene.render_image_hip(1, split_long_tiles=split, stats=st, timing=True)
        acc.append((st["stage_ms"]["blend"], st["stage_ms"]["total"]))
    a = np.median(np.asarray(acc), axis=0)
    print("split_long_tiles=%s: blend %.4f total %.4f" % (split, a[0], a[1]))
