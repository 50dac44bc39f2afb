"""Frame after frame of ONE view: is the warm frame time a fixed point?

    python tools/frame_sequence.py [--workload c3_clustered] [--frames 80]

Replays the captured frame (one hipGraph, hints fed back from frame to frame) and prints, per replay, the hipEvent time
and the words of the hints header that steer the next frame (GsxParams.hints, csrc/gsx_plan.h: kHintLongPct = the share of
a SIMD's load from which a tile is split over four helper waves; kHintXcdCost = what each XCD's tiles cost).  A view
whose time wanders although nothing changes is a feedback loop that has not settled."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3_clustered")
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--settle", type=int, default=200)
    ap.add_argument("--test-lib", action="store_true", help="libgsx_test.so: the GSX_* knobs of the environment apply")
    ap.add_argument("--quiet", action="store_true", help="the summary line only")
    a = ap.parse_args()
    if a.test_lib:
        from intro_to_gaussian_splatting_amd import _ffi
        _ffi.use_test_library()
    dev = "cuda:0"
    sc, scene = bench.build_scene(a.workload, dev)
    tile, layout = 16, "wh3"
    scene.render_image_hip(1, tile_size=tile, layout=layout)
    one = scene.capture_frame(1, tile_size=tile, layout=layout)
    for _ in range(a.settle):
        one.replay()
    torch.cuda.synchronize()
    rows, lens = [], []
    cam = scene.images[1].gsx_camera()
    ntiles = ((cam.width + tile - 1) // tile) * ((cam.height + tile - 1) // tile)
    LENS_OFFSET = (64 + 256 + 2048) * 4      # csrc/gsx_plan.h: hints_layout
    for k in range(a.frames):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        one.replay()
        e1.record()
        torch.cuda.synchronize()
        hdr = one._hints[:256].cpu().numpy().view(np.uint32)
        rows.append((e0.elapsed_time(e1), int(hdr[4]), [int(v) for v in hdr[16:24]]))
        if k >= a.frames - 4:
            lens.append(one._hints[LENS_OFFSET:LENS_OFFSET + 4 * ntiles].cpu().numpy().view(np.uint32).copy())
    one.confirm()
    for k, (ms, pct, xc) in enumerate(rows if not a.quiet else []):
        print("frame %3d  %.4f ms  long_pct %4d  xcd cost sum %9d  max/mean %.3f" % (k, ms, pct, sum(xc), max(xc) / (sum(xc) / 8.0 + 1e-9)))
    t = np.asarray([r[0] for r in rows])
    print("%s: median %.4f  min %.4f  max %.4f  sd %.4f   distinct long_pct values: %s  long tiles %s" % (
        a.workload, np.median(t), t.min(), t.max(), t.std(), sorted({r[1] for r in rows}), [int((l >> 31).sum()) for l in lens]))
    if a.quiet:
        return 0
    # tiles whose cost word (bit 31: composited by four helper waves) differs between consecutive frames
    for i in range(1, len(lens)):
        d = np.nonzero(lens[i] != lens[i - 1])[0]
        print("frames %d -> %d: %d tiles changed their cost word; long tiles %d -> %d" % (
            a.frames - len(lens) + i - 1, a.frames - len(lens) + i, d.size, int((lens[i - 1] >> 31).sum()), int((lens[i] >> 31).sum())))
        for t in d[:12]:
            print("    tile %5d (column %3d, row %2d): %s %6d -> %s %6d" % (
                t, t // ((cam.height + tile - 1) // tile), t % ((cam.height + tile - 1) // tile),
                "long" if lens[i - 1][t] >> 31 else "one ", lens[i - 1][t] & 0x7FFFFFFF, "long" if lens[i][t] >> 31 else "one ", lens[i][t] & 0x7FFFFFFF))
    return 0


if __name__ == "__main__":
    sys.exit(main())
