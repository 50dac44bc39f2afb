"""How evenly does the compositing launch load the SIMDs?  One frame of a bench workload on libgsx_test.so with the blend
probe on: per workgroup start / end (wall clock, 10 ns) and where it ran (XCC_ID, HW_ID) -> per SIMD: waves, list entries,
time of its last wave's end relative to the launch.   python tools/attic/simd_balance.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intro_to_gaussian_splatting_amd import _ffi
_ffi.use_test_library()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc, scene = bench.build_scene(wl, "cuda")
lib = _ffi.load()
for _ in range(4):
    scene.render_image_hip(1)
torch.cuda.synchronize()
half = 1 << 17              # gsx_blend.hip: kProbeSecond
buf = torch.zeros((4 * half, 4), dtype=torch.int32, device="cuda")
lib.gsx_debug_set_blend_probe(buf.data_ptr())
st = {}
scene.render_image_hip(1, stats=st, timing=True)
torch.cuda.synchronize()
lib.gsx_debug_set_blend_probe(None)
d = buf.cpu().numpy().view(np.uint32)
first = np.nonzero(d[:half, 2])[0]
first = first[d[first, 1] < 0x40000000]                 # tile workgroups of the main kernel (not helpers)
sec = d[half + first]
start, end, hw, xcc = sec[:, 3].astype(np.int64), sec[:, 1].astype(np.int64), sec[:, 2] & 0xFFFF, sec[:, 2] >> 16
t0 = start.min()
start, end = (start - t0) * 0.01, (end - t0) * 0.01   # us
span = end.max()
simd = (xcc.astype(np.int64) << 16) | (hw & 0xFF30)    # xcc, se/sh/cu, simd (wave and pipe ids masked out)
lens = d[first, 2].astype(np.int64)
print(wl, "blend stage %.3f ms; %d tile workgroups, launch span %.1f us, %d SIMDs seen" % (
    st["stage_ms"]["blend"], len(first), span, len(np.unique(simd))))
order = np.argsort(simd)
keys, idx = np.unique(simd[order], return_index=True)
last = np.maximum.reduceat(end[order], idx)
load = np.add.reduceat(lens[order], idx)
cnt = np.diff(np.append(idx, len(order)))
q = np.percentile(last / span, [0, 5, 25, 50, 75, 95, 100])
print("SIMD's last wave ends at (share of the span): min %.2f p5 %.2f p25 %.2f p50 %.2f p75 %.2f p95 %.2f max %.2f" % tuple(q))
print("mean idle tail per SIMD %.1f us (%.1f %% of the span); waves per SIMD min %d mean %.2f max %d" % (
    np.mean(span - last), 100 * np.mean(span - last) / span, cnt.min(), cnt.mean(), cnt.max()))
print("list entries per SIMD: mean %.0f, max/mean %.3f; corr(end, entries) %.2f" % (load.mean(), load.max() / load.mean(),
                                                                                 np.corrcoef(last, load)[0, 1]))
for x in range(8):
    m = (keys >> 16) == x
    print("  XCD %d: SIMDs %d, entries %.3g, last end %.1f us, mean end %.1f us" % (x, m.sum(), load[m].sum(), last[m].max(), last[m].mean()))
print("start of the last workgroup to start: %.1f us; workgroups starting after 10 us: %d" % (start.max(), int((start > 10).sum())))
# shader clock per XCD: a workgroup's s_memtime cycles over its wall-clock duration
cyc = d[first, 0].astype(np.float64)
dur = np.maximum(end - start, 0.01)
staged = (d[first, 3] & 0x7FFFFFFF).astype(np.float64)
for x in range(8):
    m = xcc == x
    print("  XCD %d: %.0f MHz while running; %.1f cycles and %.4f us per staged record-wave; staged / listed %.3f" % (
        x, cyc[m].sum() / dur[m].sum(), cyc[m].sum() / staged[m].sum(), dur[m].sum() / staged[m].sum(), staged[m].sum() / lens[m].sum()))
if os.environ.get("PLACEMENT"):
    # where do consecutive workgroups of one XCD land?  block ids (relative to the first tile block) vs SIMD
    b0 = first.min()
    rel = first - b0
    for x in (0, 1):
        m = np.nonzero((xcc == x))[0]
        m = m[np.argsort(rel[m])]
        print("XCD", x, "first blocks: (i, se, sh?, cu, simd, wave, start us)")
        for j in m[:48]:
            h = int(hw[j])
            print("   i %4d  se %d cu %2d simd %d wave %d  start %.2f  len %d" % (rel[j] >> 3, (h >> 13) & 7, (h >> 8) & 15, (h >> 4) & 3, h & 15, start[j], lens[j]))
        # does i and i + 1024 share a SIMD?  does i and i + 128?
        key = {int(rel[j] >> 3): int(simd[j]) for j in m}
        for step in (32, 64, 128, 256, 512, 1024):
            same = [key[i] == key[i + step] for i in key if i + step in key and i < 512]
            print("   share of i < 512 with SIMD(i) == SIMD(i + %d): %.2f" % (step, np.mean(same) if same else -1))
# what the hand-out intends: workgroup i of an XCD on slot i % 128 -> entries per slot
b0 = int(first.min()) & ~7         # (helpers -- and, unless they come last, the rank groups -- precede the tile workgroups)
iw = (first - b0) >> 3
for x in range(8):
    m = xcc == x
    slot = np.bincount(iw[m] % 128, weights=staged[m] + 5 * np.ceil(lens[m] / 64.0), minlength=128)   # (the schedule's cost model)
    rounds = [lens[m][(iw[m] >> 7) == r].sum() for r in range(9)]
    actual = load[(keys >> 16) == x]
    print("  XCD %d intended cost per slot: max/mean %.3f min/mean %.3f; as run per SIMD: max/mean %.3f min/mean %.3f; tiles %d" % (
        x, slot.max() / slot.mean(), slot.min() / slot.mean(), actual.max() / actual.mean(), actual.min() / actual.mean(), m.sum()))

cost = staged + 5 * np.ceil(lens / 64.0)
srt = np.sort(cost)[::-1]
print("tile cost: mean slot load %.0f (all tiles / 1024); largest tiles %s; p99 %.0f p90 %.0f p50 %.0f; long-tile helpers seen: %d" % (
    cost.sum() / 1024, srt[:8].astype(int), np.percentile(cost, 99), np.percentile(cost, 90), np.percentile(cost, 50),
    int(((d[:, 1] & 0x40000000) != 0).sum())))
durs = end - start
o = np.argsort(durs)[::-1][:8]
print("longest-running tile workgroups: us %s cost %s listed %s" % (durs[o].round(0), cost[o].astype(int), lens[o]))
if os.environ.get("STARTS"):
    print("start time percentiles (us): " + " ".join("p%d %.1f" % (p, np.percentile(start, p)) for p in (10, 50, 75, 80, 85, 90, 95, 99, 100)))
    late = start > 10
    print("late starters: %d; their block ids (relative): min %d median %d max %d of %d; corr(block id, start) %.2f" % (
        late.sum(), (first[late] - first.min()).min() if late.any() else -1, np.median(first[late] - first.min()) if late.any() else -1,
        (first[late] - first.min()).max() if late.any() else -1, first.max() - first.min(), np.corrcoef(first, start)[0, 1]))
    # when a late starter starts, has some wave of its SIMD just ended?
    ends_by_simd = {}
    for s_, e_ in zip(simd, end):
        ends_by_simd.setdefault(int(s_), []).append(e_)
    gaps = []
    for s_, st_ in zip(simd[late], start[late]):
        prev = [e_ for e_ in ends_by_simd[int(s_)] if e_ <= st_ + 0.05]
        gaps.append(st_ - max(prev) if prev else np.nan)
    gaps = np.asarray(gaps)
    print("late starters whose SIMD had a wave end before their start: %d of %d; median gap %.2f us" % (
        np.isfinite(gaps).sum(), len(gaps), np.nanmedian(gaps) if np.isfinite(gaps).any() else -1))
    print("durations of workgroups ending before 40 us: %d, (all tiles: median duration %.1f us)" % ((end < 40).sum(), np.median(end - start)))
