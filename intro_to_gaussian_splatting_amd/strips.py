"""Row-strip sharding of one frame across the GPUs of a node.

Tiles are independent once stage 1 is done, so the frame shards naturally: every rank keeps a
full replica of the Gaussians (280 MB at 5M Gaussians -- nothing against 288 GB of HBM3E), runs
the cheap projection itself, and bins / sorts / composites only the tiles of its own strip.
A strip is a run of whole tile rows of the OUTPUT TENSOR's leading axis (x for the reference's
(W,H,3) layout, y for (H,W,3)), so it is one contiguous block of the frame.  The only exchange
is the final frame gather: one collective, strips of equal (padded) size written straight into
their place in the frame buffer.  Over xGMI every strip travels its own point-to-point link to
the root, so the gather is per-link bound (about 12 MB / 153 GB/s at 4K), not ring bound.

The reference has no multi-GPU code at all; correctness is defined as:
the gathered frame equals the single-GPU frame bit for bit.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def tiles_along(extent: int, tile: int, semantics: str = "ref_cpu") -> int:
    """Tiles along an axis.  "ref_cpu": what the CPU-semantics renderer produces,
    ``len(range(0, extent - tile, tile))`` (splat/gaussian_scene.py:208,214: the last row / column is
    never rendered); "ref_cuda" / "std_3dgs": the whole frame, ``ceil(extent / tile)``."""
    if semantics != "ref_cpu":
        return -(-extent // tile)
    return max(0, -(-(extent - tile) // tile)) if extent > tile else 0


def strip_plan(n_tiles: int, world_size: int) -> Tuple[int, List[Tuple[int, int]]]:
    """Equal strips of ``ceil(n_tiles / world_size)`` tile rows; trailing ranks may get fewer or
    none.  Returns (tiles_per_strip, [(t0, t1) per rank])."""
    per = -(-n_tiles // world_size) if n_tiles > 0 else 0
    plan = []
    for r in range(world_size):
        t0 = min(r * per, n_tiles)
        plan.append((t0, min(t0 + per, n_tiles)))
    return per, plan


def balanced_plan(row_cost: Sequence[float], world_size: int, peer_extra: float = 0.0) -> List[Tuple[int, int]]:
    """Contiguous strips of tile rows whose LARGEST cost is as small as possible ("optional
    balance by prefix-sum of per-tile-row D").  ``row_cost[t]`` = cost of tile row ``t`` of the leading axis,
    e.g. ``tile_row_costs(tile_counts)``.  Returns [(t0, t1) per rank]; trailing ranks may get no rows.
    ``peer_extra``: what every row costs a rank OTHER than rank 0 on top -- the part of sending it to rank 0 that the
    compositing does not hide (rank 0 assembles the frame and sends nothing, so it can take more rows: with two ranks
    and a 4K float32 frame the single link is the bottleneck and an even split gains nothing).
    Exact for the cost given: binary search on the bottleneck, greedy fill (the classic linear partition)."""
    cost = [max(0.0, float(c)) for c in row_cost]
    n = len(cost)
    extra = max(0.0, float(peer_extra))
    if n == 0 or world_size <= 0:
        return [(0, 0)] * max(world_size, 0)

    def cut(limit: float):
        plan, t0, acc = [], 0, 0.0
        for t in range(n):
            c = cost[t] + (extra if plan else 0.0)        # (plan non-empty: this row goes to a rank behind rank 0)
            if acc + c > limit and t > t0:
                plan.append((t0, t))
                t0, acc = t, 0.0
                c = cost[t] + extra
            acc += c
        plan.append((t0, n))
        return plan

    lo, hi = max(cost) + extra, sum(cost) + extra * n
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if len(cut(mid)) <= world_size:
            hi = mid
        else:
            lo = mid
    plan = cut(hi)
    return plan + [(n, n)] * (world_size - len(plan))


def check_plan(plan: Sequence[Tuple[int, int]], n_lead: int, world_size: int) -> List[Tuple[int, int]]:
    """A strip plan is one (t0, t1) per rank, in rank order, each inside [0, n_lead], no two overlapping
    (``t0 <= t1 <= next t0``).  Anything else would silently yield a wrong frame (two ranks writing the same rows,
    rank 0's receive slices aliasing each other), so it raises."""
    if len(plan) != world_size:
        raise ValueError("the strip plan has %d entries for %d ranks" % (len(plan), world_size))
    edge = 0
    out = []
    for r, (t0, t1) in enumerate(plan):
        t0, t1 = int(t0), int(t1)
        if not (0 <= t0 <= t1 <= n_lead):
            raise ValueError("strip %d = [%d, %d) is not inside the %d tile rows of the frame" % (r, t0, t1, n_lead))
        if t1 > t0:
            if t0 < edge:
                raise ValueError("strip %d = [%d, %d) overlaps or precedes the strip before it (ends at %d)" % (r, t0, t1, edge))
            edge = t1
        out.append((t0, t1))
    return out


def tile_row_costs(tile_counts, n_lead: int, n_other: int, lead_is_x: bool = True, per_tile: float = 8.0) -> List[float]:
    """Cost of every tile row of the leading axis from the per-tile list lengths a frame reported
    (GsxParams.tile_counts, x-major: index = tx * n_tiles_y + ty): the compositing and the pair sort cost
    one unit per (Gaussian, tile) pair, every tile a small constant more (``per_tile`` pairs' worth)."""
    c = tile_counts.reshape(-1).to("cpu").to(torch.float64)
    grid = c.reshape(n_lead, n_other) if lead_is_x else c.reshape(n_other, n_lead).t()
    return (grid.sum(dim=1) + per_tile * n_other).tolist()


def _gather_strips(strip, frame, pixel_ranges, rank, world, group, all_ranks) -> None:
    """Unequal strips: every rank sends its strip straight into its place in rank 0's frame (grouped
    point-to-point operations: ncclSend / ncclRecv pairs under RCCL, each over its own xGMI link)."""
    ops = []
    if rank == 0:
        for r in range(1, world):
            a, b = pixel_ranges[r]
            if b > a:
                ops.append(dist.P2POp(dist.irecv, frame[a:b], r, group))
    elif strip.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, strip, 0, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if all_ranks:
        dist.broadcast(frame, src=0, group=group)


def render_sharded(render_fn: Callable[[Tuple[int, int, int, int], torch.Tensor, Tuple[int, int]], None],
                   width: int, height: int, tile: int, layout: str, device: torch.device,
                   group: Optional[dist.ProcessGroup] = None, all_ranks: bool = False,
                   cache: Optional[dict] = None, semantics: str = "ref_cpu",
                   plan: Optional[List[Tuple[int, int]]] = None, collective_with_one_rank: bool = False) -> Optional[torch.Tensor]:
    """Renders this rank's strip with ``render_fn(tile_window, out_strip, out_origin)`` and gathers
    the frame on rank 0 (or on every rank with ``all_ranks``).

    ``render_fn`` must fully write ``out_strip`` (zeros where nothing is rendered), exactly what
    ``GaussianScene.render_image_hip(..., tile_window=, out=, out_origin=)`` does.
    Returns the frame ((W,H,3) for "wh3", (H,W,3) for "hw3") or None on non-root ranks.
    ``cache`` (a dict owned by the caller) lets consecutive frames reuse the frame / strip buffers
    instead of allocating them per call; the returned frame is then overwritten by the next call.
    ``semantics`` only decides how many tile rows exist (see ``tiles_along``); with partial edge tiles
    the last strip's buffer extends past the frame and the surplus rows are cut off after the gather.
    ``plan`` ([(t0, t1) per rank], e.g. from ``balanced_plan``; the same on every rank) replaces the equal
    strips: each rank then renders straight into (a buffer the size of) its own rows and the strips travel
    point to point instead of through one equal-sized gather.
    ``collective_with_one_rank``: a process group of ONE rank still issues the gather (bench.py --dist-preflight: the
    collective call meets the real backend on a single GPU); by default a lone rank returns its strip as the frame.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lead, other = (width, height) if layout == "wh3" else (height, width)
    n_lead = tiles_along(lead, tile, semantics)
    n_other = tiles_along(other, tile, semantics)
    if plan is not None:
        return _render_planned(render_fn, check_plan(plan, n_lead, world), lead, other, n_other, tile, layout, device,
                               group, all_ranks, cache, rank, world)
    per, plan = strip_plan(n_lead, world)
    rows = per * tile                                  # strip extent in pixels, tile aligned
    t0, t1 = plan[rank]
    window = (t0, t1, 0, n_other) if layout == "wh3" else (0, n_other, t0, t1)
    origin = (rank * rows, 0) if layout == "wh3" else (0, rank * rows)
    covered = world * rows                             # pixels owned by strips; the rest stays zero
    is_dst = all_ranks or rank == 0 or world == 1
    if rows == 0:
        return torch.zeros((lead, other, 3), dtype=torch.float32, device=device) if is_dst else None

    def buffer(name, shape):
        if cache is None:
            return torch.empty(shape, dtype=torch.float32, device=device)
        key = (name, shape, str(device))
        if key not in cache:
            cache[key] = torch.empty(shape, dtype=torch.float32, device=device)
        return cache[key]

    frame = None
    if is_dst:
        frame = buffer("frame", (max(covered, lead), other, 3))
        if covered < lead:
            frame[covered:].zero_()                    # never-rendered last tile row(s)
    strip = frame[rank * rows:(rank + 1) * rows] if is_dst else buffer("strip", (rows, other, 3))
    render_fn(window, strip, origin)
    if world == 1 and not (collective_with_one_rank and dist.is_initialized()):
        return frame[:lead]
    if all_ranks:
        dist.all_gather_into_tensor(frame[:covered], strip, group=group)
        return frame[:lead]
    if rank == 0:
        dist.gather(strip, [frame[r * rows:(r + 1) * rows] for r in range(world)], dst=0, group=group)
        return frame[:lead]
    dist.gather(strip, None, dst=0, group=group)
    return None


def _render_planned(render_fn, plan, lead, other, n_other, tile, layout, device, group, all_ranks, cache, rank, world):
    if len(plan) != world:
        raise ValueError("the strip plan has %d entries for %d ranks" % (len(plan), world))
    pixel_ranges = [(min(t0 * tile, lead), min(t1 * tile, lead)) for t0, t1 in plan]
    is_dst = all_ranks or rank == 0 or world == 1

    def buffer(name, shape):
        if cache is None:
            return torch.empty(shape, dtype=torch.float32, device=device)
        key = (name, shape, str(device))
        if key not in cache:
            cache[key] = torch.empty(shape, dtype=torch.float32, device=device)
        return cache[key]

    frame = buffer("frame", (lead, other, 3)) if is_dst else None
    if is_dst:      # rows no strip owns (REF_CPU: the never-rendered last tile row) stay zero
        covered = sorted(r for r in pixel_ranges if r[1] > r[0])
        edge = 0
        for a, b in covered:
            if a > edge:
                frame[edge:a].zero_()
            edge = max(edge, b)
        if edge < lead:
            frame[edge:].zero_()
    a, b = pixel_ranges[rank]
    t0, t1 = plan[rank]
    strip = frame[a:b] if is_dst else buffer("strip", (b - a, other, 3))
    if b > a:
        window = (t0, t1, 0, n_other) if layout == "wh3" else (0, n_other, t0, t1)
        origin = (a, 0) if layout == "wh3" else (0, a)
        render_fn(window, strip, origin)
    if world > 1:
        _gather_strips(strip, frame, pixel_ranges, rank, world, group, all_ranks)
    return frame


def substrip_bounds(t0: int, t1: int, parts: int) -> List[int]:
    """``parts`` + 1 ascending tile coordinates that cut the strip [t0, t1) into ``parts`` runs of whole tile rows of
    (nearly) equal size; a strip of fewer rows than parts leaves the trailing runs empty."""
    n = max(0, t1 - t0)
    per = -(-n // parts) if n > 0 else 0
    return [min(t0 + k * per, t1) for k in range(parts)] + [t1]


def render_overlapped(render_fn, width: int, height: int, tile: int, layout: str, device: torch.device, parts: int = 4,
                      group: Optional[dist.ProcessGroup] = None, cache: Optional[dict] = None, semantics: str = "ref_cpu",
                      plan: Optional[List[Tuple[int, int]]] = None) -> Optional[torch.Tensor]:
    """``render_sharded`` with the frame gather OVERLAPPED with the compositing inside one frame (round-3 verdict: the
    99.5 MB float32 frame of a 4K render funnels into rank 0 over one xGMI link per peer -- 0.23 ms of an 0.6 ms
    8-GPU frame, and all of a 2-GPU frame's gain -- and nothing hid it).  Every rank cuts its strip into ``parts``
    sub-strips (``substrip_bounds``); projection, depth order and binning run once per strip, the compositing launch
    once per sub-strip (GsxParams.n_substrips), and

      * a peer sends sub-strip j -- straight into its place in rank 0's frame -- from a communication stream that waits
        for the event the library recorded behind part j, while part j + 1 is composited on the render stream;
      * rank 0 posts ALL its receives before it renders its own strip into the frame: receive group j holds sub-strip j
        of every peer (one grouped ncclRecv set under RCCL, every peer on its own link), so the transfers of the first
        parts land while rank 0 -- which sends nothing -- is still compositing.

    ``render_fn(window, out_strip, out_origin, bounds)`` renders the strip in the parts ``bounds`` describes and returns
    one ready-marker per part -- objects with ``wait_on(stream)`` (``_hip.Event``) -- or None when the strip is simply
    complete on return (CPU renderers in the gloo tests).  Only rank 0 gets the frame (no ``all_ranks``).  Same
    strips, same pixels: the assembled frame equals the single-GPU frame bit for bit."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lead, other = (width, height) if layout == "wh3" else (height, width)
    n_lead, n_other = tiles_along(lead, tile, semantics), tiles_along(other, tile, semantics)
    plan = check_plan(plan, n_lead, world) if plan is not None else strip_plan(n_lead, world)[1]
    parts = max(1, min(int(parts), 16))
    px = lambda t: min(t * tile, lead)                    # noqa: E731  (partial edge tiles: the last strip ends with the frame)
    on_gpu = torch.device(device).type == "cuda"

    def buffer(name, shape):
        if cache is None:
            return torch.empty(shape, dtype=torch.float32, device=device)
        key = (name, shape, str(device))
        if key not in cache:
            cache[key] = torch.empty(shape, dtype=torch.float32, device=device)
        return cache[key]

    def comm_stream():
        if not on_gpu:
            return None
        if cache is None:
            return torch.cuda.Stream(device)
        return cache.setdefault(("comm_stream", str(device)), torch.cuda.Stream(device))

    t0, t1 = plan[rank]
    a, b = px(t0), px(t1)
    bounds = substrip_bounds(t0, t1, parts)
    window = (t0, t1, 0, n_other) if layout == "wh3" else (0, n_other, t0, t1)
    works = []
    if rank == 0:
        frame = buffer("frame", (lead, other, 3))
        edge = 0
        for ra, rb in sorted((px(p0), px(p1)) for p0, p1 in plan if p1 > p0):     # rows no strip owns stay zero
            if ra > edge:
                frame[edge:ra].zero_()
            edge = max(edge, rb)
        if edge < lead:
            frame[edge:].zero_()
        if world > 1:
            cs = comm_stream()
            if cs is not None:
                cs.wait_stream(torch.cuda.current_stream(device))       # (the frame buffer and its zero fill come first)
            ctx = torch.cuda.stream(cs) if cs is not None else _Null()
            with ctx:
                for j in range(parts):
                    ops = []
                    for r in range(1, world):
                        rb_ = substrip_bounds(plan[r][0], plan[r][1], parts)
                        pa, pb = px(rb_[j]), px(rb_[j + 1])
                        if pb > pa:
                            ops.append(dist.P2POp(dist.irecv, frame[pa:pb], r, group))
                    if ops:
                        works += dist.batch_isend_irecv(ops)
        if b > a:
            render_fn(window, frame[a:b], (a, 0) if layout == "wh3" else (0, a), bounds)
        for w in works:
            w.wait()
        return frame
    strip = buffer("strip", (max(b - a, 1), other, 3))[:b - a]
    if b > a:
        ready = render_fn(window, strip, (a, 0) if layout == "wh3" else (0, a), bounds)
        cs = comm_stream()
        for j in range(parts):
            pa, pb = px(bounds[j]) - a, px(bounds[j + 1]) - a
            if pb <= pa:
                continue
            if cs is not None:
                # a marker per part -- or none at all (a renderer that is complete on return) or fewer than parts (a
                # one-part strip composited in one launch records nothing): then the strip's own stream is the marker
                if ready is not None and j < len(ready):
                    ready[j].wait_on(cs)
                else:
                    cs.wait_stream(torch.cuda.current_stream(device))
            ctx = torch.cuda.stream(cs) if cs is not None else _Null()
            with ctx:
                works += dist.batch_isend_irecv([dist.P2POp(dist.isend, strip[pa:pb], 0, group)])
    for w in works:
        w.wait()            # (GPU: the current stream waits for the sends -- the strip buffer may be rendered into again)
    return None


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class StripPipeline:
    """Consecutive frames of one camera, sharded as strips, with up to ``depth`` frames in flight.

    Frame i is rendered on side stream ``i % depth`` into its own strip buffer, so the latency-bound
    binning kernels of one frame overlap the compositing and the gather of the previous ones (the
    same trick the single-GPU benchmark plays with whole frames).  The gathers are all issued from
    the caller's stream, in frame order -- one collective stream, the same order on every rank --
    after waiting for the event that marks the strip as rendered; a strip buffer is handed back to
    its side stream only after the gather that read it.  On a CPU device (gloo tests) the pipeline
    degenerates to ``render_sharded`` one frame at a time.

    ``submit()`` enqueues one frame and returns the frame tensor on rank 0 (None elsewhere); the
    tensor is the pipeline's own buffer and is overwritten by the next gather.
    """

    def __init__(self, render_fn, width: int, height: int, tile: int, layout: str, device: torch.device,
                 depth: int = 3, group: Optional[dist.ProcessGroup] = None, semantics: str = "ref_cpu",
                 plan: Optional[List[Tuple[int, int]]] = None) -> None:
        self.render_fn, self.group, self.device = render_fn, group, torch.device(device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        lead, other = (width, height) if layout == "wh3" else (height, width)
        n_lead, n_other = tiles_along(lead, tile, semantics), tiles_along(other, tile, semantics)
        self.pixel_ranges = None
        if plan is not None:      # unequal strips (balanced_plan): point-to-point gather, strips of their own size
            plan = check_plan(plan, n_lead, self.world)
            self.pixel_ranges = [(min(a * tile, lead), min(b * tile, lead)) for a, b in plan]
            t0, t1 = plan[self.rank]
            self.rows, self.lead = self.pixel_ranges[self.rank][1] - self.pixel_ranges[self.rank][0], lead
            self.origin = (self.pixel_ranges[self.rank][0], 0) if layout == "wh3" else (0, self.pixel_ranges[self.rank][0])
        else:
            per, eq = strip_plan(n_lead, self.world)
            self.rows, self.lead = per * tile, lead
            t0, t1 = eq[self.rank]
            self.origin = (self.rank * self.rows, 0) if layout == "wh3" else (0, self.rank * self.rows)
        self.window = (t0, t1, 0, n_other) if layout == "wh3" else (0, n_other, t0, t1)
        covered = self.world * self.rows if plan is None else lead
        self.on_gpu = self.device.type == "cuda"
        self.depth = max(1, depth) if self.on_gpu else 1
        f32 = dict(dtype=torch.float32, device=self.device)
        self.strips = [torch.zeros((max(self.rows, 1), other, 3), **f32) for _ in range(self.depth)]
        self.frame = torch.zeros((max(covered, lead, 1), other, 3), **f32) if self.rank == 0 else None
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.depth)] if self.on_gpu else []
        self.released = [None] * self.depth      # event on the caller's stream: the gather has read strip k
        self.count = 0

    def submit(self) -> Optional[torch.Tensor]:
        k = self.count % self.depth
        self.count += 1
        strip = self.strips[k][:self.rows]
        planned = self.pixel_ranges is not None
        if self.rows > 0:
            if self.on_gpu:
                main = torch.cuda.current_stream(self.device)
                side = self.streams[k]
                if self.released[k] is not None:
                    side.wait_event(self.released[k])
                else:
                    side.wait_stream(main)            # first use: buffers were created on the caller's stream
                with torch.cuda.stream(side):
                    self.render_fn(self.window, strip, self.origin)
                main.wait_event(side.record_event())
            else:
                self.render_fn(self.window, strip, self.origin)
        if self.rows > 0 or (planned and self.world > 1):
            if self.world == 1:
                a = self.pixel_ranges[0][0] if planned else 0      # a plan need not start at tile row 0
                self.frame[a:a + self.rows].copy_(strip)
            elif planned:                              # unequal strips: every rank takes part, also with no rows
                if self.rank == 0 and self.rows > 0:
                    a, b = self.pixel_ranges[0]
                    self.frame[a:b].copy_(strip)
                _gather_strips(strip, self.frame, self.pixel_ranges, self.rank, self.world, self.group, False)
            elif self.rank == 0:
                dist.gather(strip, [self.frame[r * self.rows:(r + 1) * self.rows] for r in range(self.world)], dst=0,
                            group=self.group)
            else:
                dist.gather(strip, None, dst=0, group=self.group)
            if self.on_gpu:
                self.released[k] = torch.cuda.current_stream(self.device).record_event()
        return self.frame[:self.lead] if self.rank == 0 else None
