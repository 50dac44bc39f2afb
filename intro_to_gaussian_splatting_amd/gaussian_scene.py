"""``GaussianScene``: the reference's render surface (splat/gaussian_scene.py:25-285) on MI355X.

Host code stays Python and PyTorch-ROCm tensors hold the Gaussian parameters; every stage of
the hot path -- projection, depth ordering, tile binning, compositing -- runs in libgsx.so (HIP,
gfx950) through the C ABI of include/gsx.h.  There is no CPU path in this module: without the
library or without a GPU the render methods raise.

Surface kept from the reference:
  ``GaussianScene(colmap_path, gaussians)``, ``.images[idx]``, ``.gaussians``,
  ``.preprocess(idx) -> PreprocessedScene``            (gaussian_scene.py:70-144)
  ``.render_image(idx, tile_size=16) -> (W,H,3)``      (gaussian_scene.py:200-238, CPU semantics; host tensor)
  ``.render_points_image(idx)``                        (gaussian_scene.py:44-51)
Added: ``.render_image_hip`` (explicit layout / tile window / stats), ``.render_preprocessed``
(the argument list of the reference's native ``render_image``, splat/c/render.cu:90-101).
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _ffi
from .colmap import read_camera_file, read_image_file
from .gaussians import Gaussians
from .image import GaussianImage
from .schema import PreprocessedScene


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _check_f32(name: str, t: torch.Tensor, device: torch.device) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if t.device != device:
        raise ValueError("%s is on %s, expected %s" % (name, t.device, device))
    return t if t.is_contiguous() else t.contiguous()


class _Lru(OrderedDict):
    """A dict that forgets its least recently used entry beyond ``limit`` (device buffers keyed by stream handles or
    by views: a viewer that keeps creating streams, a sweep over hundreds of cameras or a strip plan that moves with
    every rebalance would otherwise grow them for the life of the process)."""

    def __init__(self, limit: int) -> None:
        super().__init__()
        self.limit = int(limit)

    def lookup(self, key):
        val = self.get(key)
        if val is not None:
            self.move_to_end(key)
        return val

    def store(self, key, val):
        self[key] = val
        self.move_to_end(key)
        while len(self) > self.limit:
            self.popitem(last=False)      # (freed on its own stream: torch's allocator orders the reuse behind its last kernel)
        return val


class _Workspace:
    """Caller-owned device scratch for libgsx (the library never allocates)."""

    def __init__(self, limit: int = 8) -> None:
        self.buffers = _Lru(limit)

    def get(self, device: torch.device, nbytes: int) -> torch.Tensor:
        # one buffer per (device, stream): frames in flight on different streams must not share scratch; the
        # buffers of the `limit` most recently used streams are kept (a C3 frame's is ~130 MB)
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        buf = self.buffers.lookup(key)
        if buf is None or buf.numel() < nbytes:
            buf = self.buffers.store(key, torch.empty(nbytes, dtype=torch.uint8, device=device))
        return buf


_WORKSPACE = _Workspace()
_PINNED_SLOTS = 256
_PLAIN_MIN_TILES = 16384  # windows from this many tiles take GSX_FLAG_PLAIN_FOOTPRINTS when the view allows it (render_image_hip)
_HINT_VIEWS = 64          # GsxParams.hints buffers a scene keeps (83 KB each at 1080p, 300 KB at 4K): the most recent views
_SEMANTICS = {"ref_cpu": _ffi.GSX_SEM_REF_CPU, "ref_cuda": _ffi.GSX_SEM_REF_CUDA,
              "std_3dgs": _ffi.GSX_SEM_STD_3DGS}


def _stream_handle(device: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(device: torch.device) -> None:
    if device.type != "cuda":
        raise RuntimeError(
            "the Gaussian tensors are on %s: this renderer runs only as HIP kernels on an AMD GPU "
            "(torch device 'cuda'); there is no CPU fallback" % device)


def render_preprocessed(height: int, width: int, tile_size: int, point_means: torch.Tensor,
                        point_colors: torch.Tensor, inverse_covariance_2d: torch.Tensor, min_x: torch.Tensor,
                        max_x: torch.Tensor, min_y: torch.Tensor, max_y: torch.Tensor, opacity: torch.Tensor,
                        layout: str = "wh3", instances_hint: int = 0,
                        stats: Optional[dict] = None, semantics: str = "ref_cpu", flags: int = 0) -> torch.Tensor:
    """Stage 2 on depth-sorted stage-1 arrays: the reference's native entry point
    ``render_image(image_height, image_width, tile_size, point_means, point_colors,
    inverse_covariance_2d, min_x, max_x, min_y, max_y, opacity)`` (splat/c/render.cu:90-101) with the
    CPU path's compositing semantics (``semantics="ref_cuda"``: the CUDA kernel's own semantics,
    SURVEY.md Appendix B).  Returns (W,H,3) for layout "wh3", (H,W,3) for "hw3".  ``flags``: GSX_FLAG_* of
    include/gsx.h, as they are (a caller that sets GSX_FLAG_PLAIN_FOOTPRINTS reads ``stats["n_redo"]``)."""
    lib = _ffi.load()
    dev = point_means.device
    _require_gpu(dev)
    n = int(point_means.shape[0])
    args = [_check_f32(k, v, dev) for k, v in (
        ("point_means", point_means), ("point_colors", point_colors),
        ("inverse_covariance_2d", inverse_covariance_2d), ("min_x", min_x), ("max_x", max_x),
        ("min_y", min_y), ("max_y", max_y), ("opacity", opacity))]
    params = _ffi.default_params()
    params.layout = _ffi.GSX_LAYOUT_WH3 if layout == "wh3" else _ffi.GSX_LAYOUT_HW3
    params.semantics = _SEMANTICS[semantics]
    params.flags |= int(flags)
    shape = (width, height, 3) if layout == "wh3" else (height, width, 3)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    st = _ffi.GsxFrameStats()
    cap = max(int(instances_hint), 8 * n + 4096)
    with torch.cuda.device(dev):
        for _ in range(3):
            nbytes = lib.gsx_workspace_bytes(n, width, height, tile_size, cap)
            if nbytes == 0:
                raise _ffi.GsxError(_ffi.GSX_ERR_INVALID_ARGUMENT, "gsx_workspace_bytes rejected the sizes")
            ws = _WORKSPACE.get(dev, nbytes)
            rc = lib.gsx_render_preprocessed(height, width, tile_size, *[_ptr(a) for a in args], n, _ptr(out),
                                             ctypes.byref(params), ctypes.byref(st), _ptr(ws), ws.numel(),
                                             _stream_handle(dev))
            if rc != _ffi.GSX_ERR_WORKSPACE_TOO_SMALL:
                break
            cap = int(st.n_instances * 1.25) + 4096
    _ffi.check(rc)
    if stats is not None:
        stats.update(n_visible=st.n_visible, n_instances=st.n_instances, n_tiles=st.n_tiles, n_redo=int(st.n_redo))
    return out


class NativeExtension:
    """What ``GaussianScene.compile_cuda_ext()`` returns: the object the reference gets from ``load_inline``
    (splat/gaussian_scene.py:240-261, splat/utils.py:426-434) -- ONE function, ``render_image``, with the argument
    list and return value of splat/c/render.cu:90-101.  Here nothing is compiled at run time: the function is
    ``gsx_render_preprocessed`` of the prebuilt libgsx.so under the CUDA kernel's own rules (GSX_SEM_REF_CUDA,
    (H,W,3) output), so a caller that does ``ext = scene.compile_cuda_ext(); ext.render_image(H, W, tile, ...)``
    keeps working unchanged."""

    def __init__(self) -> None:
        _ffi.load()          # fails loudly, like a failed JIT build would, when the library is absent

    @staticmethod
    def render_image(image_height: int, image_width: int, tile_size: int, point_means: torch.Tensor,
                     point_colors: torch.Tensor, inverse_covariance_2d: torch.Tensor, min_x: torch.Tensor,
                     max_x: torch.Tensor, min_y: torch.Tensor, max_y: torch.Tensor, opacity: torch.Tensor) -> torch.Tensor:
        return render_preprocessed(int(image_height), int(image_width), int(tile_size), point_means, point_colors,
                                   inverse_covariance_2d, min_x, max_x, min_y, max_y, opacity, layout="hw3",
                                   semantics="ref_cuda")


class CapturedFrame:
    """A frame recorded by ``GaussianScene.capture_frame``: ``replay()`` launches the graph,
    ``out`` is the frame buffer it writes, ``confirm()`` (synchronises) checks that the last replay
    did not exceed the pair capacity recorded in the graph and returns the frame."""

    def __init__(self, scene: "GaussianScene", graph, out: torch.Tensor, pinned: torch.Tensor, call: dict,
                 camera_buffer: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
                 capacity: int = 0, inputs: Optional[list] = None) -> None:
        self.scene, self.graph, self.out, self._pinned, self._call = scene, graph, out, pinned, call
        self._camera_buffer = camera_buffer      # device copy of the GsxCamera the recorded kernels read
        self._workspace = workspace              # scratch whose address the graph holds: lives as long as the frame
        self.capacity = int(capacity)            # (Gaussian, tile) pairs the recorded launches have room for
        # the exact parameter tensors handed to the library while recording (a non-contiguous attribute of the
        # scene is a COPY made for the call): the graph reads these addresses on every replay
        self._inputs = inputs

    def set_camera(self, image_idx: int) -> None:
        """Points the captured frame at another camera of the scene (same frame size): the next
        ``replay()`` enqueued on this stream renders that view.  Only for frames captured with
        ``movable_camera=True``.  The constants are copied on the current stream; do not call this
        while a replay of this frame is still in flight on another stream."""
        if self._camera_buffer is None:
            raise RuntimeError("this frame was captured with its camera baked in: capture it with movable_camera=True")
        cam = self.scene.images[image_idx].gsx_camera()
        first = self.scene.images[self._call["image_idx"]].gsx_camera()
        if (cam.width, cam.height) != (first.width, first.height):
            raise ValueError("a captured frame keeps its size: %dx%d, the new camera is %dx%d"
                             % (first.width, first.height, cam.width, cam.height))
        self._camera_buffer.copy_(torch.frombuffer(bytearray(bytes(cam)), dtype=torch.uint8))

    def replay(self) -> torch.Tensor:
        # (spatially ordered rows: the boxes a strip's projection trusts follow points / scales -- refreshed IN PLACE, at the
        # address the graph holds, when those were modified since; a comparison of six integers otherwise)
        g = self.scene.gaussians
        if getattr(g, "original_index", None) is not None:
            g.current_block_bounds()
        self.graph.replay()
        return self.out

    def counts(self) -> Tuple[int, int, int]:
        """(visible Gaussians, (Gaussian, tile) pairs, pairs the recorded launches have room for) of the last replay;
        synchronises.  More pairs than room: pairs were dropped, the frame has to be rendered again (``confirm()``
        raises for exactly that)."""
        torch.cuda.synchronize(self.out.device)
        st = ctypes.cast(ctypes.c_void_p(self._pinned.data_ptr()), ctypes.POINTER(_ffi.GsxFrameStats)).contents
        return int(st.n_visible), int(st.n_instances), int(st.reserved)

    def confirm(self) -> torch.Tensor:
        torch.cuda.synchronize(self.out.device)
        st = ctypes.cast(ctypes.c_void_p(self._pinned.data_ptr()), ctypes.POINTER(_ffi.GsxFrameStats)).contents
        if st.n_instances > st.reserved:
            raise _ffi.GsxError(_ffi.GSX_ERR_WORKSPACE_TOO_SMALL,
                                "the captured frame holds %d pairs but the scene now produces %d: capture it again"
                                % (st.reserved, st.n_instances))
        if getattr(self, "_plain_footprints", False) and st.n_redo > 0:
            raise _ffi.GsxError(_ffi.GSX_ERR_UNSUPPORTED,
                                "the frame was captured with GSX_FLAG_PLAIN_FOOTPRINTS and %d tiles now hold ill-conditioned "
                                "footprints: capture it again" % st.n_redo)
        return self.out


class GaussianScene:
    def __init__(self, colmap_path: str, gaussians: Gaussians) -> None:
        cameras = read_camera_file(colmap_path)
        images = read_image_file(colmap_path)
        self.images: Dict[int, GaussianImage] = {}
        for idx, image in images.items():
            self.images[idx] = GaussianImage(camera=cameras[image.camera_id], image=image,
                                             device=gaussians.device)
        self.gaussians = gaussians
        self._instances_hint = 0      # workspace sizing: largest instance count seen (+10 %)
        self._last_instances = 0      # instance count of the latest full frame
        self._cap_hints = {}          # (image, tile, window, semantics) -> pair capacity for the next frame
        self._kept_hints = {}         # same key -> Gaussians that reached a tile of the window (GsxParams.kept_hint)
        self._n_redo_seen = {}         # same key -> tiles of the last frame that held an ill-conditioned footprint (n_redo)
        self._hints = _Lru(_HINT_VIEWS)   # (same key, stream) -> [GsxParams.hints buffer, a frame has run with it?]
        self._pending = []            # speculative frames awaiting confirm_frames()
        self._part_events = {}        # (stream, K) -> K HIP events the library records behind the parts of a frame
        self._pinned_pool = None
        self._pinned_next = 0

    # ------------------------------------------------------------------ helpers
    def _colors(self, image_idx: int) -> torch.Tensor:
        """``gaussians.colors`` like the reference, or -- build extension -- the view-dependent colour
        of this camera evaluated on the GPU from ``gaussians.sh`` (gsx_sh_to_rgb)."""
        g = self.gaussians
        if getattr(g, "sh", None) is None:
            return g.colors
        lib = _ffi.load()
        dev = g.points.device
        _require_gpu(dev)
        n = int(g.points.shape[0])
        k = (int(g.sh_degree) + 1) ** 2
        pts = _check_f32("points", g.points.reshape(n, 3), dev)
        sh = _check_f32("sh", g.sh.reshape(n, k, 3), dev)
        out = torch.empty((n, 3), dtype=torch.float32, device=dev)
        center = (ctypes.c_float * 3)(*self.images[image_idx].camera_center_host)   # no device read per frame
        with torch.cuda.device(dev):
            rc = lib.gsx_sh_to_rgb(_ptr(pts), _ptr(sh), int(g.sh_degree), n, center, _ptr(out), _stream_handle(dev))
        _ffi.check(rc)
        return out

    def _inputs(self, image_idx: Optional[int] = None, inline_sh: bool = False):
        """The five parameter tensors the library reads.  With ``inline_sh`` and an SH scene the last one is
        None: the whole-path entry evaluates the colour itself from ``gaussians.sh`` (GsxParams.sh)."""
        g = self.gaussians
        dev = g.points.device
        _require_gpu(dev)
        n = int(g.points.shape[0])
        has_sh = getattr(g, "sh", None) is not None
        tensors = [
            _check_f32("points", g.points.reshape(n, 3), dev),
            _check_f32("scales", g.scales.reshape(n, 3), dev),
            _check_f32("quaternions", g.quaternions.reshape(n, 4), dev),
            _check_f32("opacity", g.opacity.reshape(n, 1), dev),
        ]
        if has_sh and inline_sh:
            tensors.append(None)
        else:
            colors = g.colors if (image_idx is None or not has_sh) else self._colors(image_idx)
            tensors.append(_check_f32("colors", colors.reshape(n, 3), dev))
        return dev, n, tensors

    def _original_index(self, dev, n: int) -> Optional[torch.Tensor]:
        """``gaussians.original_index`` (Gaussians.spatially_ordered) as the library wants it -- int32, contiguous, n
        entries, on the device of the parameters -- or None for rows in the caller's own order."""
        oi = getattr(self.gaussians, "original_index", None)
        if oi is None:
            return None
        if oi.device != dev or oi.dtype != torch.int32 or not oi.is_contiguous() or oi.numel() != n:
            raise ValueError("gaussians.original_index must be a contiguous int32 tensor of %d entries on %s" % (n, dev))
        return oi

    # ------------------------------------------------------------------ stage 1
    def preprocess(self, image_idx: int) -> PreprocessedScene:
        """Projection + depth sort on the GPU (gsx_preprocess); fields as splat/schema.py:13-25."""
        lib = _ffi.load()
        dev, n, tensors = self._inputs(image_idx)
        cam = self.images[image_idx].gsx_camera()
        # one allocation for the eleven float outputs (21 floats per Gaussian) and the order, carved into views
        flat = torch.empty(max(n, 1) * 22, dtype=torch.float32, device=dev)
        off = [0]

        def f(*shape):
            cnt = 1
            for v in shape:
                cnt *= v
            view = flat[off[0]:off[0] + cnt].view(*shape)
            off[0] += cnt
            return view

        xy, col, c2, dep, inv, rad = f(n, 2), f(n, 3), f(n, 2, 2), f(n), f(n, 2, 2), f(n)
        mnx, mxx, mny, mxy, sop = f(n), f(n), f(n), f(n), f(n, 1)
        order = f(n).view(torch.int32)
        nvis = ctypes.c_int64(0)
        params = _ffi.default_params()
        oi = self._original_index(dev, n)
        if oi is not None:          # spatially ordered rows: everything is filed and reported under the original index
            params.original_index = oi.data_ptr()
        with torch.cuda.device(dev):
            nbytes = lib.gsx_workspace_bytes(n, cam.width, cam.height, 16, 1)
            ws = _WORKSPACE.get(dev, nbytes)
            for _ in range(2):
                rc = lib.gsx_preprocess(ctypes.byref(cam), *[_ptr(t) for t in tensors], n,
                                        *[_ptr(t) for t in (xy, col, c2, dep, inv, rad, mnx, mxx, mny, mxy, sop)],
                                        _ptr(order), ctypes.byref(nvis), ctypes.byref(params), _ptr(ws), ws.numel(),
                                        _stream_handle(dev))
                # at most three Gaussians pass the cull, fewer than the call assumed: the reference's BLAS then sums its
                # products over the visible ones in another order (GSX_FLAG_SMALL_BATCH / _ONE_VISIBLE, include/gsx.h) --
                # once more, in that order
                again = _ffi.visible_rows_flag(n, int(nvis.value), params.flags) if rc == _ffi.GSX_OK else 0
                if not again:
                    break
                params.flags = _ffi.with_rows_flag(params.flags, again)
        _ffi.check(rc)
        m = nvis.value
        self.last_order = order[:m]
        return PreprocessedScene(points=xy[:m], colors=col[:m], covariance_2d=c2[:m], depths=dep[:m],
                                 inverse_covariance_2d=inv[:m], radius=rad[:m], points_xy=xy[:m],
                                 min_x=mnx[:m], min_y=mny[:m], max_x=mxx[:m], max_y=mxy[:m],
                                 sigmoid_opacity=sop[:m])

    # ------------------------------------------------------------------ whole path
    def render_image_hip(self, image_idx: int, tile_size: int = 16, layout: str = "wh3",
                         tile_window: Optional[Tuple[int, int, int, int]] = None,
                         out: Optional[torch.Tensor] = None, out_origin: Tuple[int, int] = (0, 0),
                         stats: Optional[dict] = None, timing: bool = False,
                         no_sync: bool = False, semantics: str = "ref_cpu",
                         background: Tuple[float, float, float] = (0.0, 0.0, 0.0),
                         generic_kernels: bool = False, published_rects: bool = False,
                         camera_buffer: Optional[torch.Tensor] = None,
                         tile_counts: Optional[torch.Tensor] = None, split_long_tiles: bool = True,
                         tile_schedule: Optional[bool] = None, use_hints: bool = True,
                         substrips: Optional[Sequence[int]] = None, substrip_events: Optional[list] = None,
                         _private: Optional[dict] = None) -> torch.Tensor:
        """Full forward render in libgsx (gsx_render_forward).

        semantics: "ref_cpu" (the reference's ``render_image``), "ref_cuda" (its CUDA kernel's rules
        on the same stage 1) or "std_3dgs" (build extension: the published 3DGS forward pass --
        0.3 dilation, alpha clamp 0.99, 1/255 skip, T < 1e-4 stop, ``background`` behind the last
        Gaussian, every tile of the frame; see include/gsx.h).

        layout "wh3" -> (W,H,3) indexed [x,y] like ``render_image``; "hw3" -> (H,W,3).
        tile_window (tx0,tx1,ty0,ty1) renders only those tiles (row/column strips for multi-GPU);
        ``out`` may then be a strip-sized buffer whose pixel (0,0) is frame pixel ``out_origin``.
        ``timing`` asks the library for per-stage HIP-event times (``stats["stage_ms"]``); the call
        then waits for the frame.
        ``tile_counts`` (int32 / uint32 device tensor, one entry per tile of the window, x-major) receives
        the length of every tile's Gaussian list (GsxParams.tile_counts; ``strips.balanced_plan``).
        ``use_hints`` (default): every (camera, tile size, window, rule set, stream) of this scene keeps a small
        device buffer (GsxParams.hints) in which a frame leaves the depth-sort splitters and what every tile cost
        for the NEXT frame of that view, which then skips the two kernels that would compute them on its own critical
        path; stale hints (the Gaussians or the camera changed) cost time, never a pixel.  ``use_hints=False``
        renders every frame from scratch (tests hold the two against each other).
        ``substrips`` (GsxParams.n_substrips): K + 1 ascending tile coordinates along the layout's LEADING axis (x for
        "wh3", y for "hw3") that cut the window into K parts; projection, depth order and binning run once, the
        compositing launch once per part, and ``substrip_events`` (a list the caller passes in) receives K ``_hip.Event``
        objects, event k recorded on the current stream right behind part k -- a multi-GPU rank starts sending part k
        while part k + 1 is composited (``strips.render_overlapped``).  Same pixels bit for bit.
        ``no_sync`` (GSX_FLAG_NO_SYNC) enqueues the frame without waiting for the device at all: the
        counts arrive later in pinned memory and ``confirm_frames()`` must be called (it
        synchronises) before the images are trusted -- it re-renders, on the normal path, any frame
        whose pair count exceeded the workspace capacity it was enqueued with.
        """
        lib = _ffi.load()
        dev, n, tensors = self._inputs(image_idx, inline_sh=True)
        g_ = self.gaussians
        cam = self.images[image_idx].gsx_camera()
        width, height = cam.width, cam.height
        params = _ffi.default_params()
        sh_flat = None
        if tensors[4] is None:      # SH scene: the projection kernel evaluates the view-dependent colour itself
            g = self.gaussians
            k = (int(g.sh_degree) + 1) ** 2
            sh_flat = _check_f32("sh", g.sh.reshape(n, k, 3), dev)
            params.sh, params.sh_degree = sh_flat.data_ptr(), int(g.sh_degree)
        oi = self._original_index(dev, n)
        if oi is not None:          # spatially ordered rows (Gaussians.spatially_ordered): same frame, bit for bit
            ro = getattr(g_, "row_of_index", None)
            if ro is None or ro.device != dev or ro.dtype != torch.int32 or not ro.is_contiguous() or ro.numel() != n:
                raise ValueError("gaussians.row_of_index (the inverse of original_index) must be a contiguous int32 tensor "
                                 "of %d entries on %s" % (n, dev))
            params.original_index, params.row_of_index = oi.data_ptr(), ro.data_ptr()
            # ... and a strip's projection drops whole blocks of 256 rows after reading their box (recomputed here when
            # points / scales were modified since: Gaussians.current_block_bounds)
            bb = g_.current_block_bounds() if hasattr(g_, "current_block_bounds") else getattr(g_, "block_bounds", None)
            if bb is not None:
                if bb.device != dev or bb.dtype != torch.float32 or not bb.is_contiguous() or \
                        tuple(bb.shape) != (-(-n // _ffi.GSX_BOUNDS_ROWS), 8):
                    raise ValueError("gaussians.block_bounds must be a contiguous float32 (%d, 8) tensor on %s "
                                     "(Gaussians.refresh_block_bounds)" % (-(-n // _ffi.GSX_BOUNDS_ROWS), dev))
                params.block_bounds = bb.data_ptr()
        params.layout = _ffi.GSX_LAYOUT_WH3 if layout == "wh3" else _ffi.GSX_LAYOUT_HW3
        params.semantics = _SEMANTICS[semantics]
        params.background[0], params.background[1], params.background[2] = [float(v) for v in background]
        if timing:
            params.flags |= _ffi.GSX_FLAG_TIMING
        if generic_kernels:     # tests: the any-tile-size kernels also at tile 16 (same pixels)
            params.flags |= _ffi.GSX_FLAG_GENERIC_KERNELS
        if not split_long_tiles:   # tests: every tile on one wave (same pixels as the four-wave path of long tiles)
            params.flags |= _ffi.GSX_FLAG_NO_LONG_TILE_SPLIT
        if tile_schedule is not None:   # tiles handed out by list length (default: scenes of >= 300 000 Gaussians)
            params.flags |= _ffi.GSX_FLAG_TILE_SCHEDULE if tile_schedule else _ffi.GSX_FLAG_NO_TILE_SCHEDULE
        if published_rects:     # std_3dgs: bin with the published 3-sigma squares (same pixels, longer lists)
            params.flags |= _ffi.GSX_FLAG_PUBLISHED_RECTS
        if camera_buffer is not None:   # GsxParams.camera_device: the kernels read the camera from this buffer
            if camera_buffer.device != dev or camera_buffer.dtype != torch.uint8 or \
                    camera_buffer.numel() != ctypes.sizeof(_ffi.GsxCamera):
                raise ValueError("camera_buffer must be %d bytes (uint8) on %s" % (ctypes.sizeof(_ffi.GsxCamera), dev))
            params.camera_device = camera_buffer.data_ptr()
        if tile_window is not None:
            params.tile_x0, params.tile_x1, params.tile_y0, params.tile_y1 = [int(v) for v in tile_window]
        keep_alive = []
        if substrips is not None and len(substrips) > 2:
            from . import _hip

            k_parts = len(substrips) - 1
            bounds = (ctypes.c_int32 * (k_parts + 1))(*[int(v) for v in substrips])
            evs = self._part_events.setdefault((torch.cuda.current_stream(dev).cuda_stream, k_parts),
                                               [_hip.Event() for _ in range(k_parts)])
            handles = (ctypes.c_void_p * k_parts)(*[e.handle for e in evs])
            params.n_substrips, params.substrip_axis = k_parts, 0 if layout == "wh3" else 1
            params.substrip_bounds = ctypes.cast(bounds, ctypes.POINTER(ctypes.c_int32))
            params.substrip_events = ctypes.cast(handles, ctypes.POINTER(ctypes.c_void_p))
            keep_alive += [bounds, handles]
            if substrip_events is not None:
                substrip_events[:] = evs
        if tile_counts is not None:
            if tile_counts.device != dev or tile_counts.dtype not in (torch.int32, torch.uint32) or \
                    not tile_counts.is_contiguous():
                raise ValueError("tile_counts must be a contiguous int32 tensor on %s" % dev)
            from .strips import tiles_along
            ntx, nty = tiles_along(width, tile_size, semantics), tiles_along(height, tile_size, semantics)
            wx0, wx1, wy0, wy1 = (0, ntx, 0, nty) if tile_window is None else [int(v) for v in tile_window]
            wx1, wy1 = (ntx if wx1 < 0 else min(wx1, ntx)), (nty if wy1 < 0 else min(wy1, nty))
            if tile_counts.numel() < max(0, wx1 - wx0) * max(0, wy1 - wy0):
                raise ValueError("tile_counts holds %d entries, the window has %d tiles"
                                 % (tile_counts.numel(), max(0, wx1 - wx0) * max(0, wy1 - wy0)))
            params.tile_counts = tile_counts.data_ptr()
        if out is None:
            shape = (width, height, 3) if layout == "wh3" else (height, width, 3)
            out = torch.empty(shape, dtype=torch.float32, device=dev)
        else:
            _check_f32("out", out, dev)
            if not out.is_contiguous():
                raise ValueError("out must be contiguous")
            ow, oh = (out.shape[0], out.shape[1]) if layout == "wh3" else (out.shape[1], out.shape[0])
            params.out_x0, params.out_y0, params.out_w, params.out_h = int(out_origin[0]), int(out_origin[1]), int(ow), int(oh)
        # pair capacity: 1.1 x the count this (camera, tile size, window, semantics) produced last time
        # -- the binning kernels' grids are sized by it -- or a generous guess for a first frame
        cap_key = (image_idx, tile_size, None if tile_window is None else tuple(int(v) for v in tile_window),
                   semantics + ("/published" if published_rects else ""))
        cap = self._cap_hints.get(cap_key, 0) or max(self._instances_hint, 8 * n + 4096)
        # _private (capture_frame): a pair capacity, scratch buffer and count slot owned by ONE captured
        # frame -- its graph bakes their addresses in, so they must not be shared with or recycled by
        # any other frame; such a call neither reads nor updates the scene's hints and pending list
        own = _private or {}
        if "cap" in own:
            cap = int(own["cap"])
        if "inputs" in own:         # capture_frame: the tensors whose addresses the graph bakes in stay alive with it
            passed = [t for t in tensors if t is not None] + ([sh_flat] if sh_flat is not None else [])
            scene_own = [g_.points, g_.scales, g_.quaternions, g_.opacity] + \
                ([g_.sh] if sh_flat is not None else [g_.colors])
            if any(t.data_ptr() != a.data_ptr() for t, a in zip(passed, scene_own)):
                raise ValueError("capture_frame needs contiguous float32 Gaussian tensors: a captured frame replays "
                                 "from the scene's own memory, and a non-contiguous attribute would be copied once, "
                                 "at capture time")
            own["inputs"][:] = passed + ([oi, g_.row_of_index] if oi is not None else []) + \
                ([bb] if oi is not None and bb is not None else [])
        # how many Gaussians reached a tile of this window last time: picks the depth-sort route (a hint)
        params.kept_hint = int(own.get("kept", self._kept_hints.get(cap_key, 0)))
        # GSX_FLAG_PLAIN_FOOTPRINTS: no tile of the last frame of this view held an ill-conditioned footprint, so this one runs
        # the compositing instance that cannot evaluate them -- where that pays: from _PLAIN_MIN_TILES tiles (measured: 6 %
        # of a 4K frame of 5M Gaussians, nothing at 1080p); its own n_redo says whether that held (checked below / by
        # confirm_frames / by CapturedFrame.confirm: a frame that was wrong about it is rendered again without the flag)
        n_window_tiles = (tile_window[1] - tile_window[0]) * (tile_window[3] - tile_window[2]) if tile_window is not None \
            else (-(-width // tile_size) * -(-height // tile_size) if tile_size > 0 else 0)
        plain = bool(own["plain"]) if "plain" in own else \
            (self._n_redo_seen.get(cap_key) == 0 and n_window_tiles >= _PLAIN_MIN_TILES)
        if plain and semantics == "ref_cpu" and tile_size == 16 and not generic_kernels:
            params.flags |= _ffi.GSX_FLAG_PLAIN_FOOTPRINTS
        # capture_frame: the row class (GSX_FLAG_SMALL_BATCH / _ONE_VISIBLE) the view's uncaptured frame ended up with is
        # baked into the graph -- a captured call cannot be issued again when n_visible says it assumed wrongly
        params.flags |= int(own.get("rows_flag", 0))
        # GsxParams.hints: one buffer per view and stream (a captured frame owns its own), valid once a frame has filled it
        hint_slot = None
        if use_hints:
            if "hints" in own:
                hint_slot = own["hints"]
            else:
                hkey = (cap_key, torch.cuda.current_stream(dev).cuda_stream)
                hint_slot = self._hints.lookup(hkey)
                hbytes = lib.gsx_hints_bytes(width, height, tile_size)
                if hint_slot is None or hint_slot[0].numel() != hbytes or hint_slot[0].device != dev:
                    hint_slot = self._hints.store(hkey, [torch.zeros(hbytes, dtype=torch.uint8, device=dev), False])
            params.hints = hint_slot[0].data_ptr()
            if hint_slot[1]:
                params.flags |= _ffi.GSX_FLAG_HINTS_VALID
        speculative = bool(no_sync and not timing)
        if speculative:
            params.flags |= _ffi.GSX_FLAG_NO_SYNC
            if "pinned" in own:
                pinned = own["pinned"]
            else:
                if len(self._pending) >= _PINNED_SLOTS:
                    self.confirm_frames()
                pinned = self._pinned_slot()
            st_ref = ctypes.cast(ctypes.c_void_p(pinned.data_ptr()), ctypes.POINTER(_ffi.GsxFrameStats))
            st = st_ref.contents
        else:
            st = _ffi.GsxFrameStats()
            st_ref = ctypes.byref(st)
        with torch.cuda.device(dev):
            for _ in range(6):      # (a larger workspace, the second compositing launch after all, another row class: each at most once or twice)
                nbytes = lib.gsx_workspace_bytes(n, width, height, tile_size, cap)
                if nbytes == 0:
                    raise _ffi.GsxError(_ffi.GSX_ERR_INVALID_ARGUMENT, "gsx_workspace_bytes rejected the sizes")
                ws = own.get("workspace")
                if ws is None:
                    ws = _WORKSPACE.get(dev, nbytes)
                elif ws.numel() < nbytes:
                    raise ValueError("private workspace holds %d bytes, the frame needs %d" % (ws.numel(), nbytes))
                # hand over exactly the bytes of `cap` pairs (the buffer may be larger): the library
                # derives its pair capacity -- and the binning grids -- from the size it is given
                rc = lib.gsx_render_forward(ctypes.byref(cam), *[_ptr(t) for t in tensors], n, tile_size, _ptr(out),
                                            ctypes.byref(params), st_ref, _ptr(ws), nbytes,
                                            _stream_handle(dev))
                if rc == _ffi.GSX_OK and not speculative and params.flags & _ffi.GSX_FLAG_PLAIN_FOOTPRINTS and int(st.n_redo) > 0:
                    # the view does hold ill-conditioned footprints after all: once more, with the instance that evaluates them
                    params.flags &= ~_ffi.GSX_FLAG_PLAIN_FOOTPRINTS
                    continue
                again = 0
                if rc == _ffi.GSX_OK and not speculative and not own and semantics != "std_3dgs":
                    again = _ffi.visible_rows_flag(n, int(st.n_visible), params.flags)
                if again:
                    # at most three Gaussians pass the cull, fewer than the call assumed: the reference's BLAS then sums
                    # its products over the visible ones in another order (GSX_FLAG_SMALL_BATCH / _ONE_VISIBLE,
                    # include/gsx.h) -- once more, in that order
                    params.flags = _ffi.with_rows_flag(params.flags, again)
                    continue
                if rc != _ffi.GSX_ERR_WORKSPACE_TOO_SMALL or "cap" in own:
                    break
                cap = int(st.n_instances * 1.25) + 4096
        _ffi.check(rc)
        if hint_slot is not None:
            # (stream order: the next frame on this stream finds what this one left -- or, where the library took a
            # path that does not use the buffer, the zeros it was created with: its header then says "nothing here")
            hint_slot[1] = True
        if own:
            if stats is not None and not speculative:
                stats.update(n_visible=st.n_visible, n_instances=st.n_instances, n_tiles=st.n_tiles, n_kept=st.n_kept,
                             n_redo=int(st.n_redo))
            return out
        if speculative:
            # counts are still in flight: remember what has to be confirmed
            self._pending.append((pinned, cap_key, bool(params.flags & _ffi.GSX_FLAG_PLAIN_FOOTPRINTS), dict(
                image_idx=image_idx, tile_size=tile_size, layout=layout, tile_window=tile_window, out=out,
                out_origin=out_origin, semantics=semantics, background=background,
                generic_kernels=generic_kernels, published_rects=published_rects, camera_buffer=camera_buffer,
                tile_counts=tile_counts, split_long_tiles=split_long_tiles, tile_schedule=tile_schedule,
                use_hints=use_hints)))
            if stats is not None:
                stats.update(n_visible=None, n_instances=None, n_tiles=st.n_tiles, speculative=True,
                             plain_footprints=bool(params.flags & _ffi.GSX_FLAG_PLAIN_FOOTPRINTS))
            return out
        self._note_count(cap_key, int(st.n_instances), int(st.n_kept), int(st.n_redo))
        if stats is not None:
            # (plain_footprints: the frame ran the compositing instance of GSX_FLAG_PLAIN_FOOTPRINTS)
            stats.update(n_visible=st.n_visible, n_instances=st.n_instances, n_tiles=st.n_tiles, n_kept=st.n_kept,
                         n_redo=int(st.n_redo), plain_footprints=bool(params.flags & _ffi.GSX_FLAG_PLAIN_FOOTPRINTS))
            if timing:
                stats["stage_ms"] = {k: float(st.stage_ms[i]) for i, k in enumerate(_ffi.STAGE_NAMES)}
        return out

    def _note_count(self, cap_key, n_instances: int, n_kept: int = 0, n_redo: Optional[int] = None) -> None:
        self._instances_hint = max(self._instances_hint, int(n_instances * 1.1))
        self._cap_hints[cap_key] = int(n_instances * 1.1) + 4096
        self._kept_hints[cap_key] = max(1, int(n_kept))
        if n_redo is not None:
            self._n_redo_seen[cap_key] = int(n_redo)
        if cap_key[2] is None and cap_key[3] == "ref_cpu":
            self._last_instances = n_instances

    def _ensure_pinned_pool(self) -> None:
        if self._pinned_pool is None:
            self._pinned_pool = torch.zeros((_PINNED_SLOTS, ctypes.sizeof(_ffi.GsxFrameStats)),
                                            dtype=torch.uint8).pin_memory()

    def _pinned_slot(self) -> torch.Tensor:
        """One 64-byte slot of a pinned ring (pinning memory per frame would dominate the frame)."""
        self._ensure_pinned_pool()
        slot = self._pinned_pool[self._pinned_next % _PINNED_SLOTS]
        self._pinned_next += 1
        slot.zero_()
        return slot

    def confirm_frames(self) -> int:
        """Synchronises and validates every frame rendered with ``no_sync=True`` since the last call.
        A frame whose pair count exceeded the capacity of the workspace it was enqueued with (pairs
        were dropped) is rendered again on the normal path into the same output tensor.  Returns the
        number of re-renders."""
        if not self._pending:
            return 0
        torch.cuda.synchronize(self.gaussians.points.device)
        redone = 0
        pending, self._pending = self._pending, []
        for pinned, cap_key, ran_plain, call in pending:
            st = ctypes.cast(ctypes.c_void_p(pinned.data_ptr()), ctypes.POINTER(_ffi.GsxFrameStats)).contents
            self._note_count(cap_key, int(st.n_instances), int(st.n_kept), int(st.n_redo))
            # more pairs than the workspace held (dropped), or tiles left to a second compositing launch that was not issued
            if st.n_instances > st.reserved or (ran_plain and st.n_redo > 0):
                redone += 1
                self.render_image_hip(**call)          # synchronising path, same output tensor
        return redone

    def capture_frame(self, image_idx: int, tile_size: int = 16, layout: str = "wh3",
                      semantics: str = "ref_cpu", movable_camera: bool = False,
                      headroom: float = 1.1) -> "CapturedFrame":
        """Records one whole frame of this camera (every launch, clear and the asynchronous count
        copy) into a hipGraph.  ``frame.replay()`` then re-renders it with ONE graph launch (~20 us
        of host time instead of ~120 us for ~30 separate launches) from the CURRENT contents of the
        Gaussian tensors -- all buffer addresses are baked in, and so are the camera constants unless
        ``movable_camera`` is set: then the projection kernel reads them from a device buffer
        (GsxParams.camera_device) that ``frame.set_camera(other_image_idx)`` rewrites between replays
        (same frame size; SH scenes too: the colour is evaluated inside the projection kernel with the
        camera centre of that buffer).  The graph holds room for ``headroom`` x the pair count of the captured view; a replay
        that needs more (another camera may) is reported by ``confirm()``.  Possible because the
        no-sync frame has no host dependency at all."""
        dev = self.gaussians.points.device
        _require_gpu(dev)
        cam = self.images[image_idx].gsx_camera()
        shape = (cam.width, cam.height, 3) if layout == "wh3" else (cam.height, cam.width, 3)
        out = torch.empty(shape, dtype=torch.float32, device=dev)
        cam_buf = None
        if movable_camera:
            cam_buf = torch.frombuffer(bytearray(bytes(cam)), dtype=torch.uint8).to(dev)
        kw = dict(tile_size=tile_size, layout=layout, out=out, semantics=semantics, camera_buffer=cam_buf)
        st = {}
        self.render_image_hip(image_idx, stats=st, **kw)    # normal path: the pair count of this view
        # What the graph bakes in belongs to this frame alone: the pair capacity the recorded launches are
        # sized for (headroom x this view's count), a scratch buffer of exactly that size and a pinned
        # slot for the counts.  None of them comes from (or goes back to) the scene's shared pools.
        lib = _ffi.load()
        n = int(self.gaussians.points.shape[0])
        cap = int(st["n_instances"] * max(float(headroom), 1.0)) + 4096
        nbytes = lib.gsx_workspace_bytes(n, cam.width, cam.height, tile_size, cap)
        if nbytes == 0:
            raise _ffi.GsxError(_ffi.GSX_ERR_INVALID_ARGUMENT, "gsx_workspace_bytes rejected the sizes")
        # the graph bakes in which compositing instance runs: the plain one (GSX_FLAG_PLAIN_FOOTPRINTS) only for a fixed camera
        # whose frame held no ill-conditioned footprint, where it pays (confirm() checks every replay's n_redo); a movable
        # camera may turn to such footprints
        private = dict(cap=cap, workspace=torch.empty(nbytes, dtype=torch.uint8, device=dev), kept=int(st["n_kept"]),
                       rows_flag=0 if semantics == "std_3dgs" else max(_ffi.visible_rows_flag(n, int(st["n_visible"]), 0), 0),
                       plain=(not movable_camera) and int(st.get("n_redo", 1)) == 0 and
                       int(st.get("n_tiles", 0)) >= _PLAIN_MIN_TILES,
                       pinned=torch.zeros(ctypes.sizeof(_ffi.GsxFrameStats), dtype=torch.uint8).pin_memory(), inputs=[],
                       hints=[torch.zeros(lib.gsx_hints_bytes(cam.width, cam.height, tile_size), dtype=torch.uint8,
                                          device=dev), False])
        stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(stream):   # the same call once on the capture stream, outside the capture
            self.render_image_hip(image_idx, _private=private, **kw)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            self.render_image_hip(image_idx, no_sync=True, _private=private, **kw)
        call = dict(image_idx=image_idx, **kw)
        frame = CapturedFrame(self, graph, out, private["pinned"], call, camera_buffer=cam_buf,
                              workspace=private["workspace"], capacity=cap, inputs=private["inputs"])
        frame._hints = private["hints"][0]     # the recorded kernels read and refresh this buffer on every replay
        frame._plain_footprints = bool(private["plain"])
        return frame

    def render_image(self, image_idx: int, tile_size: int = 16) -> torch.Tensor:
        """(W,H,3) float32 indexed [x,y]; same result as the reference's pure-Python
        ``render_image`` (gaussian_scene.py:200-238), computed on the GPU and -- like the reference,
        whose image is a CPU tensor (gaussian_scene.py:206) -- returned in host memory, so that
        ``plt.imshow(scene.render_image(i))`` keeps working.  ``render_image_hip`` is the same frame
        left on the device."""
        frame = self.render_image_hip(image_idx, tile_size=tile_size, layout="wh3")
        # page-locked destination (torch's caching host allocator keeps the pages registered between
        # calls): the copy runs at PCIe rate instead of through a pageable staging loop
        host = torch.empty(frame.shape, dtype=frame.dtype, pin_memory=True)
        host.copy_(frame, non_blocking=True)
        torch.cuda.current_stream(frame.device).synchronize()
        return host

    render = render_image  # the name BASELINE.json's north star uses

    def render_images(self, image_indices, tile_size: int = 16):
        """``render_image`` for a sequence of cameras -- the reference's own loop renders one image index after another
        (splat/gaussian_scene.py:200-203) --, as a generator of (W,H,3) HOST tensors with the device-to-host copy of
        frame i overlapped with the rendering of frame i + 1: two device frames and two page-locked host buffers take
        turns, the copy runs on a stream of its own behind an event.  A frame's host tensor stays valid until the
        generator has yielded two more (copy it if it has to live longer).  At 1M Gaussians / 1080p the 25 MB copy
        is longer than the render (0.36 ms): the loop runs at the copy's pace, 0.68 ms per frame measured, where
        ``render_image`` called per frame pays render + copy + two synchronisations, 0.97-1.02 ms (bench.py: host_frames)."""
        dev = self.gaussians.points.device
        _require_gpu(dev)
        copy_stream = torch.cuda.Stream(dev)
        frames, hosts, copied, checks, pending = [None, None], [None, None], [None, None], [None, None], None

        def finish(k: int) -> torch.Tensor:
            copied[k].synchronize()
            if checks[k] is not None:
                pinned, cap_key, ran_plain, call = checks[k]
                st = ctypes.cast(ctypes.c_void_p(pinned.data_ptr()), ctypes.POINTER(_ffi.GsxFrameStats)).contents
                self._note_count(cap_key, int(st.n_instances), int(st.n_kept), int(st.n_redo))
                if st.n_instances > st.reserved or (ran_plain and st.n_redo > 0):    # dropped pairs / tiles left undone: once more, synchronously
                    self.render_image_hip(**call)
                    hosts[k].copy_(frames[k])
            return hosts[k]

        for n, idx in enumerate(image_indices):
            k = n & 1
            cam = self.images[idx].gsx_camera()
            shape = (cam.width, cam.height, 3)
            if frames[k] is None or tuple(frames[k].shape) != shape:
                frames[k] = torch.empty(shape, dtype=torch.float32, device=dev)
                hosts[k] = torch.empty(shape, dtype=torch.float32, pin_memory=True)
            if copied[k] is not None:
                torch.cuda.current_stream(dev).wait_event(copied[k])     # the copy that last read this device frame
            # enqueued without waiting for the device (GSX_FLAG_NO_SYNC: the pair list is sized by this view's last count);
            # the counts land in pinned memory before the copy below has finished and are looked at when the frame is
            # handed out -- a frame that needed more pairs than it was given room for is rendered again, synchronously
            self.render_image_hip(idx, tile_size=tile_size, layout="wh3", out=frames[k], no_sync=True)
            checks[k] = self._pending.pop() if self._pending else None
            done = torch.cuda.current_stream(dev).record_event()
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done)
                hosts[k].copy_(frames[k], non_blocking=True)
                copied[k] = copy_stream.record_event()
            if pending is not None:                 # hand out the frame before this one: its copy ran under this render
                yield finish(pending)
            pending = k
        if pending is not None:
            yield finish(pending)

    def compile_cuda_ext(self) -> NativeExtension:
        """The reference's drop-in boundary (splat/gaussian_scene.py:240-261): returns an object whose
        ``render_image(image_height, image_width, tile_size, point_means, point_colors, inverse_covariance_2d,
        min_x, max_x, min_y, max_y, opacity)`` is the native entry point.  Nothing is compiled here (libgsx.so is
        built ahead of time by hipcc for gfx950); the call costs nothing and may be repeated per frame as the
        reference does."""
        return NativeExtension()

    def render_image_cuda(self, image_idx: int, tile_size: int = 16) -> torch.Tensor:
        """(H,W,3) float32 indexed [y,x] with the semantics of the reference's CUDA kernel
        (``render_image_cuda``, splat/gaussian_scene.py:263-285 -> splat/c/render.cu): the same flow as the
        reference, call for call -- ``preprocess``, ``compile_cuda_ext``, the native ``render_image`` on the
        stage-1 arrays, one device synchronisation."""
        pre = self.preprocess(image_idx)
        height, width = self.images[image_idx].height, self.images[image_idx].width
        ext = self.compile_cuda_ext()
        image = ext.render_image(height, width, tile_size, pre.points.contiguous(), pre.colors.contiguous(),
                                 pre.inverse_covariance_2d.contiguous(), pre.min_x.contiguous(), pre.max_x.contiguous(),
                                 pre.min_y.contiguous(), pre.max_y.contiguous(), pre.sigmoid_opacity.contiguous())
        torch.cuda.synchronize(image.device)
        return image

    def render_preprocessed(self, image_idx: int, pre: PreprocessedScene, tile_size: int = 16,
                            layout: str = "wh3", stats: Optional[dict] = None) -> torch.Tensor:
        cam = self.images[image_idx].gsx_camera()
        return render_preprocessed(cam.height, cam.width, tile_size, pre.points, pre.colors,
                                   pre.inverse_covariance_2d, pre.min_x, pre.max_x, pre.min_y, pre.max_y,
                                   pre.sigmoid_opacity, layout=layout, stats=stats)

    # ------------------------------------------------------------------ debug helpers
    def render_points_image(self, image_idx: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Projected (x_pix, y_pix, ndc_z) of the in-view points and their colours
        (gaussian_scene.py:44-51 -> image.py:72-89)."""
        return self.images[image_idx].project_point_to_camera_perspective_projection(
            self.gaussians.points, self.gaussians.colors)

    def get_2d_covariance(self, image_idx: int, points: torch.Tensor, covariance_3d: torch.Tensor) -> torch.Tensor:
        """(n,2,2) EWA covariance of ``points`` (n,3) with 3D covariances (n,3,3) under camera
        ``image_idx`` (gaussian_scene.py:53-68 -> utils.py:320-354), computed by gsx_covariance_2d."""
        lib = _ffi.load()
        dev = points.device
        _require_gpu(dev)
        n = int(points.shape[0])
        pts = _check_f32("points", points.reshape(n, 3), dev)
        cov = _check_f32("covariance_3d", covariance_3d.reshape(n, 3, 3), dev)
        out = torch.empty((n, 2, 2), dtype=torch.float32, device=dev)
        cam = self.images[image_idx].gsx_camera()
        with torch.cuda.device(dev):
            rc = lib.gsx_covariance_2d(ctypes.byref(cam), _ptr(pts), _ptr(cov), n, _ptr(out), _stream_handle(dev))
        _ffi.check(rc)
        return out
