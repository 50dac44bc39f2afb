"""Gaussian parameter container with the reference's attribute layout (splat/gaussians.py:9-33).

``points (N,3)``, ``colors (N,3)`` = rgb/256, ``scales (N,3)`` linear (no exp), ``quaternions
(N,4)`` (w,x,y,z), ``opacity (N,1)`` logit -- float32 PyTorch-ROCm tensors resident in HBM.
Differences from the reference, on purpose: the constructor has no file side effect (the
reference writes ``point_cloud.ply``, gaussians.py:17-18) and never tracks gradients (the
reference's path is forward-only).
"""
from __future__ import annotations

from typing import Optional

import torch


class Gaussians:
    def __init__(self, points: torch.Tensor, colors: torch.Tensor, model_path: str = ".",
                 device: Optional[str] = None) -> None:
        dev = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.device = dev
        self.model_path = model_path
        n = points.shape[0]
        f32 = torch.float32
        self.points = torch.as_tensor(points).detach().to(dev, f32).contiguous()
        self.colors = (torch.as_tensor(colors).detach().to(f32) / 256).to(dev).contiguous()
        self.scales = torch.full((n, 3), 0.001, dtype=f32, device=dev)          # gaussians.py:23
        self.quaternions = torch.zeros((n, 4), dtype=f32, device=dev)           # gaussians.py:29-30
        self.quaternions[:, 0] = 1.0
        p = 0.9999 * torch.ones((n, 1), dtype=f32)                             # gaussians.py:31-33
        self.opacity = torch.log(p / (1 - p)).to(dev)
        # build extension: spherical-harmonic colour (N,K,3); None = use `colors` like the reference
        self.sh: Optional[torch.Tensor] = None
        self.sh_degree = 0
        # build extension (spatially_ordered): int32 (N,), the ORIGINAL index of the Gaussian in every row; None = the rows
        # are in the caller's own order
        self.original_index: Optional[torch.Tensor] = None
        self.row_of_index: Optional[torch.Tensor] = None      # its inverse: row_of_index[original_index[i]] == i
        # ... and float32 (ceil(N / 256), 8): per block of 256 rows (min xyz, largest |scale|, max xyz, 0) -- GsxParams.block_bounds
        self.block_bounds: Optional[torch.Tensor] = None

    def __len__(self) -> int:
        return int(self.points.shape[0])

    def to(self, device) -> "Gaussians":
        dev = torch.device(device)
        for name in ("points", "colors", "scales", "quaternions", "opacity"):
            setattr(self, name, getattr(self, name).to(dev).contiguous())
        if self.sh is not None:
            self.sh = self.sh.to(dev).contiguous()
        if self.original_index is not None:
            self.original_index = self.original_index.to(dev).contiguous()
            self.row_of_index = self.row_of_index.to(dev).contiguous()
        if self.block_bounds is not None:
            self.block_bounds = self.block_bounds.to(dev).contiguous()
        self.device = dev
        return self

    def refresh_block_bounds(self, rows: int = 256) -> None:
        """(Re)computes ``block_bounds`` from the CURRENT points and scales of a spatially ordered container: per block of
        ``rows`` = GSX_BOUNDS_ROWS consecutive rows the box of the means and the largest |scale|.  Call it again after
        changing ``points`` or ``scales`` in place -- the library trusts the bounds (include/gsx.h: block_bounds)."""
        n = len(self)
        nb = -(-n // rows)
        p = self.points.reshape(n, 3)
        s = self.scales.reshape(n, 3).abs().amax(dim=1) if n else torch.zeros(0, device=p.device)
        pad = nb * rows - n
        big = torch.finfo(torch.float32).max
        lo_src = torch.cat([p, torch.full((pad, 3), big, device=p.device)]) if pad else p
        hi_src = torch.cat([p, torch.full((pad, 3), -big, device=p.device)]) if pad else p
        s_src = torch.cat([s, torch.zeros(pad, device=p.device)]) if pad else s
        out = torch.zeros((nb, 8), dtype=torch.float32, device=p.device)
        if nb:
            out[:, 0:3] = lo_src.reshape(nb, rows, 3).amin(dim=1)
            out[:, 4:7] = hi_src.reshape(nb, rows, 3).amax(dim=1)
            out[:, 3] = s_src.reshape(nb, rows).amax(dim=1)
        old = self.block_bounds
        if old is not None and old.shape == out.shape and old.device == out.device and old.is_contiguous():
            old.copy_(out)          # in place: a captured frame (hipGraph) has this tensor's address baked in
        else:
            self.block_bounds = out.contiguous()
        self._bounds_of = self._bounds_stamp()

    def _bounds_stamp(self):
        """What ``block_bounds`` was computed from: the identity and in-place version of ``points`` and ``scales`` (torch
        counts in-place modifications of a tensor: ``_version``)."""
        return (self.points.data_ptr(), self.points._version, tuple(self.points.shape),
                self.scales.data_ptr(), self.scales._version, tuple(self.scales.shape))

    def current_block_bounds(self) -> Optional[torch.Tensor]:
        """``block_bounds`` for the arrays AS THEY ARE NOW: recomputed first when ``points`` or ``scales`` were replaced or
        modified in place since (an optimiser step, an edit) -- stale boxes would make a strip's projection drop Gaussians
        that have moved into its window.  None for a container whose rows are in the caller's own order."""
        if self.original_index is None:
            return None
        if self.block_bounds is None or getattr(self, "_bounds_of", None) != self._bounds_stamp():
            self.refresh_block_bounds()
        return self.block_bounds

    def spatially_ordered(self, bits: int = 10) -> "Gaussians":
        """A COPY of this container with the rows of every parameter array reordered along a 3D Morton (Z-order) curve of
        the means (``bits`` per axis over the cloud's bounding box; ties by original index), carrying ``original_index``.
        Opt-in, one time, view independent.  What it buys: Gaussians that are close in space -- hence on screen -- are
        close in memory, so a call that renders a PART of the frame (a multi-GPU rank's strip, a tile window) finds the
        survivors of its window test in runs and reads whole cache lines of their scales / quaternions / opacities /
        colours instead of a scattered eighth of every line.  What it does not change: the
        frame.  The library files every depth key under the Gaussian's ORIGINAL index (GsxParams.original_index) and
        sorts them in that order, so equal depths composite in original-index order, ``GaussianScene.last_order`` and
        ``preprocess`` report original indices, and every pixel is the unordered scene's, bit for bit (tested); records
        and rectangles stay in row order (``row_of_index``, the inverse permutation, takes the sort from one to the other).
        Per block of 256 rows ``block_bounds`` holds the box of the means: a strip's projection drops far blocks unread.  The
        reference's arrays stay the default; nothing else of the reference's surface sees the permutation."""
        if self.original_index is not None:
            return self
        n = len(self)
        p = self.points.reshape(n, 3)
        g = Gaussians.__new__(Gaussians)
        g.device, g.model_path, g.sh_degree = self.device, self.model_path, self.sh_degree
        if n == 0:
            perm = torch.zeros(0, dtype=torch.int64, device=p.device)
        else:
            lo, hi = p.min(dim=0).values, p.max(dim=0).values
            span = torch.clamp(hi - lo, min=1e-30)
            cells = float(1 << bits)
            q = torch.clamp(((p - lo) / span * cells).to(torch.int64), 0, (1 << bits) - 1)
            q = torch.where(torch.isfinite(p), q, torch.zeros_like(q))         # (NaN / inf means: cell 0)
            code = torch.zeros(n, dtype=torch.int64, device=p.device)
            for b in range(bits):
                for axis in range(3):
                    code |= ((q[:, axis] >> b) & 1) << (3 * b + axis)
            perm = torch.argsort(code, stable=True)
        for name in ("points", "colors", "scales", "quaternions", "opacity"):
            setattr(g, name, getattr(self, name)[perm].contiguous())
        g.sh = None if self.sh is None else self.sh[perm].contiguous()
        g.original_index = perm.to(torch.int32).contiguous()
        g.row_of_index = torch.empty_like(g.original_index)
        g.row_of_index[perm] = torch.arange(n, dtype=torch.int32, device=perm.device)
        g.block_bounds = None
        g.refresh_block_bounds()
        return g

    def get_3d_covariance_matrix(self) -> torch.Tensor:
        """(N,3,3) Sigma = (R S)(R S)^T, computed on the GPU (gsx_covariance_3d); same result as the
        reference's method (splat/gaussians.py:54-69).  The render path does not call this (the
        projection kernel computes Sigma inline); it exists for users of the reference's API."""
        import ctypes

        from . import _ffi

        lib = _ffi.load()
        dev = self.points.device
        if dev.type != "cuda":
            raise RuntimeError("get_3d_covariance_matrix runs as a HIP kernel: the tensors are on %s "
                               "(no CPU fallback)" % dev)
        n = len(self)
        s = self.scales.reshape(n, 3).contiguous()
        q = self.quaternions.reshape(n, 4).contiguous()
        out = torch.empty((n, 3, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.gsx_covariance_3d(ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(q.data_ptr()), n,
                                       ctypes.c_void_p(out.data_ptr()),
                                       ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _ffi.check(rc)
        return out

    @classmethod
    def from_arrays(cls, points, colors_0_255, scales, quaternions, opacity_logit, device=None) -> "Gaussians":
        """Builds a container and overwrites the constructor's defaults with explicit values."""
        g = cls(torch.as_tensor(points), torch.as_tensor(colors_0_255), device=device)
        f32 = torch.float32
        g.scales = torch.as_tensor(scales).to(g.device, f32).reshape(-1, 3).contiguous()
        g.quaternions = torch.as_tensor(quaternions).to(g.device, f32).reshape(-1, 4).contiguous()
        g.opacity = torch.as_tensor(opacity_logit).to(g.device, f32).reshape(-1, 1).contiguous()
        return g

    @classmethod
    def from_ply(cls, path: str, device=None) -> "Gaussians":
        """Loads a ``.ply``: a trained 3D Gaussian Splatting checkpoint (log-scales -> linear, SH
        coefficients, logit opacity) or a plain xyz+rgb point cloud (then the reference's
        constructor defaults apply).  Build extension -- see ``ply.py``."""
        from .ply import load_gaussians

        d = load_gaussians(path)
        if "sh" not in d:
            return cls(torch.from_numpy(d["points"]), torch.from_numpy(d["colors_0_255"]), device=device)
        n = d["points"].shape[0]
        g = cls.from_arrays(d["points"], torch.zeros((n, 3)), d["scales"], d["quaternions"], d["opacity"],
                            device=device)
        g.sh = torch.from_numpy(d["sh"]).to(g.device).contiguous()
        g.sh_degree = int(d["sh_degree"])
        return g
