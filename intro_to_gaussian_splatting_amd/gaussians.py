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

    def __len__(self) -> int:
        return int(self.points.shape[0])

    def to(self, device) -> "Gaussians":
        dev = torch.device(device)
        for name in ("points", "colors", "scales", "quaternions", "opacity"):
            setattr(self, name, getattr(self, name).to(dev).contiguous())
        if self.sh is not None:
            self.sh = self.sh.to(dev).contiguous()
        self.device = dev
        return self

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
