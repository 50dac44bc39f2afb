"""Camera model: COLMAP pose + PINHOLE intrinsics -> the float32 constants stage 1 reads.

Mirrors the attribute surface of the reference's ``GaussianImage`` (splat/image.py:19-70):
``f_x f_y c_x c_y intrinsic_matrix R T height width extrinsic_matrix fovX fovY tan_fovX tan_fovY znear
zfar world2view projection_matrix full_proj_transform camera_center projection name`` and
``project_point_to_camera_perspective_projection`` (splat/image.py:72-89).  The constants are computed once on
the host with the same mix of double (``math.atan`` / ``math.tan``) and float32 tensor
arithmetic as the reference (splat/utils.py:158-159, 162-172, 189-225), so they are bit-equal
to it; the HIP kernels receive them ready-made in a ``GsxCamera`` and never re-derive them.
The principal point is carried but, as in the reference, not used by the render path.
"""
from __future__ import annotations

import ctypes
import math
from typing import Tuple

import torch

from . import _ffi
from .colmap import Camera, Image


def _rotation_from_qvec(qvec) -> torch.Tensor:
    """(w,x,y,z) -> 3x3, normalising first (splat/utils.py:132-155, float32)."""
    q = torch.tensor([float(v) for v in qvec], dtype=torch.float32)
    # Correctly rounded float32 square root (double sqrt of a float32, rounded once more, is exact).
    # torch.sqrt on the CPU is NOT correctly rounded on every host -- it is 1 ulp off on the EPYC 9575F
    # of the MI355X box while exact on the build host -- and the camera constants must not depend on
    # the host they were computed on (the golden vectors were captured where torch.sqrt is exact).
    norm = torch.tensor(math.sqrt(float(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])), dtype=torch.float32)
    q = q / norm
    w, x, y, z = q[0], q[1], q[2], q[3]
    rows = [
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ]
    return torch.stack([torch.stack(r) for r in rows])


def _fov(focal: torch.Tensor, pixels: torch.Tensor) -> torch.Tensor:
    # float32 quotient, double atan, rounded back to float32 (splat/utils.py:158-159)
    return torch.tensor([2 * math.atan(float(pixels / (2 * focal)))], dtype=torch.float32)


class GaussianImage:
    def __init__(self, camera: Camera, image: Image, device=None) -> None:
        dev = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.device = dev
        f32 = torch.float32
        f_x = torch.tensor([float(camera.params[0])], dtype=f32)
        f_y = torch.tensor([float(camera.params[1])], dtype=f32)
        c_x = torch.tensor([float(camera.params[2])], dtype=f32)
        c_y = torch.tensor([float(camera.params[3])], dtype=f32)
        R = _rotation_from_qvec(image.qvec)
        T = torch.tensor([float(v) for v in image.tvec], dtype=f32)
        height = torch.tensor([float(camera.height)], dtype=f32)
        width = torch.tensor([float(camera.width)], dtype=f32)
        fovX, fovY = _fov(f_x, width), _fov(f_y, height)
        tan_fovX, tan_fovY = torch.tan(fovX / 2), torch.tan(fovY / 2)
        zfar, znear = torch.tensor([100.0], dtype=f32), torch.tensor([0.001], dtype=f32)

        extrinsic = torch.zeros((4, 4), dtype=f32)
        extrinsic[:3, :3] = R
        extrinsic[:3, 3] = T
        extrinsic[3, 3] = 1.0
        world2view = extrinsic.t().contiguous()                      # row-vector convention

        # perspective matrix (splat/utils.py:206-225): tan in double, frustum edges in float32
        half_y, half_x = math.tan(float(fovY / 2)), math.tan(float(fovX / 2))
        top, right = half_y * znear, half_x * znear
        bottom, left = -top, -right
        P = torch.zeros((4, 4), dtype=f32)
        P[0, 0] = 2.0 * znear / (right - left)
        P[1, 1] = 2.0 * znear / (top - bottom)
        P[0, 2] = (right + left) / (right - left)
        P[1, 2] = (top + bottom) / (top - bottom)
        P[3, 2] = 1.0
        P[2, 2] = 1.0 * zfar / (zfar - znear)
        P[2, 3] = -(zfar * znear) / (zfar - znear)
        projection = P.t().contiguous()
        full_proj = world2view.unsqueeze(0).bmm(projection.unsqueeze(0)).squeeze(0)

        self.f_x, self.f_y, self.c_x, self.c_y = f_x.to(dev), f_y.to(dev), c_x.to(dev), c_y.to(dev)
        self.R, self.T = R.unsqueeze(0).to(dev), T.to(dev)
        self.height, self.width = height.to(dev), width.to(dev)
        self.fovX, self.fovY = fovX.to(dev), fovY.to(dev)
        self.tan_fovX, self.tan_fovY = tan_fovX.to(dev), tan_fovY.to(dev)
        self.zfar, self.znear = zfar.to(dev), znear.to(dev)
        self.name = image.name
        self.world2view = world2view.to(dev)
        self.projection_matrix = projection.to(dev)
        self.full_proj_transform = full_proj.to(dev)
        center = world2view.inverse()[3, :3]
        self.camera_center = center.to(dev)
        self.camera_center_host = tuple(float(v) for v in center)   # read per frame by the SH kernel's caller
        # carried for users of the reference's attribute surface; the render path does not read them
        # (splat/image.py:32-34, 39, 68-70; splat/utils.py:19-52)
        intrinsic = torch.zeros((3, 4), dtype=f32)
        intrinsic[0, 0], intrinsic[0, 2] = f_x[0], c_x[0]
        intrinsic[1, 1], intrinsic[1, 2] = f_y[0], c_y[0]
        intrinsic[2, 2] = 1.0
        self.intrinsic_matrix = intrinsic.to(dev)
        self.extrinsic_matrix = extrinsic.to(dev)
        self.projection = (intrinsic @ extrinsic).to(dev)

        cam = _ffi.GsxCamera()
        cam.world2view[:] = world2view.reshape(-1).tolist()
        cam.full_proj[:] = full_proj.reshape(-1).tolist()
        cam.tan_fovx, cam.tan_fovy = float(tan_fovX), float(tan_fovY)
        cam.fx, cam.fy = float(f_x), float(f_y)
        cam.width, cam.height = int(camera.width), int(camera.height)
        cam.camera_center[:] = [float(v) for v in center]
        self._gsx_camera = cam

    def project_point_to_camera_perspective_projection(
            self, points: torch.Tensor, colors: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(x_pix, y_pix, ndc_z) of the points with z_view >= 0.2, in input order, and their colours
        (splat/image.py:72-89), computed by gsx_project_points on the GPU."""
        lib = _ffi.load()
        dev = points.device
        if dev.type != "cuda":
            raise RuntimeError("the points are on %s: this projection runs as a HIP kernel (no CPU fallback)" % dev)
        n = int(points.shape[0])
        pts = points.to(torch.float32).reshape(n, 3).contiguous()
        out = torch.empty((n, 3), dtype=torch.float32, device=dev)
        vis = torch.empty(n, dtype=torch.uint8, device=dev)
        cam = self.gsx_camera()
        with torch.cuda.device(dev):
            rc = lib.gsx_project_points(ctypes.byref(cam), ctypes.c_void_p(pts.data_ptr()), n,
                                        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(vis.data_ptr()),
                                        ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _ffi.check(rc)
        keep = vis.bool()
        return out[keep], colors[keep]

    def gsx_camera(self) -> "_ffi.GsxCamera":
        """The C-ABI view of this camera (include/gsx.h: GsxCamera)."""
        return self._gsx_camera
