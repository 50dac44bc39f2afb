"""Binary PLY IO for Gaussian parameter sets (build extension).

The reference only reads/writes xyz + normals + uint8 rgb through the third-party ``plyfile``
(splat/utils.py:93-125) and has no loader for trained 3D Gaussian Splatting checkpoints.  This
module reads both flavours with numpy alone:

* point clouds ``x y z [nx ny nz] red green blue`` (what ``storePly`` writes), and
* trained 3DGS checkpoints ``x y z nx ny nz f_dc_0..2 f_rest_* opacity scale_0..2 rot_0..3``
  (the published format of Kerbl et al. 2023): log-scales -> linear scales (the reference's
  ``Gaussians.scales`` are linear, splat/gaussians.py:64-66), logit opacity kept as is,
  ``rot`` = (w,x,y,z), SH coefficients regrouped from channel-major ``f_rest`` to (N, K, 3).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
              "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
              "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def _read_header(fid) -> Tuple[str, int, List[Tuple[str, str]]]:
    if fid.readline().strip() != b"ply":
        raise ValueError("not a PLY file")
    fmt, count, props, in_vertex = "", 0, [], False
    while True:
        line = fid.readline()
        if not line:
            raise ValueError("PLY header is not terminated")
        tok = line.decode("ascii", "replace").split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            in_vertex = tok[1] == "vertex"
            if in_vertex:
                count = int(tok[2])
        elif tok[0] == "property" and in_vertex:
            if tok[1] == "list":
                raise ValueError("list properties on the vertex element are not supported")
            props.append((tok[2], _PLY_TYPES[tok[1]]))
        elif tok[0] == "end_header":
            break
    return fmt, count, props


def read_vertices(path: str) -> np.ndarray:
    """The vertex element as a numpy structured array."""
    with open(path, "rb") as fid:
        fmt, count, props = _read_header(fid)
        if fmt == "ascii":
            data = np.loadtxt(fid, max_rows=count, ndmin=2)
            out = np.zeros(count, dtype=[(n, t) for n, t in props])
            for k, (n, _) in enumerate(props):
                out[n] = data[:, k]
            return out
        endian = "<" if fmt == "binary_little_endian" else ">"
        return np.fromfile(fid, dtype=np.dtype([(n, endian + t) for n, t in props]), count=count)


def load_gaussians(path: str) -> Dict[str, np.ndarray]:
    """Returns float32 arrays.  Always ``points (N,3)``.  Trained checkpoint: ``scales`` (linear),
    ``quaternions``, ``opacity`` (logit, (N,1)), ``sh`` (N,K,3), ``sh_degree``.  Point cloud:
    ``colors_0_255`` (N,3)."""
    v = read_vertices(path)
    names = v.dtype.names
    f = np.float32
    out = {"points": np.stack([v["x"], v["y"], v["z"]], axis=1).astype(f)}
    if "f_dc_0" in names:
        dc = np.stack([v["f_dc_%d" % c] for c in range(3)], axis=1).astype(f)           # (N,3)
        rest_names = sorted((n for n in names if n.startswith("f_rest_")), key=lambda n: int(n[7:]))
        if len(rest_names) % 3:
            raise ValueError("f_rest_* count %d is not a multiple of 3" % len(rest_names))
        k_rest = len(rest_names) // 3
        degree = int(round(np.sqrt(k_rest + 1))) - 1
        if (degree + 1) ** 2 != k_rest + 1:
            raise ValueError("f_rest_* count %d does not match an SH degree" % len(rest_names))
        sh = np.empty((v.shape[0], k_rest + 1, 3), dtype=f)
        sh[:, 0, :] = dc
        if k_rest:
            rest = np.stack([v[n] for n in rest_names], axis=1).astype(f)                  # channel-major
            sh[:, 1:, :] = rest.reshape(-1, 3, k_rest).transpose(0, 2, 1)
        out.update(
            sh=sh, sh_degree=np.int64(degree),
            scales=np.exp(np.stack([v["scale_%d" % c] for c in range(3)], axis=1).astype(f)),
            quaternions=np.stack([v["rot_%d" % c] for c in range(4)], axis=1).astype(f),
            opacity=np.asarray(v["opacity"], dtype=f).reshape(-1, 1))
    elif "red" in names:
        out["colors_0_255"] = np.stack([v["red"], v["green"], v["blue"]], axis=1).astype(f)
    return out


def save_trained(path: str, points, sh, scales_linear, quaternions, opacity_logit) -> None:
    """Writes the trained-3DGS layout (binary little endian); inverse of ``load_gaussians``."""
    points = np.asarray(points, np.float32)
    sh = np.asarray(sh, np.float32)
    n, k = sh.shape[0], sh.shape[1]
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    names += ["f_rest_%d" % i for i in range(3 * (k - 1))]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    rec = np.zeros(n, dtype=[(nm, "<f4") for nm in names])
    rec["x"], rec["y"], rec["z"] = points[:, 0], points[:, 1], points[:, 2]
    for c in range(3):
        rec["f_dc_%d" % c] = sh[:, 0, c]
    rest = sh[:, 1:, :].transpose(0, 2, 1).reshape(n, -1)                                   # channel-major
    for i in range(rest.shape[1]):
        rec["f_rest_%d" % i] = rest[:, i]
    rec["opacity"] = np.asarray(opacity_logit, np.float32).reshape(-1)
    log_s = np.log(np.asarray(scales_linear, np.float32))
    q = np.asarray(quaternions, np.float32)
    for c in range(3):
        rec["scale_%d" % c] = log_s[:, c]
    for c in range(4):
        rec["rot_%d" % c] = q[:, c]
    with open(path, "wb") as fid:
        fid.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n)
        for nm in names:
            fid.write(b"property float %s\n" % nm.encode())
        fid.write(b"end_header\n")
        rec.tofile(fid)
