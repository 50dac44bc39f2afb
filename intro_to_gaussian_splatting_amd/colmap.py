"""Minimal COLMAP sparse-model reader: cameras and images, text or binary.

Counterpart of what ``GaussianScene.__init__`` needs from the reference's
``splat/read_colmap.py:87-239`` and ``splat/utils.py:269-290`` (``.bin`` preferred over ``.txt``).
Same record fields as the reference's namedtuples (``Camera``: id, model, width, height, params;
``Image``: id, qvec, tvec, camera_id, name, xys, point3D_ids); the render path reads camera model /
size / params and image pose / camera id / name.  Both readers are held against the reference's own
parse of a committed model (tests/golden/colmap_model*, captured by oracle/capture_golden.py).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, NamedTuple

import numpy as np


class Camera(NamedTuple):
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray


class Image(NamedTuple):
    id: int
    qvec: np.ndarray   # (w, x, y, z)
    tvec: np.ndarray
    camera_id: int
    name: str
    xys: np.ndarray = np.zeros((0, 2))            # 2D keypoints (n, 2)
    point3D_ids: np.ndarray = np.zeros(0, dtype=np.int64)


# COLMAP camera model id -> (name, number of params)
_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
           4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
           8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


def _data_lines(path: str):
    with open(path, "r") as fid:
        for raw in fid:
            line = raw.strip()
            if line and not line.startswith("#"):
                yield line


def read_cameras_text(path: str) -> Dict[int, Camera]:
    out = {}
    for line in _data_lines(path):
        tok = line.split()
        cam_id = int(tok[0])
        out[cam_id] = Camera(cam_id, tok[1], int(tok[2]), int(tok[3]), np.array([float(v) for v in tok[4:]]))
    return out


def read_images_text(path: str) -> Dict[int, Image]:
    out = {}
    with open(path, "r") as fid:
        while True:
            raw = fid.readline()
            if not raw:
                break
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            tok = line.split()
            img_id = int(tok[0])
            pts = fid.readline().split()           # the keypoint line: x y point3D_id triples
            xys = np.array([float(v) for v in pts[0::3] + pts[1::3]]).reshape(2, -1).T
            ids = np.array([int(v) for v in pts[2::3]], dtype=np.int64)
            out[img_id] = Image(img_id, np.array([float(v) for v in tok[1:5]]),
                                np.array([float(v) for v in tok[5:8]]), int(tok[8]), tok[9], xys, ids)
    return out


def _unpack(fid, fmt: str):
    size = struct.calcsize("<" + fmt)
    data = fid.read(size)
    if len(data) != size:
        raise ValueError("truncated COLMAP binary file")
    return struct.unpack("<" + fmt, data)


def read_cameras_binary(path: str) -> Dict[int, Camera]:
    out = {}
    with open(path, "rb") as fid:
        (count,) = _unpack(fid, "Q")
        for _ in range(count):
            cam_id, model_id, width, height = _unpack(fid, "iiQQ")
            name, nparams = _MODELS[model_id]
            params = np.array(_unpack(fid, "d" * nparams))
            out[cam_id] = Camera(cam_id, name, int(width), int(height), params)
    return out


def read_images_binary(path: str) -> Dict[int, Image]:
    out = {}
    with open(path, "rb") as fid:
        (count,) = _unpack(fid, "Q")
        for _ in range(count):
            rec = _unpack(fid, "idddddddi")
            name = bytearray()
            while True:
                ch = fid.read(1)
                if ch in (b"\x00", b""):
                    break
                name += ch
            (npts,) = _unpack(fid, "Q")
            raw = fid.read(24 * npts)              # (x, y, point3D_id) triples: double, double, int64
            if len(raw) != 24 * npts:
                raise ValueError("truncated COLMAP binary file")
            trip = np.frombuffer(raw, dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")]))
            xys = np.stack([trip["x"], trip["y"]], axis=1) if npts else np.zeros((0, 2))
            out[rec[0]] = Image(rec[0], np.array(rec[1:5]), np.array(rec[5:8]), rec[8], name.decode("utf-8"),
                                xys, trip["id"].astype(np.int64))
    return out


def read_camera_file(colmap_path: str) -> Dict[int, Camera]:
    binary, text = os.path.join(colmap_path, "cameras.bin"), os.path.join(colmap_path, "cameras.txt")
    if os.path.exists(binary):
        return read_cameras_binary(binary)
    if os.path.exists(text):
        return read_cameras_text(text)
    raise ValueError("no cameras.bin / cameras.txt under %s" % colmap_path)


def read_image_file(colmap_path: str) -> Dict[int, Image]:
    binary, text = os.path.join(colmap_path, "images.bin"), os.path.join(colmap_path, "images.txt")
    if os.path.exists(binary):
        return read_images_binary(binary)
    if os.path.exists(text):
        return read_images_text(text)
    raise ValueError("no images.bin / images.txt under %s" % colmap_path)
