"""Which explicit float32 operation order reproduces torch, bit for bit, on each product of the path?

TEST INFRASTRUCTURE ONLY (build container; needs torch on the CPU, not the reference's sources).

The reference's stage 1 and its weight evaluation are torch eager ops.  "left to right, no FMA" is
one reading of ``@``; what torch EXECUTES depends on the kernel each shape is routed to (MKL sgemm
for the folded (N*r, k) @ (k, c) products, ATen's own loop for small batched ones).  This script
runs every product of the path on random float32 data in the shapes, strides and views the
reference uses (file:line below), beside the candidate orders of oracle/probe_order.c, over
several N and thread counts, and prints per product the candidates with ZERO bit differences.
oracle/raster_cpu.c, oracle/cpu_ref.py and csrc/gsx_project.hip restate the winners with explicit
``fmaf``; tests/test_oracle_golden.py then checks the outcome against the reference itself.

    python oracle/probe_torch_order.py            # table
    python oracle/probe_torch_order.py --json     # machine-readable

Products probed (paths relative to /root/reference):
  view / clip     [p,1] @ world2view, [p,1] @ full_proj    splat/gaussian_scene.py:79-90,
                                                           splat/utils.py:305-310, 333-337
  normalize       F.normalize(q, p=2, dim=1)               splat/gaussians.py:59
  R @ S, M @ M^T  batched 3x3                              splat/gaussians.py:66-68
  J @ W           batched x single (W = world2view[:3,:3].T)  splat/utils.py:352-354
  (JW) @ Sigma    batched x batched                        splat/utils.py:354
  ... @ W.T       batched x single (a strided view)        splat/utils.py:354
  ... @ J^T       batched x batched (transposed view)      splat/utils.py:354
  weight          (-0.5 d) @ Q @ d.T, (1,2)@(2,2)@(2,1)    splat/utils.py:363-364
"""
from __future__ import annotations

import ctypes
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = {"l2r": 0, "fma_seq": 1, "fma_rev": 2, "pairwise": 3, "f64": 4, "l2r_rev": 5, "fma_pair": 6}


def _build():
    out = os.path.join(tempfile.mkdtemp(prefix="probe_order_"), "libprobe.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", out,
                           os.path.join(HERE, "probe_order.c"), "-lm"])
    lib = ctypes.CDLL(out)
    lib.probe_bmm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int64, ctypes.c_int]
    return lib


LIB = None


def cand_bmm(A: np.ndarray, B: np.ndarray, mode: str) -> np.ndarray:
    """A (N,I,K) float32; B (N,K,J) or (K,J).  The candidate order ``mode`` for every output element."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    n, i, k = A.shape
    j = B.shape[-1]
    C = np.empty((n, i, j), dtype=np.float32)
    LIB.probe_bmm(A.ctypes.data, B.ctypes.data, C.ctypes.data, n, i, k, j, k * j if B.ndim == 3 else 0, MODES[mode])
    return C


def bits(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def count_diff(t: torch.Tensor, c: np.ndarray) -> int:
    """Bit differences, +0 and -0 taken as equal (a zero row of J gives either)."""
    a, b = bits(t.numpy()).ravel(), bits(c).ravel()
    return int(np.count_nonzero((a != b) & ~(((a | b) & 0x7FFFFFFF) == 0)))


def probe(name: str, torch_fn, cand_fn, sizes, threads, rs) -> dict:
    res = {m: 0 for m in MODES}
    total = 0
    for nt in threads:
        torch.set_num_threads(nt)
        for n in sizes:
            args = torch_fn.make(n, rs)
            with torch.no_grad():
                out = torch_fn(*args)
            total += out.numel()
            for m in MODES:
                res[m] += count_diff(out, cand_fn(*args, m).reshape(out.shape))
    return {"product": name, "elements": total, "bit_differences": res,
            "exact": [m for m, d in res.items() if d == 0]}


def _rand(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


def _camera(rs):
    """A world2view-like 4x4 in the reference's row-vector form (last column 0,0,0,1) and in the reference's LAYOUT: a
    transposed VIEW of the row-major extrinsic matrix (splat/image.py:51-53) -- the BLAS is told about the
    transposition, and which of its kernels runs depends on it when the product has few rows."""
    q = rs.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = rs.standard_normal(3)
    return torch.from_numpy(E.astype(np.float32)).transpose(0, 1)


def few_rows(rs) -> dict:
    """Products with at most three rows (a scene with at most three visible Gaussians), the operand layouts as in the
    reference: which of these hand-written orders reproduce torch -- none of the regular candidates does."""
    f32 = np.float32
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle.cpu_ref import fma            # exact float32 FMA in numpy

    res = {"[p,1] @ world2view, 1 row: ((p0 M0 + p1 M1 fused) + M3) + p2 M2": 0,
           "[p,1] @ world2view, 2-3 rows: (p0 M0 + p2 M2) + (p1 M1 + M3), unfused": 0,
           "[p,1] @ full_proj, 1-3 rows: sequential FMA": 0,
           "X @ world2view[:3,:3] (= W.T), 1-3 batches: (k0 + k2) + k1, unfused": 0,
           "J @ world2view[:3,:3].T (= W), 1-3 batches: sequential FMA": 0}
    for _ in range(200):
        for n in (1, 2, 3):
            V = _camera(rs)
            Vn = np.ascontiguousarray(V.numpy())
            p = _rand(rs, n, 3, scale=3.0)
            h = torch.cat([p, torch.ones(n, 1)], dim=1)
            pn = p.numpy()
            got = (h @ V).numpy()
            if n == 1:
                c = np.stack([(fma(pn[:, 1], Vn[1, j], pn[:, 0] * Vn[0, j]) + Vn[3, j]) + pn[:, 2] * Vn[2, j] for j in range(4)], 1)
                res["[p,1] @ world2view, 1 row: ((p0 M0 + p1 M1 fused) + M3) + p2 M2"] += count_diff(torch.from_numpy(got), c.astype(f32))
            else:
                c = np.stack([(pn[:, 0] * Vn[0, j] + pn[:, 2] * Vn[2, j]) + (pn[:, 1] * Vn[1, j] + Vn[3, j]) for j in range(4)], 1)
                res["[p,1] @ world2view, 2-3 rows: (p0 M0 + p2 M2) + (p1 M1 + M3), unfused"] += count_diff(torch.from_numpy(got), c.astype(f32))
            F = _rand(rs, 4, 4)
            res["[p,1] @ full_proj, 1-3 rows: sequential FMA"] += count_diff(h @ F, cand_bmm(h.numpy()[:, None, :], F.numpy(), "fma_seq").reshape(n, 4))
            X = _rand(rs, n, 3, 3)
            W = V[:3, :3].T
            Wt = np.ascontiguousarray(W.T.numpy())
            Xn = X.numpy()
            c = np.empty_like(Xn)
            for i in range(3):
                for j in range(3):
                    c[:, i, j] = (Xn[:, i, 0] * Wt[0, j] + Xn[:, i, 2] * Wt[2, j]).astype(f32) + Xn[:, i, 1] * Wt[1, j]
            res["X @ world2view[:3,:3] (= W.T), 1-3 batches: (k0 + k2) + k1, unfused"] += count_diff(X @ W.T, c)
            res["J @ world2view[:3,:3].T (= W), 1-3 batches: sequential FMA"] += count_diff(
                X @ W, cand_bmm(Xn, np.ascontiguousarray(W.numpy()), "fma_seq"))
    return {"product": "few rows (N_vis <= 3), reference layouts", "elements": None, "bit_differences": res,
            "exact": [k for k, v in res.items() if v == 0]}


def main() -> None:
    global LIB
    LIB = _build()
    rs = np.random.RandomState(0)
    # four rows and up (what a product of at most three rows executes is another matter: few_rows)
    sizes = [4, 5, 7, 64, 1000, 100_000, 1_000_000]
    small = [4, 5, 7, 64, 1000, 100_000]
    threads = [1, 8]
    out = []

    # --- [p,1] @ M, M dense 4x4 (gaussian_scene.py:79-90) --------------------------------------
    def f_view(h, M):
        return h @ M

    def mk_view(n, rs):
        p = _rand(rs, n, 3, scale=3.0)
        h = torch.cat([p, torch.ones(n, 1)], dim=1)
        return h, _rand(rs, 4, 4)

    f_view.make = mk_view
    out.append(probe("[p,1] @ full_proj (N,4)@(4,4), contiguous M (a bmm result)", f_view,
                     lambda h, M, m: cand_bmm(h.numpy()[:, None, :], M.numpy(), m), sizes, threads, rs))

    def mk_view_t(n, rs):
        p = _rand(rs, n, 3, scale=3.0)
        return torch.cat([p, torch.ones(n, 1)], dim=1), _camera(rs)

    f_view_t = lambda h, M: h @ M  # noqa: E731
    f_view_t.make = mk_view_t
    out.append(probe("[p,1] @ world2view (N,4)@(4,4), M a transposed view, N >= 4", f_view_t,
                     lambda h, M, m: cand_bmm(h.numpy()[:, None, :], np.ascontiguousarray(M.numpy()), m), sizes, threads, rs))

    # few rows (N <= 3): other MKL kernels, found by exhaustive search over orders (see few_rows below)
    out.append(few_rows(rs))

    # in_view_frustum builds h with ones + slice assignment and multiplies by the view matrix (utils.py:305-307)
    def f_frustum(h, M):
        return h @ M

    def mk_frustum(n, rs):
        h = torch.ones((n, 4))
        h[:, :3] = _rand(rs, n, 3, scale=3.0)
        return h, _camera(rs)

    f_frustum.make = mk_frustum
    out.append(probe("in_view_frustum h @ world2view", f_frustum,
                     lambda h, M, m: cand_bmm(h.numpy()[:, None, :], M.numpy(), m), sizes, threads, rs))

    # --- F.normalize (gaussians.py:59): q / max(||q||, 1e-12) ------------------------------------
    def f_norm(q):
        return torch.nn.functional.normalize(q, p=2, dim=1)

    f_norm.make = lambda n, rs: (_rand(rs, n, 4),)

    def c_norm(q, m):
        qn = q.numpy()
        ss = cand_bmm(qn[:, None, :], qn[:, :, None], m).reshape(-1)
        nrm = np.maximum(np.sqrt(ss), np.float32(1e-12))
        return qn / nrm[:, None]

    out.append(probe("F.normalize(q) = q / max(sqrt(sum q^2), eps): order of the sum", f_norm, c_norm, sizes, threads, rs))

    # --- batched 3x3 products (gaussians.py:66-68, utils.py:354) --------------------------------
    def f_bmm(A, B):
        return A @ B

    f_bmm.make = lambda n, rs: (_rand(rs, n, 3, 3), _rand(rs, n, 3, 3))
    out.append(probe("batched (N,3,3) @ (N,3,3)", f_bmm, lambda A, B, m: cand_bmm(A.numpy(), B.numpy(), m),
                     small, threads, rs))

    def f_bmm_t(A, B):
        return A @ B.transpose(1, 2)

    f_bmm_t.make = f_bmm.make
    out.append(probe("batched (N,3,3) @ (N,3,3).transpose(1,2)", f_bmm_t,
                     lambda A, B, m: cand_bmm(A.numpy(), B.numpy().transpose(0, 2, 1), m), small, threads, rs))

    # J @ W: J has a zero last row and zero [0,1], [1,0]; W = world2view[:3,:3].T (a view) (utils.py:352-354)
    def mk_jw(n, rs):
        J = torch.zeros(n, 3, 3)
        J[:, 0, 0] = _rand(rs, n).abs() * 100
        J[:, 0, 2] = _rand(rs, n) * 50
        J[:, 1, 1] = _rand(rs, n).abs() * 100
        J[:, 1, 2] = _rand(rs, n) * 50
        return J, _camera(rs)

    def f_jw(J, V):
        return J @ V[:3, :3].T

    f_jw.make = mk_jw
    out.append(probe("J @ W, (N,3,3) @ (3,3) view world2view[:3,:3].T", f_jw,
                     lambda J, V, m: cand_bmm(J.numpy(), V.numpy()[:3, :3].T, m), small, threads, rs))

    def f_dense_single(A, V):
        return A @ V[:3, :3].T

    f_dense_single.make = lambda n, rs: (_rand(rs, n, 3, 3), _camera(rs))
    out.append(probe("dense (N,3,3) @ (3,3) view world2view[:3,:3].T", f_dense_single,
                     lambda A, V, m: cand_bmm(A.numpy(), V.numpy()[:3, :3].T, m), small, threads, rs))

    def f_wt(A, V):
        W = V[:3, :3].T
        return A @ W.T

    f_wt.make = f_dense_single.make
    out.append(probe("X @ W.T, (N,3,3) @ (3,3) view world2view[:3,:3]", f_wt,
                     lambda A, V, m: cand_bmm(A.numpy(), V.numpy()[:3, :3], m), small, threads, rs))

    # the whole chain as the reference writes it, J-shaped first factor (utils.py:354)
    def f_chain(J, V, S):
        W = V[:3, :3].T
        return (J @ W @ S @ W.T @ J.transpose(1, 2))[:, :2, :2]

    def mk_chain(n, rs):
        J, V = mk_jw(n, rs)
        M = _rand(rs, n, 3, 3, scale=0.05)
        return J, V, M @ M.transpose(1, 2)

    f_chain.make = mk_chain
    chain_modes = {}
    for m_single in ("l2r", "fma_seq"):
        for m_batched in ("l2r", "fma_seq"):
            def c_chain(J, V, S, _m, ms=m_single, mb=m_batched):
                W = V.numpy()[:3, :3].T
                a = cand_bmm(J.numpy(), W, ms)
                b = cand_bmm(a, S.numpy(), mb)
                c = cand_bmm(b, W.T, ms)
                d = cand_bmm(c, J.numpy().transpose(0, 2, 1), mb)
                return d[:, :2, :2]
            r = probe("chain", f_chain, c_chain, small, threads, rs)
            chain_modes["single=%s batched=%s" % (m_single, m_batched)] = r["bit_differences"]["l2r"]
    out.append({"product": "Sigma2D chain J@W@S@W.T@J^T [:2,:2]", "elements": None,
                "bit_differences": chain_modes, "exact": [k for k, v in chain_modes.items() if v == 0]})

    # --- weight (utils.py:363-364): (1,2)@(2,2) then (1,2)@(2,1), per pixel ----------------------
    def f_w(d, Q):
        return torch.stack([(-0.5 * d[i:i + 1]) @ Q[i] @ d[i:i + 1].T for i in range(d.shape[0])]).reshape(-1)

    f_w.make = lambda n, rs: (_rand(rs, n, 2, scale=20.0), _rand(rs, n, 2, 2))
    w_modes = {}
    for m1 in ("l2r", "fma_seq"):
        for m2 in ("l2r", "fma_seq"):
            def c_w(d, Q, _m, a=m1, b=m2):
                dn = d.numpy()
                t = cand_bmm((np.float32(-0.5) * dn)[:, None, :], Q.numpy(), a)
                return cand_bmm(t, dn[:, :, None], b).reshape(-1)
            r = probe("w", f_w, c_w, [20000], [1], rs)
            w_modes["first=%s second=%s" % (m1, m2)] = r["bit_differences"]["l2r"]
    out.append({"product": "weight exponent (-0.5 d) @ Q @ d.T", "elements": 20000,
                "bit_differences": w_modes, "exact": [k for k, v in w_modes.items() if v == 0]})

    if "--json" in sys.argv:
        print(json.dumps(out, indent=1))
        return
    for r in out:
        print("%-62s exact: %s" % (r["product"], ", ".join(r["exact"]) or "NONE"))
        print("    bit differences per candidate: %s" % r["bit_differences"])


if __name__ == "__main__":
    main()
