"""The three HIP runtime calls the host side needs beyond what PyTorch exposes: events the C library records on a
stream (GsxParams.substrip_events) and other streams wait for.  Bound with ctypes to the runtime instance that is
ALREADY loaded into the process (PyTorch-ROCm ships its own libamdhip64: opening another copy by name would create
events the first one does not know), found through /proc/self/maps."""
from __future__ import annotations

import ctypes

_rt = None
HIP_EVENT_DISABLE_TIMING = 0x2


def runtime():
    global _rt
    if _rt is None:
        import torch  # noqa: F401  (loads the runtime)

        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError("the HIP runtime (libamdhip64) is not loaded in this process: no ROCm build of PyTorch?")
        rt = ctypes.CDLL(path)
        rt.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        rt.hipEventDestroy.argtypes = [ctypes.c_void_p]
        rt.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        rt.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        rt.hipEventQuery.argtypes = [ctypes.c_void_p]
        for fn in (rt.hipEventCreateWithFlags, rt.hipEventDestroy, rt.hipStreamWaitEvent, rt.hipEventSynchronize, rt.hipEventQuery):
            fn.restype = ctypes.c_int
        _rt = rt
    return _rt


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError("%s failed with HIP error %d" % (what, rc))


class Event:
    """A hipEvent_t (timing disabled) whose handle can be handed to libgsx; ``wait_on(stream)`` makes a torch stream
    wait for its latest record."""

    def __init__(self) -> None:
        self.handle = ctypes.c_void_p()
        _check(runtime().hipEventCreateWithFlags(ctypes.byref(self.handle), HIP_EVENT_DISABLE_TIMING), "hipEventCreateWithFlags")

    def wait_on(self, stream) -> None:
        _check(runtime().hipStreamWaitEvent(ctypes.c_void_p(stream.cuda_stream), self.handle, 0), "hipStreamWaitEvent")

    def synchronize(self) -> None:
        _check(runtime().hipEventSynchronize(self.handle), "hipEventSynchronize")

    def __del__(self) -> None:
        try:
            if self.handle and _rt is not None:
                _rt.hipEventDestroy(self.handle)
        except Exception:
            pass
