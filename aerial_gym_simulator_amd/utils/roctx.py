"""Range markers for rocprofv3 (`rocprofv3 --marker-trace --kernel-trace ...`): the reference carries commented-out NVTX ranges
around its sensor update (sensors/warp/warp_cam.py:1,70,171); here `AGX_ROCTX=1` wraps EnvManager.step / reset / render and
the tasks' step in ROCTX ranges.  Off by default: a range push / pop pair costs ~0.3 us of host time per call, and the
position task's step is 13 us.  The library is bound at first use (librocprofiler-sdk-roctx.so, or the older libroctx64.so)."""
import ctypes
import functools
import os

_lib = None


def enabled():
    return os.environ.get("AGX_ROCTX") == "1"


def _load():
    global _lib
    if _lib is None:
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _lib = lib
                break
            except OSError:
                continue
        else:
            _lib = False
    return _lib


def push(name):
    lib = _load()
    if lib:
        lib.roctxRangePushA(name)


def pop():
    lib = _load()
    if lib:
        lib.roctxRangePop()


def ranged(name):
    """decorator: the call runs inside a ROCTX range when AGX_ROCTX=1 was set at import time, and is left alone otherwise"""
    label = name.encode()

    def deco(fn):
        if not enabled():
            return fn

        @functools.wraps(fn)
        def wrapper(*a, **k):
            push(label)
            try:
                return fn(*a, **k)
            finally:
                pop()

        return wrapper

    return deco
