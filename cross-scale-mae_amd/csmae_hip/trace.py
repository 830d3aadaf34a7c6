"""roctx ranges around the phases of a step (SURVEY §5 tracing row): forward / junction / backward (decoder, encoder) / optimizer / exchange.

`rocprofv3 --marker-trace --kernel-trace --hip-runtime-trace` then attributes every kernel to the phase whose host range its launch call fell
into (tools/step_timeline.py --markers) instead of guessing from kernel names.  The ranges are host-side calls into libroctx64 (~100 ns each
without a profiler attached); when the library is not installed they are no-ops."""
from __future__ import annotations

import contextlib
import ctypes

_lib = None
_tried = False


def _load():
    global _lib, _tried
    if not _tried:
        _tried = True
        for name in ("librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"):   # (rocprofv3 intercepts the rocprofiler-sdk one)
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                _lib = lib
                break
            except (OSError, AttributeError):
                continue
    return _lib


def push(name: str):
    lib = _load()
    if lib is not None:
        lib.roctxRangePushA(name.encode())


def pop():
    lib = _load()
    if lib is not None:
        lib.roctxRangePop()


@contextlib.contextmanager
def range_(name: str):
    push(name)
    try:
        yield
    finally:
        pop()
