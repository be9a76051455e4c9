"""ctypes binding of libbsed.so (C ABI declared in include/bsed.h).

The library is built in-tree by ``csrc/build.sh`` (``__graft_entry__.build()``) and is the ONLY
compute path: ``lib()`` raises if it is missing -- there is no eager/PyTorch fallback.
"""
import ctypes
import os
import re

# The train step keeps up to four HIP streams busy (main, GRU weight gradients, the next batch's mel transform, the EMA
# teacher) and RCCL adds its own.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4):
# beyond that two streams share a queue and their kernels serialise, which takes the overlap back -- measured on one
# MI355X box, same process otherwise: 13.28 ms per step; 13.63 once an RCCL communicator exists; 13.31 with 8 queues
# (DESIGN.md section 7).  Read by the runtime when it initialises, so it is set at import, before the first HIP call;
# an explicit setting in the environment wins; importing after the runtime is up only warns (import order is not an API:
# launch scripts and the profiling tools under tools/ export the variable themselves).
def _hw_queue_default():
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return
    import sys
    t = sys.modules.get("torch")
    if t is not None and getattr(t, "cuda", None) is not None and t.cuda.is_initialized():
        # too late: the runtime has read its configuration.  Say so instead of silently running on 4 queues.
        import warnings
        warnings.warn("bsed_amd was imported after the HIP runtime was initialised and GPU_MAX_HW_QUEUES is not set: the "
                      "train step's streams and RCCL's will share the default 4 hardware queues (2.5-3.4 % slower steps "
                      "once an RCCL communicator exists, DESIGN.md section 7).  Export GPU_MAX_HW_QUEUES=8 in the "
                      "launch environment, or import bsed_amd before the first torch.cuda call.", RuntimeWarning)
        return
    os.environ["GPU_MAX_HW_QUEUES"] = "8"


_hw_queue_default()

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BSED_LIB_PATH") or os.path.join(_HERE, "libbsed.so")  # override: A/B experiment builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "bsed.h")

_lib = None


class BsedError(RuntimeError):
    pass


class MelCfg(ctypes.Structure):
    _fields_ = [("sr", ctypes.c_int), ("n_fft", ctypes.c_int), ("hop", ctypes.c_int),
                ("n_mels", ctypes.c_int), ("fmin", ctypes.c_float), ("fmax", ctypes.c_float)]


def header_symbols():
    """Every function name declared in include/bsed.h."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bsed_[a-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BsedError(f"{LIB_PATH} is missing: build it with csrc/build.sh "
                            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        L.bsed_last_error.restype = ctypes.c_char_p
        L.bsed_build_info.restype = ctypes.c_char_p
        for name in header_symbols():
            fn = getattr(L, name)  # AttributeError = header/library mismatch: fail loudly
            if name not in ("bsed_last_error", "bsed_build_info"):
                fn.restype = ctypes.c_int
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise BsedError(f"{what} failed ({rc}): {lib().bsed_last_error().decode()}")


def _require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise BsedError("bsed_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    import torch
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise BsedError("expected a GPU tensor")
    if t.device.index != torch._C._cuda_getDevice():
        # launches go to the CURRENT device's stream (stream() below): a tensor of another GPU would be dereferenced
        # by a kernel running on the wrong device
        raise BsedError(f"tensor lives on cuda:{t.device.index} but the current device is "
                        f"cuda:{torch._C._cuda_getDevice()}: wrap the call in torch.cuda.device(...)")
    if not t.is_contiguous():
        raise BsedError("expected a contiguous tensor")
    want = torch.float32 if dtype is None else dtype
    if t.dtype != want:
        raise BsedError(f"expected dtype {want}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = None


def stream():
    """torch's current stream on the current device as a hipStream_t.  (torch.cuda.current_stream() builds a Python
    Stream object: ~8 us per call, a quarter of the host time of a launch-bound small-batch step.)"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return ctypes.c_void_p(_raw_stream(torch._C._cuda_getDevice()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Per-launch timing hook (bench.py's roofline leg): ``timer`` is an ops.KernelTimer or None.  A wrapper that knows the
# algorithmic work of its launch announces it through ``pending`` = (key, flops, bytes) right before ``call``; launches
# nobody announced are recorded under their entry-point name with zero work (small glue kernels).
timer = None
pending = None


def call(name, *args):
    """Call an int-returning entry point; ints/floats are passed with explicit ctypes."""
    global pending
    fn = getattr(lib(), name)
    if timer is None:
        check(fn(*args), name)
        return
    meta, pending = pending, None
    key, flops, nbytes = meta if meta is not None else ((name, ""), 0.0, 0.0)
    timer.launch(key, flops, lambda: check(fn(*args), name), nbytes)


c_int = ctypes.c_int
c_float = ctypes.c_float
c_u64 = ctypes.c_uint64
c_void_p = ctypes.c_void_p
c_size_t = ctypes.c_size_t
