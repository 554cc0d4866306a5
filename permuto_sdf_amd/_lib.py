"""ctypes loader for the C-ABI HIP library (include/psdf.h).  The product path has NO fallback:
if the library is missing or a symbol is absent we raise, loudly."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSDF_LIB_PATH") or os.path.join(_HERE, "lib", "libpsdf_hip.so")   # env: A/B builds of the kernels
_lib = None

c_i = ctypes.c_int
c_l = ctypes.c_int64
c_f = ctypes.c_float
c_p = ctypes.c_void_p
c_u64 = ctypes.c_uint64


class PsdfError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PsdfError(
                "permuto_sdf_amd: HIP extension %s not found. Build it with "
                "`python -m permuto_sdf_amd.build` (hipcc, gfx950). There is no CPU/PyTorch fallback." % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


class _Ptr(ctypes.c_void_p):
    """c_void_p that keeps its tensor alive for as long as the argument tuple of the call exists (a temporary
    made by `.contiguous()` must not be recycled by the caching allocator before the launch is enqueued)."""
    pass


def ptr(t):
    """Raw device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor handed to the C ABI must be contiguous"
    p = _Ptr(t.data_ptr())
    p.keepalive = t
    return p


def stream():
    """raw handle of torch's CURRENT stream.  `torch.cuda.current_stream()` builds a Stream object (10 us per call on the
    host: 1 ms of a 7 ms training step, measured with cProfile); the C binding underneath returns the pointer directly."""
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:      # a torch build without these private bindings
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(status, name):
    if status != 0:
        raise PsdfError("%s failed with status %d (%s)" % (
            name, status, "argument error" if status == -1 else "unsupported configuration" if status == -2
            else "HIP error code"))


_fns = {}


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:      # resolved once per entry point (the symbol lookup and the restype assignment cost ~1 us of every call)
        fn = getattr(lib(), name)
        fn.restype = ctypes.c_int
        _fns[name] = fn
    status = fn(*args)
    if status != 0:
        check(status, name)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PsdfError("permuto_sdf_amd ops run on the GPU only (got a %s tensor); there is no CPU path" % t.device)


# ---- small device constants without a launch each -------------------------------------------------------------------
_zero_scalars = {}
_scalar_pools = {}


def zero_scalar(device):
    """A 0-dim fp32 zero on `device` (placeholder outputs of autograd Functions).  One fill per device and process; every
    caller gets its own tensor object (detach() is a host-side view), never write into it."""
    key = str(device)
    z = _zero_scalars.get(key)
    if z is None:
        z = _zero_scalars[key] = torch.zeros((), dtype=torch.float32, device=device)
    return z.detach()


_int_pools = {}


def zeroed_int(device):
    """like zeroed_scalar, for [1] int32 counters (RaySamplesPacked.cur_nr_samples, compaction totals)"""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(1, dtype=torch.int32, device=device)
    key = str(device)
    pool = _int_pools.get(key)
    if pool is None or pool[1] >= pool[0].numel():
        pool = _int_pools[key] = [torch.zeros(1024, dtype=torch.int32, device=device), 0]
        torch.cuda.current_stream(device).synchronize()   # (see zeroed_scalar)
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1]


def zeroed_scalar(device):
    """A fresh [1] fp32 accumulator that is already zero (loss values and other wave-sum + atomic targets): slices of a
    pooled buffer, one fill per 1024 of them instead of one fill each."""
    if torch.cuda.is_current_stream_capturing():
        # inside a graph capture the zero must be a node of the graph (a memset replayed every time), not a fill that happened
        # once outside it: kernels atomicAdd into these accumulators
        return torch.zeros(1, dtype=torch.float32, device=device)
    key = str(device)
    pool = _scalar_pools.get(key)
    if pool is None or pool[1] >= pool[0].numel():
        pool = _scalar_pools[key] = [torch.zeros(1024, dtype=torch.float32, device=device), 0]
        # the fill runs on whatever stream is current NOW; later slices may be handed to work on another stream (the trainers
        # run two): wait for the fill once per 1024 accumulators rather than order every consumer after it
        torch.cuda.current_stream(device).synchronize()
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1]
