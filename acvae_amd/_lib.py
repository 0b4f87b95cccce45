"""ctypes binding of libacvae_hip.so, generated from include/acvae_hip.h at import time so that the
Python side can never drift from the C ABI.  There is NO fallback: if the library is missing or a
symbol declared in the header is not exported, importing/using the product path raises."""
import ctypes
import os
import re

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "acvae_hip.h")
LIB_PATH = os.path.join(_HERE, "libacvae_hip.so")      # the product library; tools load an A/B build with use_library()

_CT = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double,
       "uint64_t": ctypes.c_uint64}
ERRORS = {-1: "ACVAE_EINVAL (bad dims / null pointer)", -2: "ACVAE_EALIGN (alignment)",
          -3: "ACVAE_EUNSUPPORTED", -4: "ACVAE_EWORKSPACE (workspace too small)"}


def parse_header(path=HEADER):
    """-> {name: (restype, [(argname, ctype)])} for every `int|int64_t acvae_*(...)` prototype, and {enum: value}."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t)\s+(acvae_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    alist.append((a.split("*")[-1].strip(), ctypes.c_void_p))
                else:
                    toks = a.replace("const ", "").split()
                    alist.append((toks[-1], _CT[toks[0]]))
        protos[name] = (_CT[ret], alist)
    enums = {}
    for m in re.finditer(r"enum\s+\w*\s*\{([^}]*)\}", src):
        val = -1
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = item.split("=")
                val = int(v.strip(), 0)
                enums[k.strip()] = val
            else:
                val += 1
                enums[item] = val
    return protos, enums


PROTOS, ENUMS = parse_header()
_defs = dict(re.findall(r"#define\s+(ACVAE_\w+)\s+(\d+)", open(HEADER).read()))
ENUMS_TEXT_N = int(_defs["ACVAE_TEXT_NPARAMS"])
ENC_NPARAMS = int(_defs["ACVAE_ENC_NPARAMS"])
ENC_BF16 = int(_defs["ACVAE_ENC_BF16"])
FLAG_NO_PERSIST = int(_defs["ACVAE_FLAG_NO_PERSIST"])
FLAG_DEFER_PARAM_GRADS = int(_defs["ACVAE_FLAG_DEFER_PARAM_GRADS"])
FLAG_NO_ATTN_SPLIT = int(_defs["ACVAE_FLAG_NO_ATTN_SPLIT"])
FLAG_TEST_STALL = int(_defs["ACVAE_FLAG_TEST_STALL"])
_lib = None


class options:
    """Host-side defaults for the per-call `flags` of the C ABI (the library itself keeps no switches).  Tests and the
    A/B tools change them through `with _lib.override(persist=False): ...`; the environment variables of the earlier
    rounds still set the initial values."""
    persist = os.environ.get("ACVAE_DECODE_PERSIST", "1") != "0"      # persistent decode / posterior launches
    defer = os.environ.get("ACVAE_DECODE_DEFER", "1") != "0"          # decode backward: parameter gradients trail on the side stream
    attn_split = True                                                 # split-over-frames attention for few query rows
    test_stall = False                                                # ACVAE_FLAG_TEST_STALL (tests only)


class override:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: getattr(options, k) for k in self.kw}
        for k, v in self.kw.items():
            setattr(options, k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            setattr(options, k, v)
        return False


def call_flags(defer=False):
    f = 0
    if not options.persist:
        f |= FLAG_NO_PERSIST
    if not options.attn_split:
        f |= FLAG_NO_ATTN_SPLIT
    if options.test_stall:
        f |= FLAG_TEST_STALL
    if defer and options.defer:
        f |= FLAG_DEFER_PARAM_GRADS
    return f


def use_library(path):
    """Load another build of the library (tools/ab_build.py) instead of the product one.  Explicit and process-wide: must
    be called before the first library call."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("acvae_amd._lib.use_library() must come before the first library call")
    LIB_PATH = os.path.abspath(path)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the AC-VAE HIP path has no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (ret, args) in PROTOS.items():
            fn = getattr(_lib, name)            # AttributeError if the header declares what the .so lacks
            fn.restype = ret
            fn.argtypes = [t for _, t in args]
    return _lib


def _conv(x):
    if isinstance(x, torch.Tensor):
        return x.data_ptr()
    return x


def current_stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Call an `int acvae_*` entry point; tensors are passed as device pointers; raises on non-zero."""
    fn = getattr(lib(), name)
    rc = fn(*[_conv(a) for a in args])
    if PROTOS[name][0] is ctypes.c_int and rc != 0:
        what = ERRORS.get(rc, f"hipError_t {rc}" if rc > 0 else f"code {rc}")
        raise RuntimeError(f"{name} failed: {what}")
    return rc


# ---- persistent launches: status words (include/acvae_hip.h, acvae_persist_status_register)
_STATUS = {}
_STATUS_NAMES = ("decode forward", "decode backward", "posterior forward", "posterior backward")


def persist_status(dev):
    """The device's status words (page-locked int32[8], registered with the library on first use)."""
    dev = torch.device(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _STATUS.get(key)
    if t is None:
        t = torch.zeros(8, dtype=torch.int32).pin_memory()
        call("acvae_persist_status_register", key, t.data_ptr())
        _STATUS[key] = t
    return t


def check_persist_status(dev):
    """Raise if a persistent launch on `dev` gave up since the last check.  Meaningful once the streams that carried the
    launches have drained (TrainStep calls it behind its in-flight event); the launch's outputs are NaN either way."""
    t = persist_status(dev)
    if int(t[4]) != 0:
        which = [n for k, n in enumerate(_STATUS_NAMES) if int(t[k]) != 0]
        t.zero_()
        raise RuntimeError("acvae_amd: a persistent launch could not complete (" + ", ".join(which) + "): part of its grid was "
                           "never resident - is another process running persistent kernels on this GPU?  Its outputs were "
                           "overwritten with NaN.  ACVAE_DECODE_PERSIST=0 selects the per-step launches.")


# ---- workspaces of acvae_attn_fwd's split-over-frames form: zeroed once, one per (device, stream)
_ATTN_WS = {}


def attn_fwd_workspace(N, Tq, S, A, E, dev):
    """(tensor or None, bytes) for acvae_attn_fwd on the current stream.  The counters at its head must start zeroed and are
    left zeroed by every call, so the buffer is cached per stream and only re-made (zeroed) when it has to grow."""
    nbytes = call("acvae_attn_fwd_workspace_bytes", N, Tq, S, A, E)
    if nbytes <= 0 or not options.attn_split:
        return None, 0
    dev = torch.device(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), current_stream())
    t = _ATTN_WS.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.zeros(int(nbytes), dtype=torch.uint8, device=dev)
        _ATTN_WS[key] = t
    return t, t.numel()


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("acvae_amd: the HIP path needs tensors on an MI355X device (no CPU fallback)")


class _PinnedRing:
    """One page-locked staging buffer per device, handed out as a ring.  A freshly hipHostMalloc'ed block makes the
    first copy out of it wait for the GPU to drain, and torch's caching host allocator needs a fresh block whenever
    the host runs ahead of the previous copies, so the per-step host->device copies stage through memory that is
    pinned once.  A slot is reused only after the copy that last read it has passed (event), which blocks only when
    the host is a whole ring ahead of the GPU."""

    def __init__(self, nbytes=32 << 20):
        self.buf = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        self.head = 0
        self.live = []           # (start, end, event) in allocation order

    def _reserve(self, n):
        span = (n + 255) & ~255
        if self.head + span > self.buf.numel():
            self.head = 0
        s, e = self.head, self.head + span
        self.head = e
        # every live entry that overlaps the slot must have been copied out (not only the oldest one: after a wrap an
        # old tail entry the new lap never reaches can sit in front of newer ones that do overlap); entries whose copy
        # has already passed are dropped on the way so the list stays short
        keep = []
        for (ls, le, ev) in self.live:
            if ls < e and le > s:
                ev.synchronize()
            elif not ev.query():
                keep.append((ls, le, ev))
        self.live = keep
        return s, e

    def _send(self, slot, s, e, dev):
        out = slot.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.live.append((s, e, ev))
        return out

    def stage(self, t, dev):
        n = t.numel() * t.element_size()
        if n == 0 or n > self.buf.numel() // 4:
            return t.pin_memory().to(dev, non_blocking=True)
        s, e = self._reserve(n)
        slot = self.buf[s:s + n].view(t.dtype).view(t.shape)
        np.copyto(slot.numpy(), t.numpy())       # plain memcpy: torch's copy_ fans a 1 MB copy out over the intra-op pool
        return self._send(slot, s, e, dev)

    def fill(self, shape, dtype, dev, fill):
        """Reserve a slot, let ``fill(slot_tensor)`` produce the data in place (e.g. torch.randn(..., out=)), send it."""
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        if n == 0 or n > self.buf.numel() // 4:
            t = torch.empty(shape, dtype=dtype)
            fill(t)
            return t.pin_memory().to(dev, non_blocking=True)
        s, e = self._reserve(n)
        slot = self.buf[s:s + n].view(dtype).view(shape)
        fill(slot)
        return self._send(slot, s, e, dev)


_RINGS = {}


def h2d(t, dev, dtype=None):
    """Host array/tensor -> device without stalling the host: a copy out of pageable memory waits until everything
    queued before it has run, so stage through the page-locked ring and copy asynchronously."""
    t = torch.as_tensor(t)
    if t.device.type != "cpu":
        return t.to(device=dev, dtype=dtype)
    if dtype is not None:
        t = t.to(dtype)
    dev = torch.device(dev)
    if dev.type != "cuda":
        raise RuntimeError("acvae_amd: the HIP path needs a GPU device (no CPU fallback)")
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ring = _RINGS.get(key)
    if ring is None:
        ring = _RINGS[key] = _PinnedRing()
    return ring.stage(t.contiguous(), dev)


def h2d_fill(shape, dtype, dev, fill):
    """Like h2d, but the host data is produced by ``fill(tensor)`` directly in the page-locked staging slot."""
    dev = torch.device(dev)
    if dev.type != "cuda":
        raise RuntimeError("acvae_amd: the HIP path needs a GPU device (no CPU fallback)")
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ring = _RINGS.get(key)
    if ring is None:
        ring = _RINGS[key] = _PinnedRing()
    return ring.fill(tuple(shape), dtype, dev, fill)
