"""ctypes binding of libprimia_hip.so.

Prototypes are generated from include/primia_hip.h, so the header is the single source of truth
for the C ABI.  There is NO fallback: if the shared library is missing or a call fails, an
exception is raised — the product path never runs on anything but the HIP kernels.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "primia_hip.h")
LIB_PATH = os.path.join(HERE, "libprimia_hip.so")

PRIMIA_F32 = 0
PRIMIA_BF16 = 1

_ERR = {-1: "PRIMIA_ERR_ARG", -2: "PRIMIA_ERR_LAUNCH", -3: "PRIMIA_ERR_UNSUPPORTED", -4: "PRIMIA_ERR_WORKSPACE"}


class PrimiaError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    """Mirror of primia_conv_desc."""

    _fields_ = [(n, ctypes.c_int32) for n in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "Ho", "Wo")]

    @classmethod
    def make(cls, N, H, W, C, K, R, S, stride, pad):
        Ho = (H + 2 * pad - R) // stride + 1
        Wo = (W + 2 * pad - S) // stride + 1
        return cls(N, H, W, C, K, R, S, stride, pad, Ho, Wo)


_SCALARS = {
    "int": ctypes.c_int,
    "int32_t": ctypes.c_int32,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "uint32_t": ctypes.c_uint32,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "primia_stream_t": ctypes.c_void_p,
}


def parse_header(path=HEADER):
    """Return {name: (restype, [(argtype, argname), ...])} for every function the header declares."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"typedef struct.*?}\s*\w+\s*;", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|int64_t|void)\s+(primia_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    parsed.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, an = a.rsplit(" ", 1)
                    ty = ty.replace("const ", "").strip()
                    parsed.append((_SCALARS[ty], an))
        rt = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "void": None}[ret]
        out[name] = (rt, parsed)
    return out


_lib = None
_protos = None


def lib():
    """Load (once) and return the ctypes library with prototypes applied."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PrimiaError(
            f"{LIB_PATH} not found — build it with `python -m primia_amd.build` "
            "(there is no CPU fallback for the product path)"
        )
    # torch bundles its own libamdhip64.so.7; import it first so that our library binds to the SAME
    # HIP runtime instance (streams and device pointers are then interchangeable).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - symbol-export test may run without torch
        pass
    _lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (rt, args) in _protos.items():
        fn = getattr(_lib, name)  # AttributeError => header/library mismatch
        fn.restype = rt
        fn.argtypes = [a for a, _ in args]
    return _lib


def protos():
    lib()
    return _protos


def _conv(a):
    import torch

    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return ctypes.c_void_p(a.data_ptr())
    if isinstance(a, ctypes.Structure):
        return ctypes.cast(ctypes.pointer(a), ctypes.c_void_p)
    return a


def current_stream():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args, stream=None):
    """Call an `int primia_*(..., stream)` entry point on torch's current stream; raise on error."""
    fn = getattr(lib(), name)
    st = current_stream() if stream is None else stream
    keep = args  # keep Structures alive for the duration of the call
    rc = fn(*[_conv(a) for a in args], st)
    del keep
    if rc != 0:
        raise PrimiaError(f"{name} failed: {_ERR.get(rc, rc)}")


def query(name, *args):
    """Call a pure host-side query (no stream argument), return its value."""
    fn = getattr(lib(), name)
    return fn(*[_conv(a) for a in args])


def set_option(name, value):
    """primia_set_option: an explicit dispatch / tuning switch of the kernel library (csrc/options.h)."""
    rc = lib().primia_set_option(name.encode(), int(value))
    if rc != 0:
        raise PrimiaError(f"primia_set_option({name!r}) failed: {_ERR.get(rc, rc)}")


def get_option(name):
    v = ctypes.c_int(0)
    rc = lib().primia_get_option(name.encode(), ctypes.cast(ctypes.pointer(v), ctypes.c_void_p))
    if rc != 0:
        raise PrimiaError(f"primia_get_option({name!r}) failed: {_ERR.get(rc, rc)}")
    return v.value


def options():
    """{name: current value} of every option the library knows."""
    out = {}
    buf = ctypes.create_string_buffer(64)
    for i in range(lib().primia_option_count()):
        lib().primia_option_name(i, ctypes.cast(buf, ctypes.c_void_p), 64)
        out[buf.value.decode()] = get_option(buf.value.decode())
    return out


def dtype_code(dt):
    import torch

    if dt == torch.float32:
        return PRIMIA_F32
    if dt == torch.bfloat16:
        return PRIMIA_BF16
    raise PrimiaError(f"unsupported dtype {dt}")
