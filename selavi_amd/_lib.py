"""ctypes binding of libselavi_hip.so.

The signatures are derived from ``include/selavi_hip.h`` itself, so the Python side cannot drift
from the declared C ABI.  There is NO CPU fallback: if the library is missing or a call fails the
caller gets an exception (``SelaviHipError``).
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "selavi_hip.h")
LIBPATH = os.environ.get("SELAVI_HIP_LIB") or os.path.join(HERE, "libselavi_hip.so")   # override: A/B builds


class SelaviHipError(RuntimeError):
    pass


_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t,
    "double": ctypes.c_double, "float": ctypes.c_float, "slv_stream_t": ctypes.c_void_p,
    "slv_comm_t": ctypes.c_void_p,
    "unsigned": ctypes.c_uint, "uint64_t": ctypes.c_uint64,
}


def _ctype(t):
    t = t.replace("const", "").strip()
    if t.endswith("*"):
        return ctypes.c_char_p if t.replace(" ", "") == "char*" and False else ctypes.c_void_p
    return _SCALARS[t]


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(type_str, arg_name), ...])} for every ``slv_*`` declaration."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    decls = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(slv_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?[\*\s])(\w+)$", a)
                parsed.append((mm.group(1).strip(), mm.group(2)))
        decls[name] = (ret, parsed)
    return decls


_lib = None
_decls = None


def load():
    global _lib, _decls
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise SelaviHipError(
            f"{LIBPATH} is missing: run `python -m selavi_amd.build` (hipcc, gfx950). "
            "selavi_amd has no CPU fallback.")
    # ONE HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and must be in the process first, so that the
    # library's NEEDED entry resolves to that copy; loaded the other way round, the kernels of this library would launch
    # through a second runtime (/opt/rocm's) that shares nothing with the tensors torch allocates
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIBPATH)
    _decls = parse_header()
    for name, (ret, args) in _decls.items():
        fn = getattr(lib, name)            # AttributeError here == header/library drift
        fn.argtypes = [_ctype(t) for t, _ in args]
        r = ret.replace("const", "").strip()
        if r.replace(" ", "") == "char*":
            fn.restype = ctypes.c_char_p
        elif r.endswith("*"):
            fn.restype = ctypes.c_void_p
        else:
            fn.restype = _SCALARS[r]
    _lib = lib
    return lib


def declared_symbols():
    return sorted(parse_header().keys())


def check(rc, what=""):
    if rc != 0:
        msg = load().slv_last_error()
        raise SelaviHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


class _Caller:
    """``C.slv_foo(args...)`` -> calls the entry point, raises on a non-zero int return."""

    def __getattr__(self, name):
        lib = load()
        fn = getattr(lib, name)
        ret = _decls[name][0].strip()

        if ret == "int":
            def call(*a):
                rc = fn(*a)
                if rc != 0:
                    check(rc, name)
            call.__name__ = name
        else:
            call = fn
        setattr(self, name, call)
        return call


C = _Caller()


def ptr(t):
    """data_ptr of a torch tensor (must be contiguous) or 0 for None."""
    if t is None:
        return 0
    assert t.is_contiguous(), "selavi_amd kernels take dense row-major tensors"
    return t.data_ptr()


_raw_stream = None


def stream(index=None):
    """hipStream_t of torch's current stream on the current device (or on device ``index``), as an int.  Through the raw getters: the public
    torch.cuda.current_stream() costs ~8 us a call (device-index resolution, a Stream object), ~350 calls per step --
    a quarter of the host time of the launch-bound 16-clip step (tools/host_profile.py)."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        get_dev = getattr(torch._C, "_cuda_getDevice", None)
        get_raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if get_dev is not None and get_raw is not None:
            _raw_stream = lambda i=None: get_raw(get_dev() if i is None else i)
        else:
            _raw_stream = lambda i=None: torch.cuda.current_stream(i).cuda_stream
    return _raw_stream(index)
