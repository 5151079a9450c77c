"""ctypes loader for libfsr1_hip.so (the C ABI declared in include/fsr1_hip.h).

There is no fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FSR1_HIP_LIB") or os.path.join(_HERE, "libfsr1_hip.so")  # env override: tuning experiments only

_U32P = ctypes.POINTER(ctypes.c_uint32)
_F = ctypes.c_float


class fsr1_image(ctypes.Structure):
    """struct fsr1_image of include/fsr1_hip.h."""
    _fields_ = [
        ("data", ctypes.c_void_p),
        ("width", ctypes.c_int32),
        ("height", ctypes.c_int32),
        ("format", ctypes.c_int32),
        ("frames", ctypes.c_int32),
        ("row_pitch_bytes", ctypes.c_int64),
        ("frame_stride_bytes", ctypes.c_int64),
    ]


class fsr1_params(ctypes.Structure):
    """struct fsr1_params of include/fsr1_hip.h."""
    _fields_ = [
        ("render_width", ctypes.c_float),
        ("render_height", ctypes.c_float),
        ("use_rcas", ctypes.c_int32),
        ("rcas_attenuation", ctypes.c_float),
        ("hdr", ctypes.c_int32),
        ("fused", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
    ]


class fsr1_color_stages(ctypes.Structure):
    """struct fsr1_color_stages of include/fsr1_hip.h."""
    _fields_ = [
        ("stages", ctypes.c_uint32),
        ("grain_amount", ctypes.c_float),
        ("grain_bias", ctypes.c_float),
        ("frame", ctypes.c_uint32),
        ("noise_offset_x", ctypes.c_int32),
        ("noise_offset_y", ctypes.c_int32),
        ("noise", ctypes.POINTER(fsr1_image)),
    ]


_IMG = ctypes.POINTER(fsr1_image)
_STG = ctypes.POINTER(fsr1_color_stages)

# name -> (restype, argtypes); every symbol include/fsr1_hip.h declares
SYMBOLS = {
    "FsrEasuCon": (None, [_U32P] * 4 + [_F] * 6),
    "FsrEasuConOffset": (None, [_U32P] * 4 + [_F] * 8),
    "FsrRcasCon": (None, [_U32P, _F]),
    "AU1_AH1_AF1": (ctypes.c_uint32, [_F]),
    "fsr1_easu_dispatch": (ctypes.c_int, [ctypes.POINTER(fsr1_image)] * 2 + [_U32P, ctypes.c_uint32, ctypes.c_void_p]),
    "fsr1_rcas_dispatch": (ctypes.c_int, [ctypes.POINTER(fsr1_image)] * 2 + [_U32P, ctypes.c_uint32, ctypes.c_void_p]),
    "fsr1_easu_rcas_fused_dispatch": (ctypes.c_int, [ctypes.POINTER(fsr1_image)] * 2 + [_U32P, _U32P, ctypes.c_uint32, ctypes.c_void_p]),
    "fsr1_easu_dispatch_band": (ctypes.c_int, [_IMG, _IMG, _U32P, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "fsr1_rcas_dispatch_band": (ctypes.c_int, [_IMG, _IMG, _U32P, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "fsr1_easu_rcas_fused_dispatch_band": (ctypes.c_int, [_IMG, _IMG, _U32P, _U32P, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "fsr1_color_dispatch": (ctypes.c_int, [_IMG, _IMG, _STG, ctypes.c_uint32, ctypes.c_void_p]),
    "fsr1_easu_dispatch_ex": (ctypes.c_int, [_IMG, _IMG, _U32P, ctypes.c_uint32, _STG, ctypes.c_void_p]),
    "fsr1_rcas_dispatch_ex": (ctypes.c_int, [_IMG, _IMG, _U32P, ctypes.c_uint32, _STG, ctypes.c_void_p]),
    "fsr1_easu_rcas_fused_dispatch_ex": (ctypes.c_int, [_IMG, _IMG, _U32P, _U32P, ctypes.c_uint32, _STG, ctypes.c_void_p]),
    "fsr1_upscale": (ctypes.c_int, [ctypes.POINTER(fsr1_image)] * 3 + [ctypes.POINTER(fsr1_params), ctypes.c_void_p]),
    "fsr1_upscale_ex": (ctypes.c_int, [_IMG] * 3 + [ctypes.POINTER(fsr1_params), _STG, ctypes.c_void_p]),
    "fsr1_upscale_plan": (ctypes.c_int, [_IMG, ctypes.c_int32, _IMG, ctypes.POINTER(fsr1_params), ctypes.c_int32]),
    "fsr1_pipeline_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32]),
    "fsr1_pipeline_upscale": (ctypes.c_int, [ctypes.c_void_p, _IMG, _IMG, ctypes.POINTER(fsr1_params), _STG]),
    "fsr1_pipeline_fork": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "fsr1_pipeline_join": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "fsr1_pipeline_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "fsr1_pipeline_streams": (ctypes.c_int32, [ctypes.c_void_p]),
    "fsr1_pipeline_stream": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int32]),
    "fsr1_pipeline_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "fsr1_last_error": (ctypes.c_char_p, []),
    "fsr1_version": (ctypes.c_int, []),
    "fsr1_device_count": (ctypes.c_int, []),
    "fsr1_selftest": (ctypes.c_int, [_U32P]),
    "fsr1_debug_fused_run_steps": (None, [ctypes.c_int32]),
    "fsr1_debug_fused_tall_tiles": (None, [ctypes.c_int32]),
    "fsr1_debug_easu_tall_tiles": (None, [ctypes.c_int32]),
    "fsr1_timer_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "fsr1_timer_start": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "fsr1_timer_stop": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "fsr1_timer_elapsed_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "fsr1_timer_destroy": (ctypes.c_int, [ctypes.c_void_p]),
}

_lib = None


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from (csrc/ + include/): what a profile
    under profiles/ was taken of.  bench.py compares it with the running tree before quoting a profile's PMC traffic."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    files = []
    for d, exts in ((os.path.join(_HERE, "csrc"), (".hip", ".h", ".c", "Makefile")), (os.path.join(root, "include"), (".h", ".hpp"))):
        files += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts)]
    for f in sorted(files):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build(force=False):
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        subprocess.check_call(cmd + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build finished but %s is missing" % LIB_PATH)


def load():
    """Return the loaded library with prototypes set; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C fidelityfx-fsr_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: torch wheels bundle their own libamdhip64 (same SONAME as the
        # system one this library is linked to).  If torch is going to be used it must be loaded first so
        # that this library binds to the runtime torch's allocator and streams live in.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class Fsr1Error(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = load().fsr1_last_error()
        raise Fsr1Error("fsr1 error %d: %s" % (rc, msg.decode() if msg else "?"))
