"""ctypes loader for libfsr1_hip.so (the C ABI declared in include/fsr1_hip.h).

There is no fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FSR1_HIP_LIB") or os.path.join(_HERE, "libfsr1_hip.so")  # env override: tuning experiments only
# the same objects + the launch-shape test hooks of include/fsr1_hip_test.h (tests and tuning runs only; see test_hooks())
TEST_LIB_PATH = LIB_PATH[:-3] + "_test.so" if LIB_PATH.endswith(".so") else LIB_PATH + "_test"

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
    "fsr1_pipeline_reserve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "fsr1_pipeline_next_slot": (ctypes.c_int, [ctypes.c_void_p]),
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
    "fsr1_build_id": (ctypes.c_char_p, []),
    "fsr1_timer_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "fsr1_timer_start": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "fsr1_timer_stop": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "fsr1_timer_elapsed_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "fsr1_timer_destroy": (ctypes.c_int, [ctypes.c_void_p]),
}

# include/fsr1_hip_test.h: exported by libfsr1_hip_test.so only
TEST_SYMBOLS = {
    "fsr1_debug_fused_run_steps": (None, [ctypes.c_int32]),
    "fsr1_debug_fused_tall_tiles": (None, [ctypes.c_int32]),
    "fsr1_debug_easu_tall_tiles": (None, [ctypes.c_int32]),
}

_lib = None
_test_lib = None


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from (csrc/ + include/): what a profile
    under profiles/ was taken of.  bench.py compares it with the running tree before quoting a profile's PMC traffic."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    files = []
    for d, exts in ((os.path.join(_HERE, "csrc"), (".hip", ".h", ".c", ".cpp", "Makefile")), (os.path.join(root, "include"), (".h", ".hpp"))):
        files += [os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts)]
    for f in sorted(files):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_id():
    """fsr1_build_id() of the LOADED library: the source hash baked into the binary when it was built.  Equal to source_hash() unless a
    stale prebuilt binary is running beside newer sources."""
    return load().fsr1_build_id().decode()


def build(force=False):
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        subprocess.check_call(cmd + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    for path in (LIB_PATH, TEST_LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError("build finished but %s is missing" % path)


def _open(path, symbols):
    if not os.path.exists(path):
        raise RuntimeError(
            "%s not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C fidelityfx-fsr_amd/csrc`). There is no CPU fallback." % path)
    # One HIP runtime per process: torch wheels bundle their own libamdhip64 (same SONAME as the
    # system one this library is linked to).  If torch is going to be used it must be loaded first so
    # that this library binds to the runtime torch's allocator and streams live in.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Return the loaded library with prototypes set; raises if it has not been built."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH, SYMBOLS)
    return _lib


def load_test():
    """libfsr1_hip_test.so: every symbol of the product library plus the fsr1_debug_* launch-shape hooks (include/fsr1_hip_test.h)."""
    global _test_lib
    if _test_lib is None:
        _test_lib = _open(TEST_LIB_PATH, dict(SYMBOLS, **TEST_SYMBOLS))
    return _test_lib


class test_hooks:
    """`with _lib.test_hooks() as lib:` — inside the block every call of this package (api.py goes through load()) runs in
    libfsr1_hip_test.so, whose fsr1_debug_* switches `lib` exposes; on exit the switches are back at "the host's rule" and the package
    is back on the product library.  Tests and tuning runs only: the product library has no such switches."""

    def __enter__(self):
        global _lib
        load()
        self._saved = _lib
        _lib = load_test()
        return _lib

    def __exit__(self, *exc):
        global _lib
        t = load_test()
        t.fsr1_debug_fused_run_steps(0)
        t.fsr1_debug_fused_tall_tiles(-1)
        t.fsr1_debug_easu_tall_tiles(-1)
        _lib = self._saved
        return False


class Fsr1Error(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = load().fsr1_last_error()
        raise Fsr1Error("fsr1 error %d: %s" % (rc, msg.decode() if msg else "?"))
