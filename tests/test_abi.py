"""The C-ABI library loads, exports every symbol include/fsr1_hip.h declares, and validates its
arguments — no compute calls, runs without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_symbols(header="fsr1_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:void|int|uint32_t|char\s*\*|const char\s*\*)\s*\*?\s*([A-Za-z_][A-Za-z0-9_]*)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_symbols_are_exported(fsr):
    lib = ctypes.CDLL(fsr._lib.LIB_PATH)
    syms = header_symbols()
    assert {"FsrEasuCon", "FsrEasuConOffset", "FsrRcasCon", "fsr1_easu_dispatch", "fsr1_rcas_dispatch",
            "fsr1_easu_rcas_fused_dispatch", "fsr1_upscale", "fsr1_last_error"} <= set(syms)
    for s in syms:
        assert hasattr(lib, s), "include/fsr1_hip.h declares %s but libfsr1_hip.so does not export it" % s
    assert set(syms) == set(fsr._lib.SYMBOLS), "python prototypes and header disagree"


def test_test_hooks_live_in_the_test_library_only(fsr):
    """include/fsr1_hip_test.h's process-wide launch-shape switches are exported by libfsr1_hip_test.so and NOT by the product
    library, whose header promises no such state; the test library otherwise exports the whole product ABI."""
    prod = ctypes.CDLL(fsr._lib.LIB_PATH)
    test = ctypes.CDLL(fsr._lib.TEST_LIB_PATH)
    hooks = header_symbols("fsr1_hip_test.h")
    assert set(hooks) == set(fsr._lib.TEST_SYMBOLS) and len(hooks) == 3
    for s in hooks:
        assert hasattr(test, s), "libfsr1_hip_test.so does not export %s" % s
        assert not hasattr(prod, s), "the product library exports the test hook %s" % s
    for s in header_symbols():
        assert hasattr(test, s), "libfsr1_hip_test.so lacks %s" % s
    blob = open(fsr._lib.LIB_PATH, "rb").read()
    assert b"fsr1_debug_" not in blob


def test_build_id_is_the_hash_of_the_sources(fsr):
    """fsr1_build_id() — baked into the binary by csrc/Makefile — equals _lib.source_hash() of the tree it was built from: the bench
    line's way to notice a stale prebuilt library running beside newer sources."""
    assert fsr._lib.build_id() == fsr._lib.source_hash()
    test = fsr._lib.load_test()
    assert test.fsr1_build_id().decode() == fsr._lib.source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", fsr._lib.build_id())


def test_pipeline_argument_validation(fsr):
    """fsr1_pipeline_reserve / _next_slot reject a null pipeline before touching the device (no GPU needed)."""
    lib = fsr.load()
    assert lib.fsr1_pipeline_reserve(None, 1 << 20) == -1 and b"null" in lib.fsr1_last_error()
    assert lib.fsr1_pipeline_next_slot(None) == -1


def test_version(fsr):
    assert fsr.load().fsr1_version() == 100


def test_argument_validation(fsr):
    lib = fsr.load()
    img = fsr.fsr1_image(0, 16, 16, 0, 1, 0, 0)
    con = np.zeros(16, np.uint32)
    p = con.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    assert lib.fsr1_easu_dispatch(None, None, p, 0, None) == -1
    assert b"null" in lib.fsr1_last_error()
    assert lib.fsr1_easu_dispatch(ctypes.byref(img), ctypes.byref(img), p, 0, None) == -1  # null data
    bad = fsr.fsr1_image(0x1000, 16, 16, 7, 1, 0, 0)
    assert lib.fsr1_easu_dispatch(ctypes.byref(bad), ctypes.byref(bad), p, 0, None) == -2  # unsupported format
    ok = fsr.fsr1_image(0x1000, 16, 16, 0, 1, 0, 0)
    ok2 = fsr.fsr1_image(0x100000, 32, 32, 0, 1, 0, 0)
    assert lib.fsr1_easu_dispatch(ctypes.byref(ok), ctypes.byref(ok2), p, 1 << 12, None) == -1  # unknown flag (bit 11 is FSR1_FLAG_FRAMES_OVERLAP)
    assert b"unknown flag" in lib.fsr1_last_error()
    assert lib.fsr1_easu_dispatch(ctypes.byref(ok), ctypes.byref(ok2), p, (1 << 9) | (1 << 10), None) == -1  # store policies are exclusive
    assert b"OUTPUT" in lib.fsr1_last_error()
    assert lib.fsr1_easu_dispatch(ctypes.byref(ok), ctypes.byref(ok2), p, (1 << 4) | (1 << 5), None) == -1  # exclusive
    assert lib.fsr1_easu_dispatch(ctypes.byref(ok), ctypes.byref(ok), p, 0, None) == -1  # aliasing
    assert b"overlap" in lib.fsr1_last_error()
    assert lib.fsr1_easu_dispatch(ctypes.byref(ok), ctypes.byref(ok2), p, 0, None) == -1  # con0 scale = 0
    assert lib.fsr1_rcas_dispatch(ctypes.byref(ok), ctypes.byref(ok2), p, 0, None) == -1  # extents differ
    short = fsr.fsr1_image(0x1000, 16, 16, 0, 1, 64, 0)
    assert lib.fsr1_rcas_dispatch(ctypes.byref(short), ctypes.byref(ok2), p, 0, None) == -1  # pitch < row
    odd = fsr.fsr1_image(0x1004, 16, 16, 0, 1, 0, 0)
    assert lib.fsr1_rcas_dispatch(ctypes.byref(odd), ctypes.byref(ok2), p, 0, None) == -1  # misaligned


def test_no_cpu_fallback(fsr):
    """Host tensors are refused: the product has no CPU path."""
    import torch
    with pytest.raises(fsr.Fsr1Error):
        fsr.image_of(torch.zeros(4, 4, 4, dtype=torch.float16))


def test_product_does_not_link_the_oracle(fsr):
    """Nothing under the package or the library refers to oracle/."""
    blob = open(fsr._lib.LIB_PATH, "rb").read()
    assert b"oracle_" not in blob and b"ref_easu" not in blob
    pkg_dir = os.path.dirname(fsr._lib.LIB_PATH)
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".c", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "cpu_oracle" not in src and "libfsr1_oracle" not in src and "libfsr1_ref" not in src, f


def test_colour_stage_validation(fsr):
    """fsr1_color_dispatch / *_ex reject bad stage descriptors before anything is launched (no GPU needed)."""
    lib = fsr.load()
    a = fsr.fsr1_image(0x1000, 16, 16, 0, 1, 0, 0)
    b = fsr.fsr1_image(0x100000, 16, 16, 0, 1, 0, 0)
    S = fsr._lib.fsr1_color_stages

    def call(st, flags=0, out=b):
        return lib.fsr1_color_dispatch(ctypes.byref(a), ctypes.byref(out), ctypes.byref(st), flags, None)

    assert call(S(1 << 9, 0, 0, 0, 0, 0, None)) == -1 and b"unknown colour stage" in lib.fsr1_last_error()
    assert call(S(8 | 16, 0, 0, 0, 0, 0, None)) == -1 and b"exclusive" in lib.fsr1_last_error()
    assert call(S(32, 0, 0, 0, 0, 0, None)) == -1 and b"TEPD" in lib.fsr1_last_error()
    assert call(S(2, 0.5, 0, 0, 0, 0, None)) == -1 and b"noise" in lib.fsr1_last_error()
    assert call(S(1, 0, 0, 0, 0, 0, None), flags=1) == -1                      # only MATH_EXACT is a colour-pass flag
    big = fsr.fsr1_image(0x100000, 32, 16, 0, 1, 0, 0)
    assert call(S(1, 0, 0, 0, 0, 0, None), out=big) == -1 and b"extents" in lib.fsr1_last_error()
    con = np.zeros(16, np.uint32)
    p = con.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    f32 = fsr.fsr1_image(0x100000, 16, 16, 1, 1, 0, 0)
    st = S(4, 0, 0, 0, 0, 0, None)
    assert lib.fsr1_rcas_dispatch_ex(ctypes.byref(a), ctypes.byref(f32), p, 0, ctypes.byref(st), None) == -2
    assert b"format pair" in lib.fsr1_last_error()
    assert lib.fsr1_rcas_dispatch_ex(ctypes.byref(a), ctypes.byref(b), p, 1 << 5, ctypes.byref(st), None) == -2  # packed fp16 + stages


def test_header_is_plain_c(tmp_path):
    """include/fsr1_hip.h is the boundary a C host binds: it must compile as C99 and as C++ with nothing but itself."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "fsr1_hip.h"\nint main(void) { fsr1_image i = {0}; fsr1_color_stages s = {0}; fsr1_params p = {0};'
                   ' return (int)(sizeof i + sizeof s + sizeof p) == 0 || FSR1_HIP_VERSION != 100; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)])


def test_upscale_parameter_validation(fsr):
    """fsr1_upscale rejects an unknown pipeline selector and stray flag bits before anything is launched."""
    lib = fsr.load()
    a = fsr.fsr1_image(0x1000, 16, 16, 0, 1, 0, 0)
    b = fsr.fsr1_image(0x100000, 32, 32, 0, 1, 0, 0)
    m = fsr.fsr1_image(0x200000, 32, 32, 0, 1, 0, 0)
    P = fsr._lib.fsr1_params
    bad = P(16.0, 16.0, 1, 0.25, 0, 3, 0)            # fused must be 0, 1 or 2
    assert lib.fsr1_upscale(ctypes.byref(a), ctypes.byref(m), ctypes.byref(b), ctypes.byref(bad), None) == -1
    assert b"fused" in lib.fsr1_last_error()
    bad = P(16.0, 16.0, 1, 0.25, 0, 0, 1)            # HDR_SQUARE is derived from `hdr`, not passed in flags
    assert lib.fsr1_upscale(ctypes.byref(a), ctypes.byref(m), ctypes.byref(b), ctypes.byref(bad), None) == -1
    ok = P(16.0, 16.0, 1, 0.25, 0, 0, 1 << 8)        # NO_FAST_PATHS is a legal arithmetic-selection bit; no intermediary given
    assert lib.fsr1_upscale(ctypes.byref(a), None, ctypes.byref(b), ctypes.byref(ok), None) == -1
    assert b"intermediary" in lib.fsr1_last_error()


def test_upscale_plan_auto_rule(fsr):
    """fsr1_upscale_plan: the pipeline `fused = 2` (auto) takes, decided on the host without a launch — the fused quad form at
    exactly 2x, the fused launch for launch-bound frames, two dispatches for large frames at other ratios, and two
    dispatches (not an error) where the fused tile would not fit a CU's LDS but an intermediary was supplied."""
    lib = fsr.load()
    P = fsr._lib.fsr1_params

    def plan(iw, ih, ow, oh, fused=2, have_mid=1, flags=0, use_rcas=1, frames=1, stages=0):
        a = fsr.fsr1_image(0x1000, iw, ih, 0, frames, 0, 0)
        b = fsr.fsr1_image(0x10000000, ow, oh, 0, frames, 0, 0)
        return lib.fsr1_upscale_plan(ctypes.byref(a), have_mid, ctypes.byref(b), ctypes.byref(P(float(iw), float(ih), use_rcas, 0.25, 0, fused, flags)), stages)

    assert plan(1920, 1080, 3840, 2160) == 1                      # exactly 2x: the quad-form single launch
    assert plan(1920, 1080, 3840, 2160, flags=1 << 8) == 0        # NO_FAST_PATHS: generic kernels, 8.3 Mpixel -> two dispatches
    assert plan(1920, 1080, 3840, 2160, flags=1 << 5) == 0        # packed fp16, one 4K frame: the two H dispatches (93.7 vs 95.6 us)
    assert plan(1280, 720, 2560, 1440, flags=1 << 5) == 1         # ... up to 4 Mpixel its exact-2x fused launch (50.9 vs 47.0 us)
    assert plan(1920, 1080, 3840, 2160, flags=1 << 5, frames=4) == 0 and plan(1920, 1080, 3840, 2160, flags=1 << 5, frames=8) == 1  # and from 60 Mpixel up
    assert plan(3840, 2160, 7680, 4320, flags=1 << 5, frames=16) == 1 and plan(3840, 2160, 7680, 4320, flags=1 << 5) == 0
    assert plan(1280, 720, 2560, 1440, flags=(1 << 5) | (1 << 8)) == 0  # (NO_FAST_PATHS: the generic H kernels, 3.7 Mpixel -> two dispatches)
    assert plan(1920, 1080, 3840, 2160, stages=1) == 0            # colour stages neither
    assert plan(2560, 1440, 3840, 2160) == 0                      # 1.5x at 4K: two dispatches
    assert plan(1280, 720, 1920, 1080) == 1                       # 1.5x at 1080p: launch-bound, fused
    assert plan(1280, 720, 1920, 1080, frames=4) == 0             # ... but not four of them in one launch
    assert plan(2560, 1440, 3840, 2160, have_mid=0) == 1          # no intermediary: only the fused launch can run
    assert plan(2560, 1440, 3840, 2160, fused=0) == 0 and plan(2560, 1440, 3840, 2160, fused=1) == 1
    assert plan(2560, 1440, 3840, 2160, use_rcas=0) == 2          # EASU only
    # about 2x minification: EASU's tile fits the LDS, the fused tile (one-pixel apron, intermediate tile) does not
    assert plan(1920, 1080, 960, 540) == 0
    assert plan(1920, 1080, 960, 540, flags=1 << 5) == 0
    assert plan(1920, 1080, 3840, 2160, fused=3) == -1 and b"fused" in lib.fsr1_last_error()
    assert lib.fsr1_upscale_plan(None, 1, None, None, 0) == -1
    # the plan refuses what the call refuses (one validation, upscale_decide): render size outside the input, illegal flag
    # combinations and unknown bits, packed fp16 on a non-RGBA16F image or with colour stages, two dispatches without an intermediary
    def plan_raw(a, b, prm, have_mid=1, stages=0):
        return lib.fsr1_upscale_plan(ctypes.byref(a), have_mid, ctypes.byref(b), ctypes.byref(prm), stages)
    img = lambda w, h, fmt=0, frames=1: fsr.fsr1_image(0x1000, w, h, fmt, frames, 0, 0)  # noqa: E731
    assert plan_raw(img(1920, 1080), img(3840, 2160), P(2000.0, 1080.0, 1, 0.25, 0, 2, 0)) == -1 and b"render size" in lib.fsr1_last_error()
    assert plan_raw(img(1920, 1080), img(3840, 2160), P(1920.0, 1080.0, 1, 0.25, 0, 2, (1 << 4) | (1 << 5))) == -1 and b"exclusive" in lib.fsr1_last_error()
    assert plan_raw(img(1920, 1080), img(3840, 2160), P(1920.0, 1080.0, 1, 0.25, 0, 2, 1 << 20)) == -1 and b"unknown flag" in lib.fsr1_last_error()
    assert plan_raw(img(1920, 1080), img(3840, 2160), P(1920.0, 1080.0, 1, 0.25, 0, 2, 1 << 0)) == -1 and b"params.flags" in lib.fsr1_last_error()
    assert plan_raw(img(1920, 1080, 1), img(3840, 2160, 1), P(1920.0, 1080.0, 1, 0.25, 0, 2, 1 << 5)) == -2 and b"RGBA16F" in lib.fsr1_last_error()
    assert plan_raw(img(1920, 1080), img(3840, 2160), P(1920.0, 1080.0, 1, 0.25, 0, 2, 1 << 5), stages=1) == -2
    assert plan_raw(img(1920, 1080), img(3840, 2160), P(1920.0, 1080.0, 1, 0.25, 0, 0, 0), have_mid=0) == -1 and b"intermediary" in lib.fsr1_last_error()
    assert plan_raw(img(1920, 1080, 0, 2), img(3840, 2160, 0, 3), P(1920.0, 1080.0, 1, 0.25, 0, 2, 0)) == -1 and b"frame counts" in lib.fsr1_last_error()
    assert plan_raw(img(1920, 1080, 7), img(3840, 2160), P(1920.0, 1080.0, 1, 0.25, 0, 2, 0)) == -2 and b"format" in lib.fsr1_last_error()
    assert plan_raw(img(1920, 1080), img(0, 2160), P(1920.0, 1080.0, 1, 0.25, 0, 2, 0)) == -1
    # a partial viewport (dynamic resolution: render size smaller than the resource) is what the call accepts, so the plan does too
    assert plan_raw(img(1920, 1080), img(3840, 2160), P(1280.0, 720.0, 1, 0.25, 0, 2, 0)) == 0


@pytest.mark.gpu
def test_roctx_ranges_can_be_switched_on():
    """FSR1_ROCTX=1 wraps every dispatch in a roctx range (resolved with dlopen; SURVEY.md section 5 tracing): the smoke
    upscale still matches the oracle with it on."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, FSR1_ROCTX="1")
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "smoke ok" in out.stdout, out.stdout + out.stderr


def test_every_abi_symbol_is_documented():
    """INTEGRATION.md / DESIGN.md / README.md name every function include/fsr1_hip.h declares (a binder finds each one explained)."""
    docs = "".join(open(os.path.join(ROOT, f)).read() for f in ("INTEGRATION.md", "DESIGN.md", "README.md"))
    missing = [s for s in header_symbols() if s not in docs]
    assert not missing, missing
