"""TEST INFRASTRUCTURE — ctypes bindings for the CPU oracles.

Two libraries, same call shapes:
  * ``port``  oracle/libfsr1_oracle.so    plain-C restatement (oracle/fsr1_oracle.c)
  * ``ref``   oracle/_ref/libfsr1_ref.so  the reference headers compiled verbatim (oracle/build_ref.sh)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Images are numpy float32 arrays of shape (H, W, 4), RGBA.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_U32P = ctypes.POINTER(ctypes.c_uint32)
_FP = ctypes.POINTER(ctypes.c_float)
_F = ctypes.c_float
_I = ctypes.c_int

RCAS_DENOISE = 1
RCAS_ALPHA = 2
HDR_SQUARE = 4
# colour stages (same bit values as FSR1_COLOR_* in include/fsr1_hip.h)
COLOR_SRTM = 1
COLOR_LFGA = 2
COLOR_SRTM_INV = 4
COLOR_TEPD_C8 = 8
COLOR_TEPD_C10 = 16
COLOR_DITHER_FROM_NOISE = 32


def build(force=False):
    """Compile the plain-C oracle and (only where /root/reference exists) oracle/_ref."""
    so = os.path.join(_HERE, "libfsr1_oracle.so")
    src = os.path.join(_HERE, "fsr1_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libfsr1_oracle.so"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libfsr1_ref.so")
    if os.path.exists("/root/reference/ffx-fsr/ffx_fsr1.h"):
        deps = [os.path.join(_HERE, n) for n in ("ref_wrap.cpp", "ref_con.c", "ref_glsl_shim.hpp", "build_ref.sh")]
        if force or not os.path.exists(ref_so) or os.path.getmtime(ref_so) < max(os.path.getmtime(d) for d in deps):
            subprocess.check_call([os.path.join(_HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)


class _Oracle:
    def __init__(self, path, prefix, con_prefix):
        self.path = path
        self.lib = ctypes.CDLL(path)
        self.kind = "reference" if prefix == "ref_" else "port"
        g = lambda n: getattr(self.lib, n)
        self._easu_con = g(con_prefix + "FsrEasuCon")
        self._easu_con.argtypes = [_U32P] + [_F] * 6
        self._easu_con_off = g(con_prefix + "FsrEasuConOffset")
        self._easu_con_off.argtypes = [_U32P] + [_F] * 8
        self._rcas_con = g(con_prefix + "FsrRcasCon")
        self._rcas_con.argtypes = [_U32P, _F]
        self._half = g(con_prefix + "AU1_AH1_AF1")
        self._half.argtypes = [_F]
        self._half.restype = ctypes.c_uint32
        for n in ("easu_f", "easu_h"):
            fn = g(prefix + n)
            fn.argtypes = [_FP, _I, _I, _FP, _I, _I, _U32P, _I, _I, _I]
            fn.restype = None
        for n in ("rcas_f", "rcas_h"):
            fn = g(prefix + n)
            fn.argtypes = [_FP, _I, _I, _FP, _U32P, _I, _I, _I]
            fn.restype = None
        self.has_hx2 = hasattr(self.lib, prefix + "rcas_hx2")  # the reference build only: FsrRcasHx2 compiled from ffx_fsr1.h:888-984
        if self.has_hx2:
            fn = g(prefix + "rcas_hx2")
            fn.argtypes = [_FP, _I, _I, _FP, _U32P, _I, _I, _I]
            fn.restype = None
        for n in ("color_f", "color_h"):
            fn = g(prefix + n)
            fn.argtypes = [_FP, _I, _I, _FP, _I, _F, _F, ctypes.c_uint32, _FP, _I, _I, _I, _I, _I, _I, _I]
            fn.restype = None
        fn = g(prefix + "tepd_dit_f")
        fn.argtypes = [ctypes.c_uint32] * 3
        fn.restype = _F
        self._prefix = prefix
        self.threads = g(prefix + "omp_threads")()

    # ---- constant setup -------------------------------------------------------------------
    def FsrEasuCon(self, vpw, vph, inw, inh, outw, outh):
        c = np.zeros(16, np.uint32)
        self._easu_con(c.ctypes.data_as(_U32P), vpw, vph, inw, inh, outw, outh)
        return c

    def FsrEasuConOffset(self, vpw, vph, inw, inh, outw, outh, offx, offy):
        c = np.zeros(16, np.uint32)
        self._easu_con_off(c.ctypes.data_as(_U32P), vpw, vph, inw, inh, outw, outh, offx, offy)
        return c

    def FsrRcasCon(self, sharpness):
        c = np.zeros(4, np.uint32)
        self._rcas_con(c.ctypes.data_as(_U32P), sharpness)
        return c

    def AU1_AH1_AF1(self, f):
        return int(self._half(f))

    # ---- per-pixel passes -----------------------------------------------------------------
    def _easu(self, name, img, out_w, out_h, con16, flags, rows):
        img = np.ascontiguousarray(img, np.float32)
        h, w, _ = img.shape
        y0, y1 = rows if rows is not None else (0, out_h)
        out = np.zeros((out_h, out_w, 4), np.float32)
        con16 = np.ascontiguousarray(con16, np.uint32)
        getattr(self.lib, self._prefix + name)(img.ctypes.data_as(_FP), w, h, out.ctypes.data_as(_FP), out_w, out_h,
                                               con16.ctypes.data_as(_U32P), flags, y0, y1)
        return out

    def easu_f(self, img, out_w, out_h, con16, flags=0, rows=None):
        return self._easu("easu_f", img, out_w, out_h, con16, flags, rows)

    def easu_h(self, img, out_w, out_h, con16, flags=0, rows=None):
        return self._easu("easu_h", img, out_w, out_h, con16, flags, rows)

    def _rcas(self, name, img, con4, flags, rows):
        img = np.ascontiguousarray(img, np.float32)
        h, w, _ = img.shape
        y0, y1 = rows if rows is not None else (0, h)
        out = np.zeros((h, w, 4), np.float32)
        con4 = np.ascontiguousarray(con4, np.uint32)
        getattr(self.lib, self._prefix + name)(img.ctypes.data_as(_FP), w, h, out.ctypes.data_as(_FP),
                                               con4.ctypes.data_as(_U32P), flags, y0, y1)
        return out

    def rcas_f(self, img, con4, flags=0, rows=None):
        return self._rcas("rcas_f", img, con4, flags, rows)

    def rcas_h(self, img, con4, flags=0, rows=None):
        return self._rcas("rcas_h", img, con4, flags, rows)

    def rcas_hx2(self, img, con4, flags=0, rows=None):
        """The reference's packed two-pixel form FsrRcasHx2 + FsrRcasDepackHx2 (ffx_fsr1.h:880-984), reference build only."""
        if not self.has_hx2:
            raise RuntimeError("rcas_hx2 exists in the reference build (oracle/_ref) only")
        return self._rcas("rcas_hx2", img, con4, flags, rows)


    # ---- colour stages (LFGA / SRTM / TEPD) -------------------------------------------------
    def color_h(self, img, stages, **kw):
        """The same chain through the half-precision entry points (FsrSrtmH, FsrLfgaH, FsrTepdC8H ...); binary16-representable values."""
        return self.color_f(img, stages, _entry="color_h", **kw)

    def color_f(self, img, stages, amount=0.0, bias=0.0, frame=0, noise=None, noise_offset=(0, 0), rows=None, _entry="color_f"):
        """Stage chain SRTM -> LFGA -> SRTM_INV -> TEPD on an (H, W, 4) float image; noise is (S, nH, nW, 4) or (nH, nW, 4)."""
        img = np.ascontiguousarray(img, np.float32)
        h, w, _ = img.shape
        y0, y1 = rows if rows is not None else (0, h)
        out = np.zeros((h, w, 4), np.float32)
        if noise is not None:
            noise = np.ascontiguousarray(noise, np.float32)
            if noise.ndim == 3:
                noise = noise[None]
            ns, nh, nw, _ = noise.shape
            nptr = noise.ctypes.data_as(_FP)
        else:
            ns = nh = nw = 1
            nptr = ctypes.cast(None, _FP)
        getattr(self.lib, self._prefix + _entry)(img.ctypes.data_as(_FP), w, h, out.ctypes.data_as(_FP), int(stages),
                                                    float(amount), float(bias), int(frame), nptr, nw, nh, ns,
                                                    int(noise_offset[0]), int(noise_offset[1]), y0, y1)
        return out

    def FsrTepdDitF(self, x, y, f):
        return float(getattr(self.lib, self._prefix + "tepd_dit_f")(int(x), int(y), int(f)))


def port():
    build()
    return _Oracle(os.path.join(_HERE, "libfsr1_oracle.so"), "oracle_", "oracle_")


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libfsr1_ref.so"))


def ref():
    build()
    return _Oracle(os.path.join(_HERE, "_ref", "libfsr1_ref.so"), "ref_", "ref_")


def have_ref_fma():
    return os.path.exists(os.path.join(_HERE, "_ref", "libfsr1_ref_fma.so"))


def ref_fma():
    """The reference compiled a second time with contracted multiply-adds (oracle/build_ref.sh): the self-spread yardstick
    (tests/ref_self_spread.py), never a parity target."""
    build()
    return _Oracle(os.path.join(_HERE, "_ref", "libfsr1_ref_fma.so"), "ref_", "ref_")


def half_ulp_diff(a, b):
    """|a-b| in units of binary16 ULPs after rounding both to binary16 (RTNE); NaN==NaN counts 0."""
    ha = np.asarray(a, np.float32).astype(np.float16).view(np.int16).astype(np.int32)
    hb = np.asarray(b, np.float32).astype(np.float16).view(np.int16).astype(np.int32)
    # map sign-magnitude to a monotone integer line
    ka = np.where(ha < 0, -(ha & 0x7FFF), ha)
    kb = np.where(hb < 0, -(hb & 0x7FFF), hb)
    d = np.abs(ka - kb)
    both_nan = np.isnan(np.asarray(a, np.float32)) & np.isnan(np.asarray(b, np.float32))
    return np.where(both_nan, 0, d)
