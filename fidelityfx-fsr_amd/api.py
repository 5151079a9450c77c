"""Host-side mirror of the reference's operator surface for the EASU+RCAS hot path.

Names, argument order and meaning follow the reference:
  FsrEasuCon / FsrEasuConOffset / FsrRcasCon      ffx-fsr/ffx_fsr1.h:156-225, :662-672
  FSR_Filter.{OnCreate, OnCreateWindowSizeDependentResources, Upscale, OnDestroy}
                                                   sample/src/DX12/FSR_Filter.{h,cpp}
Everything that touches pixels goes through the C ABI of libfsr1_hip.so; torch is used only to own
device memory and to name the stream.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import Fsr1Error, fsr1_color_stages, fsr1_image, fsr1_params  # noqa: F401

FORMAT_RGBA16F = 0
FORMAT_RGBA32F = 1
FORMAT_RGBA8_UNORM = 2
FORMAT_R10G10B10A2_UNORM = 3

FLAG_HDR_SQUARE = 1 << 0
FLAG_RCAS_DENOISE = 1 << 1
FLAG_RCAS_PASSTHROUGH_ALPHA = 1 << 2
FLAG_MATH_EXACT = 1 << 4
FLAG_MATH_PACKED_FP16 = 1 << 5
FLAG_MATH_STRICT = 1 << 6
FLAG_NO_FAST_PATHS = 1 << 8
FLAG_OUTPUT_STREAMING = 1 << 9
FLAG_OUTPUT_CACHED = 1 << 10
FLAG_FRAMES_OVERLAP = 1 << 11

# colour stages (ffx_fsr1.h:986-1199), fixed order SRTM -> LFGA -> SRTM_INV -> TEPD
COLOR_SRTM = 1 << 0
COLOR_LFGA = 1 << 1
COLOR_SRTM_INV = 1 << 2
COLOR_TEPD_C8 = 1 << 3
COLOR_TEPD_C10 = 1 << 4
COLOR_DITHER_FROM_NOISE = 1 << 5

_U32P = ctypes.POINTER(ctypes.c_uint32)


def _u32p(a, off=0):
    return ctypes.cast(a.ctypes.data + 4 * off, _U32P)


# --------------------------------------------------------------------------------------------------
# constant setup (host)
# --------------------------------------------------------------------------------------------------
def FsrEasuCon(inputViewportInPixelsX, inputViewportInPixelsY, inputSizeInPixelsX, inputSizeInPixelsY,
               outputSizeInPixelsX, outputSizeInPixelsY):
    """-> uint32[16] = con0 | con1 | con2 | con3 (ffx_fsr1.h:156-202)."""
    c = np.zeros(16, np.uint32)
    _lib.load().FsrEasuCon(_u32p(c, 0), _u32p(c, 4), _u32p(c, 8), _u32p(c, 12), inputViewportInPixelsX,
                           inputViewportInPixelsY, inputSizeInPixelsX, inputSizeInPixelsY, outputSizeInPixelsX,
                           outputSizeInPixelsY)
    return c


def FsrEasuConOffset(inputViewportInPixelsX, inputViewportInPixelsY, inputSizeInPixelsX, inputSizeInPixelsY,
                     outputSizeInPixelsX, outputSizeInPixelsY, inputOffsetInPixelsX, inputOffsetInPixelsY):
    """-> uint32[16] (ffx_fsr1.h:205-225)."""
    c = np.zeros(16, np.uint32)
    _lib.load().FsrEasuConOffset(_u32p(c, 0), _u32p(c, 4), _u32p(c, 8), _u32p(c, 12), inputViewportInPixelsX,
                                 inputViewportInPixelsY, inputSizeInPixelsX, inputSizeInPixelsY, outputSizeInPixelsX,
                                 outputSizeInPixelsY, inputOffsetInPixelsX, inputOffsetInPixelsY)
    return c


def FsrRcasCon(sharpness):
    """-> uint32[4]; sharpness in stops, 0 = maximum (ffx_fsr1.h:662-672)."""
    c = np.zeros(4, np.uint32)
    _lib.load().FsrRcasCon(_u32p(c), sharpness)
    return c


def AU1_AH1_AF1(f):
    return int(_lib.load().AU1_AH1_AF1(f))


# --------------------------------------------------------------------------------------------------
# images
# --------------------------------------------------------------------------------------------------
def image_of(t):
    """fsr1_image descriptor of a CUDA torch tensor:
      float16 / float32 / uint8, shaped (H,W,4) or (N,H,W,4)  -> RGBA16F / RGBA32F / RGBA8_UNORM
      int32, shaped (H,W) or (N,H,W)                          -> R10G10B10A2_UNORM (one packed word per pixel)
    Row and frame strides are honoured; the innermost (x[, channel]) dims must be dense."""
    import torch
    if not t.is_cuda:
        raise Fsr1Error("image tensors must live on the GPU (got %s); there is no CPU path" % t.device)
    if t.dtype == torch.int32:
        if t.dim() == 2:
            t = t.unsqueeze(0)
        if t.dim() != 3:
            raise Fsr1Error("expected (N,H,W) packed R10G10B10A2 words, got %s" % (tuple(t.shape),))
        n, h, w = t.shape
        sn, sh, sw = t.stride()
        if sw != 1:
            raise Fsr1Error("pixels must be dense along x")
        _check_strides(n, h, w, sn, sh, 1)
        return fsr1_image(t.data_ptr(), w, h, FORMAT_R10G10B10A2_UNORM, n, sh * 4, sn * 4 if n > 1 else 0)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dim() != 4 or t.shape[-1] != 4:
        raise Fsr1Error("expected (N,H,W,4) RGBA, got %s" % (tuple(t.shape),))
    if t.dtype == torch.float16:
        fmt, es = FORMAT_RGBA16F, 2
    elif t.dtype == torch.float32:
        fmt, es = FORMAT_RGBA32F, 4
    elif t.dtype == torch.uint8:
        fmt, es = FORMAT_RGBA8_UNORM, 1
    else:
        raise Fsr1Error("unsupported dtype %s" % t.dtype)
    n, h, w, _ = t.shape
    sn, sh, sw, sc = t.stride()
    if sc != 1 or sw != 4:
        raise Fsr1Error("pixels must be RGBA-interleaved and dense along x")
    _check_strides(n, h, w, sn, sh, 4)
    return fsr1_image(t.data_ptr(), w, h, fmt, n, sh * es, sn * es if n > 1 else 0)


def _check_strides(n, h, w, sn, sh, elems_per_pixel):
    """The C ABI reads a pitch / frame stride of 0 as "tightly packed", so an expanded / broadcast view (stride 0) or an
    overlapping one must not reach it: rows and frames have to be distinct memory."""
    if h > 1 and sh < w * elems_per_pixel:
        raise Fsr1Error("row stride %d elements is smaller than a row of %d pixels (expanded or overlapping view?)" % (sh, w))
    if n > 1 and sn < sh * h:
        raise Fsr1Error("frame stride %d elements is smaller than one frame (expanded or overlapping view?)" % sn)


def _stream_ptr(stream):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def _con(c, n):
    c = np.ascontiguousarray(c, dtype=np.uint32)
    if c.size != n:
        raise Fsr1Error("expected %d constant words, got %d" % (n, c.size))
    return c


# --------------------------------------------------------------------------------------------------
# device passes
# --------------------------------------------------------------------------------------------------
class ColorStages:
    """fsr1_color_stages: which of FsrSrtmF / FsrLfgaF / FsrSrtmInvF / FsrTepdC8F|C10F run, and their inputs
    (ffx_fsr1.h:986-1199).  `noise` is a CUDA tensor (tiled grain / dither texture, (S,h,w,4) or (h,w,4))."""

    def __init__(self, stages, grain_amount=0.0, grain_bias=0.0, frame=0, noise=None, noise_offset=(0, 0)):
        self.stages, self.grain_amount, self.grain_bias, self.frame = int(stages), float(grain_amount), float(grain_bias), int(frame)
        self.noise, self.noise_offset = noise, (int(noise_offset[0]), int(noise_offset[1]))

    def c_struct(self):
        """-> (fsr1_color_stages, keep-alive tuple)"""
        img = image_of(self.noise) if self.noise is not None else None
        st = fsr1_color_stages(self.stages, self.grain_amount, self.grain_bias, self.frame & 0xFFFFFFFF, self.noise_offset[0],
                               self.noise_offset[1], ctypes.pointer(img) if img is not None else None)
        return st, (img, self.noise)


def _stages(stages):
    if stages is None:
        return None, None
    st, keep = stages.c_struct()
    return ctypes.byref(st), (st, keep)


def easu(src, dst, con=None, flags=0, stream=None, stages=None):
    """dst = EASU(src).  con defaults to FsrEasuCon(viewport = input size).  stages: optional ColorStages fused in."""
    i, o = image_of(src), image_of(dst)
    if con is None:
        con = FsrEasuCon(i.width, i.height, i.width, i.height, o.width, o.height)
    con = _con(con, 16)
    if stages is None:
        _lib.check(_lib.load().fsr1_easu_dispatch(ctypes.byref(i), ctypes.byref(o), _u32p(con), flags, _stream_ptr(stream)))
    else:
        sp, _keep = _stages(stages)
        _lib.check(_lib.load().fsr1_easu_dispatch_ex(ctypes.byref(i), ctypes.byref(o), _u32p(con), flags, sp, _stream_ptr(stream)))
    return dst


def rcas(src, dst, con=None, sharpness=0.25, flags=0, stream=None, stages=None):
    """dst = RCAS(src); con defaults to FsrRcasCon(sharpness).  stages: optional ColorStages fused in."""
    i, o = image_of(src), image_of(dst)
    con = _con(FsrRcasCon(sharpness) if con is None else con, 4)
    if stages is None:
        _lib.check(_lib.load().fsr1_rcas_dispatch(ctypes.byref(i), ctypes.byref(o), _u32p(con), flags, _stream_ptr(stream)))
    else:
        sp, _keep = _stages(stages)
        _lib.check(_lib.load().fsr1_rcas_dispatch_ex(ctypes.byref(i), ctypes.byref(o), _u32p(con), flags, sp, _stream_ptr(stream)))
    return dst


def easu_band(src, dst_band, con, origin=(0, 0), flags=0, stream=None):
    """dst_band = the window of EASU(src) whose pixel (0, 0) is output pixel `origin` = (x, y) of the full image `con` describes."""
    i, o = image_of(src), image_of(dst_band)
    con = _con(con, 16)
    _lib.check(_lib.load().fsr1_easu_dispatch_band(ctypes.byref(i), ctypes.byref(o), _u32p(con), flags, int(origin[0]), int(origin[1]), _stream_ptr(stream)))
    return dst_band


def rcas_band(src_band, dst_band, con=None, sharpness=0.25, rows_above=0, rows_below=0, flags=0, stream=None):
    """RCAS on a band: `src_band` is a view of the band's own rows inside a larger tensor; rows_above / rows_below = 1 when the
    row just above / below the view exists in that tensor and belongs to the image."""
    i, o = image_of(src_band), image_of(dst_band)
    con = _con(FsrRcasCon(sharpness) if con is None else con, 4)
    _lib.check(_lib.load().fsr1_rcas_dispatch_band(ctypes.byref(i), ctypes.byref(o), _u32p(con), flags, int(rows_above), int(rows_below), _stream_ptr(stream)))
    return dst_band


def easu_rcas_fused_band(src, dst_band, easu_con, rcas_con=None, sharpness=0.25, origin_y=0, rows_above=0, rows_below=0, flags=0, stream=None):
    """dst_band = rows [origin_y, origin_y + rows) of RCAS(EASU(src)) in one launch; rows_above / rows_below = 1 when the full
    output image `easu_con` describes has a row above / below the band."""
    i, o = image_of(src), image_of(dst_band)
    easu_con = _con(easu_con, 16)
    rcas_con = _con(FsrRcasCon(sharpness) if rcas_con is None else rcas_con, 4)
    _lib.check(_lib.load().fsr1_easu_rcas_fused_dispatch_band(ctypes.byref(i), ctypes.byref(o), _u32p(easu_con), _u32p(rcas_con), flags,
                                                              int(origin_y), int(rows_above), int(rows_below), _stream_ptr(stream)))
    return dst_band


def upscale_band(src, dst_band, out_size, band, mid=None, sharpness=0.25, flags=0, stream=None, fused=False):
    """EASU + RCAS for output rows [band[0], band[1]) of an `out_size` = (width, height) upscale of `src`, written to `dst_band`
    (band[1] - band[0] rows): what one GPU does when a single frame is split into row bands.  `mid` (optional)
    is a scratch tensor (out_size[0] pixels wide, dst_band's dtype and device) of at least band rows + 3 — one EASU row either side
    of the band plus one more so that the intermediary can start on an even row; a smaller or mismatched one is refused (it is
    never silently truncated).  fused=True takes the single launch (no intermediary); returns dst_band."""
    import torch
    ow, oh = out_size
    y0, y1 = band
    i = image_of(src)
    con = FsrEasuCon(i.width, i.height, i.width, i.height, ow, oh)
    if fused:
        return easu_rcas_fused_band(src, dst_band, con, sharpness=sharpness, origin_y=y0, rows_above=int(y0 > 0), rows_below=int(y1 < oh),
                                    flags=flags, stream=stream)
    # EASU rows the band's RCAS taps read: one either side — started on an even row, so that at exactly 2x every band (not
    # only the first) runs the exact-2x kernel, whose quads need an even origin
    m0, m1 = max(y0 - 1, 0) & ~1, min(y1 + 1, oh)
    if mid is None:
        mid = torch.empty(m1 - m0, ow, 4, dtype=dst_band.dtype, device=dst_band.device)
    else:
        if mid.dim() != 3 or mid.shape[0] < m1 - m0 or mid.shape[1] != ow or mid.shape[2] != 4:
            raise Fsr1Error("upscale_band: the intermediary must be (>= %d, %d, 4) — output rows [%d, %d), i.e. band rows + up to 3 — got %s"
                            % (m1 - m0, ow, m0, m1, tuple(mid.shape)))
        if mid.dtype != dst_band.dtype or mid.device != dst_band.device:
            raise Fsr1Error("upscale_band: the intermediary must have the band's dtype and device (%s on %s), got %s on %s"
                            % (dst_band.dtype, dst_band.device, mid.dtype, mid.device))
    mid = mid[:m1 - m0]
    # FSR_Filter.cpp:107: EASU's Sample.x is 0 when RCAS follows — the HDR square, RCAS options and the store policy of the
    # final image belong to the RCAS dispatch only (the fused launch squares once, and so must the two dispatches)
    easu_flags = flags & (FLAG_MATH_EXACT | FLAG_MATH_PACKED_FP16 | FLAG_MATH_STRICT | FLAG_NO_FAST_PATHS)
    easu_band(src, mid, con, origin=(0, m0), flags=easu_flags, stream=stream)
    rcas_band(mid[y0 - m0:y0 - m0 + (y1 - y0)], dst_band, sharpness=sharpness, rows_above=int(m0 < y0), rows_below=int(m1 > y1), flags=flags, stream=stream)
    return dst_band


def easu_rcas_fused(src, dst, easu_con=None, rcas_con=None, sharpness=0.25, flags=0, stream=None, stages=None):
    """dst = RCAS(EASU(src)) in one launch (intermediate kept in LDS).  stages: optional ColorStages fused in."""
    i, o = image_of(src), image_of(dst)
    if easu_con is None:
        easu_con = FsrEasuCon(i.width, i.height, i.width, i.height, o.width, o.height)
    easu_con = _con(easu_con, 16)
    rcas_con = _con(FsrRcasCon(sharpness) if rcas_con is None else rcas_con, 4)
    if stages is None:
        _lib.check(_lib.load().fsr1_easu_rcas_fused_dispatch(ctypes.byref(i), ctypes.byref(o), _u32p(easu_con), _u32p(rcas_con),
                                                             flags, _stream_ptr(stream)))
    else:
        sp, _keep = _stages(stages)
        _lib.check(_lib.load().fsr1_easu_rcas_fused_dispatch_ex(ctypes.byref(i), ctypes.byref(o), _u32p(easu_con), _u32p(rcas_con),
                                                                flags, sp, _stream_ptr(stream)))
    return dst


def color(src, dst, stages, flags=0, stream=None):
    """dst = stages(src): the stand-alone colour pass (FsrSrtmF -> FsrLfgaF -> FsrSrtmInvF -> FsrTepdC8F|C10F)."""
    i, o = image_of(src), image_of(dst)
    sp, _keep = _stages(stages)
    _lib.check(_lib.load().fsr1_color_dispatch(ctypes.byref(i), ctypes.byref(o), sp, flags, _stream_ptr(stream)))
    return dst


class State:
    """The fields of the sample's `State` that FSR_Filter::Upscale reads (FSR_Filter.cpp:106-133)."""

    def __init__(self, renderWidth, renderHeight, bUseRcas=True, rcasAttenuation=0.25, m_nUpscaleType=1):
        self.renderWidth = renderWidth
        self.renderHeight = renderHeight
        self.bUseRcas = bUseRcas
        self.rcasAttenuation = rcasAttenuation  # sample default, SampleRenderer.h:49
        self.m_nUpscaleType = m_nUpscaleType    # 0 = bilinear in the sample (not part of this path)


class FSR_Filter:
    """Host glue with the reference's method names (sample/src/DX12/FSR_Filter.h:30-37).

    OnCreate picks the arithmetic (slowFallback <-> fp32 FsrEasuF/FsrRcasF, the sample's
    SAMPLE_SLOW_FALLBACK permutation; otherwise the packed-fp16 FsrEasuH/FsrRcasH permutation),
    OnCreateWindowSizeDependentResources binds input/output and allocates the intermediary,
    Upscale issues EASU -> RCAS (or EASU only) on a stream.
    """

    def __init__(self):
        self._created = False
        self._flags = 0
        self.m_intermediary = None
        self._input = None
        self._output = None
        self._hdr = False
        self.fused = False

    def OnCreate(self, slowFallback=True, exact=False, fused=False, strict=False):
        """fused: False = EASU + RCAS dispatches, True = the single fused launch, "auto" = whichever is faster for the
        scale (fsr1_params.fused = 2; the intermediary is still allocated so that either can run).
        strict: FSR1_FLAG_MATH_STRICT — EASU bit-identical to FsrEasuF, the final image within 1 binary16 ULP of the reference chain."""
        _lib.load()
        if exact and strict:
            raise Fsr1Error("exact and strict are exclusive")
        self._flags = (FLAG_MATH_EXACT if exact else (FLAG_MATH_STRICT if strict else 0)) if slowFallback else FLAG_MATH_PACKED_FP16
        self.fused = fused
        self._created = True

    def OnCreateWindowSizeDependentResources(self, input, output, displayWidth, displayHeight, pState=None, hdr=False):
        import torch
        if not self._created:
            raise Fsr1Error("OnCreate has not been called")
        o = image_of(output)
        if (o.width, o.height) != (displayWidth, displayHeight):
            raise Fsr1Error("output tensor is %dx%d, display is %dx%d" % (o.width, o.height, displayWidth, displayHeight))
        self._input, self._output, self._hdr = input, output, hdr
        # FSR_Filter.cpp:72-73 creates the EASU->RCAS intermediary at display size; here it has the
        # output's format (RGBA16F/RGBA32F) and batch size.
        self.m_intermediary = None if self.fused is True else torch.empty_like(output)

    def OnDestroyWindowSizeDependentResources(self):
        self.m_intermediary = None
        self._input = self._output = None

    def OnDestroy(self):
        self.OnDestroyWindowSizeDependentResources()
        self._created = False

    def Upscale(self, displayWidth, displayHeight, pState, hdr=None, stream=None, stages=None):
        if self._input is None:
            raise Fsr1Error("OnCreateWindowSizeDependentResources has not been called")
        if not pState.m_nUpscaleType:
            raise Fsr1Error("m_nUpscaleType == 0 (bilinear) is outside the EASU+RCAS path")
        hdr = self._hdr if hdr is None else hdr
        p = fsr1_params(float(pState.renderWidth), float(pState.renderHeight), int(bool(pState.bUseRcas)),
                        float(pState.rcasAttenuation), int(bool(hdr)),
                        (2 if self.fused == "auto" else int(bool(self.fused))) if pState.bUseRcas else 0, self._flags)
        i, o = image_of(self._input), image_of(self._output)
        if (o.width, o.height) != (displayWidth, displayHeight):
            raise Fsr1Error("display size changed: call OnCreateWindowSizeDependentResources again")
        m = image_of(self.m_intermediary) if self.m_intermediary is not None else None
        sp, _keep = _stages(stages)  # optional ColorStages fused into the passes (fsr1_upscale_ex)
        _lib.check(_lib.load().fsr1_upscale_ex(ctypes.byref(i), ctypes.byref(m) if m is not None else None, ctypes.byref(o),
                                              ctypes.byref(p), sp, _stream_ptr(stream)))
        return self._output


class Pipeline:
    """fsr1_pipeline: independent frames on alternating HIP streams (include/fsr1_hip.h, "Frame pipeline").  Frame i of a sequence
    of upscale() calls runs on stream i mod `streams` with that stream's own EASU -> RCAS intermediary, so the tail of one frame
    overlaps the head of the next (a kernel boundary costs ~5 us on MI355X).

    The pipeline's streams are HIP streams torch does not know about.  With managed = True (the default) the wrapper closes the two
    gaps that leaves:
      * ordering: every submission's stream first waits for what the CURRENT torch stream holds at the time of the call (the producer
        of `src`, the last consumer of `dst`) — one event, one wait, on that slot's stream only;
      * lifetime: src / dst / stages stay referenced until the submission has left the device (or join() / synchronize() / close()),
        so torch's caching allocator cannot hand their memory to a later op on the current stream while a pipeline kernel still uses it.
    Cost of managed = True per upscale(): one event record + one stream wait in front and one event record behind, per slot used (every
    slot for a batch: the wrapper cannot know which slots the library's frame-by-frame split takes), plus the completion queries of
    the oldest submissions — a few microseconds of host time; a loop that needs the last of them uses managed = False (bench.py does).
    join() (or synchronize()) is still mandatory before the OUTPUTS are consumed on a torch stream.  managed = False is the bare C
    ABI: the caller forks, joins and keeps tensors alive itself (bench.py: its buffers live for the whole run and it synchronizes
    around every timed region).

    Aliasing (include/fsr1_hip.h): streams are ordered only within themselves — a tensor may be reused by a later submission only
    when that submission takes the same slot (next_slot()), or after join() / synchronize(); a ring of buffers whose length is a
    multiple of `streams`, walked in order, satisfies that by construction."""

    def __init__(self, streams=3, managed=True):
        self._h = ctypes.c_void_p()
        _lib.check(_lib.load().fsr1_pipeline_create(ctypes.byref(self._h), int(streams)))
        self.streams = int(streams)
        self.managed = bool(managed)
        self._slot_streams = None
        self._live = []  # (completion event, tensors) of submissions that may still be on the device

    def _slot_stream(self, slot):
        import torch
        if self._slot_streams is None:
            lib = _lib.load()
            self._slot_streams = [torch.cuda.ExternalStream(lib.fsr1_pipeline_stream(self._h, i)) for i in range(self.streams)]
        return self._slot_streams[slot]

    def next_slot(self):
        """The slot (0 .. streams - 1) the next upscale() runs on."""
        return int(_lib.load().fsr1_pipeline_next_slot(self._h))

    def reserve(self, bytes_per_stream):
        """fsr1_pipeline_reserve: size every stream's intermediary now (bytes of the largest output frame or batch), so that no later
        submission allocates — required before capturing submissions into a graph on a pipeline that has not seen that size yet."""
        _lib.check(_lib.load().fsr1_pipeline_reserve(self._h, int(bytes_per_stream)))

    def upscale(self, src, dst, sharpness=0.25, use_rcas=True, fused=0, flags=0, hdr=False, stages=None, render_size=None):
        """dst = Upscale(src) (FSR_Filter::Upscale, FSR_Filter.cpp:101-141) on the pipeline's next stream; fused: 0 two dispatches, 1 the
        single launch, 2 whichever is faster; render_size: (renderWidth, renderHeight) when smaller than the input resource."""
        import torch
        i, o = image_of(src), image_of(dst)
        rw, rh = render_size if render_size is not None else (i.width, i.height)
        p = fsr1_params(float(rw), float(rh), int(bool(use_rcas)), float(sharpness), int(bool(hdr)), int(fused) if use_rcas else 0, int(flags))
        sp, _keep = _stages(stages)
        used = ()
        if self.managed:
            # after the producer of src / the last user of dst on the caller's stream: the slot this submission takes — every slot for a
            # batch, which the library may submit frame by frame over all of them (fsr1_pipeline_upscale)
            used = [self._slot_stream(self.next_slot())] if o.frames == 1 or self.streams == 1 else [self._slot_stream(k) for k in range(self.streams)]
            for st in used:
                st.wait_stream(torch.cuda.current_stream())
        _lib.check(_lib.load().fsr1_pipeline_upscale(self._h, ctypes.byref(i), ctypes.byref(o), ctypes.byref(p), sp))
        if self.managed and not torch.cuda.is_current_stream_capturing():  # (a captured submission runs at replay time: the graph's owner keeps its tensors)
            self._trim()  # submissions that have left the device no longer need their tensors held
            self._live.append(([st.record_event() for st in used], (src, dst, stages)))
        return dst

    def fork(self, stream=None):
        _lib.check(_lib.load().fsr1_pipeline_fork(self._h, _stream_ptr(stream)))

    def _trim(self):
        """Drops the references of submissions that have left the device."""
        while self._live and all(e.query() for e in self._live[0][0]):
            self._live.pop(0)

    def join(self, stream=None):
        """Work submitted to `stream` (default: the current torch stream) afterwards starts only after everything the pipeline holds now.
        The tensors of submissions that have already finished are released here as well (and by later upscale() calls, synchronize(),
        close()); the ones still in flight stay referenced until one of those notices their completion."""
        _lib.check(_lib.load().fsr1_pipeline_join(self._h, _stream_ptr(stream)))
        self._trim()

    def synchronize(self):
        _lib.check(_lib.load().fsr1_pipeline_synchronize(self._h))
        self._live.clear()

    def close(self):
        if self._h:
            _lib.load().fsr1_pipeline_destroy(self._h)  # (synchronizes the pipeline's streams first)
            self._h = ctypes.c_void_p()
        self._live.clear()
        self._slot_streams = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def selftest():
    """fsr1_selftest: the number of operands for which (a) the packed-fp16 kernels' reciprocal (all 65536 binary16 operands) or
    (b) the EXACT variants' binary32 reciprocal rcp_ieee (all 2^32 binary32 operands) is not the correctly rounded 1/x; 0 on a
    conforming device."""
    n = ctypes.c_uint32(0)
    _lib.check(_lib.load().fsr1_selftest(ctypes.byref(n)))
    return int(n.value)


class Timer:
    """HIP-event stopwatch of the C ABI (fsr1_timer_*), recording on the stream the kernels use."""

    def __init__(self):
        self._h = ctypes.c_void_p()
        _lib.check(_lib.load().fsr1_timer_create(ctypes.byref(self._h)))

    def start(self, stream=None):
        _lib.check(_lib.load().fsr1_timer_start(self._h, _stream_ptr(stream)))

    def stop(self, stream=None):
        _lib.check(_lib.load().fsr1_timer_stop(self._h, _stream_ptr(stream)))

    def elapsed_ms(self):
        ms = ctypes.c_float()
        _lib.check(_lib.load().fsr1_timer_elapsed_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            _lib.load().fsr1_timer_destroy(self._h)
        except Exception:
            pass
