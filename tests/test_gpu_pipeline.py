"""fsr1_pipeline (include/fsr1_hip.h, "Frame pipeline"): independent frames on alternating HIP streams, each stream with its own
EASU -> RCAS intermediary.  Overlapping frames must not change a bit of any of them: every frame of a sequence equals the same
frame upscaled alone on one stream, for 1 - 4 streams, two dispatches / fused / auto / EASU only, both arithmetics, changing frame
sizes (the per-stream intermediary grows), and fork / join order the pipeline against the caller's stream."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def alone(fsr, src, ow, oh, fused, flags, use_rcas=True):
    out = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    if not use_rcas:
        fsr.easu(src, out, flags=flags | fsr.FLAG_OUTPUT_STREAMING)
    elif fused:
        fsr.easu_rcas_fused(src, out, sharpness=0.25, flags=flags)
    else:
        mid = torch.zeros_like(out)
        fsr.easu(src, mid, flags=flags & (fsr.FLAG_MATH_EXACT | fsr.FLAG_MATH_PACKED_FP16))
        fsr.rcas(mid, out, sharpness=0.25, flags=flags)
    return out


@pytest.mark.parametrize("streams", [1, 2, 3, 4])
@pytest.mark.parametrize("mode", ["two-pass", "fused", "auto", "easu"])
def test_pipelined_frames_equal_frames_upscaled_alone(fsr, streams, mode):
    shapes = [(240, 135, 480, 270), (240, 135, 480, 270), (320, 180, 480, 270), (97, 61, 194, 122), (480, 270, 960, 540), (240, 135, 480, 270)] * 3
    pipe = fsr.Pipeline(streams)
    for flags in (0, fsr.FLAG_MATH_EXACT, fsr.FLAG_MATH_PACKED_FP16):
        srcs = [dev(frames.synthetic_frame(iw, ih, k=70 + k, dtype=np.float16)) for k, (iw, ih, _, _) in enumerate(shapes)]
        outs = [torch.full((oh, ow, 4), -2.0, dtype=torch.float16, device="cuda") for (_, _, ow, oh) in shapes]
        torch.cuda.synchronize()
        for s, o in zip(srcs, outs):
            pipe.upscale(s, o, sharpness=0.25, use_rcas=mode != "easu", fused={"two-pass": 0, "fused": 1, "auto": 2, "easu": 0}[mode], flags=flags)
        pipe.synchronize()
        for k, (s, o, (iw, ih, ow, oh)) in enumerate(zip(srcs, outs, shapes)):
            if mode == "auto":  # whichever pipeline auto took, the image is the two dispatches' (fused == two-pass bit for bit)
                want = alone(fsr, s, ow, oh, False, flags)
            else:
                want = alone(fsr, s, ow, oh, mode == "fused", flags, use_rcas=mode != "easu")
            torch.cuda.synchronize()
            assert torch.equal(o.view(torch.int16), want.view(torch.int16)), "frame %d (%s, flags %d, %d streams) differs" % (k, mode, flags, streams)
    pipe.close()


def test_pipelined_4k_frames_walk_and_equal_the_two_dispatches(fsr):
    """On a pipeline of two streams the exact-2x fused launch of a 4K frame walks its columns in 4-step runs (FSR1_FLAG_FRAMES_OVERLAP),
    alone it runs tall one-step tiles: both are the two dispatches' image, and so are the pipelined two dispatches themselves."""
    iw, ih, ow, oh = 1920, 1080, 3840, 2160
    srcs = [dev(frames.synthetic_frame(iw, ih, k=90 + k, dtype=np.float16)) for k in range(4)]
    want = [alone(fsr, s, ow, oh, False, 0) for s in srcs]
    pipe = fsr.Pipeline(2)
    for fused in (1, 0, 2):
        outs = [torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda") for _ in srcs]
        torch.cuda.synchronize()
        for s, o in zip(srcs, outs):
            pipe.upscale(s, o, fused=fused)
        pipe.synchronize()
        for k in range(4):
            assert torch.equal(outs[k].view(torch.int16), want[k].view(torch.int16)), (fused, k)
    pipe.close()
    one = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu_rcas_fused(srcs[0], one, flags=fsr.FLAG_FRAMES_OVERLAP)  # the hint given by hand
    torch.cuda.synchronize()
    assert torch.equal(one.view(torch.int16), want[0].view(torch.int16))


def test_pipelined_frames_with_colour_stages_unorm_and_batches(fsr):
    """Colour stages (prologue on EASU's loads, epilogue on the pass that writes the output), RGBA8 storage and batched frames go
    through the pipeline like through fsr1_upscale_ex: every submission equals the same submission made alone on one stream."""
    rng = np.random.default_rng(7)
    noise = dev((rng.random((4, 32, 32, 4)) - np.array([0.5, 0.5, 0.5, 0.0])).astype(np.float16))
    stages = fsr.ColorStages(1 | 2 | 4, grain_amount=0.3, frame=5, noise=noise)
    iw, ih, ow, oh = 200, 120, 300, 180
    n = 3
    src16 = dev(np.stack([frames.synthetic_frame(iw, ih, k=30 + f, dtype=np.float16) for f in range(n)]))
    src8 = (src16.float().clamp(0, 1) * 255 + 0.5).to(torch.uint8)
    lib = fsr.load()

    def alone_ex(src, fused, st):
        out = torch.zeros(n, oh, ow, 4, dtype=src.dtype, device="cuda")
        f = fsr.FSR_Filter()
        f.OnCreate(slowFallback=True, fused=bool(fused))
        f.OnCreateWindowSizeDependentResources(src, out, ow, oh)
        f.Upscale(ow, oh, fsr.State(iw, ih, bUseRcas=True, rcasAttenuation=0.25), stages=st)
        torch.cuda.synchronize()
        return out

    for streams in (2, 3):
        pipe = fsr.Pipeline(streams)
        for src, st in ((src16, stages), (src16, None), (src8, None)):
            for fused in (0, 1):
                want = alone_ex(src, fused, st)
                outs = [torch.zeros(n, oh, ow, 4, dtype=src.dtype, device="cuda") for _ in range(4)]
                torch.cuda.synchronize()
                for o in outs:
                    pipe.upscale(src, o, sharpness=0.25, fused=fused, stages=st)
                pipe.synchronize()
                for k, o in enumerate(outs):
                    assert torch.equal(o, want), (streams, str(src.dtype), fused, st is not None, k)
        pipe.close()
    assert lib.fsr1_pipeline_streams(None) == 0 and lib.fsr1_pipeline_stream(None, 0) is None


def test_pipeline_is_graph_capturable(fsr):
    """fork / upscale / join record into a stream capture (the pipeline's streams join it through the fork event), so a group of
    frames replays as one hipGraph with `streams` parallel chains — the launch-bound small-frame case (bench.py --graph)."""
    iw, ih, ow, oh = 240, 135, 480, 270
    srcs = [dev(frames.synthetic_frame(iw, ih, k=40 + k, dtype=np.float16)) for k in range(6)]
    want = [alone(fsr, s, ow, oh, False, 0) for s in srcs]
    outs = [torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda") for _ in srcs]
    pipe = fsr.Pipeline(2)
    for s, o in zip(srcs, outs):  # first use allocates the per-stream intermediaries (not allowed inside a capture)
        pipe.upscale(s, o, fused=0)
    pipe.synchronize()
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            pipe.fork(cap)
            for s, o in zip(srcs, outs):
                pipe.upscale(s, o, fused=0)
            pipe.join(cap)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for k in range(6):
        assert torch.equal(outs[k].view(torch.int16), want[k].view(torch.int16)), k
    del g
    pipe.close()


def test_graph_capture_on_a_cold_pipeline_after_reserve(fsr):
    """fsr1_pipeline_reserve sizes every stream's intermediary up front, so a pipeline that has never run a frame can be captured
    into a graph; WITHOUT the reserve the capture's first submission is refused with a message that names the size (not a broken
    capture), and the pipeline stays usable."""
    iw, ih, ow, oh = 240, 135, 480, 270
    srcs = [dev(frames.synthetic_frame(iw, ih, k=20 + k, dtype=np.float16)) for k in range(6)]
    want = [alone(fsr, s, ow, oh, False, 0) for s in srcs]
    outs = [torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda") for _ in srcs]
    torch.cuda.synchronize()
    # (a) cold pipeline, no reserve: refused inside the capture
    cold = fsr.Pipeline(2)
    cap = torch.cuda.Stream()
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g0, stream=cap):
            cold.fork(cap)
            with pytest.raises(fsr.Fsr1Error, match="fsr1_pipeline_reserve"):
                cold.upscale(srcs[0], outs[0], fused=0)
            cold.upscale(srcs[0], outs[0], fused=1)  # the fused launch needs no intermediary: capturable as is
            cold.join(cap)
    g0.replay()
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int16), want[0].view(torch.int16))
    del g0
    cold.close()
    # (b) cold pipeline after reserve: captures and replays
    outs[0].zero_()
    pipe = fsr.Pipeline(3)
    pipe.reserve(ow * oh * 8)
    assert pipe.next_slot() == 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            pipe.fork(cap)
            for s, o in zip(srcs, outs):
                pipe.upscale(s, o, fused=0)
            pipe.join(cap)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    for k in range(6):
        assert torch.equal(outs[k].view(torch.int16), want[k].view(torch.int16)), k
    del g
    pipe.close()


def test_intermediary_grows_in_stream_order_without_blocking(fsr):
    """Frame sizes that grow from submission to submission: every slot's intermediary is re-allocated on its own stream (hipFreeAsync /
    hipMallocAsync) while earlier frames of the other slots are still in flight; every frame equals the frame upscaled alone, and
    next_slot() walks 0, 1, 2, 0, ..."""
    shapes = [(120, 68, 240, 136), (240, 135, 480, 270), (480, 270, 960, 540), (960, 540, 1920, 1080), (240, 135, 480, 270), (1280, 720, 2560, 1440)] * 2
    pipe = fsr.Pipeline(3)
    srcs = [dev(frames.synthetic_frame(iw, ih, k=11 + k, dtype=np.float16)) for k, (iw, ih, _, _) in enumerate(shapes)]
    outs = [torch.full((oh, ow, 4), -2.0, dtype=torch.float16, device="cuda") for (_, _, ow, oh) in shapes]
    for k, (s, o) in enumerate(zip(srcs, outs)):
        assert pipe.next_slot() == k % 3
        pipe.upscale(s, o, fused=0)
    pipe.synchronize()
    for k, (s, o, (iw, ih, ow, oh)) in enumerate(zip(srcs, outs, shapes)):
        assert torch.equal(o.view(torch.int16), alone(fsr, s, ow, oh, False, 0).view(torch.int16)), k
    pipe.close()


def test_batches_are_submitted_frame_by_frame_when_their_intermediaries_fit_the_cache(fsr):
    """A batch that takes the two dispatches goes through the pipeline frame by frame (frame f on slot (next_slot + f) mod N): same
    pixels as the batch in one launch pair on one stream, the slot counter advances by the frame count, strided batches (pitch and
    frame stride) included; a fused batch and a one-stream pipeline keep the single launch (the counter advances by one)."""
    iw, ih, ow, oh, n = 240, 135, 360, 203, 5
    src = dev(np.stack([frames.synthetic_frame(iw, ih, k=60 + f, dtype=np.float16) for f in range(n)]))
    want = torch.zeros(n, oh, ow, 4, dtype=torch.float16, device="cuda")
    mid = torch.zeros_like(want)
    fsr.easu(src, mid, con=fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh))
    fsr.rcas(mid, want, sharpness=0.25)
    torch.cuda.synchronize()
    pipe = fsr.Pipeline(3)
    big = torch.full((n, oh + 3, ow + 7, 4), 5.0, dtype=torch.float16, device="cuda")
    got = big[:, :oh, :ow]  # row pitch and frame stride larger than the image
    assert pipe.next_slot() == 0
    pipe.upscale(src, got, fused=0)
    assert pipe.next_slot() == n % 3
    pipe.upscale(src, got, fused=1)  # a fused batch: one launch, one slot
    assert pipe.next_slot() == (n + 1) % 3
    pipe.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert bool((big[:, oh:] == 5.0).all()) and bool((big[:, :, ow:] == 5.0).all())
    pipe.close()
    one = fsr.Pipeline(1)
    got.fill_(0)
    one.upscale(src, got, fused=0)
    assert one.next_slot() == 0
    one.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    one.close()


def test_managed_pipeline_orders_itself_after_the_current_torch_stream(fsr):
    """Pipeline(managed=True), the default: no fork() — each submission's stream waits for what the current torch stream holds (the
    producer of the input), and the tensors stay referenced while in flight even if the caller drops them."""
    iw, ih, ow, oh = 480, 270, 960, 540
    base = dev(frames.synthetic_frame(iw, ih, k=9, dtype=np.float16))
    want = [alone(fsr, torch.roll(base, shifts=(k, 2 * k), dims=(0, 1)).contiguous(), ow, oh, False, 0) for k in range(6)]
    torch.cuda.synchronize()
    pipe = fsr.Pipeline(3)
    side = torch.cuda.Stream()
    outs = []
    with torch.cuda.stream(side):
        for k in range(6):
            big = torch.zeros(64, 1024, 1024, device="cuda")  # work in front of the producer: an unordered reader would be early
            big.add_(1.0)
            src = torch.roll(base, shifts=(k, 2 * k), dims=(0, 1)).contiguous()
            out = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
            pipe.upscale(src, out, fused=0)
            outs.append(out)
            del src, big  # the allocator may not reuse src's memory while the pipeline still reads it
        pipe.join(side)
        sums = [o.float().sum() for o in outs]
    side.synchronize()
    for k in range(6):
        assert torch.equal(outs[k].view(torch.int16), want[k].view(torch.int16)), k
        assert float(sums[k]) == float(want[k].float().sum())
    pipe.close()


def test_pipeline_fork_and_join_order_against_the_callers_stream(fsr):
    """Inputs produced on the caller's stream right before fork(), outputs consumed on it right after join(): no host synchronisation."""
    iw, ih, ow, oh = 480, 270, 960, 540
    pipe = fsr.Pipeline(2, managed=False)  # the bare C ABI: ordering comes from fork / join alone
    base = dev(frames.synthetic_frame(iw, ih, k=3, dtype=np.float16))
    want = [alone(fsr, torch.roll(base, shifts=(k, 2 * k), dims=(0, 1)).contiguous(), ow, oh, False, 0) for k in range(6)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        srcs = []
        for k in range(6):
            big = torch.zeros(64, 1024, 1024, device="cuda")  # some work in front of the producer, so that an unordered reader would be early
            big.add_(1.0)
            srcs.append(torch.roll(base, shifts=(k, 2 * k), dims=(0, 1)).contiguous())
        outs = [torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda") for _ in range(6)]
        pipe.fork(side)
        for s, o in zip(srcs, outs):
            pipe.upscale(s, o, fused=0)
        pipe.join(side)
        sums = [o.float().sum() for o in outs]  # consumer on the caller's stream
    side.synchronize()
    for k in range(6):
        assert torch.equal(outs[k].view(torch.int16), want[k].view(torch.int16)), k
        assert float(sums[k]) == float(want[k].float().sum())
    pipe.close()


def test_pipeline_argument_validation(fsr):
    with pytest.raises(fsr.Fsr1Error):
        fsr.Pipeline(0)
    with pytest.raises(fsr.Fsr1Error):
        fsr.Pipeline(9)
    pipe = fsr.Pipeline(2)
    src = dev(frames.synthetic_frame(64, 36, k=1, dtype=np.float16))
    out = torch.zeros(72, 128, 4, dtype=torch.float16, device="cuda")
    with pytest.raises(fsr.Fsr1Error, match="render size"):
        pipe.upscale(src, out, render_size=(100, 36))
    with pytest.raises(fsr.Fsr1Error):
        pipe.upscale(src, out, flags=fsr.FLAG_MATH_EXACT | fsr.FLAG_MATH_PACKED_FP16)
    pipe.upscale(src, out)  # still usable after a refused call
    pipe.synchronize()
    assert torch.equal(out.view(torch.int16), alone(fsr, src, 128, 72, False, 0).view(torch.int16))
    pipe.close()
    pipe.close()
