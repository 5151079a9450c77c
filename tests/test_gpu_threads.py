"""The C ABI from several host threads at once, each on its own stream (the C runner drives one thread per GPU; a frame server
would drive several per GPU): launches carry their arguments by value and the library keeps no per-call state besides the
thread-local error message and the per-kernel dynamic-LDS cache, so concurrent calls must produce exactly what serial calls do."""
import importlib
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")

JOBS = [  # (in_w, in_h, out_w, out_h, flags name, pipeline)
    (320, 180, 640, 360, "exact", "two-pass"), (300, 170, 450, 255, "f", "fused"), (200, 120, 260, 156, "h", "two-pass"),
    (640, 360, 1280, 720, "f", "two-pass"), (97, 61, 131, 83, "exact", "fused"), (256, 144, 512, 288, "h", "fused"),
    (480, 270, 640, 360, "f", "two-pass"), (160, 90, 400, 225, "exact", "two-pass"),  # 2.5x: > 48 KiB of dynamic LDS
]


def run_job(fsr, job, src, stream, reps):
    iw, ih, ow, oh, fl, pipe = job
    flags = {"f": 0, "exact": fsr.FLAG_MATH_EXACT, "h": fsr.FLAG_MATH_PACKED_FP16}[fl]
    out = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    mid = torch.zeros_like(out)
    for _ in range(reps):
        if pipe == "fused":
            fsr.easu_rcas_fused(src, out, sharpness=0.25, flags=flags, stream=stream)
        else:
            fsr.easu(src, mid, flags=flags, stream=stream)
            fsr.rcas(mid, out, sharpness=0.25, flags=flags, stream=stream)
    stream.synchronize()
    return out


def test_concurrent_callers_match_serial_results(fsr):
    srcs = [torch.from_numpy(frames.synthetic_frame(j[0], j[1], k=i, dtype=np.float16)).cuda() for i, j in enumerate(JOBS)]
    main = torch.cuda.current_stream()
    serial = [run_job(fsr, j, s, main, 1) for j, s in zip(JOBS, srcs)]
    results, errors = [None] * len(JOBS), []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            results[i] = run_job(fsr, JOBS[i], srcs[i], torch.cuda.Stream(), 40)
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append((i, repr(e)))
    for _ in range(3):
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(JOBS))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for i, (a, b) in enumerate(zip(serial, results)):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), "job %d %s differs when run concurrently" % (i, JOBS[i])


def test_error_messages_are_thread_local(fsr):
    """A failing call on one thread must not disturb the message another thread reads."""
    src = torch.zeros(8, 8, 4, dtype=torch.float16, device="cuda")
    seen = {}

    def bad(name, fn):
        try:
            fn()
        except fsr.Fsr1Error as e:
            seen[name] = str(e)
    barrier = threading.Barrier(2)

    def t1():
        barrier.wait()
        for _ in range(200):
            bad("a", lambda: fsr.easu(src, torch.zeros(4, 4, 4, dtype=torch.float16, device="cuda"), flags=1 << 30))

    def t2():
        barrier.wait()
        for _ in range(200):
            bad("b", lambda: fsr.rcas(src, src))
    ts = [threading.Thread(target=t1), threading.Thread(target=t2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert "flag" in seen["a"] and "overlap" in seen["b"], seen
