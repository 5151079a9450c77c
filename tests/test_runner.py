"""The plain-C host runner (runner/fsr1_runner.c): builds against the C ABI; on a GPU box it runs the
two-pass and fused pipelines and reports the counters it gathered over RCCL."""
import json
import os
import subprocess

import pytest

from conftest import ROOT

RUNNER = os.path.join(ROOT, "runner", "fsr1_runner")


@pytest.fixture(scope="module")
def runner(fsr):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "runner")], stdout=subprocess.DEVNULL)
    return RUNNER


def test_runner_builds_and_parses_arguments(runner):
    out = subprocess.run([runner, "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--pipeline" in out.stdout
    assert subprocess.run([runner, "--pipeline", "nope"], capture_output=True).returncode == 2
    assert subprocess.run([runner, "--in", "12by7"], capture_output=True).returncode == 2
    assert subprocess.run([runner, "--bogus", "1"], capture_output=True).returncode == 2


@pytest.mark.parametrize("gpus", [1, 2, 8])
def test_runner_dry_run_host_side_of_n_gpus(runner, gpus):
    """`runner --gpus N --dry-run`: the N-thread host side with no device behind it — contiguous frame shards, the plan, the barrier
    every rank reaches before the collective, the gather of the counters, ONE JSON line with world_size_seen == N and N
    per_rank_seconds (BASELINE configs[2]'s shape at N = 8: 64 frames, 8 per GPU, 1440p -> 4K, `auto` -> two dispatches)."""
    out = subprocess.run([runner, "--dry-run", "--gpus", str(gpus), "--frames", str(8 * gpus), "--in", "2560x1440", "--out", "3840x2160", "--steps", "6",
                          "--pipeline", "auto"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == gpus and d["world_size_seen"] == gpus and d["rccl_ranks"] == 0
    assert len(d["per_rank_seconds"]) == gpus == len(d["per_gpu_ms"]) and min(d["per_rank_seconds"]) >= 0.006
    assert d["frames"] == 8 * gpus * 6 and d["scaling"] == "weak" and d["pipeline_run"] == "two-pass"
    assert abs(d["seconds"] - max(d["per_rank_seconds"])) < 1e-6  # MAX over ranks
    # uneven shards: 11 frames over 8 ranks are 2 2 2 1 1 1 1 1, all of them counted
    out = subprocess.run([runner, "--dry-run", "--gpus", "8", "--frames", "11", "--steps", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["frames"] == 22


def test_runner_dry_run_a_failing_rank_does_not_strand_the_others(runner):
    """One rank failing before the collective: every rank still reaches the barrier, nobody enters the gather, the process ends with
    exit code 1 and no JSON line (instead of hanging in an all-gather that one member never joins)."""
    out = subprocess.run([runner, "--dry-run", "--gpus", "8", "--steps", "3", "--dry-fail", "5"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "gpu 5" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # a plan the library refuses (packed fp16 with colour stages is caught by the option parser; an unusable ratio by the plan)
    out = subprocess.run([runner, "--dry-run", "--gpus", "2", "--in", "640x640", "--out", "10x10", "--steps", "1"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "fsr1_upscale_plan" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", ["two-pass", "fused", "easu", "auto"])
def test_runner_on_gpu(runner, pipeline):
    out = subprocess.run([runner, "--gpus", "1", "--frames", "3", "--in", "640x360", "--out", "1280x720", "--steps", "20",
                          "--warmup", "3", "--pipeline", pipeline], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["frames"] == 60 and d["pipeline"] == pipeline
    assert d["value"] > 1000.0  # Mpix/s; the CPU reference does ~25 on 128 cores
    # the same fields as bench.py's line; `auto` reports the pipeline it took (exactly 2x: the fused quad form) and the
    # algorithmic byte count follows that path, not the request
    assert d["steps"] == 20 and d["warmup"] == 3 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["world_size_seen"] == 1 and len(d["per_rank_seconds"]) == 1 and abs(d["ms_per_step"] - d["seconds"] * 1e3 / 20) < 1e-3
    assert d["pipeline_run"] == {"two-pass": "two-pass", "fused": "fused", "easu": "easu", "auto": "fused"}[pipeline]
    assert d["streams"] == 3 and d["intermediary"] == ("one per stream" if pipeline == "two-pass" else "none")  # default: steps alternate over three streams


@pytest.mark.gpu
def test_runner_strict_latency_and_cpu_binding(runner):
    """Round 6: --math strict (FSR1_FLAG_MATH_STRICT) from the C host, --latency N (N single frames after 1 ms of idle GPU: host clock around
    fsr1_upscale_ex + hipStreamSynchronize), and the per-GPU thread pinned to the CPUs local to its GPU (sysfs local_cpulist; --no-pin leaves it)."""
    out = subprocess.run([runner, "--gpus", "1", "--frames", "1", "--in", "640x360", "--out", "1280x720", "--steps", "20", "--warmup", "3",
                          "--math", "strict", "--pipeline", "auto", "--streams", "1", "--latency", "40"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["math"] == "strict" and d["pipeline_run"] == "fused" and d["value"] > 1000.0
    lat = d["latency_us"]
    assert lat["frames"] == 40 and 5.0 < lat["back_to_back_median"] <= lat["p90"] * 1.5 and lat["median"] <= lat["p90"] < 5000.0
    assert len(d["cpu_binding"]) == 1  # a CPU list where sysfs offers one, null otherwise
    out = subprocess.run([runner, "--gpus", "1", "--frames", "1", "--in", "320x180", "--out", "640x360", "--steps", "5", "--warmup", "1", "--no-pin"],
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["cpu_binding"] == [None]


@pytest.mark.gpu
def test_runner_with_colour_stages(runner):
    """--stages: SRTM prologue + film grain + SRTM inverse fused into the single-launch pipeline, from the C host."""
    out = subprocess.run([runner, "--gpus", "1", "--frames", "2", "--in", "640x360", "--out", "1280x720", "--steps", "10",
                          "--warmup", "2", "--pipeline", "fused", "--stages", "7", "--grain", "0.3"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["color_stages"] == 7 and d["frames"] == 20 and d["value"] > 1000.0


@pytest.mark.gpu
def test_runner_packed_fp16_and_ring(runner):
    """--math h drives FsrEasuH / FsrRcasH from the C host; --ring rotates the steps over several frame sets — at least one more than
    there are streams, so that the outputs of steps in flight together never alias (a smaller request is raised)."""
    out = subprocess.run([runner, "--gpus", "1", "--frames", "2", "--in", "640x360", "--out", "1280x720", "--steps", "12",
                          "--warmup", "2", "--math", "h", "--ring", "5"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["math"] == "h" and d["ring"] == 6 and d["streams"] == 3 and d["frames"] == 24 and d["value"] > 500.0  # 5 sets asked: rounded up to a multiple of the 3 streams
    out = subprocess.run([runner, "--gpus", "1", "--frames", "2", "--in", "640x360", "--out", "1280x720", "--steps", "12",
                          "--warmup", "2", "--ring", "2"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["ring"] == 6  # three streams: at least four sets, rounded up to a multiple of three (a set is only reused on its own slot)
    out = subprocess.run([runner, "--gpus", "1", "--frames", "2", "--in", "640x360", "--out", "1280x720", "--steps", "12",
                          "--warmup", "2", "--ring", "2", "--streams", "1"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["ring"] == 2
    fused = subprocess.run([runner, "--gpus", "1", "--frames", "1", "--in", "320x180", "--out", "640x360", "--steps", "5", "--warmup", "1",
                            "--math", "h", "--pipeline", "fused"], capture_output=True, text=True, timeout=1200)
    assert fused.returncode == 0, fused.stdout + fused.stderr
    bad = subprocess.run([runner, "--math", "h", "--stages", "2"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2


@pytest.mark.gpu
def test_runner_two_gpus_over_rccl(runner, fsr):
    """N = 2: one host thread per GPU, frames sharded in contiguous blocks, the counters gathered over RCCL (skipped on a
    single-GPU box: the driver's scaling run is the 8-GPU evidence)."""
    import ctypes
    if fsr.load().fsr1_device_count() < 2:
        pytest.skip("needs two visible GPUs")
    out = subprocess.run([runner, "--gpus", "2", "--frames", "5", "--in", "960x540", "--out", "1920x1080", "--steps", "20",
                          "--warmup", "3"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["world_size_seen"] == 2 and d["frames"] == 100 and len(d["per_gpu_ms"]) == 2
    assert all(ms > 0 for ms in d["per_gpu_ms"])


@pytest.mark.gpu
def test_runner_row_bands(runner):
    """--bands: one frame stream split into row bands (here a single band on one GPU; per-band parity with the full frame is
    tests/test_gpu_bands.py's job).  The pixel count is the whole frame's."""
    out = subprocess.run([runner, "--gpus", "1", "--bands", "--in", "960x540", "--out", "1920x1080", "--steps", "10", "--warmup", "2"],
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["bands"] == 1 and d["frames"] == 10 and d["value"] > 1000.0
    fused = subprocess.run([runner, "--gpus", "1", "--bands", "--pipeline", "fused", "--in", "960x540", "--out", "1920x1080", "--steps", "10", "--warmup", "2"],
                           capture_output=True, text=True, timeout=1200)
    assert fused.returncode == 0, fused.stdout + fused.stderr
    d = json.loads(fused.stdout.strip().splitlines()[-1])
    assert d["bands"] == 1 and d["pipeline"] == "fused" and d["value"] > 1000.0
    assert subprocess.run([runner, "--bands", "--pipeline", "easu"], capture_output=True, text=True, timeout=60).returncode == 2
