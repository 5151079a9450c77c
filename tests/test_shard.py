"""Frame sharding across ranks (the only multi-GPU logic of the path) — pure index arithmetic plus a
world_size-2 gloo run of the counter reduction bench.py performs."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_frames_for_rank_partitions(fsr):
    for total in (0, 1, 7, 8, 64, 128, 129):
        for world in (1, 2, 3, 4, 8):
            blocks = [fsr.frames_for_rank(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert fsr.frames_for_rank(64, 3, 8) == (24, 32)   # BASELINE config 3: 8 frames per GPU
    assert fsr.frames_for_rank(128, 7, 8) == (112, 128)  # config 5: 16 per GPU
    with pytest.raises(ValueError):
        fsr.frames_for_rank(8, 8, 8)


WORKER = r"""
import os, sys, importlib
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
fsr = importlib.import_module("fidelityfx-fsr_amd")
import bench
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
b, e = fsr.frames_for_rank(13, rank, world)
# each rank "processed" its frames in a rank-dependent time; the job time is the max over ranks
res = bench.reduce_counters(frames=e - b, pixels=(e - b) * 100, seconds=0.5 + rank, device=torch.device("cpu"))
assert res["frames"] == 13 and res["pixels"] == 1300 and abs(res["seconds"] - (0.5 + world - 1)) < 1e-9, res
print("rank", rank, "ok", res)
dist.destroy_process_group()
"""


def test_counter_reduction_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def _bench_line(cmd, env=None, timeout=600):
    import json
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line expected, got %d:\n%s" % (len(lines), out.stdout[-2000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 8])
@pytest.mark.parametrize("style", ["bare", "torchrun"])
def test_bench_launch_styles_stub(n, style):
    """Both ways the driver may start `bench.py --gpus N` — a bare `python bench.py --gpus N` (bench.py starts the ranks
    itself) and `python -m torch.distributed.run ... bench.py --gpus N` — give one JSON line with the contract's fields,
    the world size the process group saw and every rank's seconds.  --stub: rendezvous, barriers and the counters-only
    collectives over gloo with 1 ms of sleep per step instead of GPU work, so the plumbing runs on the CPU-only builder.
    n = 8 is the driver's largest SCALE point (one node): eight ranks, eight per_rank_seconds, world_size_seen == 8."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FSR1_BENCH_SELF_LAUNCHED")}
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "6", "--warmup", "2", "--stub"]
    if style == "bare":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(29640 + n)] + tail
    line = _bench_line(cmd, env)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert line["n_gpus"] == n and line["steps"] == 6 and line["warmup"] == 2 and line["scaling"] == "weak" and line["data"] == "stub"
    assert line["config"]["world_size_seen"] == n and len(line["per_rank_seconds"]) == n
    assert line["config"]["frames_total"] == 6 * n  # weak scaling: every rank did its own K steps
    # R regions of exactly K steps; ms_per_step = median over regions of the MAX over ranks, per_rank_seconds = each rank's own median
    reg = line["config"]["region_ms_per_step"]
    assert line["config"]["regions"] == 7 and reg["min"] <= reg["median"] <= reg["max"] and reg["median"] == line["ms_per_step"]
    assert reg["min"] >= 1.0  # a step sleeps 1 ms
    assert line["ms_per_step"] >= min(line["per_rank_seconds"]) * 1e3 / 6 - 1e-3
    # (median of per-region maxima vs maximum of per-rank medians: equal on an idle host; eight sleeping ranks on a busy 8-CPU builder scatter)
    assert abs(line["ms_per_step"] - max(line["per_rank_seconds"]) * 1e3 / 6) < 0.5 * line["ms_per_step"]
    if n > 1:
        assert ("self-launched" in line["config"]["launch"]) == (style == "bare")


def test_bench_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--stub"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and "does not match" in out.stderr


@pytest.mark.gpu
def test_bench_self_launch_two_ranks_on_the_visible_gpus():
    """The real bench step under the N-rank launch on whatever this box has: two ranks, each driving the HIP path, counters over
    RCCL when two GPUs are visible, over gloo with the ranks sharing the one GPU otherwise (RCCL refuses two ranks on a device)."""
    torch = pytest.importorskip("torch")
    two = torch.cuda.device_count() >= 2
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--workload", "540p_to_1080p"]
    if not two:
        cmd += ["--backend", "gloo", "--oversubscribe"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FSR1_BENCH_SELF_LAUNCHED")}
    line = _bench_line(cmd, env, timeout=900)
    assert line["n_gpus"] == 2 and line["config"]["world_size_seen"] == 2 and len(line["per_rank_seconds"]) == 2
    assert line["config"]["oversubscribed"] == (not two) and line["value"] > 0 and "roofline" in line
    # round 5: per-rank clock / power travel in one more counters-only collective; at N > 1 only the headline regions run
    assert len(line["per_rank_mhz"]) == 2 and len(line["per_rank_watts"]) == 2
    assert line.get("also_measured") is None and "skipped at n_gpus > 1" in line["also_measured_note"]
    assert "latency_us" not in line and "cpu_baseline" not in line and line["parity_class"] == "F-strict"  # (round 6: the default arithmetic of the bench)
    # round 6: the telemetry comes from a dedicated window (>= 200 samples per rank), with the power cap, the partition mode, the CPUs each rank is
    # pinned to and every rank's host submission ceiling
    t = line["telemetry"]
    assert len(t["samples_per_rank"]) == 2 and len(t["power_cap_w"]) == 2 and len(t["at_power_cap"]) == 2
    assert t["reliable"] == (min(t["samples_per_rank"]) >= 200) and ("unreliable" in t) == (not t["reliable"])
    assert "cpu_binding" in line["config"] and len(line["host_submission"]["per_rank_steps_per_s"]) == 2 and line["host_submission"]["margin"] > 0


@pytest.mark.gpu
def test_bench_two_gpus_over_rccl():
    """bench.py --gpus 2 launched the way the driver launches it (one process per GPU, RCCL): skipped on a single-GPU box."""
    import json
    import subprocess
    import sys
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    from conftest import ROOT
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "10", "--no-cpu-baseline", "--workload", "540p_to_1080p"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("style", ["bare", "torchrun"])
def test_bench_rccl_group_at_one_rank(style):
    """--collective always: the RCCL communicator, barrier, all-reduce and all-gather of the N-rank path, run on the one GPU a box
    has (world size 1) — the part of the multi-GPU path that is not index arithmetic, exercised on hardware."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--collective", "always", "--steps", "50", "--warmup", "10", "--no-cpu-baseline", "--no-also",
            "--no-cold-rcas", "--workload", "540p_to_1080p"]
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable] + tail if style == "bare" else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["world_size_seen"] == 1 and line["config"]["collective_backend"].startswith("rccl")
    assert line["value"] > 0 and len(line["per_rank_seconds"]) == 1
