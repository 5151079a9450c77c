"""bench.py's self-consistency check of its own line (no GPU): kernels that run back to back on one stream cannot take longer
than the step that contains them.  Round 3's driver line violated exactly that (a 20-launch stopwatch on a GPU that had just
idled read EASU 54.95 us + RCAS 30.2 us inside a 65.17 us step, fused 77.82 us inside a 59.53 us step) and printed a roofline
fraction that followed from neither the committed rocprofv3 summaries nor its own headline."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _line(easu_us, rcas_us, step_ms, fused_us, fused_step_ms):
    roof = lambda name, us, alg: {"kernel": name, "bound": "hbm", "achieved": alg / us / 1e3, "peak": 8000.0, "unit": "GB/s",  # noqa: E731
                                  "frac": alg / us / 1e3 / 8000.0, "avg_kernel_us": us, "algorithmic_bytes": alg,
                                  "read_only": {"frac": 0.05}, "valu": {"frac": 0.6}}
    k = {"easu": roof("easu", easu_us, 82944000), "rcas": roof("rcas", rcas_us, 132710400)}
    k["rcas"]["cold_input"] = {"avg_kernel_us": rcas_us * 1.3, "frac": 0.5}
    return {"ms_per_step": step_ms, "roofline": copy.deepcopy(k["easu"]), "kernels": k,
            "also_measured": {"fused": {"ms_per_step": fused_step_ms, "avg_kernel_us": fused_us, "hbm_frac": 0.17, "valu_frac": 0.6}}}


def test_round3_driver_line_is_flagged():
    line = _line(54.95, 30.2, 0.06517, 77.82, 0.05953)  # BENCH_r03.json's figures
    bad = bench.check_stopwatch(line)
    assert len(bad) == 2 and line["stopwatch_suspect"] is True and len(line["stopwatch_violations"]) == 2
    # nothing derived from the suspect kernel times is printed
    assert "frac" not in line["roofline"] and "achieved" not in line["roofline"] and "read_only" not in line["roofline"]
    for r in line["kernels"].values():
        assert "frac" not in r and "valu" not in r
    assert "frac" not in line["kernels"]["rcas"]["cold_input"]
    assert "hbm_frac" not in line["also_measured"]["fused"] and "valu_frac" not in line["also_measured"]["fused"]
    # the raw measurements stay, so that the reader sees what was wrong
    assert line["kernels"]["easu"]["avg_kernel_us"] == 54.95 and line["also_measured"]["fused"]["avg_kernel_us"] == 77.82


def test_consistent_line_passes_untouched():
    line = _line(40.12, 24.68, 0.06517, 59.87, 0.05953)  # profiles/r03_*'s rocprofv3 averages inside the same steps
    before = copy.deepcopy(line)
    assert bench.check_stopwatch(line) == [] and line["stopwatch_suspect"] is False
    del line["stopwatch_suspect"]
    assert line == before


def test_only_the_fused_figure_suspect():
    line = _line(40.12, 24.68, 0.06517, 77.82, 0.05953)
    bad = bench.check_stopwatch(line)
    assert len(bad) == 1 and "fused" in bad[0] and line["stopwatch_suspect"] is True


def test_median():
    assert bench.median([3, 1, 2]) == 2 and bench.median([4, 1, 2, 3]) == 2.5 and bench.median([7]) == 7
    assert bench.reduce_regions([0.5, 0.25], None) == [0.5, 0.25]  # identity without a process group


def test_pipelined_line_is_judged_against_the_one_stream_step():
    """--streams 2: steps overlap, so the headline step (60.2 us) is shorter than the kernels it contains (40.1 + 24.7 us); the check
    compares the kernels with the in-order step the line carries beside it (config.one_stream) — consistent — and still flags a broken
    stopwatch there."""
    line = _line(40.12, 24.68, 0.0602, 59.87, 0.0564)
    line["config"] = {"streams": 2, "one_stream": {"ms_per_step": 0.0656, "value": 126400.0}}
    line["also_measured"]["fused"]["one_stream"] = {"ms_per_step": 0.0605, "value": 137100.0}
    assert bench.check_stopwatch(line) == [] and line["stopwatch_suspect"] is False
    bad = _line(54.95, 30.2, 0.0602, 77.82, 0.0564)
    bad["config"] = {"streams": 2, "one_stream": {"ms_per_step": 0.0656}}
    bad["also_measured"]["fused"]["one_stream"] = {"ms_per_step": 0.0605}
    assert len(bench.check_stopwatch(bad)) == 2 and bad["stopwatch_suspect"] is True


def test_telemetry_summary_and_gather_without_a_process_group():
    """bench.Telemetry reduces its samples over labelled spans (median MHz, mean W) and never fails a run when a box exposes no
    sysfs files; bench.gather_pairs is the identity at world size 1 and maps "not read" to null (round 5: per-rank clock / power)."""
    import bench

    class _NoGpuTorch:
        class cuda:
            @staticmethod
            def get_device_properties(i):
                raise RuntimeError("no GPU here")
    t = bench.Telemetry(_NoGpuTorch, 0)
    t.close()
    assert t.summary("headline") == (None, None, 0)
    t.samples = [(0.5, 100.0, 200.0), (1.5, 2100.0, 1300.0), (1.6, 2200.0, 1400.0), (1.7, 2000.0, None), (3.0, 90.0, 150.0)]
    t.spans = {"headline": [(1.0, 2.0)], "idle": [(2.5, 3.5)]}
    assert t.summary("headline") == (2100.0, 1350.0, 3)
    assert t.summary("idle") == (90.0, 150.0, 1)
    assert t.summary("nothing") == (None, None, 0)
    import torch
    mhz, watts = bench.gather_pairs(2100.0, None, torch.device("cpu"))
    assert mhz == [2100.0] and watts == [None]


def test_telemetry_block_marks_a_short_sample_unreliable():
    """Round 6: the per-rank clock / power figures come from a dedicated 1.2 s window of the headline's load (>= 200 samples at 2 ms whatever
    --steps is); a line whose window held fewer samples on any rank says so instead of quoting a figure that under-reads a short load
    (round 5's K = 20 line: 57 samples of a 0.1 s run, 20 % low).  The block also carries the power cap, whether each rank sat at it, the
    SMU's throttle-accumulator deltas and the compute-partition mode."""
    import bench
    import torch
    rows = bench.gather_row([2180.0, 1352.0, 600, 1400.0, 1234.0, 0.0], torch.device("cpu"))
    assert rows == [[2180.0, 1352.0, 600.0, 1400.0, 1234.0, 0.0]]
    good = bench.telemetry_block(rows + [[2100.0, 1100.0, 580, 1400.0, None, None]], "sysfs", 2.0, bench.TELEMETRY_WINDOW_S, "SPX")
    assert good["reliable"] is True and "unreliable" not in good
    assert good["samples_per_rank"] == [600, 580] and good["power_cap_w"] == [1400.0, 1400.0]
    assert good["at_power_cap"] == [True, False] and good["compute_partition"] == "SPX"
    assert good["throttle_accumulator_delta"]["ppt"] == [1234.0, None]
    short = bench.telemetry_block([[2049.0, 1082.0, 57, 1400.0, None, None]], "sysfs", 2.0, bench.TELEMETRY_WINDOW_S, "SPX")
    assert short["reliable"] is False and "unreliable" in short
    none = bench.telemetry_block([[None, None, None, None, None, None]], None, 2.0, bench.TELEMETRY_WINDOW_S, None)
    assert none["reliable"] is False and none["at_power_cap"] == [None]
    assert bench.TELEMETRY_WINDOW_S / 0.002 >= 2 * bench.TELEMETRY_MIN_SAMPLES  # the window is sized for twice the minimum
