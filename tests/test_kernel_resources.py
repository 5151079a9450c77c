"""Register budgets of the built gfx950 kernels, read from the code objects inside libfsr1_hip.so (tools/kernel_meta.py) — no GPU.

The kernels' speed rests on how many waves a SIMD holds (MI355X: 512 VGPRs per lane and SIMD: <= 64 registers -> 8 waves,
<= 72 -> 7, <= 80 -> 6), and a spilled register in a filter loop costs more than any of the tunings DESIGN.md records.  These are
the budgets the measurements of DESIGN.md section 3 were taken at; a compiler or source change that moves one shows up here, before
a GPU is involved."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "fidelityfx-fsr_amd", "libfsr1_hip.so")


@pytest.fixture(scope="module")
def meta(fsr):
    # (the `fsr` fixture builds the library when it is missing: this check never skips — the waves_per_eu budgets on the EASU and
    #  fused templates turn a kernel that outgrows them into silent spills, and this is the guard)
    assert os.path.exists(LIB)
    import kernel_meta
    m = kernel_meta.kernel_meta(LIB)
    assert len(m) > 100, "expected every kernel instantiation of the library"
    return m


def test_no_kernel_spills_or_uses_scratch(meta):
    bad = {k: v for k, v in meta.items() if v["vgpr_spills"] or v["scratch_bytes"]}
    assert not bad, bad


@pytest.mark.parametrize("kernel,max_vgpr", [
    # the headline's kernels: exact-2x EASU (row-pair form) at seven waves, RCAS with the 2-row ring at eight
    ("fsr1::easu_kernel<0, false, false, 0, true, false, 0, 16, 4, false>", 72),
    ("fsr1::easu_kernel<0, false, false, 0, true, false, 0, 32, 4, false>", 72),  # the 64 x 32 tile of large / overlapped launches
    ("fsr1::rcas_kernel<0, false, false, false, 0, 2>", 64),
    # EXACT exact-2x EASU with the quad's analyses shared: seven waves
    ("fsr1::easu_kernel<0, true, false, 0, true, false, 0, 16, 4, false>", 72),
    # generic EASU (pitched layout): eight waves
    ("fsr1::easu_kernel<0, false, false, 0, false, false, 48, 16, 4, false>", 64),
    # ... and its 512-thread form on 64 x 32 tiles (round 5): four workgroups of eight waves per CU = eight waves per SIMD
    ("fsr1::easu_kernel<0, false, false, 0, false, false, 48, 32, 8, false>", 64),
    ("fsr1::easu_kernel<0, false, false, 0, false, false, 56, 32, 8, false>", 64),
    # F-strict (round 6): the default arithmetic + the rounding-boundary test + the in-workgroup re-evaluation in the reference's order.
    # 64 x 16 exact-2x tiles at seven waves; 64 x 32 at six (26.4 KB of LDS: six workgroups per CU anyway); generic at eight
    ("fsr1::easu_kernel<0, false, false, 0, true, false, 0, 16, 4, true>", 72),
    ("fsr1::easu_kernel<0, false, false, 0, true, false, 0, 32, 4, true>", 80),
    ("fsr1::easu_kernel<0, false, false, 0, false, false, 48, 32, 8, true>", 64),
    ("fsr1::easu_kernel<0, false, false, 0, false, false, 56, 32, 8, true>", 64),
    # fused exact-2x launch, one-step and walking form: seven workgroups per CU is what their LDS admits
    ("fsr1::fused_s2_kernel<0, false, false, 4, false>", 72),
    ("fsr1::fused_s2_kernel<0, false, true, 4, false>", 72),
    ("fsr1::fused_s2_kernel<0, true, true, 4, false>", 72),
    # the tall one-step tile (512 threads): three workgroups of eight waves per CU = six waves per SIMD
    ("fsr1::fused_s2_kernel<0, false, false, 8, false>", 80),
    ("fsr1::fused_s2_kernel<0, true, false, 8, false>", 80),
    # ... and their F-strict forms: the walking one at seven waves, the one-step tile at six
    ("fsr1::fused_s2_kernel<0, false, true, 4, true>", 72),
    ("fsr1::fused_s2_kernel<0, false, false, 4, true>", 80),
    # packed fp16
    ("fsr1::easu_h_kernel<true>", 64),
    ("fsr1::rcas_h_kernel<false>", 64),
])
def test_occupancy_critical_kernels_keep_their_register_budget(meta, kernel, max_vgpr):
    assert kernel in meta, "kernel instantiation not found: %s" % kernel
    assert meta[kernel]["vgpr"] <= max_vgpr, (kernel, meta[kernel])


def test_walking_fused_kernel_fits_seven_workgroups_of_sgprs(meta):
    """256-thread workgroups are admitted per CU up to floor(800 / (ceil(sgpr / 16) * 16 + 16)) (MI355X_MICROARCH.md, residency):
    seven workgroups need at most 96 SGPRs."""
    for k, v in meta.items():
        if k.startswith("fsr1::fused_s2_kernel<") and (k.endswith(", 4, false>") or k.endswith(", true, 4, true>")):
            assert 800 // ((v["sgpr"] + 15) // 16 * 16 + 16) >= 7, (k, v)


def test_no_experiment_switches_in_the_product_sources():
    """Tuning experiments live as patches under tools/experiments_r0x/, not as `#ifdef FSR1_...` blocks in the kernels (a mistyped -D
    would build a wrong-answer library), and the library reads no tuning value from the environment on a launch path."""
    import glob
    import re
    hits = []
    for f in glob.glob(os.path.join(ROOT, "fidelityfx-fsr_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*")):
        if not f.endswith((".hip", ".h", ".hpp", ".c")):
            continue
        for n, line in enumerate(open(f), 1):
            if re.match(r"\s*#\s*ifdef\s+FSR1_", line) or re.match(r"\s*#\s*if\s+defined\s*\(?\s*FSR1_", line):
                hits.append("%s:%d: %s" % (os.path.relpath(f, ROOT), n, line.strip()))
            if re.match(r"\s*#\s*ifndef\s+FSR1_", line) and not re.search(r"_(H|HPP)\b", line):
                hits.append("%s:%d: %s" % (os.path.relpath(f, ROOT), n, line.strip()))
            if "getenv(" in line and "FSR1_ROCTX" not in line:
                hits.append("%s:%d: %s" % (os.path.relpath(f, ROOT), n, line.strip()))
    assert not hits, hits


def test_strict_threshold_and_its_fixture_agree():
    """tests/golden/strict_worst_tiles.json (the adversarial searches' worst input tiles, tests/test_gpu_strict.py) names the threshold its
    distances are judged against: it must be the kernels' kEasuStrictK, and every recorded distance must lie below it with the margin the
    header states (the threshold is 1.5 x the largest distance any search has produced)."""
    import json
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "include", "fsr1_device_easu.hpp")).read()
    k = float(re.search(r"constexpr float kEasuStrictK = ([0-9.]+)f;", src).group(1))
    doc = json.load(open(os.path.join(root, "tests", "golden", "strict_worst_tiles.json")))
    assert doc["threshold"] == k
    assert len(doc["tiles"]) >= 10
    worst = max(t["max_d_measured"] for t in doc["tiles"])
    assert worst * 1.5 <= k + 0.5, "kEasuStrictK = %g is less than 1.5 x the largest recorded distance %.1f" % (k, worst)
    for t in doc["tiles"]:
        T = t["T"]
        assert len(t["rgb_bits"]) == T and all(len(r) == T and all(len(p) == 3 for p in r) for r in t["rgb_bits"])
        assert (t["in"][0] * t["num"]) % t["den"] == 0 and (t["in"][1] * t["num"]) % t["den"] == 0
