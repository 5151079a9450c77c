"""Constant setup: bit-exact against the reference (golden words generated from the reference's
A_CPU build, tests/golden/con_kat.json) — product library, oracle port and,
where present, oracle/_ref."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

with open(os.path.join(GOLDEN, "con_kat.json")) as fh:
    KAT = json.load(fh)


def words(c):
    return ["%08x" % v for v in np.asarray(c, np.uint32)]


@pytest.mark.parametrize("case", KAT["easu"], ids=lambda c: "x".join(str(int(v)) for v in c["args"]))
def test_easu_con(fsr, port, case):
    assert words(fsr.FsrEasuCon(*case["args"])) == case["con"]
    assert words(port.FsrEasuCon(*case["args"])) == case["con"]


@pytest.mark.parametrize("case", KAT["easu_offset"], ids=lambda c: "x".join(str(v) for v in c["args"]))
def test_easu_con_offset(fsr, port, case):
    assert words(fsr.FsrEasuConOffset(*case["args"])) == case["con"]
    assert words(port.FsrEasuConOffset(*case["args"])) == case["con"]


@pytest.mark.parametrize("case", KAT["rcas"], ids=lambda c: str(c["stops"]))
def test_rcas_con(fsr, port, case):
    assert words(fsr.FsrRcasCon(case["stops"])) == case["con"]
    assert words(port.FsrRcasCon(case["stops"])) == case["con"]


def test_survey_appendix_b1_words(fsr):
    # SURVEY.md Appendix B.1, copied by hand (independent of the generated JSON)
    c = fsr.FsrEasuCon(2560, 1440, 2560, 1440, 3840, 2160)
    assert words(c[:4]) == ["3f2aaaab", "3f2aaaaa", "be2aaaaa", "be2aaaac"]  # reciprocal-multiply quirk: x != y
    assert words(c[4:8]) == ["39cccccd", "3a360b61", "39cccccd", "ba360b61"]
    c = fsr.FsrEasuConOffset(1280, 720, 1920, 1080, 2560, 1440, 16, 8)
    assert words(c[:4]) == ["3f000000", "3f000000", "417c0000", "40f80000"]
    assert words(fsr.FsrRcasCon(0.2)) == ["3f5edc67", "3af63af6", "00000000", "00000000"]  # truncating half: 3af6 not 3af7
    assert words(fsr.FsrRcasCon(0.25)) == ["3f5744fd", "3aba3aba", "00000000", "00000000"]


def test_half_truncation_kat(fsr, port):
    for case in KAT["half"]:
        f = np.array([int(case["f_bits"], 16)], np.uint32).view(np.float32)[0]
        assert "%04x" % fsr.AU1_AH1_AF1(f) == case["h"], case
        assert "%04x" % port.AU1_AH1_AF1(f) == case["h"], case


def test_half_truncation_sweep(fsr, port, request):
    """Every sign/exponent with a spread of mantissas: the arithmetic restatements of the table
    conversion agree with each other (and with the reference build where it is available)."""
    import cpu_oracle
    refo = cpu_oracle.ref() if cpu_oracle.have_ref() else None
    mant = np.array([0, 1, 0x1fff, 0x2000, 0x2001, 0x3fffff, 0x400000, 0x7fffff, 0x555555, 0x2aaaaa, 0x001000, 0x7fe000], np.uint32)
    for se in range(512):
        for m in mant:
            f = np.array([(se << 23) | int(m)], np.uint32).view(np.float32)[0]
            a = fsr.AU1_AH1_AF1(f)
            assert a == port.AU1_AH1_AF1(f), hex((se << 23) | int(m))
            if refo is not None:
                assert a == refo.AU1_AH1_AF1(f), hex((se << 23) | int(m))


def test_ref_cpu_and_gpu_builds_agree(ref):
    """The reference's A_CPU and A_GPU builds of FsrEasuCon produce the same words."""
    import ctypes
    fn = ref.lib.ref_easu_con_gpu
    fn.argtypes = [ctypes.POINTER(ctypes.c_uint32)] + [ctypes.c_float] * 6
    for case in KAT["easu"]:
        c = np.zeros(16, np.uint32)
        fn(c.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), *case["args"])
        assert words(c) == case["con"]
