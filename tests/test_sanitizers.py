"""Sanitizer job (SURVEY.md section 5 lists none in the reference): the CPU oracle and the product's host constant code
(fidelityfx-fsr_amd/csrc/fsr1_con.c) built with AddressSanitizer + UndefinedBehaviorSanitizer and run over ragged, 1x1 and
ratio-extreme shapes — where clamp-to-edge gathers and out-of-bounds-is-zero loads do all the work — plus a bit-for-bit
cross-check of the two constant-setup implementations."""
import os
import subprocess

from conftest import ROOT


def test_oracle_and_host_constants_under_asan_ubsan():
    d = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", d, "sanitize_check"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([os.path.join(d, "sanitize_check")], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failure(s)" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
