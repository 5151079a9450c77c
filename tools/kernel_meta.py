#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch figures from the code objects embedded in a HIP shared library.

  python tools/kernel_meta.py fidelityfx-fsr_amd/libfsr1_hip.so  ->  JSON {demangled kernel name: {vgpr, sgpr, lds_static_bytes, scratch_bytes, vgpr_spills}}

These come from the AMDGPU metadata notes of the gfx950 code objects (what the loader uses), not from the profiler's
trace columns — rocprofv3 on this image reports VGPR 24 / LDS 0 for kernels that use 48 VGPRs and dynamic LDS.
Dynamic LDS is chosen per launch by the host (fsr1_api.hip) and is not part of the code object.
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

def _llvm_bin():
    """llvm-objdump / llvm-readelf of the ROCm tree: $ROCM_PATH, hipconfig --rocmpath, /opt/rocm, then PATH."""
    roots = [os.environ.get("ROCM_PATH"), "/opt/rocm"]
    try:
        roots.insert(1, subprocess.run(["hipconfig", "--rocmpath"], capture_output=True, text=True, timeout=20).stdout.strip())
    except (OSError, subprocess.SubprocessError):
        pass
    for r in roots:
        if r and os.path.exists(os.path.join(r, "lib", "llvm", "bin", "llvm-objdump")):
            return os.path.join(r, "lib", "llvm", "bin")
    found = shutil.which("llvm-objdump")
    if found:
        return os.path.dirname(found)
    raise FileNotFoundError("llvm-objdump not found under $ROCM_PATH, `hipconfig --rocmpath`, /opt/rocm or PATH")


LLVM = None


def kernel_meta(lib):
    global LLVM
    if LLVM is None:
        LLVM = _llvm_bin()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "gfx9" not in f:  # the code objects llvm-objdump --offloading extracts are named after their architecture (gfx950 here)
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count")[1:]:
                g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(.*", "", dem).replace("void ", "").strip()
                out[dem] = {"vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"), "lds_static_bytes": g("group_segment_fixed_size"),
                            "scratch_bytes": g("private_segment_fixed_size"), "vgpr_spills": g("vgpr_spill_count")}
    return out


if __name__ == "__main__":
    json.dump(kernel_meta(sys.argv[1]), sys.stdout, indent=1, sort_keys=True)
    print()
