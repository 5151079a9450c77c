"""Host geometry of the column-walking fused exact-2x kernel (fidelityfx-fsr_amd/csrc/fsr1_fused_s2.hip): how many 16-row steps a
workgroup takes and how many runs cover an image.  The two host functions are lifted from the source and compiled for the host
(plain integer arithmetic), so the rule the A/B measurements of DESIGN.md section 3.3 led to is pinned without a GPU:
one 4K frame stays on one-step tiles, four 4K frames or one 8K frame walk four steps, the 16-frame 8K batch eight; whatever the
number, the runs cover every output row and none starts outside the image."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def geometry(tmp_path_factory):
    text = open(os.path.join(ROOT, "fidelityfx-fsr_amd", "csrc", "fsr1_fused_s2.hip")).read()
    consts = re.search(r"constexpr int kFs2OutW = .*?;\nconstexpr int kFs2QH = .*?;", text, flags=re.S)
    steps = re.search(r"#ifndef FSR1_FUSED_S2_MAX_STEPS.*?\nint fused_s2_run_steps\(int width, int height, int frames\) \{.*?\n\}\n", text, flags=re.S)
    geo = re.search(r"void fused_s2_geometry\(int width, int height, int steps, int\* tiles_x, int\* tiles_y\) \{.*?\n\}\n", text, flags=re.S)
    assert consts and steps and geo, "host geometry functions not found in fsr1_fused_s2.hip"
    tmp = tmp_path_factory.mktemp("walk")
    src = tmp / "walk.cpp"
    src.write_text("#include <cstdio>\n#include <cstdlib>\n" + consts.group(0) + "\n" + steps.group(0) + geo.group(0) +
                   "int main(int argc, char** argv) { int w = atoi(argv[1]), h = atoi(argv[2]), f = atoi(argv[3]);\n"
                   "  int s = fused_s2_run_steps(w, h, f), tx, ty; fused_s2_geometry(w, h, s, &tx, &ty);\n"
                   "  std::printf(\"%d %d %d %d\\n\", s, tx, ty, kFs2Step); return 0; }\n")
    exe = tmp / "walk"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", str(exe), str(src)])

    def run(w, h, frames, forced=None):
        env = {k: v for k, v in os.environ.items() if k != "FSR1_FUSED_S2_STEPS"}
        if forced is not None:
            env["FSR1_FUSED_S2_STEPS"] = str(forced)
        return [int(x) for x in subprocess.check_output([str(exe), str(w), str(h), str(frames)], text=True, env=env).split()]
    return run


@pytest.mark.parametrize("w,h,frames,steps", [(3840, 2160, 1, 1), (1920, 1080, 1, 1), (320, 180, 1, 1), (3840, 2160, 2, 2), (3840, 2160, 4, 4),
                                             (7680, 4320, 1, 4), (7680, 4320, 16, 8), (3840, 2160, 64, 8)])
def test_steps_follow_the_size_of_the_launch(geometry, w, h, frames, steps):
    s, tx, ty, step = geometry(w, h, frames)
    assert s == steps and step == 16
    assert tx == -(-w // 62)
    run = 16 * s - 2
    assert (ty - 1) * run < h <= ty * run  # the runs cover every row, and the last one starts inside the image


@pytest.mark.parametrize("forced", [1, 2, 3, 5, 9, 11, 40])
def test_forced_steps_still_cover_the_image(geometry, forced):
    for (w, h) in ((3840, 2160), (194, 320), (62, 14), (63, 15), (1, 1)):
        s, tx, ty, _ = geometry(w, h, 3, forced)
        run = 16 * s - 2
        assert s == forced and tx * 62 >= w and (ty - 1) * run < h <= ty * run
