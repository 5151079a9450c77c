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
    consts = re.search(r"constexpr int kFs2OutW = .*?;\nconstexpr int kFs2QH = .*?;", open(os.path.join(ROOT, "fidelityfx-fsr_amd", "csrc", "fsr1_device.h")).read(), flags=re.S)
    steps = re.search(r"constexpr int kFs2MaxSteps = .*?\nint fused_s2_run_steps\(int width, int height, int frames, int cus, int wgs_per_cu, bool overlapped, bool strict\) \{.*?\n\}\n", text, flags=re.S)
    geo = re.search(r"bool fused_s2_tall_tiles\(int width, int height, int frames, int steps, int cus, int fmt\) \{.*?\n\}\n\nvoid fused_s2_geometry\(int width, int height, int steps, int\* tiles_x, int\* tiles_y, int step_rows\) \{.*?\n\}\n", text, flags=re.S)
    assert consts and steps and geo, "host geometry functions not found in fsr1_fused_s2.hip"
    tmp = tmp_path_factory.mktemp("walk")
    src = tmp / "walk.cpp"
    # the launch-shape overrides come from the test library's translation unit (csrc/fsr1_test_hooks.cpp), compiled alongside
    src.write_text("#include <cstdio>\n#include <cstdlib>\n#include \"fsr1_hip_test.h\"\n#include \"fsr1_overrides.h\"\nusing namespace fsr1;\n" + consts.group(0) + "\n" + steps.group(0) + geo.group(0) +
                   "int main(int argc, char** argv) { int w = atoi(argv[1]), h = atoi(argv[2]), f = atoi(argv[3]), cus = atoi(argv[4]);\n"
                   "  fsr1_debug_fused_run_steps(atoi(argv[5]));  // the test hook (include/fsr1_hip_test.h); 0 = the rule\n"
                   "  int s = fused_s2_run_steps(w, h, f, cus, argc > 6 ? atoi(argv[6]) : 7, argc > 7 && atoi(argv[7]), argc > 8 && atoi(argv[8])), tx, ty; fused_s2_geometry(w, h, s, &tx, &ty, kFs2Step);\n"
                   "  int tall = fused_s2_tall_tiles(w, h, f, s, cus, 0), ttx = 0, tty = 0; if (tall) fused_s2_geometry(w, h, s, &ttx, &tty, 2 * kFs2Step);\n"
                   "  std::printf(\"%d %d %d %d %d %d %d\\n\", s, tx, ty, kFs2Step, tall, ttx, tty); return 0; }\n")
    exe = tmp / "walk"
    csrc = os.path.join(ROOT, "fidelityfx-fsr_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", csrc, "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src), os.path.join(csrc, "fsr1_test_hooks.cpp")])

    def run(w, h, frames, forced=0, cus=256, wgs=7, overlapped=0, strict=0):
        return [int(x) for x in subprocess.check_output([str(exe), str(w), str(h), str(frames), str(cus), str(forced), str(wgs), str(overlapped), str(strict)], text=True).split()]
    return run


@pytest.mark.parametrize("w,h,frames,steps", [(3840, 2160, 1, 1), (1920, 1080, 1, 1), (320, 180, 1, 1), (3840, 2160, 2, 2), (3840, 2160, 4, 4),
                                             (7680, 4320, 1, 4), (7680, 4320, 16, 8), (3840, 2160, 64, 8)])
def test_steps_follow_the_size_of_the_launch(geometry, w, h, frames, steps):
    s, tx, ty, step = geometry(w, h, frames)[:4]
    assert s == steps and step == 16
    assert tx == -(-w // 62)
    run = 16 * s - 2
    assert (ty - 1) * run < h <= ty * run  # the runs cover every row, and the last one starts inside the image


def test_steps_scale_with_the_compute_units(geometry):
    """The rule counts residencies of the device it runs on (hipDeviceAttributeMultiprocessorCount), not of the MI355X it was
    measured on: a 32-CU partition walks the same 4K frame in longer runs, a 304-CU part in the same one-step tiles."""
    assert geometry(3840, 2160, 1, cus=32)[0] == 8 and geometry(3840, 2160, 1, cus=304)[0] == 1 and geometry(3840, 2160, 4, cus=128)[0] == 8
    # the packed-fp16 twin holds five workgroups per CU instead of seven: the same launches are more residencies long
    assert geometry(3840, 2160, 1, wgs=5)[0] == 1 and geometry(3840, 2160, 4, wgs=5)[0] == 6 and geometry(7680, 4320, 16, wgs=5)[0] == 8
    assert geometry(3840, 2160, 1, forced=200)[0] == 64 and geometry(3840, 2160, 1, forced=-3)[0] == 1  # the hook clamps to 0 .. 64


def test_overlapped_launches_walk_longer_runs(geometry):
    """FSR1_FLAG_FRAMES_OVERLAP (frames pipelined over several streams, fsr1_pipeline): the tail of a launch is filled by the next
    frame's head, so a single 4K frame walks 4 steps, a 1440p output 2, a 1080p output stays on one-step tiles, an 8K frame 8 — the
    measured optima of profiles/ab_r04/r4c4_two_stream_walk.log; and the runs still cover the image."""
    assert geometry(3840, 2160, 1, overlapped=1)[0] == 4 and geometry(2560, 1440, 1, overlapped=1)[0] == 2
    assert geometry(1920, 1080, 1, overlapped=1)[0] == 1 and geometry(7680, 4320, 1, overlapped=1)[0] == 8
    assert geometry(3840, 2160, 1, overlapped=1)[4] == 0  # a walking launch is never tall
    for (w, h) in ((3840, 2160), (2560, 1440), (194, 320)):
        s, tx, ty = geometry(w, h, 1, overlapped=1)[:3]
        run = 16 * s - 2
        assert tx * 62 >= w and (ty - 1) * run < h <= ty * run


def test_strict_launches_walk_at_most_two_steps(geometry):
    """F-strict (round 6): every step ends with the serial re-evaluation of its queued pixels while the other waves wait — runs of at most
    two steps, alone or overlapped (profiles/ab_r06/r6c5_strict_fused_shapes.log); a forced number of steps is still taken."""
    assert geometry(3840, 2160, 1, strict=1)[0] == 1 and geometry(3840, 2160, 1, overlapped=1, strict=1)[0] == 2
    assert geometry(7680, 4320, 16, strict=1)[0] == 2 and geometry(3840, 2160, 1, forced=5, strict=1)[0] == 5


def test_tall_tiles_for_one_step_launches_that_fill_the_chip(geometry):
    """A one-step launch with at least four residencies of 512-thread workgroups takes the 62 x 30 tile (32 EASU rows per step): one 4K
    frame does, 1440p and smaller outputs keep the 256-thread tile, walking launches (steps > 1) are never tall; the tall tiles cover
    every row and the last one starts inside the image."""
    s, tx, ty, step, tall, ttx, tty = geometry(3840, 2160, 1)
    assert (s, tall, ttx, tty) == (1, 1, 62, 72) and (tty - 1) * 30 < 2160 <= tty * 30
    assert geometry(2560, 1440, 1)[4] == 0 and geometry(1920, 1080, 1)[4] == 0 and geometry(320, 180, 1)[4] == 0
    assert geometry(3840, 2160, 4)[4] == 0 and geometry(7680, 4320, 1)[4] == 0  # these walk (4 steps)
    assert geometry(3840, 2160, 1, forced=2)[4] == 0
    assert geometry(3840, 2160, 1, cus=32)[4] == 0  # a 32-CU partition walks the frame instead (8 steps)
    assert geometry(2560, 1440, 1, cus=128)[4:] == [1, 42, 48]


@pytest.mark.parametrize("forced", [1, 2, 3, 5, 9, 11, 40])
def test_forced_steps_still_cover_the_image(geometry, forced):
    for (w, h) in ((3840, 2160), (194, 320), (62, 14), (63, 15), (1, 1)):
        s, tx, ty = geometry(w, h, 3, forced)[:3]
        run = 16 * s - 2
        assert s == forced and tx * 62 >= w and (ty - 1) * run < h <= ty * run
