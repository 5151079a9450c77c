"""easu_lane_column (include/fsr1_device_easu.hpp): the lane -> output-column permutation of the generic EASU kernels.

The hardware fact it encodes (MI355X_MICROARCH.md, LDS): a wave64 ds_read_b128 is served in four groups of sixteen lanes,
{0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32.  The function must be a permutation of 0..63 that stays inside
each half-wave and hands every such group sixteen CONSECUTIVE columns — then the texels a group reads at any ratio >= 1x span
at most sixteen 16-byte records and never wrap around the 64 LDS banks.  The function's source is lifted from the header and
compiled for the host (it is plain integer arithmetic), so the test runs without a GPU and the device header stays untouched."""
import os
import re
import subprocess

from conftest import ROOT

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def lane_columns(tmp_path):
    text = open(os.path.join(ROOT, "include", "fsr1_device_easu.hpp")).read()
    m = re.search(r"__device__ __forceinline__ int easu_lane_column\(int lane\) \{.*?\n\}\n", text, flags=re.S)
    assert m, "easu_lane_column not found in include/fsr1_device_easu.hpp"
    src = tmp_path / "lane_column.cpp"
    src.write_text("#include <cstdio>\n#define __device__\n#define __forceinline__ inline\n" + m.group(0) +
                   "int main() { for (int l = 0; l < 64; ++l) std::printf(\"%d\\n\", easu_lane_column(l)); return 0; }\n")
    exe = tmp_path / "lane_column"
    subprocess.check_call(["g++", "-O1", "-o", str(exe), str(src)])
    return [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]


def test_lane_column_is_a_half_wave_permutation_with_consecutive_columns_per_lds_group(tmp_path):
    col = lane_columns(tmp_path)
    assert sorted(col) == list(range(64))
    assert sorted(col[:32]) == list(range(32)) and sorted(col[32:]) == list(range(32, 64))  # a row's store covers the same bytes
    for g in GROUPS:
        cols = sorted(col[l] for l in g)
        assert cols == list(range(cols[0], cols[0] + 16)) and cols[0] % 16 == 0, (g, cols)


def test_consecutive_columns_keep_a_group_inside_sixteen_records(tmp_path):
    """What the permutation buys: for every upscale ratio the texels floor(c * in/out + b) read by a group's sixteen columns span
    fewer than sixteen records, so their bank residues (record index mod 16) are distinct; in lane order they are not."""
    col = lane_columns(tmp_path)
    for in_w, out_w in ((2560, 3840), (2259, 3840), (2954, 3840), (1920, 3840), (3839, 3840)):
        scale = in_w / out_w
        for g in GROUPS:
            tex = {int((col[l] * scale) // 1) for l in g}
            assert max(tex) - min(tex) < 16
            assert len({t % 16 for t in tex}) == len(tex)
    # the identity mapping at 1.5x: lanes 0-3, 12-15, 20-27 reach texels 0 .. 18 and two of them share a bank residue
    g = GROUPS[0]
    tex = {int(l * (2560 / 3840)) for l in g}
    assert len({t % 16 for t in tex}) < len(tex)
