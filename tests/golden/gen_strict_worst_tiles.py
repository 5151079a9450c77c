"""tests/golden/strict_worst_tiles.json from the adversarial searches' logs (tools/experiments_r06/strict_adversarial.py ->
profiles/ab_r06/r06_strict_adversarial.json: the worst tile of each of the eight ratios; r06_strict_adversarial_islands.json: the two tiles
beyond d = 35): the input tile (T x T texels, R G B as binary16 bit patterns) on which the default arithmetic and the reference's operation
order were furthest apart, with the distance that was measured (units of 2^-24 x the window's magnitude; F-strict's threshold is 56 since
this search, 48 before), the input size of the search and the tile position it was found at (at ratios other than 2x the sub-texel
position is rounded in binary32 from the absolute pixel coordinate, so the distance is a property of the tile AT that position).  Data only:
inputs and a measured figure."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = {"what": "worst input tiles of the F-strict adversarial searches (round 6), binary16 bit patterns, row-major [T][T][3]", "threshold": 56, "tiles": []}
for name, keep in (("r06_strict_adversarial.json", lambda r: True), ("r06_strict_adversarial_islands.json", lambda r: r["max_d"] > 35)):
    src = json.load(open(os.path.join(ROOT, "profiles", "ab_r06", name)))
    for r in src["runs"]:
        if keep(r):
            num, den = (int(x) for x in r["ratio"].split("/"))
            out["tiles"].append({"num": num, "den": den, "T": r["tile_texels"], "max_d_measured": r["max_d"], "in": r["in"], "at_tile": r["found"]["tile"],
                                 "rgb_bits": r["worst_tile_rgb_binary16_bits"]})
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "strict_worst_tiles.json"), "w"), separators=(",", ":"))
print(len(out["tiles"]), "tiles")
