"""Turns the text dumps of oracle/lua/gen_golden.lua (vectors produced by the REFERENCE under LuaJIT) into
tests/golden_lua/<case>.npz, laid out like tests/golden/<case>.npz.

    python oracle/lua/import_golden.py <dir with the .txt dumps>

tests/test_oracle.py::test_oracle_matches_reference_held_vectors compares the C oracle with every file it finds
there, bit for bit; without them that test is skipped and parity stays "unpinned" (no Lua interpreter exists in
this pipeline)."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path):
    rows = {}
    with open(path) as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            vals = np.array([float(v) for v in parts[2:]], dtype=np.float64)
            assert len(vals) == int(parts[1]), (path, parts[0])
            rows[parts[0]] = vals
    out = {}
    for tag in ("white", "yolk"):
        out["init_" + tag] = np.array([rows["init_%s_%s" % (tag, f)] for f in ("x", "y", "mass_t", "inv_mass", "radius")])
    steps = sorted({int(k.split("_")[1][4:]) for k in rows if k.startswith("white_step")})
    for s in steps:
        for tag in ("white", "yolk"):
            out["%s_step%d" % (tag, s)] = np.array([rows["%s_step%d_%s" % (tag, s, f)] for f in ("x", "y", "vx", "vy")])
        out["centroid_step%d" % s] = rows["centroid_step%d" % s].reshape(-1, 2)
    out["snap_steps"] = np.array(steps)
    return out


def main():
    src = sys.argv[1]
    dst = os.path.join(ROOT, "tests", "golden_lua")
    os.makedirs(dst, exist_ok=True)
    for path in sorted(glob.glob(os.path.join(src, "*.txt"))):
        name = os.path.splitext(os.path.basename(path))[0]
        np.savez_compressed(os.path.join(dst, name + ".npz"), **load(path))
        print("wrote", name)


if __name__ == "__main__":
    main()
