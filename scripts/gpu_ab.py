"""Developer A/B timing of the packed pipeline in ONE process (rule: variants interleaved, same device).

    python scripts/gpu_ab.py --batches 4096 --overlap 4 --warmup 10 --steps 60 --variants walk=1 walk=2
    python scripts/gpu_ab.py --batches 16384 --overlap 1 --variants walk=0 walk=2

A variant is a comma-separated list of key=value: walk (EGG_OPT_LEVEL_WALK), packed (EGG_OPT_PACKED), gp
(EGG_OPT_GROUP_PARTICLES).  Every variant steps its own handler of the same scene; rounds are interleaved; per
variant: wall ms per step, HIP-event ms of the white stream, and the per-kind launch times (EGG_OPT_TIMING = 2, which
fences the launches of a kind off from each other: read the SHARES, the wall column has its own rounds without it).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from egg_fluid_simulation_amd import WHITE, YOLK, SimulationHandler, _ffi  # noqa: E402

OPTS = {"walk": _ffi.OPT_LEVEL_WALK, "packed": _ffi.OPT_PACKED, "gp": _ffi.OPT_GROUP_PARTICLES}
for name in ("OPT_EXEC", "OPT_PASS"):
    if hasattr(_ffi, name):
        OPTS[name[4:].lower()] = getattr(_ffi, name)


def make(args, variant):
    kvs = dict(kv.split("=") for kv in variant.split(",") if kv)
    os.environ["EGGSIM_TUNE"] = kvs.pop("tune", "0")  # (read by egg_create: developer experiments inside the kernels)
    h = SimulationHandler()
    for k, v in kvs.items():
        h.set_option(OPTS[k], float(v))
    xs, ys, _ = bench.grid_positions(args.batches, overlap=args.overlap)
    h.add_many(xs, ys, 50, 15, None, args.yolk_n or None)
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=4096)
    ap.add_argument("--overlap", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--yolk-n", type=int, default=0, help="yolk particles per batch (0: the default 15; 2: as good as none -- the white stream then has the chip to itself)")
    ap.add_argument("--variants", nargs="+", default=["walk=1", "walk=2"])
    args = ap.parse_args()
    per = max(1, args.steps // args.rounds)
    hs = [make(args, v) for v in args.variants]
    for h in hs:
        for _ in range(args.warmup):
            h.step(1 / 60, 2, 3)
        h.synchronize()
    wall = [[] for _ in hs]
    kms = [[] for _ in hs]
    kmy = [[] for _ in hs]
    for r in range(args.rounds):  # every handler advances by the same steps per round: same scene state per round
        for i, h in enumerate(hs):
            h.set_option(_ffi.OPT_TIMING, 1)
            h.synchronize()
            t0 = time.perf_counter()
            for _ in range(per):
                h.step(1 / 60, 2, 3)
            h.synchronize()
            wall[i].append(1e3 * (time.perf_counter() - t0) / per)
            st = h.stats()
            kms[i].append(st["kernel_ms_sum"][WHITE] / max(1, st["timed_steps"]))
            kmy[i].append(st["kernel_ms_sum"][YOLK] / max(1, st["timed_steps"]))
    # per-kind leg: a few more steps with events around every launch
    kinds = []
    for i, h in enumerate(hs):
        h.set_option(_ffi.OPT_TIMING, 2)
        n = max(4, per // 2)
        for _ in range(n):
            h.step(1 / 60, 2, 3)
        h.synchronize()
        st = h.stats()
        kinds.append({_ffi.PK_KINDS[k]: st["pk_kernel_ms"][WHITE][k] / n for k in range(len(_ffi.PK_KINDS)) if st["pk_kernel_launches"][WHITE][k]})
        kinds[-1].update({"yolk:" + _ffi.PK_KINDS[k]: st["pk_kernel_ms"][YOLK][k] / n for k in range(len(_ffi.PK_KINDS)) if st["pk_kernel_launches"][YOLK][k]})
    for i, v in enumerate(args.variants):
        st = hs[i].stats()
        print("%-24s wall ms/step: median %.3f min %.3f | white stream ms/step %.3f yolk %.3f | levels/pass %s variants %s redo %d host_ms %s" %
              (v, float(np.median(wall[i])), min(wall[i]), float(np.median(kms[i])), float(np.median(kmy[i])), st["max_levels"], st["pk_variants"], st["redo_steps"],
               ["%.2f" % (x / max(1, st["steps"])) for x in st["host_ms"]]))
        print("    per kind ms/step: " + "  ".join("%s %.3f" % (k.replace("egg_pk_", "").replace("_kernel", ""), t) for k, t in kinds[i].items()))
    ref = [hs[0].download(WHITE, f) for f in ("x", "y")]
    for i in range(1, len(hs)):
        same = all(np.array_equal(hs[i].download(WHITE, f), r) for f, r in zip(("x", "y"), ref))
        print("variant %s vs %s: positions %s" % (args.variants[i], args.variants[0], "bit-equal" if same else "DIFFER"))


if __name__ == "__main__":
    main()
