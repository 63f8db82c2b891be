"""How often does a collision pass visit exactly the pair stream of the pass before it?  (CPU, oracle trace.)

VERDICT r02 item 2a proposed to skip the level walk + sort for tiles whose pair stream equals the previous pass's.  The
stream of a fresh pass depends only on which spatial-hash cell every particle is in, so this measures how often NO
particle of an island changes its cell between two passes.  One island of the workload per run:

    python tests/stream_coherence.py 1 1 400     # one default blob: an island of BASELINE configs 2 / 4 / 5
    python tests/stream_coherence.py 4 4 400     # four coincident blobs: an island (site) of config 3

Prints, per sampled step, the visited pairs of the six white passes ("=" marks a pass whose stream equals the previous
fresh pass's) and whether all six equal the previous step's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # (kept under tests/: only tests may use the oracle)
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as om  # noqa: E402  (a CPU-only developer script: the oracle is the thing measured here)


def run(n_batches, overlap, steps):
    xs, ys, _ = bench.grid_positions(n_batches, overlap=overlap)
    o = om.Oracle()
    o.set_budget_particles(0, 4096 * 157)
    o.set_budget_particles(1, 4096 * 15)
    for x, y in zip(xs, ys):
        o.add(float(x), float(y), 50, 15)
    o.set_trace(True)
    report = set(range(0, steps, max(1, steps // 12))) | {steps - 1, steps - 2}
    prev, same_pass, same_step, n_pass, n_step = None, 0, 0, 0, 0
    for step in range(steps):
        o.update(1 / 60)
        tr, ps = o.trace(), o.pass_stats()
        streams = {}
        for seq, s in enumerate(ps):
            if s["which"] == 0:
                m = tr["pass_seq"] == seq
                streams[(s["sub_step"], s["pass_"])] = (tr["self_i"][m].copy(), tr["other_i"][m].copy())
        line, last = [], None
        for k in sorted(streams):
            a, b = streams[k]
            same = last is not None and len(a) == len(last[0]) and np.array_equal(a, last[0]) and np.array_equal(b, last[1])
            n_pass += 1
            same_pass += int(same)
            line.append("%s:%d%s" % (k, len(a), "=" if same else ""))
            if not (k[0] > 0 and k[1] == 0):  # (the stale pass is not a reference for the next one)
                last = (a, b)
        whole = prev is not None and all(np.array_equal(streams[k][0], prev[k][0]) and np.array_equal(streams[k][1], prev[k][1]) for k in streams)
        n_step += 1
        same_step += int(whole)
        if step in report:
            print("step %3d  %s | same as the previous step: %s" % (step, " ".join(line), whole))
        prev = streams
    print("passes whose stream equals the previous fresh pass's: %d of %d; steps equal to the previous step: %d of %d" % (same_pass, n_pass, same_step, n_step))


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
