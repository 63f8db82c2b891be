"""Developer: long randomised session against the oracle (bit-exact check every 250 steps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from egg_fluid_simulation_amd import SimulationHandler
from oracle import oracle as om
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
h, o = SimulationHandler(), om.Oracle()
live = {}
def add():
    x, y = rng.uniform(0, 700, 2)
    wr = float(rng.choice([20.0, 35.0, 50.0])); yr = float(rng.choice([9.0, 15.0]))
    a = h.add(float(x), float(y), wr, yr); b = o.add(float(x), float(y), wr, yr)
    assert a == b
    live[a] = [float(x), float(y), *rng.uniform(-4, 4, 2)]
for _ in range(6): add()
t0 = time.time()
for step in range(n_steps):
    r = rng.random()
    if r < 0.002 and len(live) < 9: add()
    elif r < 0.004 and len(live) > 3:
        v = int(rng.choice(sorted(live))); h.remove(v); o.remove(v); del live[v]
    for i, t in live.items():
        if rng.random() < 0.002: t[0], t[1] = (float(q) for q in rng.uniform(0, 700, 2))
        t[0] += t[2]; t[1] += t[3]
        if not (0 < t[0] < 700): t[2] = -t[2]
        if not (0 < t[1] < 700): t[3] = -t[3]
        h.set_target_position(i, t[0], t[1]); o.set_target_position(i, t[0], t[1])
    h.step(1 / 60, 2, 3); o.step(1 / 60, 2, 3)
    if step % 250 == 249:
        for w in (0, 1):
            for f in ("x", "y", "vx", "vy"):
                assert np.array_equal(h.download(w, f), o.field(w, f)), (step, w, f)
        assert h.stats()["pair_solves"] == o.total_visited
        if step % 2500 == 2499:
            s = h.stats()
            print("step %d ok (%.0f s): batches %d, retiles %d, redo %d, tiles %s" % (step + 1, time.time() - t0, len(live), s["retiles"], s["redo_steps"], s["n_tiles"]), flush=True)
print("soak passed: %d steps bit-exact" % n_steps)
