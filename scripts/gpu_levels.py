"""Developer: depth of the pair-dependency DAG (levels per collision pass) and pairs per pass for a bench scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from egg_fluid_simulation_amd import SimulationHandler, _ffi
nb, overlap = int(sys.argv[1]), int(sys.argv[2])
xs, ys, side = bench.grid_positions(nb, overlap=overlap)
h = SimulationHandler()
h.set_option(_ffi.OPT_PACKED, 1)
h.set_option(_ffi.OPT_TIMING, 2)
h.add_many(xs, ys, 50, 15)
for k in range(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    h.step(1 / 60, 2, 3)
    s = h.stats()
    if k % 5 == 4:
        print("step %d: levels white %d yolk %d, most pairs in a pass %s, tiles %s" % (k + 1, s["max_levels"][0], s["max_levels"][1], s["max_pass_visits"], s["n_tiles"]))
s = h.stats()
for name, ms, n in zip(_ffi.PK_KINDS, s["pk_kernel_ms"][0], s["pk_kernel_launches"][0]):
    if n: print("  white %-28s %8.1f us per launch group, %d" % (name, 1e3 * ms / max(1, s["steps"]) , n))
