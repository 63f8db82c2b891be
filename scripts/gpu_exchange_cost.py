"""Developer: host-side cost of the per-step neighbour exchange (2 ranks over gloo on one box), split into
post() / step_begin / finish() / step_end.  Launch: python -m torch.distributed.run --nproc-per-node 2 ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch, torch.distributed as dist
import bench
from egg_fluid_simulation_amd import SimulationHandler
from egg_fluid_simulation_amd.sharding import BoundaryExchange

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
h = SimulationHandler(device=0)
xs, ys, side = bench.grid_positions(256, column_offset=rank)
h.add_many(xs, ys, 50, 15)
halo = BoundaryExchange(h, rank, world, slab_lo=100.0 + bench.PITCH * rank * side - bench.PITCH / 2,
                        slab_hi=100.0 + bench.PITCH * (rank + 1) * side - bench.PITCH / 2, group=dist)
T = np.zeros(4)
for it in range(230):
    t0 = time.perf_counter(); h.step_begin(1 / 60, 2, 3)
    t1 = time.perf_counter(); halo.post(claims_fixed=True)
    t2 = time.perf_counter(); halo.finish()
    t3 = time.perf_counter(); h.step_end(True)
    t4 = time.perf_counter()
    if it >= 30:
        T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
if rank == 0:
    print("per step, us: step_begin %.1f  post %.1f  finish %.1f  step_end %.1f  (total %.1f)" % (*(T / 200 * 1e6), T.sum() / 200 * 1e6))
dist.destroy_process_group()
