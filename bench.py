#!/usr/bin/env python3
"""Benchmark of the XPBD particle step on MI355X (driver contract: one JSON line on rank 0).

    python bench.py                         # 1 GPU, BASELINE config 3: 4096 batches, 4 coincident per site
    python bench.py --batches 256 --overlap 1      # BASELINE config 2 (the latency case; also reported by default)
    python bench.py --batches 16384 --overlap 1    # one GPU's share of BASELINE config 4 / the whole of it on one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one SimulationHandler:_step (dt = 1/60, 2 sub-steps x 3 collision passes,
simulation_handler.lua:1722) over every particle of the workload.  Inputs are resident in HBM
before the timed region (the handler owns the particle arrays on the device).

Workload at N = 1: the largest single-GPU configuration of BASELINE.json -- config 3 (4096 batches, every
four consecutive batches on one centre: 704,512 particles in 1024 dense islands).  Config 2 (256 separate
batches, one island per CU: a latency test) is timed in the same run and reported as `latency_config2`.

Multi-GPU: the plane is cut into N x-slabs of `--batches` batches each (weak scaling); every
rank steps its slab with its own handler and the ranks exchange the cell boxes of their
slab-boundary batches once per step (egg_fluid_simulation_amd/sharding.py) -- the only
cross-rank data the path needs while no batch crosses a cut.

value = pair solves (visited pairs, simulation_handler.lua:1657) per second over ALL ranks;
steps_per_sec is the lock-step rate of the whole job.  The reference Lua path cannot be timed
(no Lua interpreter in this pipeline); cpu_baseline times the sequential C restatement
(oracle/, single thread because the reference is single-threaded) on sites of the same workload.

roofline: SURVEY.md 8d's byte model prices a step at 592 B per particle.  Large scenes run the PACKED pipeline
(one launch per phase, csrc/eggsim_packed.hip): `kernel_ms` is then the HIP-event time of ALL launches that step the
white particles in one step (events around the white stream's launch sequence, inside the timed region; the yolk
launches run beside them on their own stream); a short profiling leg after the timed region puts events around
every launch: `kernels` lists them, `dominant_kernel` names the kind with the largest share.
Small scenes run the fused step kernel: one launch per step, timed inside the timed region.  `traffic` and
`valu` are rocprofv3 counter measurements of the same workload committed under profiles/ (they cannot be taken
from inside the process); they are attached only when the workload matches.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PARTICLE_STEP = 592.0  # SURVEY.md 8d / BASELINE.md: FP64 SoA, state round-trips HBM once per phase
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec
PITCH = 160.0                         # px between batch centres (BASELINE config 2: islands never touch)
COUNTER_FILES = [os.path.join(ROOT, "profiles", n) for n in ("r03_counters.json", "r02_counters.json")]  # written by scripts/collect_counters.sh; newest first


def site_pitch(overlap=1):
    """k coincident blobs spread to ~sqrt(k) times the radius: keep the sites' islands apart"""
    return PITCH * (1.0 if overlap <= 1 else 1.25 * math.ceil(math.sqrt(overlap)))


def grid_positions(n_batches, column_offset=0, overlap=1):
    """batch centres on a PITCH grid; overlap > 1 puts that many consecutive batches on the same centre
    (BASELINE config 3: forced spatial overlap)"""
    n_sites = (n_batches + overlap - 1) // overlap
    side = int(math.ceil(math.sqrt(n_sites)))
    k = np.arange(n_batches) // overlap
    pitch = site_pitch(overlap)
    xs = 100.0 + pitch * (k % side + column_offset * side)
    ys = 100.0 + pitch * (k // side)
    return xs.astype(np.float64), ys.astype(np.float64), side


def workload_name(batches, overlap):
    if overlap == 1:
        tag = {256: "BASELINE config 2", 16384: "BASELINE config 4 on one GPU", 65536: "BASELINE config 5 on one GPU"}.get(batches)
        return "%s%d non-overlapping batches per GPU" % (tag + ": " if tag else "", batches)
    tag = "BASELINE config 3: " if (batches, overlap) == (4096, 4) else "BASELINE config 3 layout: "
    return "%s%d batches per GPU, %d coincident per site" % (tag, batches, overlap)


def cpu_baseline(n_batches, overlap, budget_s=12.0):
    """The CPU oracle on sites of the same workload, single thread, bounded to ~budget_s seconds.  Sites are
    independent islands, so the oracle steps a SAMPLE of them (its cost per step grows with the island count);
    pair-solves/s is a rate and needs no extrapolation."""
    from oracle.oracle import Oracle
    sample = min(n_batches, 64 if overlap == 1 else 8 * overlap)
    xs, ys, _ = grid_positions(sample, overlap=overlap)
    o = Oracle()
    for x, y in zip(xs, ys):
        o.add(float(x), float(y), 50, 15)
    warm = 2
    for _ in range(warm):
        o.update(1 / 60)
    v0 = o.total_visited
    t0 = time.perf_counter()
    steps = 0
    while True:
        o.update(1 / 60)
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or steps >= 2000:
            break
    pairs = o.total_visited - v0
    return {"value": pairs / dt, "unit": "pair-solves/s", "cores": 1, "kind": "port",
            "batch_steps_per_sec": sample * steps / dt,
            "sample": "oracle/eggsim_oracle.c (sequential C restatement of the Lua step; the Lua reference itself "
                      "cannot run here): %d batches of the same layout (%d per site), %d timed steps after %d warm-up, "
                      "1 thread of %d host cores; islands are independent, the whole workload costs the CPU "
                      "%.0f x this sample per step" % (sample, overlap, steps, warm, os.cpu_count() or 0, n_batches / sample)}


def pk_symbol(kind, variants):
    """the kernel symbol rocprofv3 shows for a phase of the packed pipeline (egg_stats.pk_variants says which variant runs)"""
    from egg_fluid_simulation_amd import _ffi
    if kind == "egg_pk_levels_kernel":
        names = [n for bit, n in ((_ffi.PK_VARIANT_LEVELS_OOO, "egg_pk_levels_ooo_kernel"), (_ffi.PK_VARIANT_LEVELS_INORDER, "egg_pk_levels_mr16_kernel")) if variants & bit]
    elif kind == "egg_pk_exec_kernel":
        names = [n for bit, n in ((_ffi.PK_VARIANT_EXEC_CHAIN, "egg_pk_exec_chain_kernel"), (_ffi.PK_VARIANT_EXEC, "egg_pk_exec_kernel")) if variants & bit]
    elif kind == "egg_pk_sort_kernel":
        names = [n for bit, n in ((_ffi.PK_VARIANT_SORT_DIRECT, "egg_pk_sort_direct_kernel"), (_ffi.PK_VARIANT_SORT_LDS, "egg_pk_sort_kernel")) if variants & bit]
    elif kind == "egg_pk_pass_kernel":
        names = ["egg_pk_levexec_kernel"]
    else:
        names = [kind]
    return " / ".join(names) if names else kind


def git_commit_of(path):
    """the commit that last changed a committed measurement file (so that a number read from it can be traced)"""
    import subprocess
    try:
        out = subprocess.run(["git", "-C", ROOT, "log", "-n", "1", "--format=%h", "--", path], capture_output=True, text=True, timeout=10)
        return out.stdout.strip() or None
    except Exception:
        return None


def steady_window(make_handler, warmup, steps, n_white):
    """config 3 never settles (its dependency chains deepen for hundreds of steps): the same scene timed at a later window"""
    from egg_fluid_simulation_amd import WHITE, _ffi
    h = make_handler()
    for _ in range(warmup):
        h.step(1 / 60, 2, 3)
    h.set_option(_ffi.OPT_TIMING, 1)
    h.synchronize()
    s0 = h.stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        h.step(1 / 60, 2, 3)
    h.synchronize()
    dt = time.perf_counter() - t0
    s1 = h.stats()
    kernel_ms = s1["kernel_ms_sum"][WHITE] / max(1, s1["timed_steps"])
    achieved = ALGO_BYTES_PER_PARTICLE_STEP * n_white / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    out = {"warmup": warmup, "steps": steps, "ms_per_step": 1e3 * dt / steps, "steps_per_sec": steps / dt,
           "pair_solves_per_sec": (s1["pair_solves"] - s0["pair_solves"]) / dt, "kernel_ms": kernel_ms,
           "achieved_gbs": achieved, "frac": achieved / HBM_PEAK_GBS, "levels_per_pass": s1.get("max_levels"),
           "redo_steps": s1["redo_steps"] - s0["redo_steps"]}
    del h
    return out


def timed_run(h, one_step, steps, warmup, torch, dist, timing_option):
    from egg_fluid_simulation_amd import _ffi
    for _ in range(warmup):
        one_step()
    h.set_option(_ffi.OPT_TIMING, timing_option)  # HIP events around the step launches, on the streams they run on
    h.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    s0 = h.stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    h.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    return time.perf_counter() - t0, s0, h.stats()


def latency_config2(torch, device, steps=200, warmup=30):
    """BASELINE config 2 (256 separate batches, one island per CU): the step latency of the fused kernel"""
    from egg_fluid_simulation_amd import WHITE, SimulationHandler
    h = SimulationHandler(device=device)
    xs, ys, _ = grid_positions(256)
    h.add_many(xs, ys, 50, 15)
    elapsed, s0, s1 = timed_run(h, lambda: h.step(1 / 60, 2, 3), steps, warmup, torch, None, 1)
    n_w, n_y = h.get_n_particles()
    kernel_ms = s1["kernel_ms_sum"][WHITE] / max(1, s1["timed_steps"])
    out = {"workload": workload_name(256, 1), "particles": n_w + n_y, "steps": steps,
           "ms_per_step": 1e3 * elapsed / steps, "steps_per_sec": steps / elapsed,
           "pair_solves_per_sec": (s1["pair_solves"] - s0["pair_solves"]) / elapsed,
           "kernel": "egg_step_kernel_multi_wide (one launch per step, white + yolk tiles)" if s1["fused_launch"] else "egg_step_kernel*",
           "kernel_ms": kernel_ms,
           "roofline_frac": ALGO_BYTES_PER_PARTICLE_STEP * (n_w + n_y) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else None}
    del h
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batches", type=int, default=4096, help="batches per GPU (4096 with --overlap 4 = BASELINE config 3)")
    ap.add_argument("--overlap", type=int, default=4, help="coincident batches per site (4 = BASELINE config 3, 1 = separate batches)")
    ap.add_argument("--tile-target", type=int, default=0, help="pack independent islands into tiles of this size")
    ap.add_argument("--no-fuse", action="store_true", help="one launch per particle type even on a full chip (A/B testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the config-2 latency block")
    ap.add_argument("--packed", type=int, default=-1, help="packed pipeline: -1 automatic, 0 never, 1 always (A/B testing)")
    ap.add_argument("--group-particles", type=int, default=0, help="packed pipeline: particles per executor wave (0 = default)")
    ap.add_argument("--profile-steps", type=int, default=-1, help="steps of the per-kernel timing leg (a replay of the timed region on a fresh scene; -1: as many as --steps, 0: none)")
    ap.add_argument("--no-windows", action="store_true", help="skip the later timing windows of config 3 (config3_protocol, config3_late)")
    ap.add_argument("--late-warmup", type=int, default=600, help="warm-up steps of the config3_late window")
    args = ap.parse_args()
    if args.profile_steps < 0:
        args.profile_steps = args.steps

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" %
                         (args.gpus, args.gpus, world))

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; EGG_BENCH_BACKEND=gloo lets the multi-rank path be rehearsed with several
        # ranks on ONE GPU (RCCL refuses duplicate devices)
        backend = os.environ.get("EGG_BENCH_BACKEND", "nccl")
        n_dev = max(1, torch.cuda.device_count())
        local_rank = local_rank % n_dev
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)

    from egg_fluid_simulation_amd import WHITE, YOLK, SimulationHandler, _ffi
    from egg_fluid_simulation_amd.sharding import BoundaryExchange

    xs, ys, side = grid_positions(args.batches, column_offset=rank, overlap=args.overlap)

    def make_handler():
        hh = SimulationHandler(device=local_rank)  # raises without a GPU: no CPU path
        if args.tile_target:
            hh.set_option(_ffi.OPT_TILE_TARGET_PARTICLES, args.tile_target)
        if args.no_fuse:
            hh.set_option(_ffi.OPT_FUSE_TYPES, 0)
        if args.packed >= 0:
            hh.set_option(_ffi.OPT_PACKED, args.packed)
        if args.group_particles:
            hh.set_option(_ffi.OPT_GROUP_PARTICLES, args.group_particles)
        hh.add_many(xs, ys, 50, 15)
        return hh

    h = make_handler()
    n_white, n_yolk = h.get_n_particles()
    pitch = site_pitch(args.overlap)
    halo = BoundaryExchange(h, rank, world, slab_lo=100.0 + pitch * rank * side - pitch / 2,
                            slab_hi=100.0 + pitch * (rank + 1) * side - pitch / 2, group=dist) if world > 1 else None

    def one_step():
        if halo is None:
            h.step(1 / 60, 2, 3)
            return
        # the kernels are launched first; the claims of this step (fixed by step_begin's tiling) are collected,
        # sent to the neighbours and checked while they run; a conflict (a batch about to cross a cut)
        # raises -- the tiled benchmark never produces one
        h.step_begin(1 / 60, 2, 3)
        halo.post(claims_fixed=True)
        halo.finish()
        h.step_end(True)

    elapsed, s0, s1 = timed_run(h, one_step, args.steps, args.warmup, torch, dist, 1)

    pairs = float(s1["pair_solves"] - s0["pair_solves"])
    follows = float(s1["follow_solves"] - s0["follow_solves"])
    kernel_ms_white = s1["kernel_ms_sum"][WHITE] / max(1, s1["timed_steps"])
    kernel_ms_yolk = s1["kernel_ms_sum"][YOLK] / max(1, s1["timed_steps"])
    if dist is not None:
        red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([pairs, follows, float(n_white + n_yolk)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        pairs, follows, total_particles = (float(v) for v in agg.tolist())
    else:
        total_particles = float(n_white + n_yolk)

    # per-kernel leg (outside the timed region): HIP events around every launch of the packed pipeline.  The scenes are
    # not stationary (config 3's dependency chains deepen step by step), so the leg REPLAYS the timed region's steps
    # on a fresh scene -- the same steps a `rocprofv3 --kernel-trace --stats` of this command averages over
    per_kernel = None
    if rank == 0 and s1["packed"][WHITE] and args.profile_steps > 0:
        hp = h
        if world == 1:
            hp = make_handler()
            for _ in range(args.warmup):
                hp.step(1 / 60, 2, 3)
        # (several ranks: rank 0 carries on with its own slab, without the exchange)
        hp.set_option(_ffi.OPT_TIMING, 2)
        for _ in range(args.profile_steps):
            hp.step(1 / 60, 2, 3)
        hp.synchronize()
        sp = hp.stats()
        if hp is not h:
            del hp
        per_kernel = []
        for w, tag in ((WHITE, "white"), (YOLK, "yolk")):
            for k, name in enumerate(_ffi.PK_KINDS):
                n = sp["pk_kernel_launches"][w][k]
                if n:
                    groups = n / max(1, sp["packed"][w])  # launches of one kind: one per class of the type
                    per_kernel.append({"kernel": pk_symbol(name, sp["pk_variants"][w]), "type": tag, "avg_ms": sp["pk_kernel_ms"][w][k] / groups,
                                       "launches_per_step": groups / args.profile_steps,
                                       "ms_per_step": sp["pk_kernel_ms"][w][k] / args.profile_steps})

    if rank == 0:
        steps_per_sec = args.steps / elapsed
        fused = bool(s1.get("fused_launch"))
        packed = bool(s1["packed"][WHITE])
        note = "592 B/particle/step byte model of SURVEY.md 8d; this path is bound by the reference's sequential pair order " \
               "(config.levels_per_pass dependent levels per collision pass at ~0.3 us each, plus the walk that finds " \
               "them), not by HBM bytes: see `traffic`, `valu` and DESIGN.md section 4"
        if packed and per_kernel:
            white = [k for k in per_kernel if k["type"] == "white"]
            # all launches stepping the white particles: HIP events around the white stream's launches of a step, inside
            # the timed region (the per-launch events of the profiling leg add up to a little more: they fence the
            # launches off from each other)
            kernel_ms = kernel_ms_white
            sum_of_launches = sum(k["ms_per_step"] for k in white)
            dom = max(white, key=lambda k: k["ms_per_step"])
            algo_bytes = ALGO_BYTES_PER_PARTICLE_STEP * n_white
            kernel_name = "packed pipeline, white particles: %d launches per step (%s)" % (
                round(sum(k["launches_per_step"] for k in white)), ", ".join(sorted({k["kernel"] for k in white})))
            dominant = {"kernel": dom["kernel"], "avg_launch_ms": dom["avg_ms"], "launches_per_step": dom["launches_per_step"],
                        "share_of_launch_time": dom["ms_per_step"] / sum_of_launches,
                        "sum_of_white_launches_ms_per_step": sum_of_launches}
        else:
            kernel_ms = kernel_ms_white
            algo_bytes = ALGO_BYTES_PER_PARTICLE_STEP * ((n_white + n_yolk) if fused else n_white)
            kernel_name = "egg_step_kernel_multi* (one launch, white + yolk tiles)" if fused else "egg_step_kernel* (white launch)"
            dominant = None
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        # rocprofv3 counters cannot be taken from inside the process: they are read from the newest committed counter file
        # that holds this workload, and the line says which file (and which commit of it) they come from
        traffic, valu, traffic_source = None, None, None
        for cf in COUNTER_FILES if world == 1 else []:
            try:
                whole = json.load(open(cf))
                c = whole.get("%d_%d" % (args.batches, args.overlap))
            except Exception:
                whole, c = {}, None
            if c:
                traffic, valu = c.get("hbm_bytes_per_step"), c.get("valu")
                traffic_source = {"file": os.path.relpath(cf, ROOT), "commit": git_commit_of(cf) or (whole.get("_source") or {}).get("code_commit"),
                                  "how": "scripts/collect_counters.sh (rocprofv3 --pmc, one pass per counter set, summed over every launch of a step; 2 x FETCH_SIZE + WRITE_SIZE); NOT measured in this run"}
                break
        out = {
            "metric": "pair_solves_per_sec", "value": pairs / elapsed, "unit": "pair-solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "steps_per_sec": steps_per_sec,
            "constraint_solves_per_sec": (pairs + follows) / elapsed,
            "particles": int(total_particles),
            "config": {"workload": workload_name(args.batches, args.overlap) +
                                   " (white r=50, yolk r=15) on a %.0f px grid, default config, dt=1/60, 2 sub-steps x 3 "
                                   "collision passes" % site_pitch(args.overlap),
                       "batches_per_gpu": args.batches, "coincident_per_site": args.overlap,
                       "particles_per_gpu": int(n_white + n_yolk),
                       "parallelism": "slab%d" % world, "tiles": s1["n_tiles"], "retiles": s1["retiles"] - s0["retiles"],
                       "redo_steps": s1["redo_steps"] - s0["redo_steps"], "path": "packed" if packed else "fused",
                       # packed path: the longest chain of dependent pairs in one collision pass at the end of the run
                       # (white, yolk) -- the levels its executor runs one after the other, i.e. the latency floor
                       "levels_per_pass": s1.get("max_levels")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "valu": valu,
                         "traffic_source": traffic_source, "valu_source": traffic_source,
                         "kernel": kernel_name, "kernel_ms": kernel_ms, "dominant_kernel": dominant,
                         "kernel_ms_timed_region": kernel_ms_white,
                         "kernel_ms_yolk_stream": None if fused else kernel_ms_yolk,
                         "algorithmic_bytes_per_launch": algo_bytes, "kernels": per_kernel, "note": note},
        }
        if world == 1:
            del h
            if (args.batches, args.overlap) == (4096, 4) and not args.no_windows:
                # the headline config is a transient: besides the driver's window (value), the BASELINE.md protocol window
                # and a late one, each with its own ms_per_step / frac / levels_per_pass
                if (args.warmup, args.steps) != (10, 100):
                    out["config3_protocol"] = steady_window(make_handler, 10, 100, n_white)
                else:
                    out["config3_protocol"] = "the timed region of this line (10 warm-up + 100 steps)"
                out["config3_late"] = steady_window(make_handler, args.late_warmup, 50, n_white)
            if not args.no_latency:
                out["latency_config2"] = latency_config2(torch, local_rank)
            out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(args.batches, args.overlap)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
