"""Round-2 parity cases of the device path (through the C ABI) against the CPU oracle:
  * (BASELINE configs 3, 4 and 5 at full size: tests/test_gpu_fullsize.py, every island against the oracle)
  * the mass guard of the pair loop (simulation_handler.lua:1601) and what `n_collided` counts there;
  * particles closer than math.eps (math.lua:53-56: normalize returns (0, 0));
  * a yolk-only mass change between two fused launches (L:1420-1430, L:1731-1744);
  * the boundary's refusals: explicit particle counts <= 1 create nothing (L:79-85), state-changing calls
    between egg_step_begin and egg_step_end are rejected.
Everything is compared bit for bit; the north star's 1e-4 relative tolerance is therefore met with zero difference."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WHITE, YOLK = 0, 1
N_W, N_Y = 157, 15  # particles per batch at white r = 50, yolk r = 15, particle r = 4 (L:52-58)


@pytest.fixture(scope="module")
def egg():
    import egg_fluid_simulation_amd as e
    return e


def _same(h, o, tag):
    for w in (WHITE, YOLK):
        for f in ("x", "y", "vx", "vy"):
            a, b = h.download(w, f), o.field(w, f)
            assert a.shape == b.shape and np.array_equal(a, b), (tag, w, f)
    assert h.stats()["pair_solves"] == o.total_visited, tag


def _grid_xy(side, pitch=160.0, x0=100.0):
    k = np.arange(side * side)
    return x0 + pitch * (k % side), x0 + pitch * (k // side)


MASS_TWEAKS = [
    dict(max_mass=1e9),                 # inverse masses 1e-9 .. 1.7e-8: the guard w_i + w_j < 1e-8 splits the pairs
    dict(min_mass=3e8, max_mass=1e9),   # every pair fails the guard; no particle follows (w <= eps, L:1458)
    dict(min_mass=1.0, max_mass=2.0e8 / 0.9),  # only the few heaviest pairs (mass_t near 1) are guarded
]


@pytest.mark.parametrize("tweak", MASS_TWEAKS, ids=["mixed", "all_guarded", "few_guarded"])
@pytest.mark.parametrize("centers", [[(300.0, 300.0)], [(300.0, 300.0), (340.0, 320.0), (300.0, 350.0)],
                                     [(100.0 + 90.0 * k, 100.0 + 35.0 * (k % 3)) for k in range(12)]],
                         ids=["one_batch_exact_budget", "three_overlapping", "twelve_tiles"])
def test_mass_guard_pairs_match_oracle(egg, oracle_mod, tweak, centers):
    """L:1601: a pair whose inverse masses add up to less than eps is marked in `collided` but neither
    projected nor counted in n_collided -- so it does not use up budget either (one batch: the yolk budget of
    12 pairs per pass binds, L:1657)"""
    from egg_fluid_simulation_amd.default_config import default_configs
    w, y = default_configs()
    w.update(tweak)
    y.update(tweak)
    h = egg.SimulationHandler(w, y)
    o = oracle_mod.Oracle()
    o.set_config(WHITE, dict(oracle_mod.DEFAULT_WHITE, **tweak))
    o.set_config(YOLK, dict(oracle_mod.DEFAULT_YOLK, **tweak))
    for cx, cy in centers:
        assert h.add(cx, cy, 50, 15) == o.add(cx, cy, 50, 15)
    guarded = 0
    for step in range(8):
        for i, (cx, cy) in enumerate(centers):
            h.set_target_position(i + 1, cx + 5.0 * step, cy + 3.0 * step)
            o.set_target_position(i + 1, cx + 5.0 * step, cy + 3.0 * step)
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
        _same(h, o, (tweak, step))
    inv = o.field(WHITE, "inv_mass")
    guarded = int((inv[:, None] + inv[None, :] < 1e-8).sum())
    assert guarded > 0  # the case does reach the guard


def test_particles_closer_than_eps(egg, oracle_mod):
    """sub-eps distances through the device path: coincident twins of two batches on one centre (different
    batches: collision only), and a batch so small that all its particles lie within 1e-9 of each other
    (same batch: the dead cohesion block fires on exact coincidence, the collision's normalize() returns
    (0, 0) below eps, math.lua:53-56)"""
    h, o = egg.SimulationHandler(), oracle_mod.Oracle()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "only 2 white / 2 yolk particles will be created"
        for x, y, wr, yr, wn, yn in ((0.0, 0.0, 40.0, 40.0, 2, 2), (0.0, 0.0, 40.0, 40.0, 2, 2),
                                     (200.0, 50.0, 1e-9, 1e-9, 12, 6), (-300.0, -35.0, 50.0, 15.0, 157, 15)):
            assert h.add(x, y, wr, yr, None, None, wn, yn) == o.add(x, y, wr, yr, wn, yn)
    for step in range(6):
        for i in (1, 2, 3, 4):
            tx, ty = 3.0 * step * max(0, i - 2), 2.0 * step  # batches 1 and 2 (the coincident twins) share their target
            h.set_target_position(i, tx, ty)
            o.set_target_position(i, tx, ty)
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
        _same(h, o, step)
    x, y = h.download(WHITE, "x"), h.download(WHITE, "y")
    assert x[0] == x[2] and y[0] == y[2] and x[1] == x[3] and y[1] == y[3]  # coincident twins never separate


@pytest.mark.skipif(os.environ.get("EGGSIM_PACKED") == "1", reason="asserts the one-launch mode; EGGSIM_PACKED=1 forces the packed pipeline")
def test_yolk_only_mass_change_between_fused_launches(egg, oracle_mod):
    """A yolk min/max mass change keeps the cell size, so no re-tiling (and none of its stream syncs) happens
    between the re-derivation kernel (L:1420-1430) and the next step -- which runs the yolk tiles inside the
    launch on the white stream.  The new inverse masses must be what that launch reads."""
    from egg_fluid_simulation_amd.default_config import default_configs
    n = 300
    xs = 100.0 + 160.0 * (np.arange(n) % 20)
    ys = 100.0 + 160.0 * (np.arange(n) // 20)
    h, o = egg.SimulationHandler(), oracle_mod.Oracle()
    ids = h.add_many(xs, ys, 50, 15)
    for a, b in zip(xs, ys):
        o.add(float(a), float(b), 50, 15)
    for step in range(6):
        if step in (2, 4):
            tweak = dict(min_mass=0.5 + step, max_mass=3.0 * step)
            w, y = default_configs()
            y.update(tweak)
            h.set_yolk_config(y)
            o.set_config(YOLK, dict(oracle_mod.DEFAULT_YOLK, **tweak))
        h.set_target_positions(ids, xs + 2.0 * step, ys + 1.0 * step)
        for i, a, b in zip(ids, xs, ys):
            o.set_target_position(int(i), float(a + 2.0 * step), float(b + 1.0 * step))
        retiles = h.stats()["retiles"]
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
        assert h.stats()["fused_launch"] == 1
        _same(h, o, step)
        assert np.array_equal(h.download(YOLK, "inv_mass"), o.field(YOLK, "inv_mass"))
    del retiles


def test_explicit_small_counts_create_nothing(egg):
    """L:79-85: a count <= 1 throws before anything is created; only nil means 'use the default'"""
    h = egg.SimulationHandler()
    a = h.add(0, 0, 50, 15)
    for bad in (0, 1, -3):
        with pytest.raises(egg.EggError, match="white particle count cannot be 1 or negative"):
            h.add(10, 10, 50, 15, None, None, bad, 15)
        with pytest.raises(egg.EggError, match="yolk particle count cannot be 1 or negative"):
            h.add(10, 10, 50, 15, None, None, 157, bad)
    assert h.list_ids() == [a] and h.get_n_particles() == (N_W, N_Y)
    assert h.add(10, 10, 50, 15) == a + 1  # no id was consumed by the refused calls
    # the C entry point itself refuses an explicit 0 (the wrapper's check is not the only one)
    import ctypes as C
    from egg_fluid_simulation_amd import _ffi
    out = C.c_int64()
    rc = h._lib.egg_add(h._h, 0.0, 0.0, 50.0, 15.0, 0, _ffi.DEFAULT_COUNT, C.byref(out))
    assert rc == _ffi.EGG_ERR_INVALID_ARGUMENT and h.list_ids() == [a, a + 1]


def test_state_changes_are_refused_while_a_step_is_in_flight(egg, oracle_mod):
    h, o = egg.SimulationHandler(), oracle_mod.Oracle()
    for s in (h, o):
        s.add(100.0, 100.0, 50, 15)
        s.add(400.0, 100.0, 50, 15)
    h.step_begin(1 / 60, 2, 3)
    for call in (lambda: h.add(700.0, 100.0, 50, 15), lambda: h.remove(1), lambda: h.step(1 / 60, 2, 3),
                 lambda: h.update(1 / 60), lambda: h.set_target_position(1, 5.0, 5.0),
                 lambda: h.set_white_config(h.get_white_config()), lambda: h.step_begin(1 / 60, 2, 3)):
        with pytest.raises(egg.EggError, match="in flight"):
            call()
    h.step_end(True)
    o.step(1 / 60, 2, 3)
    _same(h, o, "after the refused calls")
    assert h.list_ids() == [1, 2]


# ------------------------------------------------------------------------------------------------------------------
# The packed pipeline (csrc/eggsim_packed.hip: one launch per phase, level-sorted pair execution) is chosen
# automatically for large scenes only.  Here it is FORCED (EGG_OPT_PACKED = 1) on scenes the sequential oracle can
# follow, so that every kernel of it is held to the oracle bit for bit: goldens, every (sub-steps, passes) shape it
# supports, overlapping / coincident batches, moving targets with re-tiling every step, live config changes, adds and
# removes, and the fused kernel stepping the same scene (both paths must give the same bits).

WALKS = [1, 2]  # EGG_OPT_LEVEL_WALK: the in-order walk (egg_pk_levels_mr16_kernel) and the out-of-order one (egg_pk_levels_ooo_kernel)


def _packed(egg, walk=0, **kw):
    from egg_fluid_simulation_amd import _ffi
    h = egg.SimulationHandler(**kw)
    h.set_option(_ffi.OPT_PACKED, 1)
    h.set_option(_ffi.OPT_LEVEL_WALK, walk)
    return h


def _walk_used(h, walk):
    from egg_fluid_simulation_amd import _ffi
    v = h.stats()["pk_variants"][WHITE]
    want = {1: _ffi.PK_VARIANT_LEVELS_INORDER, 2: _ffi.PK_VARIANT_LEVELS_OOO}[walk]
    return (v & (_ffi.PK_VARIANT_LEVELS_INORDER | _ffi.PK_VARIANT_LEVELS_OOO)) == want


@pytest.mark.parametrize("walk", WALKS)
@pytest.mark.parametrize("name", ["cfg1_moving", "four_batches", "substeps_3_2", "substeps_2_1"])
def test_packed_pipeline_matches_golden(egg, name, walk):
    from conftest import load_golden, replay_golden
    g = load_golden(name)
    h = _packed(egg, walk)

    def state(hh, w):
        return np.array([hh.download(w, f) for f in ("x", "y", "vx", "vy")])

    def check(step, tag, arr):
        assert np.array_equal(arr, g["%s_step%d" % (tag, step)]), (name, step, tag)

    replay_golden(g, h, state, check)
    assert h.stats()["pair_solves"] == int(g["visits"].sum())
    # one batch: the yolk budget binds -> that type runs the exact-budget fused tile, the white type the packed pipeline
    assert h.stats()["packed"][WHITE] >= 1 and _walk_used(h, walk)


@pytest.mark.parametrize("walk", WALKS)
@pytest.mark.parametrize("S,C", [(1, 1), (1, 3), (2, 1), (2, 2), (2, 3), (3, 2), (4, 3)])
def test_packed_pipeline_substep_and_pass_shapes(egg, oracle_mod, S, C, walk):
    """fresh passes, the stale first pass of every later sub-step (L:1905-1912), one pass per sub-step with two
    sub-steps (two live hash generations); three or more generations fall back to the fused kernel"""
    n = 14
    xs = 100.0 + 95.0 * (np.arange(n) % 5)
    ys = 100.0 + 95.0 * (np.arange(n) // 5)
    h, o = _packed(egg, walk), oracle_mod.Oracle()
    ids = h.add_many(xs, ys, 50, 15)
    for a, b in zip(xs, ys):
        o.add(float(a), float(b), 50, 15)
    for k in range(8):
        dx, dy = 30.0 * np.cos(0.4 * k), 30.0 * np.sin(0.4 * k)
        h.set_target_positions(ids, xs + dx, ys + dy)
        for i, a, b in zip(ids, xs, ys):
            o.set_target_position(int(i), float(a + dx), float(b + dy))
        h.step(1 / 60, S, C)
        o.step(1 / 60, S, C)
        _same(h, o, (S, C, k))
    assert h.stats()["packed"][WHITE] >= 1, h.stats()  # (the blobs merge into one 2198-particle island)
    assert _walk_used(h, walk)


@pytest.mark.parametrize("walk", WALKS)
def test_packed_pipeline_dense_and_sparse_islands_with_hand_made_chaos(egg, oracle_mod, walk):
    """coincident batches (dense islands, lists beyond LDS), separate blobs, a blob chasing a teleporting target through
    the others (islands merge and split, claims fail and steps are re-run), a live config change, a remove and an add"""
    from egg_fluid_simulation_amd.default_config import default_configs
    rng = np.random.default_rng(11)
    h, o = _packed(egg, walk), oracle_mod.Oracle()
    centers = [(300.0, 300.0)] * 4 + [(900.0 + 170.0 * (k % 4), 200.0 + 170.0 * (k // 4)) for k in range(12)]
    ids = []
    for x, y in centers:
        ids.append(h.add(x, y, 50, 15))
        assert o.add(x, y, 50, 15) == ids[-1]
    tx, ty = 300.0, 300.0
    for step in range(30):
        if step % 7 == 3:
            tx, ty = (float(v) for v in rng.uniform(200, 1400, 2))
        h.set_target_position(ids[0], tx, ty)
        o.set_target_position(ids[0], tx, ty)
        if step == 10:
            w, y = default_configs()
            w.update(dict(damping=0.3, max_mass=3.0, collision_strength=0.9))
            h.set_white_config(w)
            o.set_config(WHITE, dict(oracle_mod.DEFAULT_WHITE, damping=0.3, max_mass=3.0, collision_strength=0.9))
        if step == 15:
            h.remove(ids[6])
            o.remove(ids[6])
        if step == 18:
            assert h.add(1000.0, 900.0, 35, 9) == o.add(1000.0, 900.0, 35, 9)
        S, C = [(2, 3), (2, 3), (1, 3), (2, 2)][step % 4]
        h.step(1 / 60, S, C)
        o.step(1 / 60, S, C)
        if step % 5 == 4:
            _same(h, o, step)
    _same(h, o, "end")
    assert h.stats()["packed"][WHITE] >= 1 and _walk_used(h, walk)


def test_fused_pass_stream_outgrows_its_lds_level_array(egg, oracle_mod):
    """The out-of-order walk keeps a tile's levels in LDS, sized from the last step's longest pair stream + 25 %.  Two more
    batches dropped onto a settled island of four more than double its stream: the launch reports that (fail_levlds),
    the step is re-run with a larger array, and the result is the sequential one."""
    h, o = _packed(egg, 2), oracle_mod.Oracle()
    for _ in range(4):
        assert h.add(400.0, 400.0, 50, 15) == o.add(400.0, 400.0, 50, 15)
    for step in range(12):
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
    _same(h, o, "settled")
    redo_before = h.stats()["redo_steps"]
    for _ in range(2):
        assert h.add(400.0, 400.0, 50, 15) == o.add(400.0, 400.0, 50, 15)
    for step in range(3):
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
        _same(h, o, step)
    st = h.stats()
    assert st["redo_steps"] > redo_before and st["max_tile_particles"][WHITE] == 6 * N_W, st
    assert _walk_used(h, 2)


def test_fused_pass_with_four_dense_islands_to_an_executor(egg, oracle_mod):
    """Four dense islands to an executor (EGG_OPT_GROUP_PARTICLES = 2560; not the automatic choice): the fused pass then
    runs eight waves per group (two walk a tile, one executes, one helps).  Six sites of four coincident batches -- groups of four and of two islands -- with the first
    steps' coincident particles (distance 0: the reference path inside the executor) and a moving target."""
    from egg_fluid_simulation_amd import _ffi
    h, o = _packed(egg, 2), oracle_mod.Oracle()
    h.set_option(_ffi.OPT_GROUP_PARTICLES, 2560)
    centers = [(300.0 + 260.0 * (k % 3), 300.0 + 260.0 * (k // 3)) for k in range(6) for _ in range(4)]
    ids = []
    for x, y in centers:
        ids.append(h.add(x, y, 50, 15))
        assert o.add(x, y, 50, 15) == ids[-1]
    for step in range(12):
        h.set_target_position(ids[0], 300.0 + 2.0 * step, 300.0)
        o.set_target_position(ids[0], 300.0 + 2.0 * step, 300.0)
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
        if step % 3 == 2:
            _same(h, o, step)
    _same(h, o, "end")
    st = h.stats()
    assert st["pk_variants"][WHITE] & _ffi.PK_VARIANT_PASS_FUSED and st["max_tile_particles"][WHITE] == 4 * N_W


@pytest.mark.parametrize("walk", WALKS)
def test_packed_pipeline_inverse_masses_between_half_eps_and_eps(egg, oracle_mod, walk):
    """every inverse mass in [eps / 2, eps): the tile-wide 'all pairs take the fast path' test of the list kernel holds
    (w >= eps / 2), no PAIR fails the mass guard (w_i + w_j >= eps, L:1601) and no particle follows its target
    (w <= eps, L:1458).  Four coincident batches: more partners per particle than the list kernel stages in LDS, so
    the entries come from its second enumeration -- which must not test pairs against a lone w_j (n_collided counts
    every visited pair here)."""
    from egg_fluid_simulation_amd.default_config import default_configs
    tweak = dict(min_mass=1.0 / 0.7e-8, max_mass=1.0 / 0.6e-8)
    w, y = default_configs()
    w.update(tweak)
    y.update(tweak)
    h = _packed(egg, walk, white_config=w, yolk_config=y)
    o = oracle_mod.Oracle()
    o.set_config(WHITE, dict(oracle_mod.DEFAULT_WHITE, **tweak))
    o.set_config(YOLK, dict(oracle_mod.DEFAULT_YOLK, **tweak))
    centers = [(300.0, 300.0)] * 4 + [(700.0, 300.0), (700.0, 480.0)]
    for cx, cy in centers:
        assert h.add(cx, cy, 50, 15) == o.add(cx, cy, 50, 15)
    for step in range(4):
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
        _same(h, o, step)  # (positions and the pair-solve count)
    inv = o.field(WHITE, "inv_mass")
    assert (inv >= 0.5e-8).all() and (inv < 1e-8).all()
    assert h.stats()["packed"][WHITE] >= 1 and h.stats()["max_tile_particles"][WHITE] == 4 * N_W


def test_packed_and_fused_paths_give_the_same_bits(egg):
    from egg_fluid_simulation_amd import _ffi
    n = 400
    xs = 100.0 + 160.0 * (np.arange(n) % 20)
    ys = 100.0 + 160.0 * (np.arange(n) // 20)
    out = []
    for packed in (0, 1):
        h = egg.SimulationHandler()
        h.set_option(_ffi.OPT_PACKED, packed)
        ids = h.add_many(xs, ys, 50, 15)
        for k in range(6):
            h.set_target_positions(ids, xs + 6.0 * k, ys - 4.0 * k)
            h.step(1 / 60, 2, 3)
        st = h.stats()
        assert (st["packed"][WHITE] >= 1) == bool(packed)
        out.append([h.download(w, f) for w in (WHITE, YOLK) for f in ("x", "y", "vx", "vy")] + [st["pair_solves"]])
    for a, b in zip(out[0][:-1], out[1][:-1]):
        assert np.array_equal(a, b)
    assert out[0][-1] == out[1][-1]


def test_one_flying_blob_among_resting_ones(egg, oracle_mod):
    """A blob sent across a field of resting blobs: its swept claim is many times wider than the others' (the host's
    bucket grid treats it apart, eggsim_host_tiling.hip retile) and it merges with whatever it crosses on the way."""
    xs, ys = _grid_xy(6)
    h = egg.SimulationHandler()
    o = oracle_mod.Oracle()
    ids = h.add_many(xs, ys, 50, 15)
    for x, y in zip(xs, ys):
        o.add(float(x), float(y), 50, 15)
    flyer = h.add(-700.0, 420.0, 50, 15)
    assert o.add(-700.0, 420.0, 50, 15) == flyer
    widest = 1
    for step in range(24):
        if step == 2:
            for s in (h, o):
                s.set_target_position(flyer, 1900.0, 470.0)
        h.update(1 / 60)
        o.update(1 / 60)
        widest = max(widest, h.stats()["max_tile_particles"][WHITE] // N_W)
    _same(h, o, "flyer")
    assert h.get_position(flyer)[0] > 300.0  # it is inside the field (or beyond)
    assert widest >= 2  # ... and shared a tile with the blobs it crossed
    assert ids.size == 36
