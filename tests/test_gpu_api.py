"""Behaviour of the SimulationHandler surface on the device path: the reference's defaults,
warnings, errors, accumulator, add/remove and live config changes (simulation_handler.lua:27-419)."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WHITE, YOLK = 0, 1


@pytest.fixture(scope="module")
def egg():
    import egg_fluid_simulation_amd as e
    return e


def test_add_defaults_counts_and_ids(egg):
    h = egg.SimulationHandler()
    a = h.add(10, 20)  # white radius 15 * 4 = 60, yolk 60 / 5 = 12 (L:41-47)
    b = h.add(500, 20, 50, 15)
    assert (a, b) == (1, 2) and h.list_ids() == [1, 2]
    assert h.get_n_particles(a) == (225, 9) and h.get_n_particles(b) == (157, 15)
    assert h.get_n_particles() == (382, 24)
    assert h.get_target_position(a) == (10.0, 20.0)
    x, y = h.get_position(a)
    assert abs(x - 10) < 1 and abs(y - 20) < 1


def test_error_and_warning_conventions(egg):
    h = egg.SimulationHandler()
    a = h.add(0, 0, 50, 15)
    with pytest.raises(egg.EggError, match="white radius cannot be 0 or negative"):
        h.add(0, 0, -1, 5)
    with pytest.raises(egg.EggError, match="yolk particle count cannot be 1 or negative"):
        h.add(0, 0, 50, 15, None, None, 20, 1)
    with pytest.warns(egg.EggWarning, match="only 7 white / 3 yolk"):
        c = h.add(300, 0, 10, 6)  # ceil(100/16) = 7 < 10, ceil(36/16) = 3 < 5: warning, batch still created
    assert h.get_n_particles(c) == (7, 3)
    with pytest.warns(egg.EggWarning, match="set_target_position: no batch with id"):
        h.set_target_position(99, 1, 2)  # warning, no throw (L:259)
    with pytest.warns(egg.EggWarning, match="remove: no batch with id"):
        h.remove(99)
    for fn in (h.get_position, h.get_target_position, h.get_n_particles):
        with pytest.raises(egg.EggError, match="no batch with id"):
            fn(99)
    with pytest.raises(egg.EggError, match="`step_delta` is not a number > 0"):
        h.update(0.1, -1)
    with pytest.raises(egg.EggError, match="`n_substeps` is not a number > 0"):
        h.update(0.1, 1 / 60, 0)
    with pytest.raises(egg.EggError, match="`n_collision_steps` is not a number > 0"):
        h.update(0.1, 1 / 60, 2, 0)
    assert h.get_position(a)[0] == pytest.approx(0, abs=1)


def test_update_accumulator(egg):
    h = egg.SimulationHandler()
    h.add(0, 0, 50, 15)
    assert sum(h.update(1 / 60) for _ in range(100)) == 100 and h.elapsed == 0.0
    assert h.update(1.0) == 5 and h.elapsed == 0.0  # death-spiral guard (L:203-213)
    assert h.update(0.01) == 0 and h.interpolation_alpha == pytest.approx(0.6)
    assert h.update(0.01, 1 / 60, 1.2, 2.5) == 1  # counts are ceil'ed (L:181-182)


def test_remove_and_add_between_steps_match_oracle(egg, oracle_mod):
    h, o = egg.SimulationHandler(), oracle_mod.Oracle()
    ids = [h.add(120.0 * k, 0, 50, 15) for k in range(4)]
    for k in range(4):
        o.add(120.0 * k, 0, 50, 15)
    for _ in range(5):
        h.update(1 / 60)
        o.update(1 / 60)
    h.remove(ids[1])
    o.remove(ids[1])
    n = h.add(1000, 50, 50, 15)
    assert n == o.add(1000, 50, 50, 15) == 5
    assert h.list_ids() == [1, 3, 4, 5]
    for _ in range(5):
        h.update(1 / 60)
        o.update(1 / 60)
    for w in (WHITE, YOLK):
        assert np.array_equal(h.download(w, "x"), o.positions(w)[0])
        assert np.array_equal(h.download(w, "batch_id"), o.field(w, "batch_id"))
    for i in h.list_ids():
        assert h.get_position(i) == o.get_position(i)


def test_live_config_change_rederives_mass_and_radius(egg, oracle_mod):
    from egg_fluid_simulation_amd.default_config import default_configs
    h, o = egg.SimulationHandler(), oracle_mod.Oracle()
    h.add(0, 0, 50, 15)
    o.add(0, 0, 50, 15)
    for _ in range(3):
        h.update(1 / 60)
        o.update(1 / 60)
    w, _ = default_configs()
    w.update(max_mass=3.0, min_radius=3.0, max_radius=5.0, damping=0.3, collision_strength=0.9)
    h.set_white_config(w)
    o.set_config(WHITE, dict(oracle_mod.DEFAULT_WHITE, max_mass=3.0, min_radius=3.0, max_radius=5.0, damping=0.3,
                             collision_strength=0.9))
    assert h.get_white_config()["max_mass"] == 3.0
    for _ in range(4):
        h.update(1 / 60)
        o.update(1 / 60)
    for f in ("x", "y", "radius", "inv_mass"):
        assert np.array_equal(h.download(WHITE, f), o.field(WHITE, f)), f


def test_instance_record_and_last_positions(egg, oracle_mod):
    h, o = egg.SimulationHandler(), oracle_mod.Oracle()
    h.add(50, 60, 50, 15)
    o.add(50, 60, 50, 15)
    for _ in range(3):
        h.update(1 / 60)
        o.update(1 / 60)
    rec = h.download_instance_data(WHITE)
    assert rec.shape == (157, 7)
    for col, f in enumerate(("x", "y", "last_x", "last_y", "vx", "vy", "radius")):
        assert np.array_equal(rec[:, col], o.field(WHITE, f)), f


def test_environment_reductions_match_the_reference_fields(egg, oracle_mod):
    """what :draw() sizes and places its canvases with: AABB incl. radius, centroids (summed in particle order),
    largest radius and speed -- bit for bit the fields the reference's env holds after each step"""
    h, o = egg.SimulationHandler(), oracle_mod.Oracle()
    for w in (WHITE, YOLK):
        e = h.get_environment(w)
        assert e["min_x"] == float("inf") and e["max_y"] == -float("inf") and e["centroid_x"] == 0.0
    cs = [(-30.0, 12.0), (400.0, 300.0), (415.0, 310.0), (900.0, -250.0)]
    for k, (x, y) in enumerate(cs):
        h.add(x, y, 50 if k % 2 == 0 else 35, 15)
        o.add(x, y, 50 if k % 2 == 0 else 35, 15)
    for step in range(12):
        for i, (x, y) in enumerate(cs):
            h.set_target_position(i + 1, x + 3.0 * step, y - 2.0 * step)
            o.set_target_position(i + 1, x + 3.0 * step, y - 2.0 * step)
        h.update(1 / 60)
        o.update(1 / 60)
        if step in (0, 5, 11):
            for w in (WHITE, YOLK):
                mine, ref = h.get_environment(w), o.env(w)
                for key in mine:
                    assert mine[key] == ref[key], (step, w, key, mine[key], ref[key])


def test_plain_c_caller_matches_the_oracle(egg, oracle_mod, tmp_path):
    """tests/c/abi_roundtrip.c drives the library from C99 (no Python, no ctypes in the product path of that
    process); its printed batch positions are the oracle's to the last digit"""
    import subprocess
    from test_abi import _build_c_caller
    exe = _build_c_caller(tmp_path)
    out = subprocess.run([exe, "7"], capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0] == "create 0"
    o = oracle_mod.Oracle()
    a = o.add(400.0, 300.0, 50.0, 15.0)
    b = o.add(470.0, 320.0, 35.0, 9.0)
    for k in range(7):
        o.set_target_position(a, 400.0 + 3.0 * k, 300.0 - 2.0 * k)
        o.update(1 / 60)
    for line, i in zip(out[1:3], (a, b)):
        _, bid, x, y = line.split()
        assert int(bid) == i and (float(x), float(y)) == o.get_position(i), line
    assert out[3].startswith("unknown id:") and "no batch with id" in out[3]
    # ... and its egg_render image is the CPU model's of the draw path (oracle/render_model.py), pixel for pixel
    from oracle import render_model as model
    states = [{k: o.field(w, k) for k in ("x", "y", "last_x", "last_y", "vx", "vy", "radius")} for w in (WHITE, YOLK)]
    ref, _ = model.render(states, [o.env(w) for w in (WHITE, YOLK)], model.DEFAULT_RENDER,
                          [np.ones((s["x"].size, 4), np.float32) for s in states], (240, 200), 1.0, (300.0, 200.0))
    for line in out[4:8]:
        _, px, py, r, g, b, a = line.split()
        assert np.array_equal(np.float32([r, g, b, a]), ref[int(py), int(px)]), line
    assert ref[100, 100, 3] == 1.0 and ref[5, 5, 3] == 0.0  # an egg and the empty corner
    assert float(out[8].split()[2]) == float(np.cumsum(ref.reshape(-1).astype(np.float64))[-1])  # (cumsum: added up in order, like the C loop)


def test_unsupported_configuration_fails_loudly(egg):
    """one collision pass per sub-step keeps a hash generation per sub-step alive; the device path holds 8"""
    h = egg.SimulationHandler()
    h.add(0, 0, 50, 15)
    with pytest.raises(egg.EggError, match="un-cleared hash generations"):
        h.update(1 / 60, 1 / 60, 9, 1)


def test_batch_hand_over_device_to_device_matches_the_host_path(egg, oracle_mod):
    """egg_export_batch / egg_import_batch with DEVICE buffers (what sharding.py hands to RCCL: the particle state never
    touches host memory) against the same hand-over through numpy arrays, and both against one oracle that never
    moved anything"""
    import torch
    centers = [(200.0, 200.0), (230.0, 215.0), (600.0, 200.0)]
    o = oracle_mod.Oracle()
    for x, y in centers:
        o.add(x, y, 50, 15)
    results = []
    for path in ("device", "host"):
        a, b = egg.SimulationHandler(), egg.SimulationHandler()
        ids = a.add_many_keyed([c[0] for c in centers], [c[1] for c in centers], [1, 2, 3], 50, 15)
        for _ in range(3):
            a.step(1 / 60, 2, 3)
        # the two overlapping batches move to handler b, the third stays
        for lid in ids[:2]:
            nw, ny = a.get_n_particles(int(lid))
            if path == "device":
                buf = torch.empty(9 * (nw + ny), dtype=torch.float64, device="cuda")
                info = a.export_batch_to(int(lid), buf.data_ptr(), buf.data_ptr() + 8 * 9 * nw)
                torch.cuda.synchronize()
                wire = buf.clone()  # (stands for the RCCL transfer)
                b.import_batch_from(info, wire.data_ptr(), wire.data_ptr() + 8 * 9 * nw)
            else:
                info, ws, ys = a.export_batch(int(lid))
                b.import_batch(info, ws, ys)
            a.remove(int(lid))
        for _ in range(3):
            a.step(1 / 60, 2, 3)
            b.step(1 / 60, 2, 3)
        results.append([np.concatenate([b.download(w, f), a.download(w, f)]) for w in (0, 1) for f in ("x", "y", "vx", "vy")])
    for _ in range(6):
        o.step(1 / 60, 2, 3)
    want = [o.field(w, f) for w in (0, 1) for f in ("x", "y", "vx", "vy")]
    # (the oracle's budget counts all three batches; the split handlers each their own: 0.05 N^2 must not bind either way)
    assert all(not s["cut"] or s["which"] == 1 for s in o.pass_stats())
    for got in results:
        for g, w in zip(got[:4], want[:4]):  # white: no budget cut anywhere
            assert np.array_equal(g, w)
    for g_dev, g_host in zip(*results):
        assert np.array_equal(g_dev, g_host)


def test_device_group_behind_the_c_abi_matches_one_oracle(egg, oracle_mod):
    """egg_group_*: three device handles of ONE process (all on GPU 0 here: the box has one) behind one group, x-slabs cut
    at 600 and 1200.  A column of blobs is driven across both cuts and through the blobs resting there (islands span
    devices: the launched step is discarded, the island handed to the lower device, the step re-run), another blob just
    strays into the next slab.  Every particle against ONE oracle that holds all batches."""
    centers = [(300.0, 200.0), (300.0, 330.0), (900.0, 260.0), (1500.0, 200.0), (1500.0, 330.0), (450.0, 700.0), (1000.0, 700.0)]
    centers += [(150.0 + 330.0 * k, 1100.0) for k in range(5)]  # (twelve batches: the yolk budget 0.05 N^2 no longer binds, L:1752-1753)
    g = egg.SimulationGroup([0, 0, 0], cuts=[0.0, 600.0, 1200.0, 1800.0])
    o = oracle_mod.Oracle()
    ids = [g.add(x, y, 50, 15) for x, y in centers]
    assert ids == [o.add(x, y, 50, 15) for x, y in centers] == list(range(1, 13))
    assert [g.owner(i)[0] for i in ids] == [0, 0, 1, 2, 2, 0, 1, 0, 0, 1, 1, 2]
    for k in range(70):
        tx = min(300.0 + 22.0 * k, 1560.0)
        for i, y in ((1, 200.0), (2, 330.0)):  # the slab-0 column runs to the right through slab 1 into slab 2
            g.set_target_position(i, tx, y)
            o.set_target_position(i, tx, y)
        g.set_target_position(6, 450.0 + min(6.0 * k, 330.0), 700.0)  # strays into slab 1 without meeting anything
        o.set_target_position(6, 450.0 + min(6.0 * k, 330.0), 700.0)
        if k % 2:
            assert g.update(1 / 60) == 1
        else:
            g.step(1 / 60, 2, 3)
        o.update(1 / 60)
    c = g.counters()
    assert c["migrations"] >= 4 and c["discarded_steps"] >= 1, c
    assert g.owner(1)[0] == 2 and g.owner(6)[0] == 1
    for which in (0, 1):
        x, y, b = o.field(which, "x"), o.field(which, "y"), o.field(which, "batch_id")
        got = g.particles(which)
        assert sorted(got) == ids
        for i in ids:
            assert np.array_equal(got[i][0], x[b == i]) and np.array_equal(got[i][1], y[b == i]), (which, i)
    for i in ids:
        assert g.get_position(i) == o.get_position(i)
    with pytest.raises(egg.EggError):
        g.get_position(99)
