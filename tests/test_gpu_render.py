"""Image-level regression of the headless renderer through the C ABI against the CPU model
oracle/render_model.py, which restates the reference's draw path: _update_canvases / _draw_canvases
(simulation_handler.lua:1995-2175) and its four shaders.

PARITY UNPINNED against the running reference (LOVE / OpenGL do not exist in this pipeline, and the reference's
own result depends on the GPU's canvas format, MSAA and rasteriser); see the model's header.  Model and kernels
compute every float32 expression in the same order without contraction (division and square root correctly rounded
on both sides), so the images are compared for EQUALITY, bit for bit -- the north star's 1e-4 is met with zero
difference; only the 38 x 38 density texture goes through exp() and gets 1e-7.  The particle STATE the two sides
draw is bit-identical (device solver vs oracle), which the tests assert first."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WHITE, YOLK = 0, 1
ATOL = 0.0  # see above


@pytest.fixture(scope="module")
def egg():
    import egg_fluid_simulation_amd as e
    return e


@pytest.fixture(scope="module")
def model():
    from oracle import render_model
    return render_model


def _state(o, which):
    return {k: o.field(which, k) for k in ("x", "y", "last_x", "last_y", "vx", "vy", "radius")}


def _scene(egg, oracle_mod, steps=6, fast=True):
    """three blobs, one of them chasing a far target (smeared, rotated quads), stepped on both sides"""
    h = egg.SimulationHandler()
    o = oracle_mod.Oracle()
    spots = [(100.0, 100.0), (300.0, 140.0), (190.0, 330.0)]
    ids = [h.add(x, y, 50, 15) for x, y in spots]
    for x, y in spots:
        o.add(x, y, 50, 15)
    if fast:
        for s in (h, o):
            s.set_target_position(ids[1], 900.0, -400.0)
    for _ in range(steps):
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
    for w in (WHITE, YOLK):
        for f in ("x", "y", "vx", "vy"):
            assert np.array_equal(h.download(w, f), o.field(w, f))
    return h, o, ids


def _model_render(model, o, size, t, origin=(0.0, 0.0), cfgs=None, colors=None, params=None, canvas_sizes=None,
                  clear=(0, 0, 0, 0)):
    states = [_state(o, w) for w in (WHITE, YOLK)]
    envs = [o.env(w) for w in (WHITE, YOLK)]
    cfgs = cfgs or model.DEFAULT_RENDER
    if colors is None:
        colors = [np.ones((states[w]["x"].size, 4), np.float32) for w in (WHITE, YOLK)]  # L:985-990
    return model.render(states, envs, cfgs, colors, size, t, origin, params, canvas_sizes, clear)


def _close(a, b, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    diff = np.abs(a.astype(np.float64) - b.astype(np.float64))
    assert diff.max() <= ATOL, (what, float(diff.max()), int((diff > ATOL).sum()))


def test_particle_texture_matches_the_shader(egg, model):
    h = egg.SimulationHandler()
    tex = h.particle_texture()
    ref = model.particle_texture(4.0, 4.0)
    assert tex.shape == ref.shape == (38, 38)  # (4 * 4 + 3) * 2 (L:626-635)
    assert np.abs(tex - ref).max() <= 1e-7
    assert tex[:3].max() == 0 and tex[:, :3].max() == 0 and tex[-3:].max() == 0  # the transparent padding
    assert 0.99 < tex[18:20, 18:20].min() <= 1.0  # the gaussian's peak sits between the four centre texels


def test_draw_matches_the_model(egg, oracle_mod, model):
    h, o, _ = _scene(egg, oracle_mod)
    size, origin, t = (520, 470), (-40.0, -30.0), 0.35
    image = h.draw(size, origin, interpolation_alpha=t, clear=(0.1, 0.2, 0.3, 1.0))
    ref, canvases = _model_render(model, o, size, t, origin, clear=(0.1, 0.2, 0.3, 1.0))
    for w in (WHITE, YOLK):
        canvas, (x0, y0) = h.render_canvas(w)
        _close(canvas, canvases[w], "canvas %d" % w)
        env = o.env(w)
        assert (x0, y0) == (env["centroid_x"] - 0.5 * canvas.shape[1], env["centroid_y"] - 0.5 * canvas.shape[0])  # L:2131-2132
        assert canvas[..., 3].max() > 0.9  # blobs are dense: the screen blend saturates towards 1
    _close(image, ref, "screen")
    # the picture is not trivial: background, white body, yolk, outline and the shadowed rim are all there
    assert np.unique(np.round(image[..., :3], 2).reshape(-1, 3), axis=0).shape[0] > 50
    assert np.allclose(image[0, 0], (0.1, 0.2, 0.3, 1.0))
    # same call again: every pixel the same bits (tiles blend their particles in index order, whatever the scheduling)
    again = h.draw(size, origin, interpolation_alpha=t, clear=(0.1, 0.2, 0.3, 1.0))
    assert np.array_equal(image, again)


def test_interpolation_alpha_comes_from_update(egg, oracle_mod, model):
    h, o, _ = _scene(egg, oracle_mod, steps=2, fast=False)
    assert h.update(0.004) == 0 and o.update(0.004) == 0  # accumulates, no step: alpha = 0.004 * 60 (L:216)
    t = o.interpolation_alpha
    assert 0.2 < t < 0.3 and h.interpolation_alpha == t
    image = h.draw((450, 420))
    ref, _ = _model_render(model, o, (450, 420), t)
    _close(image, ref, "screen")


def test_particle_colours_premultiplied_loop_and_config_keys(egg, oracle_mod, model):
    """use_particle_color: particles carry their batch's colour (L:978-984), set_*_color rewrites it (L:1110-1129); the
    non-instanced draw loop premultiplies by alpha (L:2035-2041); outline / lighting keys away from their defaults"""
    import copy
    white, yolk = egg.default_configs()
    white = dict(white, outline_thickness=2.5, highlight_strength=0.6, shadow_strength=0.7, texture_scale=10.0, motion_blur=0.002)
    yolk = dict(yolk, outline_thickness=0.0, highlight_strength=1.5, shadow_strength=0.4)
    h = egg.SimulationHandler(copy.deepcopy(white), copy.deepcopy(yolk))
    h._use_particle_color = True
    o = oracle_mod.Oracle()
    spots = [(100.0, 100.0), (230.0, 120.0)]
    a = h.add(*spots[0], 50, 15, white_color=[0.9, 0.5, 0.4, 0.8], yolk_color=[0.3, 0.9, 0.2, 1.0])
    b = h.add(*spots[1], 50, 15)
    for x, y in spots:
        o.add(x, y, 50, 15)
    for s in (h, o):
        s.set_target_position(b, 500.0, 420.0)
    for _ in range(5):
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
    h.set_yolk_color(b, 0.2, 0.3, 1.0, 0.6)
    n = [(157, 15)[w] for w in (WHITE, YOLK)]
    colors = [np.concatenate([np.tile(np.float32(ca), (n[w], 1)), np.tile(np.float32(cb), (n[w], 1))])
              for w, (ca, cb) in enumerate([([0.9, 0.5, 0.4, 0.8], white["color"]), ([0.3, 0.9, 0.2, 1.0], [0.2, 0.3, 1.0, 0.6])])]
    # batch b was created without a yolk colour: it shares the config's table, set_yolk_color changed config.color (L:49-50)
    assert h.get_yolk_config()["color"] == [0.2, 0.3, 1.0, 0.6]
    cfgs = [dict(model.DEFAULT_RENDER[0], **{k: white[k] for k in ("outline_thickness", "highlight_strength", "shadow_strength", "texture_scale", "motion_blur")}),
            dict(model.DEFAULT_RENDER[1], color=(0.2, 0.3, 1.0, 0.6), **{k: yolk[k] for k in ("outline_thickness", "highlight_strength", "shadow_strength")})]
    for instancing in (True, False):
        params = dict(use_particle_color=True, use_instancing=instancing)
        image = h.draw((560, 520), (-30.0, -40.0), interpolation_alpha=1.0, use_instancing=instancing)
        ref, canvases = _model_render(model, o, (560, 520), 1.0, (-30.0, -40.0), cfgs, colors, params)
        for w in (WHITE, YOLK):
            _close(h.render_canvas(w)[0], canvases[w], "canvas %d instancing %s" % (w, instancing))
        _close(image, ref, "screen instancing %s" % instancing)
    # yolk has no outline: love's colour is still the white's when the yolk canvas is drawn (L:2137-2142) -- and the model
    # agrees, so the check above covered it; make sure the quirk is really in the picture
    cfg_plain = [cfgs[0], dict(cfgs[1], outline_thickness=1.0)]
    ref_plain, _ = _model_render(model, o, (560, 520), 1.0, (-30.0, -40.0), cfg_plain, colors, dict(use_particle_color=True))
    assert np.abs(ref_plain - image).max() > 0.05


def test_lighting_switch_and_explicit_canvas_sizes(egg, oracle_mod, model):
    h, o, _ = _scene(egg, oracle_mod, steps=3, fast=False)
    h._use_lighting = False
    sizes = [(400, 390), (333, 301)]
    image = h.draw((500, 480), canvas_sizes=sizes, interpolation_alpha=0.0)
    ref, canvases = _model_render(model, o, (500, 480), 0.0, params=dict(use_lighting=False), canvas_sizes=sizes)
    for w in (WHITE, YOLK):
        canvas, _ = h.render_canvas(w)
        assert canvas.shape[:2] == (sizes[w][1], sizes[w][0])
        _close(canvas, canvases[w], "canvas %d" % w)
    _close(image, ref, "screen")


def test_canvases_only_grow(egg, oracle_mod, model):
    h, o, ids = _scene(egg, oracle_mod, steps=2, fast=False)
    h.draw((64, 64))
    first = [h.render_canvas(w)[0].shape[:2] for w in (WHITE, YOLK)]
    expect = [model.canvas_size(o.env(w), model.DEFAULT_RENDER[w]) for w in (WHITE, YOLK)]
    assert first == [(hh, ww) for ww, hh in expect]
    for s in (h, o):
        s.remove(ids[2])  # the bounds shrink, the canvas does not (L:1957-1970)
        s.step(1 / 60, 2, 3)
    h.draw((64, 64))
    for w in (WHITE, YOLK):
        fresh = model.canvas_size(o.env(w), model.DEFAULT_RENDER[w])
        assert fresh[1] < first[w][0] - 100  # a canvas made for the two remaining blobs would be much lower
        assert h.render_canvas(w)[0].shape[:2] == (first[w][0], max(first[w][1], fresh[0]))


def test_nothing_is_drawn_without_canvases(egg):
    """before the first _step there are no canvases (L:1997-1999, L:2118); the same while a type has no particles"""
    h = egg.SimulationHandler()
    h.add(100.0, 100.0, 50, 15)
    image = h.draw((96, 80), clear=(0.5, 0.25, 0.125, 1.0))
    assert image.shape == (80, 96, 4) and np.all(image == np.float32([0.5, 0.25, 0.125, 1.0]))
    with pytest.raises(egg.EggError):
        h.render_canvas(WHITE)
    h.step(1 / 60, 2, 3)
    assert h.draw((96, 80), origin=(50.0, 60.0))[..., 3].max() > 0.5


def test_colour_aliasing_through_the_c_abi(egg):
    """egg_set_color on a batch without its own colour table changes config.color (L:49-50, L:349-350); a batch that got
    a colour argument at add, or a config set afterwards, keeps the tables apart"""
    import ctypes as C
    from egg_fluid_simulation_amd import _ffi
    h = egg.SimulationHandler()
    lib, hd = h._lib, h._h
    a, b = h.add(0.0, 0.0, 50, 15), h.add(300.0, 0.0, 50, 15, white_color=[0.1, 0.2, 0.3, 1.0])
    cfg = _ffi.EggRenderConfig()
    assert lib.egg_set_color(hd, b, WHITE, 0.5, 0.5, 0.5, 1.0) == 0
    lib.egg_get_render_config(hd, WHITE, C.byref(cfg))
    assert np.allclose(list(cfg.color), [0.961, 0.961, 0.953, 1.0])  # b owns its table
    assert lib.egg_set_color(hd, a, WHITE, 2.0, -1.0, 0.25, 0.5) == 0  # clamped (L:300-319)
    lib.egg_get_render_config(hd, WHITE, C.byref(cfg))
    assert list(cfg.color) == [1.0, 0.0, 0.25, 0.5]
    lib.egg_set_render_config(hd, WHITE, C.byref(cfg))  # a new table for the config (L:1307-1311)
    assert lib.egg_set_color(hd, a, WHITE, 0.0, 0.0, 0.0, 1.0) == 0
    lib.egg_get_render_config(hd, WHITE, C.byref(cfg))
    assert list(cfg.color) == [1.0, 0.0, 0.25, 0.5]
    assert lib.egg_set_color(hd, 99, YOLK, 0.0, 0.0, 0.0, 1.0) == _ffi.EGG_WARN_UNKNOWN_ID


def test_config2_sized_scene_renders_every_blob(egg):
    """BASELINE config 2's 256 batches fill the reference's 2560 px canvas cap (L:1953-1954) exactly: 44 k particles,
    every 16 x 16 tile sorted and blended; checked through properties (no model run at this size)"""
    from bench import grid_positions
    xs, ys, side = grid_positions(256)
    h = egg.SimulationHandler()
    h.add_many(xs, ys, 50, 15)
    for _ in range(3):
        h.step(1 / 60, 2, 3)
    lo, hi = xs.min() - 120.0, xs.max() + 120.0
    size = int(hi - lo)
    image = h.draw((size, size), origin=(lo, ys.min() - 120.0))
    canvas, (x0, y0) = h.render_canvas(WHITE)
    assert max(canvas.shape[:2]) <= 2560
    # every blob's centre is opaque white-ish, the gaps between blobs stay clear
    cx = np.round(xs - lo).astype(int)
    cy = np.round(ys - (ys.min() - 120.0)).astype(int)
    assert image[cy, cx, 3].min() > 0.99
    # (a quad reaches 48 px beyond its particle: between two neighbours at the 160 px pitch the densities join, in the
    # middle of four blobs -- 113 px from each centre -- nothing is drawn)
    gx, gy = int(round(0.5 * (xs[0] + xs[1]) - lo)), int(round(0.5 * (ys[0] + ys[side]) - (ys.min() - 120.0)))
    assert image[gy, gx, 3] == 0 and image[0, 0, 3] == 0 and image[cy[0], gx, 3] > 0
    assert np.array_equal(image, h.draw((size, size), origin=(lo, ys.min() - 120.0)))


def test_device_matches_the_committed_image(egg):
    """tests/golden/render_small.npz (oracle/gen_golden_render.py): the device steps the generator's scene itself and
    must draw the committed image bit for bit -- no model run in between"""
    import os
    import warnings
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, "render_small.npz"))
    h = egg.SimulationHandler()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", egg.EggWarning)  # "only 14 white / 6 yolk particles will be created" (L:114-120)
        h.add(100.0, 100.0, 20.0, 8.0, white_n_particles=14, yolk_n_particles=6)
        b = h.add(130.0, 96.0, 20.0, 8.0, white_n_particles=14, yolk_n_particles=6)
    h.set_target_position(b, 300.0, 160.0)
    for _ in range(5):
        h.step(1 / 60, 2, 3)
    for w in (WHITE, YOLK):
        assert np.array_equal(h.download(w, "x"), g["s%d_x" % w]) and np.array_equal(h.download(w, "vy"), g["s%d_vy" % w])
    image = h.draw(tuple(int(v) for v in g["screen"]), tuple(g["origin"]), interpolation_alpha=float(g["alpha"]))
    assert np.array_equal(image, g["image"])
    assert [h.render_canvas(w)[0].shape[:2] for w in (WHITE, YOLK)] == [tuple(s) for s in g["canvas_shapes"]]


def test_canvas_cap_clips_the_splat(egg, oracle_mod, model):
    """two blobs 2460 px apart: the canvas stops at 2560 px (L:1953-1954), centred on the common centroid, so both blobs
    hang over its left / right edge and their quads are clipped there -- on both sides alike"""
    h = egg.SimulationHandler()
    o = oracle_mod.Oracle()
    for x in (0.0, 2460.0):
        h.add(x, 50.0, 50, 15)
        o.add(x, 50.0, 50, 15)
    for _ in range(2):
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
    image = h.draw((2760, 300), origin=(-150.0, -100.0), interpolation_alpha=1.0)
    ref, canvases = _model_render(model, o, (2760, 300), 1.0, (-150.0, -100.0))
    for w in (WHITE, YOLK):
        canvas, (x0, _) = h.render_canvas(w)
        assert canvas.shape[1] == 2560
        _close(canvas, canvases[w], "canvas %d" % w)
    white, (x0, _) = h.render_canvas(WHITE)
    assert x0 == pytest.approx(1230.0 - 1280.0, abs=1.0)
    assert white[:, 0, 3].max() > 0.5 and white[:, -1, 3].max() > 0.5  # cut through the blobs' density
    _close(image, ref, "screen")


def test_drawing_does_not_disturb_the_solver(egg):
    """draw() between steps (its kernels share the white stream, its reductions read both state buffers) leaves the
    particle state of a moving scene exactly where a run without drawing puts it"""
    def run(draw):
        h = egg.SimulationHandler()
        ids = [h.add(100.0 + 130.0 * k, 100.0 + 40.0 * (k % 2), 50, 15) for k in range(5)]
        for step in range(12):
            for k, i in enumerate(ids):
                h.set_target_position(i, 100.0 + 130.0 * k + 15.0 * step, 100.0 + 40.0 * (k % 2) - 9.0 * step)
            h.update(1 / 60)
            if draw and step % 2 == 0:
                h.draw((320, 240), origin=(0.0, 0.0), interpolation_alpha=0.5)
        return [h.download(w, f) for w in (WHITE, YOLK) for f in ("x", "y", "vx", "vy", "last_x", "last_y")]
    plain, drawn = run(False), run(True)
    assert all(np.array_equal(a, b) for a, b in zip(plain, drawn))


def test_other_particle_radii_change_texture_and_quads(egg, oracle_mod, model):
    """white radii 3..5 px (per-particle radius from the mass distribution, L:955-962), yolk radius 6: the density
    texture is sized by the LARGER max_radius (L:626-629), every quad by its own particle's radius"""
    white, yolk = egg.default_configs()
    white = dict(white, min_radius=3.0, max_radius=5.0)
    yolk = dict(yolk, min_radius=6.0, max_radius=6.0)
    h = egg.SimulationHandler(dict(white), dict(yolk))
    from oracle.oracle import CONFIG_KEYS  # the oracle takes the solver keys only
    o = oracle_mod.Oracle({k: white[k] for k in CONFIG_KEYS}, {k: yolk[k] for k in CONFIG_KEYS})
    for x, y in ((100.0, 100.0), (260.0, 130.0)):
        assert h.add(x, y, 50, 15) == o.add(x, y, 50, 15)
    for s in (h, o):
        s.set_target_position(2, 420.0, 260.0)
    for _ in range(4):
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
    assert np.array_equal(h.download(WHITE, "x"), o.field(WHITE, "x"))
    radii = h.download(WHITE, "radius")
    assert radii.min() >= 3.0 and radii.max() <= 5.0 and np.unique(radii).size > 20
    assert h.particle_texture().shape == (54, 54)  # (6 * 4 + 3) * 2
    image = h.draw((480, 400), (-20.0, -20.0), interpolation_alpha=0.75)
    states = [_state(o, w) for w in (WHITE, YOLK)]
    ref, canvases = model.render(states, [o.env(w) for w in (WHITE, YOLK)], model.DEFAULT_RENDER,
                                 [np.ones((s["x"].size, 4), np.float32) for s in states], (480, 400), 0.75, (-20.0, -20.0),
                                 max_radius=(5.0, 6.0))
    for w in (WHITE, YOLK):
        _close(h.render_canvas(w)[0], canvases[w], "canvas %d" % w)
    _close(image, ref, "screen")


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_render_settings_match_the_model(egg, oracle_mod, model, seed):
    """random render keys, thresholds, colours, screen placement and interpolation on one small scene"""
    import copy
    rng = np.random.default_rng(seed)
    white, yolk = egg.default_configs()
    cfgs = []
    for base in (white, yolk):
        cfgs.append(dict(color=[float(v) for v in rng.uniform(0.2, 1.0, 4)], outline_color=[float(v) for v in rng.uniform(0.0, 1.0, 4)],
                         outline_thickness=float(rng.choice([0.0, 0.5, 1.0, 2.0, 3.7])), highlight_strength=float(rng.choice([0.0, 0.5, 2.0])),
                         shadow_strength=float(rng.choice([0.0, 0.3, 1.5])), texture_scale=float(rng.uniform(6.0, 14.0)),
                         motion_blur=float(rng.choice([0.0, 0.0003, 0.004]))))
    h = egg.SimulationHandler(dict(white, **copy.deepcopy(cfgs[0])), dict(yolk, **copy.deepcopy(cfgs[1])))
    h._use_particle_color = bool(rng.integers(2))
    h._use_lighting = bool(rng.integers(2))
    h._thresholding_threshold = float(rng.uniform(0.15, 0.6))
    h._thresholding_smoothness = float(rng.uniform(0.005, 0.05))
    o = oracle_mod.Oracle()
    spots = [(float(x), float(y)) for x, y in rng.uniform(60.0, 260.0, (3, 2))]
    batch_colors = []
    for x, y in spots:
        wc, yc = [float(v) for v in rng.uniform(0, 1, 4)], [float(v) for v in rng.uniform(0, 1, 4)]
        h.add(x, y, 30.0, 12.0, white_color=wc, yolk_color=yc)
        o.add(x, y, 30.0, 12.0)
        batch_colors.append((wc, yc))
    target = (float(rng.uniform(0, 400)), float(rng.uniform(0, 400)))
    for s in (h, o):
        s.set_target_position(2, *target)
    for _ in range(int(rng.integers(2, 7))):
        h.step(1 / 60, 2, 3)
        o.step(1 / 60, 2, 3)
    assert np.array_equal(h.download(WHITE, "x"), o.field(WHITE, "x")) and np.array_equal(h.download(YOLK, "vy"), o.field(YOLK, "vy"))
    n = [o.n_particles(w) // 3 for w in (WHITE, YOLK)]
    if h._use_particle_color:
        colors = [np.concatenate([np.tile(np.float32(bc[w]), (n[w], 1)) for bc in batch_colors]) for w in (WHITE, YOLK)]
    else:
        colors = [np.ones((3 * n[w], 4), np.float32) for w in (WHITE, YOLK)]
    size = (int(rng.integers(200, 420)), int(rng.integers(180, 400)))
    origin = (float(rng.uniform(-60, 60)), float(rng.uniform(-60, 60)))
    t = float(rng.uniform(0, 1))
    instancing = bool(rng.integers(2))
    params = dict(use_particle_color=h._use_particle_color, use_lighting=h._use_lighting, use_instancing=instancing,
                  threshold=h._thresholding_threshold, smoothness=h._thresholding_smoothness)
    clear = tuple(float(v) for v in rng.uniform(0, 1, 4))
    image = h.draw(size, origin, interpolation_alpha=t, clear=clear, use_instancing=instancing)
    ref, canvases = _model_render(model, o, size, t, origin, cfgs, colors, params, clear=clear)
    for w in (WHITE, YOLK):
        _close(h.render_canvas(w)[0], canvases[w], "canvas %d" % w)
    _close(image, ref, "screen")


def test_colour_table_aliasing_survives_draw_calls():
    """L:49-50, L:349-350: a batch added WITHOUT a colour shares its type's config colour table, so set_*_color on it
    retints the config (and every other batch sharing the table) -- also after several draw() calls: the Python host
    sends its render config only where the reference calls set_*_config (egg_set_render_config = a NEW colour table,
    L:1307-1311, which ends the sharing, as test_colour_aliasing_through_the_c_abi checks)."""
    import ctypes as C
    import egg_fluid_simulation_amd as e
    from egg_fluid_simulation_amd import _ffi
    h = e.SimulationHandler()
    lib = h._lib
    a = h.add(300.0, 300.0, 50, 15)          # no colour given: shares the config's table
    b = h.add(600.0, 300.0, 50, 15, [0.2, 0.4, 0.6, 1.0], None)  # a white colour of its own
    h.step(1 / 60, 2, 3)
    cfg = _ffi.EggRenderConfig()
    assert lib.egg_get_render_config(h._h, 0, C.byref(cfg)) == 0
    before = list(cfg.color)
    for _ in range(3):
        h.draw((320, 240), origin=(150.0, 150.0))
    assert lib.egg_set_color(h._h, a, 0, 0.9, 0.1, 0.1, 1.0) == 0
    assert lib.egg_get_render_config(h._h, 0, C.byref(cfg)) == 0
    assert [round(v, 6) for v in cfg.color] == [0.9, 0.1, 0.1, 1.0] and list(cfg.color) != before  # the shared table changed
    # the batch with its own colour does not touch the config
    assert lib.egg_set_color(h._h, b, 0, 0.1, 0.9, 0.1, 1.0) == 0
    assert lib.egg_get_render_config(h._h, 0, C.byref(cfg)) == 0
    assert [round(v, 6) for v in cfg.color] == [0.9, 0.1, 0.1, 1.0]
