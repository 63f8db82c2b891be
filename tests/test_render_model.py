"""The CPU model of the reference's draw path (oracle/render_model.py) held to its committed fixture and to properties
of the shaders it restates (simulation_handler_*.glsl, simulation_handler.lua:1995-2175).  No GPU needed."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import oracle as om
from oracle import render_model as model

F = np.float32
FIELDS = ("x", "y", "last_x", "last_y", "vx", "vy", "radius")


def _fixture():
    g = np.load(os.path.join(GOLDEN_DIR, "render_small.npz"))
    states = [{k: g["s%d_%s" % (w, k)] for k in FIELDS} for w in (0, 1)]
    envs = [dict(zip(om.ENV_KEYS, g["env%d" % w])) for w in (0, 1)]
    return g, states, envs


def test_model_reproduces_its_fixture():
    g, states, envs = _fixture()
    colors = [np.ones((s["x"].size, 4), F) for s in states]
    image, canvases = model.render(states, envs, model.DEFAULT_RENDER, colors, tuple(g["screen"]), float(g["alpha"]),
                                   tuple(g["origin"]))
    assert np.array_equal(image, g["image"])
    assert [c.shape[:2] for c in canvases] == [tuple(s) for s in g["canvas_shapes"]]


def test_fixture_state_is_the_oracles():
    """the fixture's particle state is what the oracle computes today for the generator's scene"""
    from oracle.gen_golden_render import scene
    g, states, envs = _fixture()
    o = scene()
    for w in (0, 1):
        for k in FIELDS:
            assert np.array_equal(o.field(w, k), states[w][k]), (w, k)
        assert np.array_equal(np.float64([o.env(w)[k] for k in om.ENV_KEYS]), g["env%d" % w])


def test_particle_texture_is_the_shaders_gaussian():
    tex = model.particle_texture(4.0, 4.0)
    assert tex.shape == (38, 38) and tex.dtype == F  # (4 * 4 + 3) * 2 (L:626-635)
    assert np.array_equal(tex, tex.T) and np.array_equal(tex, tex[::-1]) and np.array_equal(tex, tex[:, ::-1])
    assert not tex[:3].any() and not tex[-3:].any()  # transparent padding
    # texel (19, 19): uv = (19.5 - 3) / 32, 1 - dist = 2 * |uv - 0.5| * sqrt(2)
    q = 2 * np.hypot(16.5 / 32 - 0.5, 16.5 / 32 - 0.5)
    assert tex[19, 19] == F(np.exp(-4 * np.pi / 3 * q * q))
    assert model.particle_texture(4.0, 6.0).shape == (54, 54)  # the larger of the two max_radius counts (L:626-629)


def _one_particle(vx=0.0, vy=0.0, color=(1, 1, 1, 1), instancing=True, t=1.0):
    state = dict(x=np.float64([50.0]), y=np.float64([40.0]), last_x=np.float64([30.0]), last_y=np.float64([40.0]),
                 vx=np.float64([vx]), vy=np.float64([vy]), radius=np.float64([1.0]))
    env = dict(centroid_x=50.0, centroid_y=40.0, last_centroid_x=50.0, last_centroid_y=40.0)
    cfg = dict(texture_scale=8.0, motion_blur=0.01)
    return model.splat(state, env, cfg, [color], (100, 80), t, model.particle_texture(), instancing)


def test_splat_of_one_particle():
    c = _one_particle()
    # the quad is 16 x 16 px around the canvas centre (50, 40): nothing outside, the texture's symmetry inside
    assert not c[:, :42].any() and not c[:, 58:].any() and not c[:32].any() and not c[48:].any()
    assert c[39, 49, 3] == c[40, 50, 3] == c[39, 50, 3] == c[40, 49, 3] > 0.94
    assert np.array_equal(c[..., 0], c[..., 3])  # white particle: all four channels carry the density
    # interpolation: alpha = 0 draws the particle at its last position (20 px to the left)
    c0 = _one_particle(t=0.0)
    assert np.array_equal(c0[:, 22:38], c[:, 42:58])
    # motion blur stretches the quad ALONG the velocity: (1 + |v| * blur) = 3 times as long, as wide as before
    cx = _one_particle(vx=200.0)
    cy = _one_particle(vy=-200.0)
    # (the texture's 3 transparent padding texels of 38 trim the visible part: 48 px * 32 / 38, and a texel of filter reach)
    assert (cx[40, :, 3] > 0).sum() == 42 and (cx[:, 50, 3] > 0).sum() == 14
    assert (cy[:, 50, 3] > 0).sum() == 42 and (cy[40, :, 3] > 0).sum() == 14
    assert (_one_particle()[40, :, 3] > 0).sum() == 14
    # colour: instanced draw multiplies by the straight rgba, the draw loop by the premultiplied one (L:2035-2041)
    a = _one_particle(color=(0.5, 1.0, 0.25, 0.5))
    b = _one_particle(color=(0.5, 1.0, 0.25, 0.5), instancing=False)
    assert np.array_equal(a[..., 3], b[..., 3]) and np.array_equal(a[..., 1] * F(0.5), b[..., 1])


def test_screen_blend_accumulates_towards_one():
    state = dict(x=np.float64([50.0] * 40), y=np.float64([40.0] * 40), last_x=np.float64([50.0] * 40), last_y=np.float64([40.0] * 40),
                 vx=np.zeros(40), vy=np.zeros(40), radius=np.float64([1.0] * 40))
    env = dict(centroid_x=50.0, centroid_y=40.0, last_centroid_x=50.0, last_centroid_y=40.0)
    c = model.splat(state, env, dict(texture_scale=8.0, motion_blur=0.0), np.ones((40, 4), F), (100, 80), 1.0, model.particle_texture())
    one = _one_particle()[..., 3]
    assert c[..., 3].max() <= 1.0 and c[40, 50, 3] > 0.999999
    # 1 - (1 - a)^40, up to rounding
    assert np.allclose(c[..., 3], 1 - (1 - one.astype(np.float64)) ** 40, atol=1e-5)


def test_composite_thresholds_outlines_and_tints():
    canvas = np.zeros((60, 60, 4), F)
    yy, xx = np.mgrid[0:60, 0:60]
    canvas[..., :] = np.clip(1.3 - np.hypot(xx - 29.5, yy - 29.5) / 15.0, 0, 1).astype(F)[..., None]  # a soft disc
    env = dict(centroid_x=50.0, centroid_y=50.0)
    flat = dict(color=(0.2, 0.4, 0.6, 1.0), outline_color=(1.0, 0.0, 0.0, 1.0), outline_thickness=0.0,
                highlight_strength=0.0, shadow_strength=0.0)
    screen = np.zeros((100, 100, 4), F)
    model.composite(screen, [canvas], [env], [flat])
    # no outline pass: love's colour was never set, the canvas is drawn WHITE (L:2137-2142)
    inside, outside = screen[50, 50], screen[50, 22]
    assert np.array_equal(inside, F([1, 1, 1, 1])) and not outside.any()
    assert not screen[:20].any() and not screen[:, 80:].any()  # the canvas quad covers [20, 80) only
    # with the outline pass: tinted body, a ring of the outline colour where the density is between 0.15 and 0.3
    screen2 = np.zeros((100, 100, 4), F)
    model.composite(screen2, [canvas], [env], [dict(flat, outline_thickness=2.0)])
    assert np.allclose(screen2[50, 50], (0.2, 0.4, 0.6, 1.0))
    ring = (screen2[..., 0] > 0.9) & (screen2[..., 1] < 0.1)
    assert ring.sum() > 50
    r = np.hypot(*(np.argwhere(ring) - 49.5).T)
    assert r.min() > 14.0 and r.max() < 19.5  # density 0.3 at r = 15, 0.15 at r = 17.25
    # the shadow darkens the side whose normal faces (-0.5, 0.75) -- lower left --, the highlight brightens the side
    # facing (1, -1) -- upper right --, and neither touches the opposite side
    base = dict(flat, outline_thickness=1.0)
    plain, shaded, shiny = (np.zeros((100, 100, 4), F) for _ in range(3))
    model.composite(plain, [canvas], [env], [base])
    model.composite(shaded, [canvas], [env], [dict(base, shadow_strength=4.0)])
    model.composite(shiny, [canvas], [env], [dict(base, highlight_strength=1.0)])
    assert plain[58, 43, 2] - shaded[58, 43, 2] > 0.9 and plain[42, 57, 2] == shaded[42, 57, 2]
    assert shiny[42, 57, 2] - plain[42, 57, 2] > 0.4 and abs(shiny[58, 43, 2] - plain[58, 43, 2]) < 1e-6
    assert np.array_equal(plain[..., 3], shaded[..., 3]) and np.array_equal(plain[..., 3], shiny[..., 3])  # alpha untouched
    # _use_lighting off: both switched off whatever the strengths (L:2149-2156)
    off = np.zeros((100, 100, 4), F)
    model.composite(off, [canvas], [env], [dict(base, shadow_strength=4.0, highlight_strength=1.0)], params=dict(use_lighting=False))
    assert np.array_equal(off, plain)
