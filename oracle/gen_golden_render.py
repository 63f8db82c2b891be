"""Generates tests/golden/render_small.npz: a small scene's particle state (from the CPU oracle) and the image the
CPU model of the draw path (oracle/render_model.py) makes of it.  Test infrastructure; the fixture pins the model
(and through it the HIP renderer) against accidental change -- it is NOT a reference-held vector: the reference draws
through LOVE / OpenGL, which does not exist here (parity unpinned, see render_model.py).

    python -m oracle.gen_golden_render
"""
import os

import numpy as np

from oracle import oracle as om
from oracle import render_model as model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCREEN, ORIGIN, ALPHA = (144, 120), (28.0, 40.0), 0.25
FIELDS = ("x", "y", "last_x", "last_y", "vx", "vy", "radius")


def scene():
    om.build()
    o = om.Oracle()
    o.add(100.0, 100.0, 20.0, 8.0, 14, 6)  # small counts: the "few particles" warning case of add (L:114-120)
    o.add(130.0, 96.0, 20.0, 8.0, 14, 6)
    o.set_target_position(2, 300.0, 160.0)
    for _ in range(5):
        o.step(1 / 60, 2, 3)
    return o


def main():
    o = scene()
    states = [{k: o.field(w, k) for k in FIELDS} for w in (0, 1)]
    envs = [o.env(w) for w in (0, 1)]
    colors = [np.ones((s["x"].size, 4), np.float32) for s in states]
    image, canvases = model.render(states, envs, model.DEFAULT_RENDER, colors, SCREEN, ALPHA, ORIGIN)
    out = dict(image=image, screen=np.int32(SCREEN), origin=np.float64(ORIGIN), alpha=np.float64(ALPHA),
               canvas_shapes=np.int32([c.shape[:2] for c in canvases]))
    for w in (0, 1):
        for k in FIELDS:
            out["s%d_%s" % (w, k)] = states[w][k]
        out["env%d" % w] = np.float64([envs[w][k] for k in om.ENV_KEYS])
    path = os.path.join(ROOT, "tests", "golden", "render_small.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; image", image.shape, "alpha max", float(image[..., 3].max()))


if __name__ == "__main__":
    main()
