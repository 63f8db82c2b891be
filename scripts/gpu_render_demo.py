"""Developer: draws a small scene with the HIP renderer, compares it with the CPU model (max difference, differing
pixels), writes the picture as a PNG, and times draw() on BASELINE config 2's 256 batches."""
import os, sys, time, zlib, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from egg_fluid_simulation_amd import SimulationHandler
from oracle import oracle as om, render_model as model


def write_png(path, rgba):
    """8-bit RGBA PNG of a float image over a dark background"""
    a = rgba[..., 3:4]
    rgb = np.clip(rgba[..., :3] + np.float32([0.12, 0.13, 0.16]) * (1 - a), 0, 1)
    img = (rgb * 255 + 0.5).astype(np.uint8)
    raw = b"".join(b"\x00" + img[j].tobytes() for j in range(img.shape[0]))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", img.shape[1], img.shape[0], 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
h, o = SimulationHandler(), om.Oracle()
spots = [(120.0, 120.0), (320.0, 160.0), (210.0, 350.0)]
ids = [h.add(x, y, 50, 15) for x, y in spots]
for x, y in spots:
    o.add(x, y, 50, 15)
for s in (h, o):
    s.set_target_position(ids[1], 470.0, 110.0)
for _ in range(8):
    h.step(1 / 60, 2, 3)
    o.step(1 / 60, 2, 3)
size, origin, t = (560, 500), (-30.0, -30.0), 0.5
image = h.draw(size, origin, interpolation_alpha=t)
states = [{k: o.field(w, k) for k in ("x", "y", "last_x", "last_y", "vx", "vy", "radius")} for w in (0, 1)]
ref, canvases = model.render(states, [o.env(w) for w in (0, 1)], model.DEFAULT_RENDER,
                             [np.ones((states[w]["x"].size, 4), np.float32) for w in (0, 1)], size, t, origin)
for w in (0, 1):
    c = h.render_canvas(w)[0]
    print("canvas %d %s: max |device - model| = %.3g, differing values %d of %d" % (
        w, c.shape, np.abs(c - canvases[w]).max(), int((c != canvases[w]).sum()), c.size))
print("screen %s: max |device - model| = %.3g, differing values %d of %d" % (
    image.shape, np.abs(image - ref).max(), int((image != ref).sum()), image.size))
write_png(os.path.join(out, "render_three_blobs.png"), image)

xs, ys, side = bench.grid_positions(256)
h = SimulationHandler()
h.add_many(xs, ys, 50, 15)
for _ in range(3):
    h.step(1 / 60, 2, 3)
lo = xs.min() - 120.0
n = int(xs.max() + 120.0 - lo)
img = h.draw((n, n), origin=(lo, ys.min() - 120.0))
t0 = time.perf_counter()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(reps):
    img = h.draw((n, n), origin=(lo, ys.min() - 120.0))
dt = (time.perf_counter() - t0) / reps
print("256 batches (%d particles), %d x %d screen: %.2f ms per draw() incl. the %d MB copy to the host" % (
    sum(h.get_n_particles()), n, n, 1e3 * dt, img.nbytes >> 20))
write_png(os.path.join(out, "render_config2.png"), img[::2, ::2])
