"""CPU model of the reference's draw path, for image-level regression of the HIP renderer.

TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).

What it restates (reference = /root/reference, "L:" = simulation_handler.lua):
  * the particle density texture: simulation_handler_particle_texture.glsl:7-17 drawn by
    _initialize_particle_texture (L:620-682);
  * _update_canvases (L:1995-2113): every particle's textured, velocity-aligned, smeared quad
    (simulation_handler_instanced_draw.glsl:14-46, and the non-instanced loop L:2009-2052) splatted with the
    "screen", "premultiplied" blend into one canvas per type, centred on the interpolated centroid;
  * resize_canvas_maybe (L:1935-1975): the canvas size from the environment's bounds;
  * _draw_canvases (L:2117-2175): per type the outline pass (simulation_handler_outline.glsl) and the
    thresholding / lighting pass (simulation_handler_lighting.glsl), alpha-blended onto the screen.

PARITY UNPINNED: the reference draws through LOVE / OpenGL, which does not exist in this pipeline, and its
result depends on the GPU (canvas format rgba8 first when available, L:539-556; 4x MSAA, L:452; the driver's
rasteriser and texture filter).  This model and the HIP kernels restate the passes as published, in IEEE
float32, on float32 canvases sampled at pixel centres without multisampling; rotation uses the normalised
velocity where the vertex shader writes cos/sin(atan(vy, vx)); pow(x, 48) is six multiplications.  Both
sides compute every expression in the same order, so the HIP image is expected to equal this model's to the
last bit wherever no transcendental is involved (the only one is exp in the 38 x 38 texture).
"""
import math

import numpy as np

F = np.float32
MAX_CANVAS = 2560  # L:1953-1954

# render keys of the default config tables (simulation_handler_default_config.lua:22-36, 54-68)
DEFAULT_RENDER = [
    dict(color=(0.961, 0.961, 0.953, 1.0), outline_color=(0.973, 0.796, 0.529, 1.0), outline_thickness=1.0,
         highlight_strength=0.0, shadow_strength=1.0, texture_scale=12.0, motion_blur=0.0003),
    dict(color=(0.969, 0.682, 0.141, 1.0), outline_color=(0.984, 0.522, 0.271, 1.0), outline_thickness=1.0,
         highlight_strength=1.0, shadow_strength=0.0, texture_scale=12.0, motion_blur=0.0003),
]
DEFAULT_PARAMS = dict(threshold=0.3, smoothness=0.01, use_particle_color=False, use_lighting=True,
                      use_instancing=True)  # L:444-449, L:494-496


def particle_texture(max_radius_white=4.0, max_radius_yolk=4.0, resolution_factor=4, padding=3):
    """alpha of the (size x size) density texture; all four channels hold it (alpha blend onto a cleared canvas)"""
    radius = max(max_radius_white, max_radius_yolk) * resolution_factor  # L:626-629
    size = int((radius + padding) * 2)  # L:634-635
    c = (np.arange(size, dtype=np.float64) + 0.5 - (size - 2 * radius) / 2) / (2 * radius)  # quad uv at pixel centres
    u, v = np.meshgrid(c, c, indexing="xy")
    inside = (u >= 0) & (u < 1) & (v >= 0) & (v < 1)
    q = 2.0 * np.sqrt((u - 0.5) ** 2 + (v - 0.5) ** 2)  # 1 - dist, particle_texture.glsl:13
    g = np.exp((-4.0 * math.pi / 3.0) * q * q)
    return np.where(inside, g, 0.0).astype(F)


def canvas_size(env, cfg, current=(0, 0)):
    """resize_canvas_maybe (L:1935-1975): canvases only grow"""
    padding = env["max_radius"] * cfg["texture_scale"] * (1 + max(1.0, env["max_velocity"]) * cfg["motion_blur"])
    w = min(math.ceil((env["max_x"] - env["min_x"]) + 2 * padding), MAX_CANVAS)
    h = min(math.ceil((env["max_y"] - env["min_y"]) + 2 * padding), MAX_CANVAS)
    return max(int(w), current[0]), max(int(h), current[1])


def _bilinear_zero(tex, tu, tv):
    """GL_LINEAR sample of a single-channel texture with "clampzero" wrap at texel coordinates tu, tv (float32 arrays)"""
    size = tex.shape[0]
    i0 = np.floor(tu)
    j0 = np.floor(tv)
    fu = (tu - i0).astype(F)
    fv = (tv - j0).astype(F)
    i0 = i0.astype(np.int64)
    j0 = j0.astype(np.int64)

    def at(i, j):
        ok = (i >= 0) & (i < size) & (j >= 0) & (j < size)
        return np.where(ok, tex[np.clip(j, 0, size - 1), np.clip(i, 0, size - 1)], F(0))

    one = F(1)
    top = at(i0, j0) * (one - fu) + at(i0 + 1, j0) * fu
    bot = at(i0, j0 + 1) * (one - fu) + at(i0 + 1, j0 + 1) * fu
    return (top * (one - fv) + bot * fv).astype(F)


def splat(state, env, cfg, colors, size, t, texture, use_instancing=True):
    """_update_canvases for one type.  state: dict of float64 arrays x, y, last_x, last_y, vx, vy, radius in particle
    order; colors: (n, 4) particle rgba; size = (w, h).  Returns the (h, w, 4) float32 canvas."""
    w, h = size
    canvas = np.zeros((h, w, 4), F)
    tsize = F(texture.shape[0])
    t32 = F(t)
    one, half = F(1), F(0.5)
    # frame interpolation of the centroid, in Lua doubles (L:2057-2058); the translation reaches the GPU as float32
    pcx = env["last_centroid_x"] * (1 - t) + env["centroid_x"] * t
    pcy = env["last_centroid_y"] * (1 - t) + env["centroid_y"] * t
    tx, ty = F(w / 2 - pcx), F(h / 2 - pcy)
    x, y = state["x"].astype(F), state["y"].astype(F)
    lx, ly = state["last_x"].astype(F), state["last_y"].astype(F)
    vx, vy = state["vx"].astype(F), state["vy"].astype(F)
    rad = state["radius"].astype(F)
    scale, blur = F(cfg["texture_scale"]), F(cfg["motion_blur"])
    col = np.asarray(colors, F).reshape(-1, 4)
    for k in range(x.size):
        ox = lx[k] * (one - t32) + x[k] * t32  # mix(previous, current, alpha), instanced_draw.glsl:36
        oy = ly[k] * (one - t32) + y[k] * t32
        speed = np.sqrt(vx[k] * vx[k] + vy[k] * vy[k], dtype=F)
        base = rad[k] * scale
        sx, sy = base * (one + speed * blur), base
        c, s = (vx[k] / speed, vy[k] / speed) if speed > 0 else (one, F(0))
        cx, cy = ox + tx, oy + ty
        ex = abs(c) * sx + abs(s) * sy  # half extents of the rotated quad's bounding box
        ey = abs(s) * sx + abs(c) * sy
        if not (math.isfinite(cx) and math.isfinite(cy) and ex < 1e6 and ey < 1e6):
            continue
        i0, i1 = max(0, int(math.floor(cx - ex - half)) - 1), min(w - 1, int(math.ceil(cx + ex - half)) + 1)
        j0, j1 = max(0, int(math.floor(cy - ey - half)) - 1), min(h - 1, int(math.ceil(cy + ey - half)) + 1)
        if i1 < i0 or j1 < j0:
            continue
        px = (np.arange(i0, i1 + 1).astype(F) + half)[None, :]
        py = (np.arange(j0, j1 + 1).astype(F) + half)[:, None]
        dx, dy = px - cx, py - cy
        u = (dx * c + dy * s) * (one / sx)  # quad coordinates in [-1, 1]
        v = (dy * c - dx * s) * (one / sy)
        inside = (np.abs(u) <= one) & (np.abs(v) <= one)
        g = _bilinear_zero(texture, (u * half + half) * tsize - half, (v * half + half) * tsize - half)
        g = np.where(inside, g, F(0))
        r_, g_, b_, a_ = col[k]
        if use_instancing:
            src = np.stack([g * r_, g * g_, g * b_, g * a_], axis=-1)  # texture * color_override
        else:
            src = np.stack([g * (r_ * a_), g * (g_ * a_), g * (b_ * a_), g * a_], axis=-1)  # L:2035-2041
        dst = canvas[j0:j1 + 1, i0:i1 + 1]
        canvas[j0:j1 + 1, i0:i1 + 1] = src + dst * (one - src)  # "screen", "premultiplied"
    return canvas


def _sample_clamp(canvas, u, v):
    """GL_LINEAR, clamp to edge, of an (h, w, 4) canvas at texture coordinates u, v (float32 arrays of one shape)"""
    h, w = canvas.shape[:2]
    one, half = F(1), F(0.5)
    tu, tv = u * F(w) - half, v * F(h) - half
    i0, j0 = np.floor(tu), np.floor(tv)
    fu, fv = (tu - i0).astype(F)[..., None], (tv - j0).astype(F)[..., None]
    i0, j0 = i0.astype(np.int64), j0.astype(np.int64)
    ia, ib = np.clip(i0, 0, w - 1), np.clip(i0 + 1, 0, w - 1)
    ja, jb = np.clip(j0, 0, h - 1), np.clip(j0 + 1, 0, h - 1)
    top = canvas[ja, ia] * (one - fu) + canvas[ja, ib] * fu
    bot = canvas[jb, ia] * (one - fu) + canvas[jb, ib] * fu
    return (top * (one - fv) + bot * fv).astype(F)


def _smoothstep(e0, e1, x):
    t = np.clip((x - e0) / (e1 - e0), F(0), F(1)).astype(F)
    return t * t * (F(3) - F(2) * t)


def _normalize3(x, y, z):
    length = np.sqrt((x * x + y * y) + z * z).astype(F)
    return x / length, y / length, z / length


def _blend_alpha(dst, src):
    """love "alpha", "alphamultiply": rgb = src.rgb * src.a + dst.rgb * (1 - src.a), a = src.a + dst.a * (1 - src.a)"""
    a = src[..., 3:4]
    out = np.empty_like(dst)
    out[..., :3] = src[..., :3] * a + dst[..., :3] * (F(1) - a)
    out[..., 3:4] = a + dst[..., 3:4] * (F(1) - a)
    return out


def composite(screen, canvases, envs, cfgs, origin=(0.0, 0.0), params=None):
    """_draw_canvases onto `screen` ((H, W, 4) float32, modified and returned).  World px = screen px + origin."""
    p = dict(DEFAULT_PARAMS)
    p.update(params or {})
    H, W = screen.shape[:2]
    psx, psy = F(1) / F(W), F(1) / F(H)  # pixel_size = 1 / love_ScreenSize
    thr, smooth = F(p["threshold"]), F(p["smoothness"])
    color = np.ones(4, F)  # love's current colour; only set inside the outline branch (L:2137-2142)
    sx = (np.arange(W).astype(F) + F(0.5))[None, :]
    sy = (np.arange(H).astype(F) + F(0.5))[:, None]
    for canvas, env, cfg in zip(canvases, envs, cfgs):
        ch, cw = canvas.shape[:2]
        cx0 = F(env["centroid_x"] - 0.5 * cw - origin[0])  # L:2131-2132
        cy0 = F(env["centroid_y"] - 0.5 * ch - origin[1])
        dx, dy = sx - cx0, sy - cy0
        cover = (dx >= 0) & (dx < F(cw)) & (dy >= 0) & (dy < F(ch))
        u = np.broadcast_to(dx / F(cw), (H, W)).astype(F)
        v = np.broadcast_to(dy / F(ch), (H, W)).astype(F)
        thickness = cfg["outline_thickness"]
        if thickness > 0:
            center = _sample_clamp(canvas, u, v)
            steps = int(math.ceil(thickness)) + 1
            step_size = F(thickness) / F(steps)
            diag = F(math.sqrt(2.0) / 2.0)
            dirs = [(F(1), F(0)), (F(-1), F(0)), (F(0), F(1)), (F(0), F(-1)),
                    (diag, diag), (-diag, diag), (diag, -diag), (-diag, -diag)]
            max_alpha = np.zeros((H, W), F)
            for ddx, ddy in dirs:
                for step in range(1, steps + 1):
                    reach = F(step) * step_size
                    s = _sample_clamp(canvas, u + (ddx * reach) * psx, v + (ddy * reach) * psy)
                    max_alpha = np.maximum(max_alpha, s[..., 3])
            max_alpha = np.minimum(max_alpha, F(1))
            e0 = F(0.5) * thr
            oa = _smoothstep(e0, e0 + F(0.035), max_alpha)
            oc = np.asarray(cfg["outline_color"], F)
            src = oc[None, None, :] * oa[..., None]
            drawn = cover & (center[..., 3] != 0)  # `discard`
            screen[...] = np.where(drawn[..., None], _blend_alpha(screen, src), screen)
            color = np.asarray(cfg["color"], F)
        # thresholding + lighting
        data = _sample_clamp(canvas, u, v)
        value = _smoothstep(thr - smooth, thr + smooth, data[..., 3])
        if p["use_particle_color"]:
            center = np.concatenate([data[..., :3], value[..., None]], axis=-1) * color
        else:
            center = value[..., None] * color[None, None, :]

        def a_at(ox, oy):
            return _sample_clamp(canvas, u + F(ox) * psx, v + F(oy) * psy)[..., 3]

        tl, tm, tr = a_at(-1, -1), a_at(0, -1), a_at(1, -1)
        ml, mr = a_at(-1, 0), a_at(1, 0)
        bl, bm, br = a_at(-1, 1), a_at(0, 1), a_at(1, 1)
        two = F(2)
        gx = -tl + tr - two * ml + two * mr - bl + br
        gy = -tl - two * tm - tr + bl + two * bm + br
        nx, ny, nz = _normalize3(-gx, -gy, np.ones_like(gx))
        specular = np.zeros((H, W), F)
        if cfg["highlight_strength"] > 0 and p["use_lighting"]:
            lx_, ly_, lz_ = _normalize3(F(1), F(-1), F(1))
            hx, hy, hz = _normalize3(lx_ + F(0), ly_ + F(0), lz_ + F(1))
            d = np.maximum((nx * hx + ny * hy) + nz * hz, F(0))
            d2 = d * d
            d4 = d2 * d2
            d8 = d4 * d4
            d16 = d8 * d8
            d32 = d16 * d16
            specular = specular + F(cfg["highlight_strength"]) * (d32 * d16)  # pow(., 48)
        shadow = np.zeros((H, W), F)
        if cfg["shadow_strength"] > 0 and p["use_lighting"]:
            lx_, ly_, lz_ = _normalize3(F(-0.5), F(0.75), F(0))
            sh = (nx * lx_ + ny * ly_) + nz * lz_
            shadow = _smoothstep(F(0), F(1), np.clip(sh * F(cfg["shadow_strength"]), F(0), F(1)))
        src = np.empty((H, W, 4), F)
        src[..., :3] = center[..., :3] - shadow[..., None] + specular[..., None]
        src[..., 3] = center[..., 3]
        screen[...] = np.where(cover[..., None], _blend_alpha(screen, src), screen)
    return screen


def render(states, envs, cfgs, colors, screen_size, t, origin=(0.0, 0.0), params=None, canvas_sizes=None,
           clear=(0.0, 0.0, 0.0, 0.0), max_radius=(4.0, 4.0)):
    """SimulationHandler:draw() (L:158-161) onto a cleared (H, W) screen.  Returns (screen, [white canvas, yolk canvas])."""
    p = dict(DEFAULT_PARAMS)
    p.update(params or {})
    W, H = screen_size
    screen = np.empty((H, W, 4), F)
    screen[...] = np.asarray(clear, F)
    if any(s["x"].size == 0 for s in states):
        return screen, [None, None]  # one canvas is nil: neither pass draws anything (L:1997-1999, L:2118)
    tex = particle_texture(max_radius[0], max_radius[1])
    sizes = canvas_sizes or [canvas_size(envs[w], cfgs[w]) for w in range(2)]
    canvases = [splat(states[w], envs[w], cfgs[w], colors[w], sizes[w], t, tex, p["use_instancing"]) for w in range(2)]
    composite(screen, canvases, envs, cfgs, origin, p)
    return screen, canvases
