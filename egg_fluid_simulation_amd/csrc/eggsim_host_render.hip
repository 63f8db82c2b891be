// eggsim_host_render.hip -- host side of the headless renderer (csrc/eggsim_render.hip): egg_render*, render configs
// and colours behind include/eggsim.h.  See eggsim_host.h.
#include "eggsim_host.h"

extern "C" {

// ------------------------------------------------------------------ headless renderer (eggsim_render.hip)

int egg_default_render_config(int which, egg_render_config *cfg) {  // simulation_handler_default_config.lua:22-36, 54-68
    if (!cfg || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    const float white[2][4] = {{0.961f, 0.961f, 0.953f, 1.0f}, {0.973f, 0.796f, 0.529f, 1.0f}};
    const float yolk[2][4] = {{0.969f, 0.682f, 0.141f, 1.0f}, {0.984f, 0.522f, 0.271f, 1.0f}};
    memcpy(cfg->color, which == EGG_WHITE ? white[0] : yolk[0], sizeof cfg->color);
    memcpy(cfg->outline_color, which == EGG_WHITE ? white[1] : yolk[1], sizeof cfg->outline_color);
    cfg->outline_thickness = 1;
    cfg->highlight_strength = which == EGG_WHITE ? 0 : 1;
    cfg->shadow_strength = which == EGG_WHITE ? 1 : 0;
    cfg->texture_scale = 12;
    cfg->motion_blur = 0.0003;
    return EGG_OK;
}

int egg_set_render_config(egg_handle *h, int which, const egg_render_config *cfg) {
    if (!h || !cfg || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    if (!(cfg->outline_thickness >= 0) || !(cfg->texture_scale > 0) || !std::isfinite(cfg->motion_blur) ||
        !std::isfinite(cfg->highlight_strength) || !std::isfinite(cfg->shadow_strength) || !(cfg->outline_thickness <= 256))
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_set_render_config: value out of range");
    h->render.cfg[which] = *cfg;
    // config.color is a new table now -- whatever its values: set_*_config deep-copies (L:1307-1311) --, so batches that
    // shared the old one keep it for themselves.  (Call this where the reference calls set_*_config, not once per frame.)
    for (Batch &b : h->batches) b.own_color[which] = true;
    return EGG_OK;
}

int egg_get_render_config(const egg_handle *h, int which, egg_render_config *cfg) {
    if (!h || !cfg || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    *cfg = h->render.cfg[which];
    return EGG_OK;
}

int egg_set_render_flags(egg_handle *h, int32_t use_particle_color, int32_t use_lighting) {
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    h->render.use_particle_color = use_particle_color != 0;
    h->render.use_lighting = use_lighting != 0;
    return EGG_OK;
}

static float clamp01(double v) { return (float)clampd(v, 0, 1); }

int egg_set_add_color(egg_handle *h, int64_t id, int which, double r, double g, double b, double a) {
    if (!h || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    Batch *B = find_batch(h, id);
    if (!B) return fail(h, EGG_ERR_UNKNOWN_ID, "egg_set_add_color: no batch with id `%lld`", (long long)id);
    if (std::isnan(r) || std::isnan(g) || std::isnan(b) || std::isnan(a))  // L:87-103
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.add: %s color component is not a number", which == EGG_WHITE ? "white" : "yolk");
    B->own_color[which] = true;
    if (h->render.use_particle_color) {  // add does not clamp (L:978-984)
        const float c[4] = {(float)r, (float)g, (float)b, (float)a};
        memcpy(B->pcolor[which], c, sizeof c);
    }
    return EGG_OK;
}

int egg_set_color(egg_handle *h, int64_t id, int which, double r, double g, double b, double a) {
    if (!h || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    if (std::isnan(r) || std::isnan(g) || std::isnan(b) || std::isnan(a))
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_set_color: a colour component is not a number");
    Batch *B = find_batch(h, id);
    if (!B)
        return fail(h, EGG_WARN_UNKNOWN_ID, "In SimulationHandler.%s: no batch with id `%lld`",
                    which == EGG_WHITE ? "set_white_color" : "set_egg_yolk_color", (long long)id);
    const float c[4] = {clamp01(r), clamp01(g), clamp01(b), clamp01(a)};  // _assert_color (L:300-319)
    memcpy(B->pcolor[which], c, sizeof c);
    if (!B->own_color[which]) memcpy(h->render.cfg[which].color, c, sizeof c);  // the shared table (L:49-50, L:349-350)
    return EGG_OK;
}

int egg_default_render_params(egg_render_params *p) {
    if (!p) return EGG_ERR_INVALID_ARGUMENT;
    memset(p, 0, sizeof *p);
    p->screen_w = 800;
    p->screen_h = 600;
    p->interpolation_alpha = std::numeric_limits<double>::quiet_NaN();
    p->threshold = 0.3;    // L:444
    p->smoothness = 0.01;  // L:445
    p->use_instancing = 1;
    return EGG_OK;
}

namespace {

// the density texture: simulation_handler_particle_texture.glsl drawn by _initialize_particle_texture (L:620-682)
int render_texture(egg_handle *h) {
    egg_handle::Render &R = h->render;
    const double radius = std::max(h->sys[0].cfg.max_radius, h->sys[1].cfg.max_radius) * 4;  // L:626-629, L:455
    if (radius == R.texture_radius && R.tsize > 0) return EGG_OK;
    const double padding = 3;  // L:454
    const double size_d = (radius + padding) * 2;
    if (!(radius > 0) || !(size_d <= 1024))
        return fail(h, EGG_ERR_UNSUPPORTED, "particle texture of %g px", size_d);
    const int size = (int)size_d;
    R.texture_host.assign((size_t)size * size, 0.0f);
    for (int j = 0; j < size; ++j)
        for (int i = 0; i < size; ++i) {
            // the quad of 2 radius x 2 radius px sits in the middle of the canvas; uv at the pixel centre
            const double u = (i + 0.5 - (size - 2 * radius) / 2) / (2 * radius), v = (j + 0.5 - (size - 2 * radius) / 2) / (2 * radius);
            if (!(u >= 0 && u < 1 && v >= 0 && v < 1)) continue;
            const double q = 2.0 * std::sqrt((u - 0.5) * (u - 0.5) + (v - 0.5) * (v - 0.5));  // 1 - dist
            R.texture_host[(size_t)j * size + i] = (float)std::exp((-4.0 * kPi / 3.0) * q * q);
        }
    hipStream_t st = h->sys[0].stream;
    HIP_TRY(h, R.texture.reserve((size_t)size * size, false, st));
    HIP_TRY(h, hipMemcpyAsync(R.texture.p, R.texture_host.data(), (size_t)size * size * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    R.tsize = size;
    R.texture_radius = radius;
    return EGG_OK;
}

// pass 1 of one type into R.canvas[which] (cw x ch), centred on the interpolated centroid
int render_splat(egg_handle *h, int which, const egg_environment &env, double t, int cw, int ch, int use_instancing) {
    egg_handle::Render &R = h->render;
    System &s = h->sys[which];
    hipStream_t st = h->sys[0].stream;
    int rc = upload_atoms(h, which);
    if (rc != EGG_OK) return rc;
    const size_t na = s.atoms.size();
    std::vector<float> colors(4 * na);
    for (size_t k = 0; k < na; ++k) memcpy(&colors[4 * k], h->batches[(size_t)s.atoms[k].batch].pcolor[which], 16);
    HIP_TRY(h, R.atom_color.reserve(na, false, st));
    HIP_TRY(h, hipMemcpyAsync(R.atom_color.p, colors.data(), na * 16, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipStreamSynchronize(st));  // (`colors` is pageable and goes out of scope)

    EggRenderArgs A;
    memset(&A, 0, sizeof A);
    A.x = s.x[s.cur].p;
    A.y = s.y[s.cur].p;
    A.last_x = s.x[s.cur ^ 1].p;  // positions at the start of the most recent _step (L:1795-1815)
    A.last_y = s.y[s.cur ^ 1].p;
    A.vx = s.vx[s.cur].p;
    A.vy = s.vy[s.cur].p;
    A.radius = s.radius.p;
    A.atom_offset = s.d_atom_offset.p;
    A.atom_color = R.atom_color.p;
    A.n = (int32_t)s.n;
    A.n_atoms = (int32_t)na;
    A.t = (float)t;
    // frame interpolation of the centroid in doubles like the Lua (L:2057-2058); the transform reaches the GPU as floats
    const double pcx = env.last_centroid_x * (1 - t) + env.centroid_x * t, pcy = env.last_centroid_y * (1 - t) + env.centroid_y * t;
    A.tx = (float)(cw / 2.0 - pcx);
    A.ty = (float)(ch / 2.0 - pcy);
    A.texture_scale = (float)R.cfg[which].texture_scale;
    A.motion_blur = (float)R.cfg[which].motion_blur;
    A.premultiply = use_instancing ? 0 : 1;
    A.cw = cw;
    A.ch = ch;
    A.tiles_x = (cw + EGG_RENDER_TILE - 1) / EGG_RENDER_TILE;
    A.tiles_y = (ch + EGG_RENDER_TILE - 1) / EGG_RENDER_TILE;
    const size_t nt = (size_t)A.tiles_x * A.tiles_y;
    HIP_TRY(h, R.tiles.reserve(3 * nt + 8, false, st));
    HIP_TRY(h, R.totals.reserve(4, false, st));
    A.tile_count = R.tiles.p;
    A.tile_start = R.tiles.p + nt;
    A.tile_cursor = R.tiles.p + 2 * nt + 1;
    A.totals = R.totals.p;
    A.texture = R.texture.p;
    A.tsize = R.tsize;
    HIP_TRY(h, R.canvas[which].reserve((size_t)cw * ch, false, st));
    A.canvas = R.canvas[which].p;
    HIP_TRY(h, hipMemsetAsync(A.tile_count, 0, nt * 4, st));
    const dim3 pgrid((unsigned)((s.n + 255) / 256)), pblock(256);
    hipLaunchKernelGGL(egg_render_count_kernel, pgrid, pblock, 0, st, A);
    hipLaunchKernelGGL(egg_render_scan_kernel, dim3(1), dim3(1024), 0, st, A);
    HIP_TRY(h, hipGetLastError());
    uint32_t totals[2] = {0, 0};
    HIP_TRY(h, hipMemcpyAsync(totals, A.totals, sizeof totals, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    size_t padded = 1;
    while (padded < totals[1]) padded <<= 1;
    const size_t lds = egg_render_splat_lds_bytes(R.tsize, padded);
    if (lds > h->lds_limit)
        return fail(h, EGG_ERR_UNSUPPORTED, "%u particles overlap one %d x %d px canvas tile; the renderer sorts at most %zu in LDS",
                    totals[1], EGG_RENDER_TILE, EGG_RENDER_TILE, (h->lds_limit - egg_render_splat_lds_bytes(R.tsize, 0)) / 8);
    HIP_TRY(h, R.entries.reserve((size_t)totals[0] + 8, false, st));
    A.entries = R.entries.p;
    hipLaunchKernelGGL(egg_render_fill_kernel, pgrid, pblock, 0, st, A);
    hipLaunchKernelGGL(egg_render_splat_kernel, dim3((unsigned)nt), dim3(256), lds, st, A);
    HIP_TRY(h, hipGetLastError());
    h->stats.kernel_launches += 4;
    return EGG_OK;
}

}  // namespace

int egg_render(egg_handle *h, const egg_render_params *p, float *rgba) {
    if (!h || !p) return EGG_ERR_INVALID_ARGUMENT;
    REJECT_IN_FLIGHT(h, "egg_render");
    if (p->screen_w <= 0 || p->screen_h <= 0 || (int64_t)p->screen_w * p->screen_h > ((int64_t)1 << 28))
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render: screen of %d x %d px", p->screen_w, p->screen_h);
    if (!(p->threshold >= 0 && p->threshold <= 1) || !(p->smoothness >= 0) || !std::isfinite(p->origin_x) || !std::isfinite(p->origin_y))
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render: threshold / smoothness / origin out of range");
    for (int w = 0; w < 2; ++w)
        if (p->canvas_w[w] < 0 || p->canvas_h[w] < 0 || p->canvas_w[w] > 16384 || p->canvas_h[w] > 16384)
            return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render: canvas size out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    egg_handle::Render &R = h->render;
    hipStream_t st = h->sys[0].stream;
    HIP_TRY(h, hipStreamSynchronize(h->sys[1].stream));
    const size_t npx = (size_t)p->screen_w * p->screen_h;
    HIP_TRY(h, R.screen.reserve(npx, false, st));
    hipLaunchKernelGGL(egg_render_clear_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, R.screen.p, npx,
                       make_float4(p->clear[0], p->clear[1], p->clear[2], p->clear[3]));
    HIP_TRY(h, hipGetLastError());
    h->stats.kernel_launches++;
    // the canvases exist from the first _step on and only while both types have particles (L:1936-1938, L:1997-1999, L:2118)
    const bool drawable = h->stats.steps > 0 && h->sys[0].n > 0 && h->sys[1].n > 0;
    R.canvas_valid = false;
    if (drawable) {
        const double t = std::isnan(p->interpolation_alpha) ? h->interpolation_alpha : clampd(p->interpolation_alpha, 0, 1);
        int rc = render_texture(h);
        if (rc != EGG_OK) return rc;
        egg_environment env[2];
        EggCompositeArgs C;
        memset(&C, 0, sizeof C);
        for (int w = 0; w < 2; ++w) {
            rc = egg_get_environment(h, w, &env[w]);
            if (rc != EGG_OK) return rc;
            const egg_render_config &cfg = R.cfg[w];
            int cw = p->canvas_w[w], ch = p->canvas_h[w];
            if (cw == 0 || ch == 0) {  // resize_canvas_maybe (L:1935-1975)
                const double padding = env[w].max_radius * cfg.texture_scale * (1 + std::max(1.0, env[w].max_velocity) * cfg.motion_blur);
                const double nw = std::min(std::ceil((env[w].max_x - env[w].min_x) + 2 * padding), 2560.0);
                const double nh = std::min(std::ceil((env[w].max_y - env[w].min_y) + 2 * padding), 2560.0);
                if (!(nw >= 1) || !(nh >= 1)) return fail(h, EGG_ERR_UNSUPPORTED, "egg_render: particle bounds are not finite");
                R.canvas_w[w] = std::max(R.canvas_w[w], (int)nw);
                R.canvas_h[w] = std::max(R.canvas_h[w], (int)nh);
                if (cw == 0) cw = R.canvas_w[w];
                if (ch == 0) ch = R.canvas_h[w];
            }
            rc = render_splat(h, w, env[w], t, cw, ch, p->use_instancing);
            if (rc != EGG_OK) return rc;
            // _draw_canvases places the canvas around the CURRENT centroid (L:2131-2132), pass 1 drew around the interpolated one
            R.canvas_x0[w] = env[w].centroid_x - 0.5 * cw;
            R.canvas_y0[w] = env[w].centroid_y - 0.5 * ch;
            EggCompositeLayer &L = C.layer[w];
            L.canvas = R.canvas[w].p;
            L.w = cw;
            L.h = ch;
            L.x0 = (float)(R.canvas_x0[w] - p->origin_x);
            L.y0 = (float)(R.canvas_y0[w] - p->origin_y);
            L.color = make_float4(cfg.color[0], cfg.color[1], cfg.color[2], cfg.color[3]);
            L.outline_color = make_float4(cfg.outline_color[0], cfg.outline_color[1], cfg.outline_color[2], cfg.outline_color[3]);
            L.outline_thickness = (float)cfg.outline_thickness;
            L.highlight_strength = (float)cfg.highlight_strength;
            L.shadow_strength = (float)cfg.shadow_strength;
        }
        C.screen = R.screen.p;
        C.screen_w = p->screen_w;
        C.screen_h = p->screen_h;
        C.n_layers = 2;
        C.threshold = (float)p->threshold;
        C.smoothness = (float)p->smoothness;
        C.use_particle_color = R.use_particle_color;
        C.use_lighting = R.use_lighting;
        hipLaunchKernelGGL(egg_render_composite_kernel, dim3((unsigned)((p->screen_w + 15) / 16), (unsigned)((p->screen_h + 15) / 16)),
                           dim3(256), 0, st, C);
        HIP_TRY(h, hipGetLastError());
        h->stats.kernel_launches++;
        R.canvas_valid = true;
        R.last_w[0] = C.layer[0].w;
        R.last_h[0] = C.layer[0].h;
        R.last_w[1] = C.layer[1].w;
        R.last_h[1] = C.layer[1].h;
    }
    if (rgba) HIP_TRY(h, hipMemcpyAsync(rgba, R.screen.p, npx * 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return EGG_OK;
}

int egg_render_canvas(egg_handle *h, int which, float *rgba, int64_t cap_pixels, int32_t *w, int32_t *hgt, double *x0, double *y0) {
    if (!h || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    egg_handle::Render &R = h->render;
    if (!R.canvas_valid) return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render_canvas: no canvas (egg_render has not drawn anything)");
    HIP_TRY(h, hipSetDevice(h->device));
    const int cw = R.last_w[which], ch = R.last_h[which];
    if (w) *w = cw;
    if (hgt) *hgt = ch;
    if (x0) *x0 = R.canvas_x0[which];
    if (y0) *y0 = R.canvas_y0[which];
    if (rgba) {
        if (cap_pixels < (int64_t)cw * ch)
            return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render_canvas: buffer holds %lld of %lld pixels", (long long)cap_pixels, (long long)cw * ch);
        HIP_TRY(h, hipMemcpy(rgba, R.canvas[which].p, (size_t)cw * ch * 16, hipMemcpyDeviceToHost));
    }
    return EGG_OK;
}

int egg_render_particle_texture(egg_handle *h, float *alpha, int64_t cap, int32_t *size) {
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = render_texture(h);
    if (rc != EGG_OK) return rc;
    const int n = h->render.tsize;
    if (size) *size = n;
    if (alpha) {
        if (cap < (int64_t)n * n) return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render_particle_texture: buffer too small");
        HIP_TRY(h, hipMemcpy(alpha, h->render.texture.p, (size_t)n * n * 4, hipMemcpyDeviceToHost));  // what the kernels sample
    }
    return EGG_OK;
}

}  // extern "C"
