// eggsim_host_abi.hip -- the extern "C" entry points of include/eggsim.h (the renderer's are in
// eggsim_host_render.hip); state, tiling and the step live in the other eggsim_host_*.hip units (eggsim_host.h).
//
// The reference keeps everything in Lua tables and runs `_step` on the host
// (simulation_handler.lua, "L:").  Here the particle arrays live in HBM as SoA
// (double buffered: the inactive buffer is the reference's last_update_x/y,
// L:1795-1818), the host keeps only batch bookkeeping, and a step is one
// kernel launch per particle type and tile size class.
#include "eggsim_host.h"

// ===================================================================== C ABI

extern "C" {

int egg_default_config(int which, egg_config *cfg) {  // simulation_handler_default_config.lua:1-70
    if (!cfg || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    const double base_damping = 0.1, particle_radius = 4, base_mass = 1;
    cfg->damping = base_damping;
    cfg->follow_strength = 1 - 0.004;
    cfg->collision_overlap_factor = 2;
    cfg->min_mass = base_mass;
    cfg->min_radius = particle_radius;
    cfg->max_radius = particle_radius;
    if (which == EGG_WHITE) {
        cfg->cohesion_strength = 1 - 0.2;
        cfg->cohesion_interaction_distance_factor = 2;
        cfg->collision_strength = 1 - 0.0025;
        cfg->max_mass = base_mass * 1.8;
    } else {
        cfg->cohesion_strength = 1 - 0.002;
        cfg->cohesion_interaction_distance_factor = 3;
        cfg->collision_strength = 1 - 0.001;
        cfg->max_mass = base_mass * 1.35;
    }
    cfg->max_collision_fraction = 0.05;   // L:448
    cfg->mass_distribution_variance = 4;  // L:447
    cfg->eps = 1e-8;                      // math.lua:2
    return EGG_OK;
}

const char *egg_last_error(const egg_handle *h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int egg_create(const egg_config *white, const egg_config *yolk, int device, egg_handle **out) {
    if (!white || !out) return fail(nullptr, EGG_ERR_INVALID_ARGUMENT, "egg_create: null argument");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, EGG_ERR_NO_DEVICE, "no HIP device available (%s); libeggsim has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count)
        return fail(nullptr, EGG_ERR_NO_DEVICE, "device ordinal %d out of range (0..%d)", device, count - 1);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, EGG_ERR_DEVICE, "hipSetDevice: %s", hipGetErrorString(e));
    egg_handle *h = new egg_handle();
    h->device = device;
    (void)hipGetDeviceProperties(&h->prop, device);
    h->sys[0].cfg = *white;
    h->sys[1].cfg = yolk ? *yolk : *white;
    for (int w = 0; w < 2; ++w) (void)egg_default_render_config(w, &h->render.cfg[w]);
    for (int w = 0; w < 2; ++w)
        if (!(h->sys[w].cfg.eps >= 0x1p-300 && h->sys[w].cfg.eps <= 1.0)) {
            delete h;
            return fail(nullptr, EGG_ERR_INVALID_ARGUMENT, "egg_config.eps must be in [2^-300, 1]");
        }
    // a workgroup gets 64 KiB of dynamic LDS by default; ask for as much of the CU's 160 KiB as the
    // runtime grants for this kernel
    for (size_t want = kLdsMax; want > h->lds_limit; want -= 16 * 1024) {
        e = hipFuncSetAttribute((const void *)egg_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)egg_step_kernel_gl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)egg_step_kernel_occ, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)egg_step_kernel_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)egg_step_kernel_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)egg_step_kernel_multi_occ, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)egg_step_kernel_multi_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)egg_step_kernel_mg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)egg_step_kernel_gl_mg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess)
            for (const void *f : {(const void *)egg_pk_lists_fresh_kernel, (const void *)egg_pk_lists_stale_kernel,
                                  (const void *)egg_pk_exec_kernel, (const void *)egg_pk_exec_chain_kernel,
                                  (const void *)egg_pk_levexec_kernel,
                                  (const void *)egg_pk_sort_kernel, (const void *)egg_pk_levels_mr16_kernel,
                                  (const void *)egg_pk_levels_ooo_kernel, (const void *)egg_render_splat_kernel})
                if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
        if (e == hipSuccess) {
            h->lds_limit = want;
            break;
        }
    }
    (void)hipGetLastError();
    if (const char *e_pk = getenv("EGGSIM_PACKED")) h->opt_packed = atoi(e_pk);  // developer / test override of EGG_OPT_PACKED
    if (const char *e_tn = getenv("EGGSIM_TUNE")) h->opt_tune = atoi(e_tn);
    if (const char *e_lw = getenv("EGGSIM_LEVEL_WALK")) h->opt_level_walk = std::min(2, std::max(0, atoi(e_lw)));  // developer / test override of EGG_OPT_LEVEL_WALK
    {
        // egg_pk_levels_ooo_kernel ranks the entries of a pair stream with one LDS atomic add per batch and relies on the
        // hardware serving same-address lanes of ONE instruction in ascending lane order.  That is what gfx950 does, but
        // no manual promises it: probe it here (256 workgroups x 64 trials of pseudo-random keys, ~20 us) and use the
        // in-order walk everywhere if a single lane disagrees.
        unsigned long long *bad = nullptr, host_bad = ~0ull;
        if (hipMalloc((void **)&bad, sizeof *bad) == hipSuccess) {
            if (hipMemset(bad, 0, sizeof *bad) == hipSuccess) {
                hipLaunchKernelGGL(egg_pk_probe_lds_order_kernel, dim3(256), dim3(64), 0, 0, 64, bad);
                if (hipMemcpy(&host_bad, bad, sizeof host_bad, hipMemcpyDeviceToHost) != hipSuccess) host_bad = ~0ull;
            }
            (void)hipFree(bad);
        }
        const hipError_t pe = hipGetLastError();
        h->lds_lane_ordered = host_bad == 0;
        if (h->simd_claims.reserve(4096, false, 0) != hipSuccess || hipMemset(h->simd_claims.p, 0, 4096 * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            h->lds_lane_ordered = false;  // (without the table the fused pass is not used either: it is chosen with the out-of-order walk)
        }
        if (getenv("EGGSIM_DEBUG")) fprintf(stderr, "eggsim: LDS atomic lane-order probe: %llu mismatches (%s)\n", host_bad, hipGetErrorString(pe));
    }
    if (const char *e_gp = getenv("EGGSIM_GROUP_PARTICLES")) h->opt_group_particles = std::max(1, atoi(e_gp));
    for (int w = 0; w < 2; ++w) {
        System &s = h->sys[w];
        s.margin = h->opt_margin;
        bool ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreate(&s.ev0) == hipSuccess && hipEventCreate(&s.ev1) == hipSuccess &&
                  s.stage_down.reserve(2 * kStatInts * sizeof(int32_t)) == hipSuccess && reserve_out(h, s, 0) == EGG_OK;
        if (!ok) {
            egg_destroy(h);
            return fail(nullptr, EGG_ERR_DEVICE, "device resource allocation failed");
        }
        memset(s.stage_down.p, 0, 2 * kStatInts * sizeof(int32_t));
        s.h_status = (EggStatus *)s.stage_down.p;
    }
    // the reference primes its environments with _step(0, 1, 1) on zero particles (L:562); the
    // observable effect is that mass/radius of particles added later are not re-derived
    for (int w = 0; w < 2; ++w) {
        System &s = h->sys[w];
        s.has_env = true;
        s.env_min_mass = s.cfg.min_mass;
        s.env_max_mass = s.cfg.max_mass;
        s.env_min_radius = s.cfg.min_radius;
        s.env_max_radius = s.cfg.max_radius;
    }
    *out = h;
    return EGG_OK;
}

void egg_destroy(egg_handle *h) {
    if (!h) return;
    if (const char *e = getenv("EGGSIM_HOST_PROFILE"); e && atoi(e) && h->stats.retiles > 0) {
        fprintf(stderr, "eggsim retile ms per call (%lld calls):", (long long)h->stats.retiles);
        for (double v : g_retile_ms) fprintf(stderr, " %.4f", v / (double)h->stats.retiles);
        fprintf(stderr, "\n");
    }
    (void)hipSetDevice(h->device);
    for (int w = 0; w < 2; ++w) {
        System &s = h->sys[w];
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.ev0) (void)hipEventDestroy(s.ev0);
        if (s.ev1) (void)hipEventDestroy(s.ev1);
        for (auto &ps : s.pk_stamps) {
            (void)hipEventDestroy(ps.a);
            (void)hipEventDestroy(ps.b);
        }
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete h;
}

int egg_set_config(egg_handle *h, int which, const egg_config *cfg) {
    if (!h || !cfg || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    if (!(cfg->eps >= 0x1p-300 && cfg->eps <= 1.0)) return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_config.eps must be in [2^-300, 1]");
    REJECT_IN_FLIGHT(h, "egg_set_config");
    h->sys[which].cfg = *cfg;
    return EGG_OK;
}

int egg_get_config(const egg_handle *h, int which, egg_config *cfg) {
    if (!h || !cfg || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    *cfg = h->sys[which].cfg;
    return EGG_OK;
}

static int add_many_impl(egg_handle *h, int64_t n, const double *xs, const double *ys, double white_radius,
                         double yolk_radius, int64_t white_n, int64_t yolk_n, const int64_t *keys, int64_t *out_ids);

int egg_add_many(egg_handle *h, int64_t n, const double *xs, const double *ys, double white_radius,
                 double yolk_radius, int64_t white_n, int64_t yolk_n, int64_t *out_ids) {
    return add_many_impl(h, n, xs, ys, white_radius, yolk_radius, white_n, yolk_n, nullptr, out_ids);
}

int egg_add_many_keyed(egg_handle *h, int64_t n, const double *xs, const double *ys, double white_radius,
                       double yolk_radius, int64_t white_n, int64_t yolk_n, const int64_t *keys, int64_t *out_ids) {
    if (!h || !keys) return EGG_ERR_INVALID_ARGUMENT;
    int64_t last = h->next_key - 1;
    for (int64_t k = 0; k < n; ++k) {
        if (keys[k] <= last)
            return fail(h, EGG_ERR_INVALID_ARGUMENT,
                        "egg_add_many_keyed: keys must ascend and exceed every key in the handler (use egg_import_batch "
                        "to insert in the middle)");
        last = keys[k];
    }
    return add_many_impl(h, n, xs, ys, white_radius, yolk_radius, white_n, yolk_n, keys, out_ids);
}

static int add_many_impl(egg_handle *h, int64_t n, const double *xs, const double *ys, double white_radius,
                         double yolk_radius, int64_t white_n, int64_t yolk_n, const int64_t *keys, int64_t *out_ids) {
    if (!h || n < 0 || (n > 0 && (!xs || !ys))) return EGG_ERR_INVALID_ARGUMENT;
    REJECT_IN_FLIGHT(h, "egg_add");
    (void)hipSetDevice(h->device);
    const egg_config &wc = h->sys[0].cfg, &yc = h->sys[1].cfg;
    // L:33-58
    double white_particle_radius = mixd(wc.min_radius, wc.max_radius, 0.5);
    double yolk_particle_radius = mixd(yc.min_radius, yc.max_radius, 0.5);
    if (std::isnan(white_radius)) white_radius = white_particle_radius * 15;
    if (std::isnan(yolk_radius)) yolk_radius = white_radius * (10.0 / 50);
    // only the sentinel means "not given" (L:52-58); an explicit 0 or negative count reaches the `<= 1` check below
    if (white_n == EGG_DEFAULT_COUNT)
        white_n = (int64_t)std::ceil((kPi * (white_radius * white_radius)) /
                                     (kPi * (white_particle_radius * white_particle_radius)));
    if (yolk_n == EGG_DEFAULT_COUNT)
        yolk_n = (int64_t)std::ceil((kPi * (yolk_radius * yolk_radius)) /
                                    (kPi * (yolk_particle_radius * yolk_particle_radius)));
    // L:71-85
    if (!(white_radius > 0)) return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.add: white radius cannot be 0 or negative");
    if (!(yolk_radius > 0)) return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.add: yolk radius cannot be 0 or negative");
    if (white_n <= 1) return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.add: white particle count cannot be 1 or negative");
    if (yolk_n <= 1) return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.add: yolk particle count cannot be 1 or negative");
    for (int64_t k = 0; k < n; ++k)
        if (!std::isfinite(xs[k]) || !std::isfinite(ys[k]))
            return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.add: position is not a finite number");
    if (h->sys[0].n + white_n * n > 2000000000ll || h->sys[1].n + yolk_n * n > 2000000000ll)
        return fail(h, EGG_ERR_UNSUPPORTED, "more than 2e9 particles of one type");
    if (n == 0) return EGG_OK;

    ParticleTemplate tw, ty;
    make_template(wc, white_radius, white_n, tw);
    make_template(yc, yolk_radius, yolk_n, ty);
    int rc = append_particles(h, h->sys[0], tw, n, xs, ys);
    if (rc != EGG_OK) return rc;
    rc = append_particles(h, h->sys[1], ty, n, xs, ys);
    if (rc != EGG_OK) return rc;
    for (int64_t k = 0; k < n; ++k) {
        Batch b;
        b.id = (int64_t)h->batches.size() + 1;
        b.alive = true;
        b.target_x = xs[k];
        b.target_y = ys[k];
        b.white_radius = white_radius;
        b.yolk_radius = yolk_radius;
        b.n[0] = white_n;
        b.n[1] = yolk_n;
        b.key = keys ? keys[k] : h->next_key;
        h->next_key = b.key + 1;
        if (h->render.use_particle_color)  // L:978-990: the batch colour (here: the config's) or plain white
            for (int w = 0; w < 2; ++w) memcpy(b.pcolor[w], h->render.cfg[w].color, sizeof b.pcolor[w]);
        h->order.push_back((int32_t)h->batches.size());
        h->batches.push_back(b);
        h->n_alive++;
        if (out_ids) out_ids[k] = b.id;
    }
    if (white_n < 10 || yolk_n < 5) {  // L:114-120: warning only
        fail(h, EGG_WARN_FEW_PARTICLES,
             "In SimulationHandler.add: only %lld white / %lld yolk particles will be created; consider "
             "increasing the radius or decreasing the particle size",
             (long long)white_n, (long long)yolk_n);
        return EGG_WARN_FEW_PARTICLES;
    }
    return EGG_OK;
}

int egg_add(egg_handle *h, double x, double y, double white_radius, double yolk_radius, int64_t white_n,
            int64_t yolk_n, int64_t *out_id) {
    return egg_add_many(h, 1, &x, &y, white_radius, yolk_radius, white_n, yolk_n, out_id);
}

int egg_remove(egg_handle *h, int64_t id) {  // L:140-155, L:1037-1106
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    REJECT_IN_FLIGHT(h, "egg_remove");
    Batch *b = find_batch(h, id);
    if (!b) return fail(h, EGG_WARN_UNKNOWN_ID, "In SimulationHandler.remove: no batch with id `%lld`", (long long)id);
    (void)hipSetDevice(h->device);
    for (int w = 0; w < 2; ++w) {
        System &s = h->sys[w];
        int rc = upload_atoms(h, w);  // make sure s.atoms reflects the current layout
        if (rc != EGG_OK) return rc;
        const Atom *at = nullptr;
        for (const Atom &a : s.atoms)
            if (a.batch == (int32_t)(id - 1)) at = &a;
        if (!at) return fail(h, EGG_ERR_INTERNAL, "egg_remove: atom not found");
        s.aabb_on_device = false;
        // order-preserving compaction: shift the tail down over the removed range
        const size_t from = (size_t)at->offset + (size_t)at->count, tail = (size_t)s.n - from;
        if (tail) {
            DevBuf<double> tmp;
            HIP_TRY(h, tmp.reserve(tail, false, s.stream));
            double *arrays[] = {s.x[0].p, s.x[1].p, s.y[0].p, s.y[1].p, s.vx[0].p, s.vx[1].p, s.vy[0].p,
                                s.vy[1].p, s.inv_mass.p, s.radius.p, s.mass_t.p};
            for (double *a : arrays) {
                HIP_TRY(h, hipMemcpyAsync(tmp.p, a + from, tail * 8, hipMemcpyDeviceToDevice, s.stream));
                HIP_TRY(h, hipMemcpyAsync(a + at->offset, tmp.p, tail * 8, hipMemcpyDeviceToDevice, s.stream));
            }
            HIP_TRY(h, hipStreamSynchronize(s.stream));
        }
        s.n -= at->count;
        s.atoms_dirty = s.tiling_dirty = true;
        s.aabb_valid = false;
    }
    b->alive = false;
    h->n_alive--;
    h->order.erase(std::remove(h->order.begin(), h->order.end(), (int32_t)(id - 1)), h->order.end());
    return EGG_OK;
}

int egg_set_target(egg_handle *h, int64_t id, double x, double y) {  // L:254-264
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    REJECT_IN_FLIGHT(h, "egg_set_target");
    Batch *b = find_batch(h, id);
    if (!b)
        return fail(h, EGG_WARN_UNKNOWN_ID, "In SimulationHandler.set_target_position: no batch with id `%lld`",
                    (long long)id);
    const bool moved = b->target_x != x || b->target_y != y;
    b->target_x = x;
    b->target_y = y;
    if (moved) {
        for (int w = 0; w < 2; ++w) {
            h->sys[w].targets_dirty = true;
            h->sys[w].claims_stale = true;  // claims are swept towards the target
        }
    }
    return EGG_OK;
}

int egg_set_targets_many(egg_handle *h, int64_t n, const int64_t *ids, const double *xs, const double *ys) {
    if (!h || n < 0 || (n > 0 && (!ids || !xs || !ys))) return EGG_ERR_INVALID_ARGUMENT;
    int rc = EGG_OK;
    for (int64_t k = 0; k < n; ++k) {
        int r = egg_set_target(h, ids[k], xs[k], ys[k]);
        if (r != EGG_OK) rc = r;
    }
    return rc;
}

int egg_get_target(const egg_handle *h, int64_t id, double *x, double *y) {  // L:268-278
    if (!h || !x || !y) return EGG_ERR_INVALID_ARGUMENT;
    const Batch *b = find_batch(h, id);
    if (!b)
        return fail(const_cast<egg_handle *>(h), EGG_ERR_UNKNOWN_ID,
                    "In SimulationHandler.get_target_position: no batch with id `%lld`", (long long)id);
    *x = b->target_x;
    *y = b->target_y;
    return EGG_OK;
}

int egg_step(egg_handle *h, double delta, int32_t n_substeps, int32_t n_collision_steps) {
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    if (n_substeps < 1 || n_collision_steps < 1 || std::isnan(delta))
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_step: invalid arguments");
    REJECT_IN_FLIGHT(h, "egg_step");
    (void)hipSetDevice(h->device);
    return do_step(h, delta, n_substeps, n_collision_steps);
}

int egg_update(egg_handle *h, double delta, double step_delta, int32_t n_substeps, int32_t n_collision_steps,
               int32_t *out_n_steps) {  // L:168-222
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    if (out_n_steps) *out_n_steps = 0;
    REJECT_IN_FLIGHT(h, "egg_update");
    if (std::isnan(delta)) return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.update: `delta` is not a number");
    if (step_delta < 0 || std::isnan(step_delta))
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.update: `step_delta` is not a number > 0");
    if (step_delta == 0)  // the reference would loop forever here (elapsed >= 0 always holds)
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.update: `step_delta` is 0");
    if (n_substeps < 1)
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.update: `n_substeps` is not a number > 0");
    if (n_collision_steps < 1)
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.update: `n_collision_steps` is not a number > 0");
    (void)hipSetDevice(h->device);
    h->elapsed = h->elapsed + delta;  // L:200
    const double step = step_delta;
    int n_steps = 0;
    const double max_n_steps = std::max(4.0, 4 * std::ceil((1.0 / 60) / step_delta));  // L:203
    while (h->elapsed >= step) {
        int rc = do_step(h, step, n_substeps, n_collision_steps);
        if (rc != EGG_OK) return rc;
        h->elapsed = h->elapsed - step;
        n_steps = n_steps + 1;
        if (n_steps > max_n_steps) {  // L:208-213: death-spiral guard
            h->elapsed = 0;
            break;
        }
    }
    h->interpolation_alpha = clampd(h->elapsed / step, 0, 1);  // L:216
    if (out_n_steps) *out_n_steps = n_steps;
    return EGG_OK;
}

int egg_prepare_step(egg_handle *h, double step_delta, int32_t n_substeps, int32_t n_collision_steps) {
    if (!h || n_substeps < 1 || n_collision_steps < 1 || !(step_delta >= 0)) return EGG_ERR_INVALID_ARGUMENT;
    REJECT_IN_FLIGHT(h, "egg_prepare_step");
    (void)hipSetDevice(h->device);
    return do_step(h, step_delta, n_substeps, n_collision_steps, kPrepare);
}

int egg_step_begin(egg_handle *h, double delta, int32_t n_substeps, int32_t n_collision_steps) {
    if (!h || n_substeps < 1 || n_collision_steps < 1 || std::isnan(delta)) return EGG_ERR_INVALID_ARGUMENT;
    if (h->in_flight) return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_step_begin: a step is already in flight");
    (void)hipSetDevice(h->device);
    int rc = do_step(h, delta, n_substeps, n_collision_steps, kBegin);
    if (rc == EGG_OK) {
        h->in_flight = true;
        h->flight_delta = delta;
        h->flight_s = n_substeps;
        h->flight_c = n_collision_steps;
    }
    return rc;
}

int egg_step_end(egg_handle *h, int32_t commit) {
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    if (!h->in_flight) return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_step_end: no step in flight");
    (void)hipSetDevice(h->device);
    h->in_flight = false;
    if (!commit) {
        // discard: the launches wrote the inactive buffers only; wait for them and forget
        for (int w = 0; w < 2; ++w) {
            HIP_TRY(h, hipStreamSynchronize(h->sys[w].stream));
            h->sys[w].aabb_on_device = false;
            h->sys[w].out_copied = false;
        }
        return EGG_OK;
    }
    return do_step(h, h->flight_delta, h->flight_s, h->flight_c, kEnd);
}

int egg_step_peek_visits(egg_handle *h, int64_t max_pass_visits[2], double budget[2]) {
    if (!h || !max_pass_visits || !budget) return EGG_ERR_INVALID_ARGUMENT;
    if (!h->in_flight) return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_step_peek_visits: no step in flight");
    (void)hipSetDevice(h->device);
    const double sub_delta = std::max(h->flight_delta / h->flight_s, h->sys[0].cfg.eps);
    for (int w = 0; w < 2; ++w) {
        System &s = h->sys[w];
        max_pass_visits[w] = 0;
        budget[w] = make_env(s.cfg, sub_delta, h->budget_particles[w] >= 0 ? h->budget_particles[w] : s.n).budget;
        if (s.n == 0 || s.classes.empty()) continue;
        HIP_TRY(h, wait_step(s.wait_stream ? s.wait_stream : s.stream));  // the status block was copied behind the kernels
        const int np = std::min(h->flight_s * h->flight_c, EGG_MAX_PASSES);
        for (int p = 0; p < np; ++p) max_pass_visits[w] = std::max(max_pass_visits[w], (int64_t)s.h_status->visits[p]);
    }
    return EGG_OK;
}

int egg_synchronize(egg_handle *h) {
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(h->device);
    for (int w = 0; w < 2; ++w) HIP_TRY(h, hipStreamSynchronize(h->sys[w].stream));
    return EGG_OK;
}

int egg_get_positions_many(egg_handle *h, int64_t n, const int64_t *ids, double *xs, double *ys) {
    if (!h || n < 0 || (n > 0 && (!ids || !xs || !ys))) return EGG_ERR_INVALID_ARGUMENT;
    if (n == 0) return EGG_OK;
    (void)hipSetDevice(h->device);
    for (int w = 0; w < 2; ++w) {
        int rc = upload_atoms(h, w);
        if (rc != EGG_OK) return rc;
    }
    // atom of each live batch: both types list live batches in the same (creation) order
    std::vector<int32_t> atom_of_batch(h->batches.size(), -1);
    for (size_t k = 0; k < h->sys[0].atoms.size(); ++k) atom_of_batch[(size_t)h->sys[0].atoms[k].batch] = (int32_t)k;
    std::vector<int32_t> wo((size_t)n), wc((size_t)n), yo((size_t)n), yc((size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        const Batch *b = find_batch(h, ids[k]);
        if (!b)
            return fail(h, EGG_ERR_UNKNOWN_ID, "In SimulationHandler.get_position: no batch with id `%lld`",
                        (long long)ids[k]);
        int32_t a = atom_of_batch[(size_t)ids[k] - 1];
        wo[(size_t)k] = h->sys[0].atoms[(size_t)a].offset;
        wc[(size_t)k] = h->sys[0].atoms[(size_t)a].count;
        yo[(size_t)k] = h->sys[1].atoms[(size_t)a].offset;
        yc[(size_t)k] = h->sys[1].atoms[(size_t)a].count;
    }
    System &W = h->sys[0], &Y = h->sys[1];
    DevBuf<int32_t> d_idx;
    DevBuf<double> d_out;
    HIP_TRY(h, d_idx.reserve((size_t)n * 4, false, W.stream));
    HIP_TRY(h, d_out.reserve((size_t)n * 2, false, W.stream));
    HIP_TRY(h, hipMemcpy(d_idx.p, wo.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(d_idx.p + n, wc.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(d_idx.p + 2 * n, yo.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(d_idx.p + 3 * n, yc.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(h, hipStreamSynchronize(Y.stream));
    const int threads = 64;
    hipLaunchKernelGGL(egg_centroid_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, W.stream,
                       W.x[W.cur].p, W.y[W.cur].p, Y.x[Y.cur].p, Y.y[Y.cur].p, d_idx.p, d_idx.p + n, d_idx.p + 2 * n,
                       d_idx.p + 3 * n, (int)n, d_out.p, d_out.p + n);
    HIP_TRY(h, hipGetLastError());
    h->stats.kernel_launches++;
    HIP_TRY(h, hipMemcpyAsync(xs, d_out.p, (size_t)n * 8, hipMemcpyDeviceToHost, W.stream));
    HIP_TRY(h, hipMemcpyAsync(ys, d_out.p + n, (size_t)n * 8, hipMemcpyDeviceToHost, W.stream));
    HIP_TRY(h, hipStreamSynchronize(W.stream));
    return EGG_OK;
}

int egg_get_position(egg_handle *h, int64_t id, double *x, double *y) {
    return egg_get_positions_many(h, 1, &id, x, y);
}

int egg_get_bounds_many(egg_handle *h, int64_t n, const int64_t *ids, double *lo_x, double *lo_y, double *hi_x,
                        double *hi_y) {
    if (!h || n < 0 || (n > 0 && (!ids || !lo_x || !lo_y || !hi_x || !hi_y))) return EGG_ERR_INVALID_ARGUMENT;
    if (n == 0) return EGG_OK;
    (void)hipSetDevice(h->device);
    double cell[2];
    for (int w = 0; w < 2; ++w) {
        System &s = h->sys[w];
        int rc = upload_atoms(h, w);
        if (rc != EGG_OK) return rc;
        cell[w] = cell_size_of(s.cfg);
        const size_t na = s.atoms.size();
        if (!s.tiling_dirty && s.h_claim.size() == na) continue;  // the claims of the formed tiles answer the query below
        if (h->in_flight && !s.aabb_valid && na)  // the running step kernel owns the device-side box buffer
            return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_get_bounds_many: cell boxes are not available while a step is in flight");
        if (!s.aabb_valid && s.aabb_on_device && s.tiled_cell_size == cell[w]) {
            rc = fetch_end_aabb(h, s);
            if (rc != EGG_OK) return rc;
        }
        if (!s.aabb_valid && na) {
            hipLaunchKernelGGL(egg_atom_bounds_kernel, dim3((unsigned)na), dim3(EGG_WAVE), 0, s.stream, s.x[s.cur].p,
                               s.y[s.cur].p, s.d_atom_offset.p, s.d_atom_count.p, (int)na, cell[w], d_aabb(s));
            HIP_TRY(h, hipGetLastError());
            h->stats.kernel_launches++;
            s.aabb.resize(na);
            HIP_TRY(h, hipMemcpyAsync(s.aabb.data(), d_aabb(s), na * sizeof(Box), hipMemcpyDeviceToHost, s.stream));
            HIP_TRY(h, hipStreamSynchronize(s.stream));
            // these are the cells of the CURRENT positions at the CURRENT cell size
            s.aabb_valid = true;
            if (s.tiled_cell_size != cell[w]) s.tiling_dirty = true;
            s.tiled_cell_size = cell[w];
        }
    }
    std::vector<int32_t> atom_of_batch(h->batches.size(), -1);
    for (size_t k = 0; k < h->sys[0].atoms.size(); ++k) atom_of_batch[(size_t)h->sys[0].atoms[k].batch] = (int32_t)k;
    for (int64_t k = 0; k < n; ++k) {
        if (!find_batch(h, ids[k]))
            return fail(h, EGG_ERR_UNKNOWN_ID, "egg_get_bounds_many: no batch with id `%lld`", (long long)ids[k]);
        const int32_t a = atom_of_batch[(size_t)ids[k] - 1];
        // the claim of the upcoming step when the tiles are current (egg_prepare_step), else the occupied cells
        const System &sw = h->sys[0], &sy = h->sys[1];
        const bool cw = !sw.tiling_dirty && sw.h_claim.size() == sw.atoms.size();
        const bool cy = !sy.tiling_dirty && sy.h_claim.size() == sy.atoms.size();
        const Box &bw = cw ? sw.h_claim[(size_t)a] : sw.aabb[(size_t)a], &by = cy ? sy.h_claim[(size_t)a] : sy.aabb[(size_t)a];
        lo_x[k] = std::min(bw.lo_x * cell[0], by.lo_x * cell[1]);
        lo_y[k] = std::min(bw.lo_y * cell[0], by.lo_y * cell[1]);
        hi_x[k] = std::max((bw.hi_x + 1.0) * cell[0], (by.hi_x + 1.0) * cell[1]);
        hi_y[k] = std::max((bw.hi_y + 1.0) * cell[0], (by.hi_y + 1.0) * cell[1]);
    }
    return EGG_OK;
}

int egg_get_claims_many(egg_handle *h, int64_t n, const int64_t *ids, double *boxes, double *cell_sizes) {
    if (!h || n < 0 || (n > 0 && (!ids || !boxes))) return EGG_ERR_INVALID_ARGUMENT;
    std::vector<double> lx((size_t)n), ly((size_t)n), hx((size_t)n), hy((size_t)n);
    // reuse the bounds path to make sure boxes / claims are current
    int rc = egg_get_bounds_many(h, n, ids, lx.data(), ly.data(), hx.data(), hy.data());
    if (rc != EGG_OK) return rc;
    std::vector<int32_t> atom_of_batch(h->batches.size(), -1);
    for (size_t k = 0; k < h->sys[0].atoms.size(); ++k) atom_of_batch[(size_t)h->sys[0].atoms[k].batch] = (int32_t)k;
    for (int w = 0; w < 2; ++w) {
        const System &s = h->sys[w];
        const double cell = cell_size_of(s.cfg);
        if (cell_sizes) cell_sizes[w] = cell;
        const bool claims = !s.tiling_dirty && s.h_claim.size() == s.atoms.size();
        for (int64_t k = 0; k < n; ++k) {
            const int32_t a = atom_of_batch[(size_t)ids[k] - 1];
            const Box &b = claims ? s.h_claim[(size_t)a] : s.aabb[(size_t)a];
            double *o = boxes + 8 * k + 4 * w;
            o[0] = b.lo_x * cell;
            o[1] = b.lo_y * cell;
            o[2] = (b.hi_x + 1.0) * cell;
            o[3] = (b.hi_y + 1.0) * cell;
        }
    }
    return EGG_OK;
}

int egg_get_n_particles(const egg_handle *h, int64_t id, int64_t *n_white, int64_t *n_yolk) {  // L:409-419
    if (!h || !n_white || !n_yolk) return EGG_ERR_INVALID_ARGUMENT;
    if (id < 0) {
        *n_white = h->sys[0].n;
        *n_yolk = h->sys[1].n;
        return EGG_OK;
    }
    const Batch *b = find_batch(h, id);
    if (!b)
        return fail(const_cast<egg_handle *>(h), EGG_ERR_UNKNOWN_ID,
                    "In SimulationHandler:get_n_particles: no batch with id `%lld`", (long long)id);
    *n_white = b->n[0];
    *n_yolk = b->n[1];
    return EGG_OK;
}

int egg_list_ids(const egg_handle *h, int64_t cap, int64_t *ids, int64_t *n) {  // L:399-405
    if (!h || !n) return EGG_ERR_INVALID_ARGUMENT;
    int64_t k = 0;
    for (int32_t bi : h->order) {
        const Batch &b = h->batches[(size_t)bi];
        if (ids && k < cap) ids[k] = b.id;
        ++k;
    }
    *n = k;
    return EGG_OK;
}

int egg_get_elapsed(const egg_handle *h, double *elapsed, double *interpolation_alpha) {
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    if (elapsed) *elapsed = h->elapsed;
    if (interpolation_alpha) *interpolation_alpha = h->interpolation_alpha;
    return EGG_OK;
}

int egg_download_particles(egg_handle *h, int which, int field, double *dst, int64_t cap) {
    if (!h || !dst || (which != EGG_WHITE && which != EGG_YOLK) || field < 0 || field >= EGG_N_FIELDS)
        return EGG_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(h->device);
    System &s = h->sys[which];
    if (cap < s.n) return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_download_particles: buffer holds %lld of %lld particles",
                               (long long)cap, (long long)s.n);
    if (s.n == 0) return EGG_OK;
    const double *src = nullptr;
    switch (field) {
        case EGG_FIELD_X: src = s.x[s.cur].p; break;
        case EGG_FIELD_Y: src = s.y[s.cur].p; break;
        case EGG_FIELD_VX: src = s.vx[s.cur].p; break;
        case EGG_FIELD_VY: src = s.vy[s.cur].p; break;
        case EGG_FIELD_LAST_X: src = s.x[s.cur ^ 1].p; break;  // positions at the start of the last _step
        case EGG_FIELD_LAST_Y: src = s.y[s.cur ^ 1].p; break;
        case EGG_FIELD_RADIUS: src = s.radius.p; break;
        case EGG_FIELD_INV_MASS: src = s.inv_mass.p; break;
        case EGG_FIELD_MASS_T: src = s.mass_t.p; break;
        default: break;
    }
    if (field == EGG_FIELD_BATCH_ID) {
        int rc = upload_atoms(h, which);
        if (rc != EGG_OK) return rc;
        for (const Atom &a : s.atoms)
            for (int32_t k = 0; k < a.count; ++k) dst[a.offset + k] = (double)h->batches[(size_t)a.batch].id;
        return EGG_OK;
    }
    HIP_TRY(h, hipMemcpyAsync(dst, src, (size_t)s.n * 8, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(h, hipStreamSynchronize(s.stream));
    return EGG_OK;
}

// ---- multi-GPU hand-over: a batch leaves one handler and enters another with its full state ----

static const int kExportFields = 9;  // x y vx vy last_x last_y inv_mass radius mass_t

int egg_export_batch(egg_handle *h, int64_t id, egg_batch_info *info, double *white_state, double *yolk_state) {
    if (!h || !info) return EGG_ERR_INVALID_ARGUMENT;
    const Batch *b = find_batch(h, id);
    if (!b) return fail(h, EGG_ERR_UNKNOWN_ID, "egg_export_batch: no batch with id `%lld`", (long long)id);
    (void)hipSetDevice(h->device);
    info->key = b->key;
    info->target_x = b->target_x;
    info->target_y = b->target_y;
    info->white_radius = b->white_radius;
    info->yolk_radius = b->yolk_radius;
    info->n_white = b->n[0];
    info->n_yolk = b->n[1];
    for (int w = 0; w < 2; ++w) {
        double *dst = w == 0 ? white_state : yolk_state;
        if (!dst) continue;
        System &s = h->sys[w];
        int rc = upload_atoms(h, w);
        if (rc != EGG_OK) return rc;
        const Atom *at = nullptr;
        for (const Atom &a : s.atoms)
            if (a.batch == (int32_t)(id - 1)) at = &a;
        if (!at) return fail(h, EGG_ERR_INTERNAL, "egg_export_batch: atom not found");
        const double *src[kExportFields] = {s.x[s.cur].p,     s.y[s.cur].p,     s.vx[s.cur].p, s.vy[s.cur].p, s.x[s.cur ^ 1].p,
                                            s.y[s.cur ^ 1].p, s.inv_mass.p, s.radius.p,    s.mass_t.p};
        for (int f = 0; f < kExportFields; ++f)
            HIP_TRY(h, hipMemcpyAsync(dst + (size_t)f * at->count, src[f] + at->offset, (size_t)at->count * 8,
                                      hipMemcpyDefault, s.stream));  // (dst: host or device memory)
        HIP_TRY(h, hipStreamSynchronize(s.stream));
    }
    return EGG_OK;
}

int egg_import_batch(egg_handle *h, const egg_batch_info *info, const double *white_state, const double *yolk_state,
                     int64_t *out_id) {
    if (!h || !info || !white_state || !yolk_state || info->n_white < 1 || info->n_yolk < 1)
        return EGG_ERR_INVALID_ARGUMENT;
    REJECT_IN_FLIGHT(h, "egg_import_batch");
    (void)hipSetDevice(h->device);
    // position in the layout order
    size_t pos = 0;
    while (pos < h->order.size() && h->batches[(size_t)h->order[pos]].key < info->key) ++pos;
    if (pos < h->order.size() && h->batches[(size_t)h->order[pos]].key == info->key)
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_import_batch: key %lld is already present", (long long)info->key);
    for (int w = 0; w < 2; ++w) {
        System &s = h->sys[w];
        const int64_t cnt = w == 0 ? info->n_white : info->n_yolk;
        const double *src = w == 0 ? white_state : yolk_state;
        int64_t at = 0;
        for (size_t k = 0; k < pos; ++k) at += h->batches[(size_t)h->order[k]].n[w];
        int rc = reserve_particles(h, s, s.n + cnt);
        if (rc != EGG_OK) return rc;
        const size_t tail = (size_t)(s.n - at);
        double *arrays[11] = {s.x[s.cur].p, s.y[s.cur].p, s.vx[s.cur].p, s.vy[s.cur].p, s.x[s.cur ^ 1].p, s.y[s.cur ^ 1].p,
                              s.inv_mass.p, s.radius.p,   s.mass_t.p,    s.vx[s.cur ^ 1].p, s.vy[s.cur ^ 1].p};
        const int field_of[11] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 2, 3};
        DevBuf<double> tmp;
        if (tail) HIP_TRY(h, tmp.reserve(tail, false, s.stream));
        for (int a = 0; a < 11; ++a) {
            if (tail) {  // open a gap: order-preserving shift of the tail
                HIP_TRY(h, hipMemcpyAsync(tmp.p, arrays[a] + at, tail * 8, hipMemcpyDeviceToDevice, s.stream));
                HIP_TRY(h, hipMemcpyAsync(arrays[a] + at + cnt, tmp.p, tail * 8, hipMemcpyDeviceToDevice, s.stream));
            }
            HIP_TRY(h, hipMemcpyAsync(arrays[a] + at, src + (size_t)field_of[a] * cnt, (size_t)cnt * 8,
                                      hipMemcpyDefault, s.stream));  // (src: host or device memory)
        }
        HIP_TRY(h, hipStreamSynchronize(s.stream));
        s.n += cnt;
        s.atoms_dirty = s.targets_dirty = s.tiling_dirty = true;
        s.aabb_valid = s.aabb_on_device = false;
        s.disp_valid = false;
    }
    Batch b;
    b.id = (int64_t)h->batches.size() + 1;
    b.alive = true;
    b.key = info->key;
    b.target_x = info->target_x;
    b.target_y = info->target_y;
    b.white_radius = info->white_radius;
    b.yolk_radius = info->yolk_radius;
    b.n[0] = info->n_white;
    b.n[1] = info->n_yolk;
    h->order.insert(h->order.begin() + (long)pos, (int32_t)h->batches.size());
    h->batches.push_back(b);
    h->n_alive++;
    if (info->key >= h->next_key) h->next_key = info->key + 1;
    if (out_id) *out_id = b.id;
    return EGG_OK;
}

int egg_selftest_arith(egg_handle *h, int64_t n_operand_pairs, uint64_t seed, int64_t *mismatches) {
    if (!h || !mismatches || n_operand_pairs < 0) return EGG_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(h->device);
    DevBuf<unsigned long long> d;
    HIP_TRY(h, d.reserve(1, false, h->sys[0].stream));
    HIP_TRY(h, hipMemsetAsync(d.p, 0, sizeof(unsigned long long), h->sys[0].stream));
    const int threads = 256, blocks = 1024;
    int per_thread = (int)((n_operand_pairs + (int64_t)threads * blocks - 1) / ((int64_t)threads * blocks));
    hipLaunchKernelGGL(egg_selftest_arith_kernel, dim3(blocks), dim3(threads), 0, h->sys[0].stream,
                       (unsigned long long)seed, per_thread, d.p);
    HIP_TRY(h, hipGetLastError());
    unsigned long long bad = 0;
    HIP_TRY(h, hipMemcpyAsync(&bad, d.p, sizeof bad, hipMemcpyDeviceToHost, h->sys[0].stream));
    HIP_TRY(h, hipStreamSynchronize(h->sys[0].stream));
    *mismatches = (int64_t)bad;
    return EGG_OK;
}

int egg_get_environment(egg_handle *h, int which, egg_environment *out) {
    if (!h || !out || which < 0 || which > 1) return EGG_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    System &s = h->sys[which];
    const double inf = std::numeric_limits<double>::infinity();
    *out = egg_environment{inf, inf, -inf, -inf, 0, 0, 0, 0, 0, 0};  // L:1358-1390
    if (h->stats.steps == 0 || s.n == 0) return EGG_OK;
    auto key = [](double d) {
        unsigned long long u;
        memcpy(&u, &d, 8);
        return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    };
    auto unkey = [](unsigned long long k) {
        unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
        double d;
        memcpy(&d, &u, 8);
        return d;
    };
    // scratch: 6 ordered keys + 4 sums, in the type's scratch-free status staging area would alias live data,
    // so a small dedicated buffer
    HIP_TRY(h, s.d_env.reserve(16, false, s.stream));
    unsigned long long init[6] = {key(inf), key(inf), key(-inf), key(-inf), key(0.0), key(0.0)};
    HIP_TRY(h, hipMemcpyAsync(s.d_env.p, init, sizeof init, hipMemcpyHostToDevice, s.stream));
    const int n = (int)s.n;
    const int blocks = std::min(1024, (n + 255) / 256);
    hipLaunchKernelGGL(egg_env_bounds_kernel, dim3((unsigned)blocks), dim3(256), 0, s.stream, s.x[s.cur].p, s.y[s.cur].p,
                       s.vx[s.cur].p, s.vy[s.cur].p, s.radius.p, n, s.d_env.p);
    hipLaunchKernelGGL(egg_env_sums_kernel, dim3(4), dim3(EGG_WAVE), 0, s.stream, s.x[s.cur].p, s.y[s.cur].p,
                       s.x[s.cur ^ 1].p, s.y[s.cur ^ 1].p, n, (double *)(s.d_env.p + 6));
    HIP_TRY(h, hipGetLastError());
    h->stats.kernel_launches += 2;
    unsigned long long back[10];
    HIP_TRY(h, hipMemcpyAsync(back, s.d_env.p, sizeof back, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(h, hipStreamSynchronize(s.stream));
    double sums[4];
    memcpy(sums, back + 6, sizeof sums);
    out->min_x = unkey(back[0]);
    out->min_y = unkey(back[1]);
    out->max_x = unkey(back[2]);
    out->max_y = unkey(back[3]);
    out->max_radius = unkey(back[4]);
    out->max_velocity = unkey(back[5]);
    out->centroid_x = sums[0] / (double)n;
    out->centroid_y = sums[1] / (double)n;
    out->last_centroid_x = sums[2] / (double)n;
    out->last_centroid_y = sums[3] / (double)n;
    return EGG_OK;
}


int egg_get_stats(egg_handle *h, egg_stats *out) {
    if (!h || !out) return EGG_ERR_INVALID_ARGUMENT;
    *out = h->stats;
    return EGG_OK;
}

int egg_set_option(egg_handle *h, int option, double value) {
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    switch (option) {
        case EGG_OPT_CLAIM_MARGIN_CELLS:
            if (!(value >= 1 && value <= 4096)) return fail(h, EGG_ERR_INVALID_ARGUMENT, "margin must be in [1, 4096]");
            h->opt_margin = (int)value;
            for (int w = 0; w < 2; ++w) {
                h->sys[w].margin = h->opt_margin;
                h->sys[w].tiling_dirty = true;
            }
            return EGG_OK;
        case EGG_OPT_TILE_TARGET_PARTICLES:
            if (!(value >= 0 && value <= kMaxTileParticles)) return fail(h, EGG_ERR_INVALID_ARGUMENT, "tile target out of range");
            h->opt_tile_target = (int)value;
            h->sys[0].tiling_dirty = h->sys[1].tiling_dirty = true;
            return EGG_OK;
        case EGG_OPT_TIMING:
            h->opt_timing = value >= 2 ? 2 : (value != 0);
            memset(h->stats.pk_kernel_ms, 0, sizeof h->stats.pk_kernel_ms);
            memset(h->stats.pk_kernel_launches, 0, sizeof h->stats.pk_kernel_launches);
            h->stats.kernel_ms_sum[0] = h->stats.kernel_ms_sum[1] = 0;
            h->stats.timed_steps = 0;
            return EGG_OK;
        case EGG_OPT_THREADS_PER_PARTICLE:
            if (!(value >= 0 && value <= 4) || value != (int)value)
                return fail(h, EGG_ERR_INVALID_ARGUMENT, "threads per particle must be 0 (automatic), 1, 2, 3 or 4");
            h->opt_spread = (int)value;
            return EGG_OK;
        case EGG_OPT_FUSE_TYPES:
            h->opt_no_fuse = value == 0;
            return EGG_OK;
        case EGG_OPT_FORCE_GLOBAL_STATE:
            h->opt_force_global_state = value != 0;
            h->sys[0].tiling_dirty = h->sys[1].tiling_dirty = true;
            return EGG_OK;
        case EGG_OPT_BUDGET_PARTICLES_WHITE:
        case EGG_OPT_BUDGET_PARTICLES_YOLK:
            h->budget_particles[option == EGG_OPT_BUDGET_PARTICLES_WHITE ? 0 : 1] = value < 0 ? -1 : (int64_t)value;
            return EGG_OK;
        case EGG_OPT_SPIN_SLEEP:
            h->opt_spin_sleep = value < 0 ? -1 : (value != 0);
            return EGG_OK;
        case EGG_OPT_PACKED:
            h->opt_packed = value < 0 ? -1 : (value != 0);
            h->sys[0].tiling_dirty = h->sys[1].tiling_dirty = true;
            return EGG_OK;
        case EGG_OPT_LEVEL_WALK:
            if (!(value == 0 || value == 1 || value == 2)) return fail(h, EGG_ERR_INVALID_ARGUMENT, "level walk must be 0 (by regime), 1 (in order) or 2 (out of order)");
            if (value == 2 && !h->lds_lane_ordered) return fail(h, EGG_ERR_UNSUPPORTED, "the out-of-order level walk needs same-address LDS atomics served in lane order; this device's probe failed");
            h->opt_level_walk = (int)value;
            h->sys[0].tiling_dirty = h->sys[1].tiling_dirty = true;
            return EGG_OK;
        case EGG_OPT_GROUP_PARTICLES:
            if (!(value >= 0 && value <= 10240)) return fail(h, EGG_ERR_INVALID_ARGUMENT, "group particles must be in [0, 10240]");
            h->opt_group_particles = (int)value;
            h->sys[0].tiling_dirty = h->sys[1].tiling_dirty = true;
            return EGG_OK;
        case EGG_OPT_FORCE_SINGLE_TILE:
            h->opt_force_single = value != 0;
            h->sys[0].tiling_dirty = h->sys[1].tiling_dirty = true;
            return EGG_OK;
        default:
            return fail(h, EGG_ERR_INVALID_ARGUMENT, "unknown option %d", option);
    }
}

}  // extern "C"
