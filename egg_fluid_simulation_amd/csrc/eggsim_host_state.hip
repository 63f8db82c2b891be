// eggsim_host_state.hip -- particle creation (Fibonacci spiral, Butterworth mass factor: L:907-997, on the host in
// double like LuaJIT does), atoms, the device buffers of a particle type.  See eggsim_host.h.
#include "eggsim_host.h"

namespace egghost {

std::string g_create_error;

int fail(egg_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h)
        h->error = buf;
    else
        g_create_error = buf;
    return code;
}


Batch *find_batch(egg_handle *h, int64_t id) {
    if (id < 1 || id > (int64_t)h->batches.size()) return nullptr;
    Batch *b = &h->batches[(size_t)id - 1];
    return b->alive ? b : nullptr;
}
const Batch *find_batch(const egg_handle *h, int64_t id) { return find_batch(const_cast<egg_handle *>(h), id); }

// ---------------------------------------------------------------- particles


void make_template(const egg_config &cfg, double batch_radius, int64_t n_particles, ParticleTemplate &out) {
    out.dx.resize((size_t)n_particles);
    out.dy.resize((size_t)n_particles);
    out.t.resize((size_t)n_particles);
    out.inv_mass.resize((size_t)n_particles);
    out.radius.resize((size_t)n_particles);
    const double n = (double)n_particles;
    const double golden_ratio = (1 + std::sqrt(5.0)) / 2;
    const double golden_angle = 2 * kPi / (golden_ratio * golden_ratio);
    const double variance = cfg.mass_distribution_variance;
    auto butterworth = [variance](double t) {
        double u = variance * (t - 0.5);
        double u2 = u * u;
        return 1 / (1 + u2 * u2);
    };
    for (int64_t k = 1; k <= n_particles; ++k) {
        const double i = (double)k;
        double r = std::sqrt((i - 1) / n);
        double theta = i * golden_angle;
        out.dx[(size_t)k - 1] = r * batch_radius * std::cos(theta);
        out.dy[(size_t)k - 1] = r * batch_radius * std::sin(theta);
        double left = (i - 0.5) / n;
        double right = (i + 0.5) / n;
        double center = 0.5 * (left + right);
        double half_width = 0.5 * (right - left);
        double t1 = center - half_width / std::sqrt(3.0);
        double t2 = center + half_width / std::sqrt(3.0);
        double t = 0.5 * (butterworth(t1) + butterworth(t2));
        out.t[(size_t)k - 1] = t;
        double mass = mixd(cfg.min_mass, cfg.max_mass, t);
        out.inv_mass[(size_t)k - 1] = 1 / mass;
        out.radius[(size_t)k - 1] = mixd(cfg.min_radius, cfg.max_radius, t);
    }
}

int reserve_particles(egg_handle *h, System &s, int64_t need) {
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(h, s.x[b].reserve((size_t)need, true, s.stream));
        HIP_TRY(h, s.y[b].reserve((size_t)need, true, s.stream));
        HIP_TRY(h, s.vx[b].reserve((size_t)need, true, s.stream));
        HIP_TRY(h, s.vy[b].reserve((size_t)need, true, s.stream));
    }
    HIP_TRY(h, s.inv_mass.reserve((size_t)need, true, s.stream));
    HIP_TRY(h, s.radius.reserve((size_t)need, true, s.stream));
    HIP_TRY(h, s.mass_t.reserve((size_t)need, true, s.stream));
    return EGG_OK;
}

int append_particles(egg_handle *h, System &s, const ParticleTemplate &tp, int64_t n_batches, const double *cx,
                     const double *cy) {
    const size_t per = tp.dx.size();
    const size_t total = per * (size_t)n_batches;
    int rc = reserve_particles(h, s, s.n + (int64_t)total);
    if (rc != EGG_OK) return rc;
    std::vector<double> buf(total);
    auto upload = [&](double *dst) -> hipError_t {
        return hipMemcpy(dst + s.n, buf.data(), total * sizeof(double), hipMemcpyHostToDevice);
    };
    for (int64_t b = 0; b < n_batches; ++b)
        for (size_t k = 0; k < per; ++k) buf[(size_t)b * per + k] = cx[b] + tp.dx[k];
    HIP_TRY(h, upload(s.x[0].p));
    HIP_TRY(h, upload(s.x[1].p));
    for (int64_t b = 0; b < n_batches; ++b)
        for (size_t k = 0; k < per; ++k) buf[(size_t)b * per + k] = cy[b] + tp.dy[k];
    HIP_TRY(h, upload(s.y[0].p));
    HIP_TRY(h, upload(s.y[1].p));
    std::fill(buf.begin(), buf.end(), 0.0);
    HIP_TRY(h, upload(s.vx[0].p));
    HIP_TRY(h, upload(s.vx[1].p));
    HIP_TRY(h, upload(s.vy[0].p));
    HIP_TRY(h, upload(s.vy[1].p));
    for (int64_t b = 0; b < n_batches; ++b) std::copy(tp.inv_mass.begin(), tp.inv_mass.end(), buf.begin() + b * per);
    HIP_TRY(h, upload(s.inv_mass.p));
    for (int64_t b = 0; b < n_batches; ++b) std::copy(tp.radius.begin(), tp.radius.end(), buf.begin() + b * per);
    HIP_TRY(h, upload(s.radius.p));
    for (int64_t b = 0; b < n_batches; ++b) std::copy(tp.t.begin(), tp.t.end(), buf.begin() + b * per);
    HIP_TRY(h, upload(s.mass_t.p));
    s.n += (int64_t)total;
    s.atoms_dirty = s.targets_dirty = s.tiling_dirty = true;
    s.aabb_valid = s.aabb_on_device = false;
    return EGG_OK;
}

// ------------------------------------------------------------------- atoms


// End of a step launch: a step kernel runs for a fraction of a millisecond, and waking up from a blocking
// wait costs a noticeable part of that, so poll the stream for a short while before blocking.
hipError_t wait_step(hipStream_t stream) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0;; ++it) {
        const hipError_t e = hipStreamQuery(stream);
        if (e != hipErrorNotReady) return e;
        if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) break;
    }
    return hipStreamSynchronize(stream);
}

int upload_atoms(egg_handle *h, int which) {
    System &s = h->sys[which];
    if (s.atoms_dirty) {
        s.atoms.clear();
        int64_t off = 0;
        for (int32_t b : h->order) {
            const Batch &B = h->batches[(size_t)b];
            if (!B.alive) continue;
            Atom a;
            a.batch = (int32_t)b;
            a.offset = (int32_t)off;
            a.count = (int32_t)B.n[which];
            off += B.n[which];
            s.atoms.push_back(a);
        }
        const size_t na = s.atoms.size();
        std::vector<int32_t> o(na), c(na), bb(na);
        for (size_t k = 0; k < na; ++k) {
            o[k] = s.atoms[k].offset;
            c[k] = s.atoms[k].count;
            bb[k] = s.atoms[k].batch;
        }
        HIP_TRY(h, s.d_atom_offset.reserve(na + 1, false, s.stream));
        HIP_TRY(h, s.d_atom_count.reserve(na + 1, false, s.stream));
        HIP_TRY(h, s.d_atom_batch.reserve(na + 1, false, s.stream));
        {
            int rc = reserve_out(h, s, na);
            if (rc != EGG_OK) return rc;
        }
        HIP_TRY(h, s.d_atom_fail.reserve(na + 1, false, s.stream));
        if (na) {
            HIP_TRY(h, hipMemcpy(s.d_atom_offset.p, o.data(), na * 4, hipMemcpyHostToDevice));
            HIP_TRY(h, hipMemcpy(s.d_atom_count.p, c.data(), na * 4, hipMemcpyHostToDevice));
            HIP_TRY(h, hipMemcpy(s.d_atom_batch.p, bb.data(), na * 4, hipMemcpyHostToDevice));
        }
        s.atoms_dirty = false;
        s.targets_dirty = true;
        s.tiling_dirty = true;
        s.disp_valid = false;
        s.aabb_valid = s.aabb_on_device = false;
    }
    if (s.targets_dirty) {
        const size_t na = s.atoms.size();
        s.h_tx.resize(na);
        s.h_ty.resize(na);
        s.h_fd.resize(na);
        for (size_t k = 0; k < na; ++k) {
            const Batch &B = h->batches[(size_t)s.atoms[k].batch];
            s.h_tx[k] = B.target_x;
            s.h_ty[k] = B.target_y;
            // target_distance = 2 * batch_id_to_radius[batch_id], radius = sqrt(batch radius) (L:1454, L:1790)
            s.h_fd[k] = 2 * std::sqrt(which == EGG_WHITE ? B.white_radius : B.yolk_radius);
        }
        s.targets_dirty = false;
        s.meta_dirty = true;
    }
    return EGG_OK;
}

double cell_size_of(const egg_config &c) {  // L:1756-1760
    double max_factor = std::max(c.collision_overlap_factor, c.cohesion_interaction_distance_factor);
    return std::max(1.0, c.max_radius * max_factor);
}



// Groups atoms into tiles.  Atoms whose margin-padded cell boxes come within one cell of
// each other may interact during the step and must share a tile ("islands": connected
// components of that relation); independent islands may additionally be packed into one
// tile to fill a wave.  Each atom's claim box is its padded box: the kernel verifies that
// no particle leaves it, which proves that particles of different tiles never occupy
// device buffer for `na` atoms; a fresh allocation gets both status blocks initialised (afterwards every
// launch re-initialises the block the next launch will use)
int reserve_out(egg_handle *h, System &s, size_t na) {
    const int32_t *before = s.d_out.p;
    HIP_TRY(h, s.d_out.reserve(2 * kStatInts + 8 * na + 8, false, s.stream));
    if (s.d_out.p != before) {
        EggStatus init;
        memset(&init, 0, sizeof init);
        init.min_slack = std::numeric_limits<int32_t>::max();
        for (int p = 0; p < 2; ++p) HIP_TRY(h, hipMemcpy(d_stat(s, p), &init, sizeof init, hipMemcpyHostToDevice));
    }
    return EGG_OK;
}

int fetch_end_aabb(egg_handle *h, System &s) {
    // after a committed step d_atom_aabb holds the atoms' cells at the new positions
    const size_t na = s.atoms.size();
    s.aabb.resize(na);
    s.disp.resize(4 * na);
    if (na) {
        HIP_TRY(h, hipMemcpyAsync(s.aabb.data(), d_aabb(s), na * sizeof(Box), hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(h, hipMemcpyAsync(s.disp.data(), d_disp(s), 4 * na * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(h, hipStreamSynchronize(s.stream));
    }
    s.aabb_valid = true;
    s.disp_valid = true;  // both describe the step that produced the current positions
    return EGG_OK;
}

}  // namespace egghost
