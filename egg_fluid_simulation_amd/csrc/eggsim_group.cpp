// eggsim_group.cpp -- several GPUs behind the C ABI: egg_group_* (include/eggsim.h).
//
// A LuaJIT / C host is ONE process, so its multi-GPU form is one process driving one egg_handle per device.  This
// file is a pure client of the single-device entry points (egg_create .. egg_step_begin / egg_step_end,
// egg_get_claims_many, egg_export_batch / egg_import_batch): the same protocol egg_fluid_simulation_amd/sharding.py runs
// between processes, without the messages.
//
//   * the plane is cut into x-slabs, one per device; a batch lives on the device whose slab holds its target when it
//     is added; ids are global, every handle lays its particles out in ascending global id (egg_add_many_keyed), so
//     results equal one handle holding everything, bit for bit;
//   * _step: every device launches its step (egg_step_begin fixes the claims of the step), the host then tests the
//     claims of batches on DIFFERENT devices against each other while the kernels run.  Claims of two batches closer
//     than one spatial-hash cell (simulation_handler.lua:1568-1578: what can interact) on different devices = the
//     sequential Gauss-Seidel order of the reference would cross devices: every device discards the launched step
//     (double-buffered state: free), the ISLANDS of such batches move to the lowest device involved
//     (egg_export_batch -> egg_import_batch), and the step is run again;
//   * an island that has left its slab's halo without meeting anything moves to the slab it is in after the step;
//   * the collision budget 0.05 N^2 (L:1752-1753) counts the particles of all devices (EGG_OPT_BUDGET_PARTICLES_*);
//     the visits of the step in flight are added up over the devices BEFORE it is committed, and a budget that could
//     bind is refused (exact-budget mode needs all particles of a type in one tile).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/eggsim.h"

struct egg_group {
    std::vector<egg_handle *> h;
    std::vector<double> cuts;  // n + 1 ascending x positions; slab k = [cuts[k], cuts[k + 1])
    double halo = 64.0;
    struct Rec {
        int owner = -1;
        int64_t local = 0;  // id inside the owning handle
        bool alive = false;
    };
    std::vector<Rec> batch;  // index = global id - 1
    bool budget_stale = true;
    double elapsed = 0, alpha = 0;
    int64_t migrations = 0, discarded_steps = 0;
    int64_t committed_visits[2] = {0, 0};
    std::string error;
};

namespace {

int gfail(egg_group *g, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->error = buf;
    return code;
}

#define GTRY(g, k, expr)                                                                                          \
    do {                                                                                                          \
        const int _rc = (expr);                                                                                   \
        if (_rc < 0) return gfail(g, _rc, "device %d: %s", (int)(k), egg_last_error((g)->h[(size_t)(k)]));       \
    } while (0)

int slab_of(const egg_group *g, double x) {
    int k = 0;
    const int n = (int)g->h.size();
    while (k + 1 < n && x >= g->cuts[(size_t)k + 1]) ++k;
    return k;
}

struct Claim {
    int64_t gid;
    int owner;
    double box[8];  // white lo_x lo_y hi_x hi_y, yolk ...
};

// the claims of every live batch for the step being prepared / in flight, per owner
int gather_claims(egg_group *g, std::vector<Claim> &out, double cell[2]) {
    out.clear();
    cell[0] = cell[1] = 0;
    const int n = (int)g->h.size();
    std::vector<std::vector<int64_t>> gids((size_t)n), lids((size_t)n);
    for (size_t i = 0; i < g->batch.size(); ++i)
        if (g->batch[i].alive) {
            gids[(size_t)g->batch[i].owner].push_back((int64_t)i + 1);
            lids[(size_t)g->batch[i].owner].push_back(g->batch[i].local);
        }
    for (int k = 0; k < n; ++k) {
        const size_t m = lids[(size_t)k].size();
        if (!m) continue;
        std::vector<double> boxes(8 * m);
        double cs[2] = {0, 0};
        GTRY(g, k, egg_get_claims_many(g->h[(size_t)k], (int64_t)m, lids[(size_t)k].data(), boxes.data(), cs));
        cell[0] = std::max(cell[0], cs[0]);
        cell[1] = std::max(cell[1], cs[1]);
        for (size_t j = 0; j < m; ++j) {
            Claim c;
            c.gid = gids[(size_t)k][j];
            c.owner = k;
            memcpy(c.box, &boxes[8 * j], sizeof c.box);
            out.push_back(c);
        }
    }
    return EGG_OK;
}

bool near(const Claim &a, const Claim &b, const double cell[2]) {
    for (int t = 0; t < 2; ++t) {
        const double *p = a.box + 4 * t, *q = b.box + 4 * t, c = cell[t];
        if (p[0] - q[2] < c && q[0] - p[2] < c && p[1] - q[3] < c && q[1] - p[3] < c) return true;
    }
    return false;
}

// Islands = batches chained by claims less than a cell apart, over ALL devices (sweep over the claims' low x).
// plan[gid] = the device an island must move to: the lowest device of an island that spans several, or -- for an island
// on one device that lies wholly outside that slab's halo -- the slab its middle is in.  conflicts: islands spanning devices.
int plan_moves(egg_group *g, const std::vector<Claim> &cl, const double cell[2], std::map<int64_t, int> &plan, int &conflicts) {
    plan.clear();
    conflicts = 0;
    const size_t n = cl.size();
    std::vector<size_t> order(n), parent(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::iota(parent.begin(), parent.end(), (size_t)0);
    auto lo = [&](size_t i) { return std::min(cl[i].box[0], cl[i].box[4]); };
    auto hi = [&](size_t i) { return std::max(cl[i].box[2], cl[i].box[6]); };
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return lo(a) < lo(b); });
    auto find = [&](size_t v) {
        while (parent[v] != v) v = parent[v] = parent[parent[v]];
        return v;
    };
    const double reach = std::max(cell[0], cell[1]);
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            const size_t a = order[i], b = order[j];
            if (lo(b) - hi(a) >= reach) break;  // (sorted by low x: nothing further right can touch a -- boxes of similar width)
            if (near(cl[a], cl[b], cell)) {
                const size_t ra = find(a), rb = find(b);
                if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
            }
        }
    // (the early exit above assumes no box hides a much wider one to its left: make up for it with the widest box)
    double widest = 0;
    for (size_t i = 0; i < n; ++i) widest = std::max(widest, hi(i) - lo(i));
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j) {
            const size_t a = order[i], b = order[j];
            if (lo(b) - lo(a) >= widest + reach) break;
            if (find(a) != find(b) && near(cl[a], cl[b], cell)) {
                const size_t ra = find(a), rb = find(b);
                parent[std::max(ra, rb)] = std::min(ra, rb);
            }
        }
    std::map<size_t, std::vector<size_t>> members;
    for (size_t i = 0; i < n; ++i) members[find(i)].push_back(i);
    const int nd = (int)g->h.size();
    for (auto &kv : members) {
        const std::vector<size_t> &m = kv.second;
        int lowest = nd, highest = -1;
        double x0 = std::numeric_limits<double>::infinity(), x1 = -x0;
        for (size_t i : m) {
            lowest = std::min(lowest, cl[i].owner);
            highest = std::max(highest, cl[i].owner);
            x0 = std::min(x0, lo(i));
            x1 = std::max(x1, hi(i));
        }
        int dest = -1;
        if (lowest != highest) {
            dest = lowest;
            ++conflicts;
        } else if ((lowest > 0 && x1 < g->cuts[(size_t)lowest] - g->halo) || (lowest + 1 < nd && x0 > g->cuts[(size_t)lowest + 1] + g->halo)) {
            dest = slab_of(g, 0.5 * (x0 + x1));
        }
        if (dest >= 0)
            for (size_t i : m)
                if (cl[i].owner != dest) plan[cl[i].gid] = dest;
    }
    return EGG_OK;
}

int move_batches(egg_group *g, const std::map<int64_t, int> &plan) {
    for (const auto &kv : plan) {
        egg_group::Rec &r = g->batch[(size_t)kv.first - 1];
        const int from = r.owner, to = kv.second;
        int64_t nw = 0, ny = 0;
        GTRY(g, from, egg_get_n_particles(g->h[(size_t)from], r.local, &nw, &ny));
        std::vector<double> ws(9 * (size_t)nw), ys(9 * (size_t)ny);
        egg_batch_info info;
        GTRY(g, from, egg_export_batch(g->h[(size_t)from], r.local, &info, ws.data(), ys.data()));
        int64_t lid = 0;
        GTRY(g, to, egg_import_batch(g->h[(size_t)to], &info, ws.data(), ys.data(), &lid));
        GTRY(g, from, egg_remove(g->h[(size_t)from], r.local));
        r.owner = to;
        r.local = lid;
        g->migrations++;
    }
    return EGG_OK;
}

int sync_budget(egg_group *g) {
    if (!g->budget_stale) return EGG_OK;
    int64_t tw = 0, ty = 0;
    for (size_t k = 0; k < g->h.size(); ++k) {
        int64_t nw = 0, ny = 0;
        GTRY(g, k, egg_get_n_particles(g->h[k], -1, &nw, &ny));
        tw += nw;
        ty += ny;
    }
    for (size_t k = 0; k < g->h.size(); ++k) {
        GTRY(g, k, egg_set_option(g->h[k], EGG_OPT_BUDGET_PARTICLES_WHITE, (double)tw));
        GTRY(g, k, egg_set_option(g->h[k], EGG_OPT_BUDGET_PARTICLES_YOLK, (double)ty));
    }
    g->budget_stale = false;
    return EGG_OK;
}

// hands islands over until no island spans devices (tiles and claims of the step re-formed every round)
int rebalance(egg_group *g, double delta, int S, int C) {
    for (int round = 0; round < 2 * (int)g->h.size() + 4; ++round) {
        for (size_t k = 0; k < g->h.size(); ++k) GTRY(g, k, egg_prepare_step(g->h[k], delta, S, C));
        std::vector<Claim> cl;
        double cell[2];
        int rc = gather_claims(g, cl, cell);
        if (rc != EGG_OK) return rc;
        std::map<int64_t, int> plan;
        int conflicts = 0;
        plan_moves(g, cl, cell, plan, conflicts);
        if (plan.empty()) return EGG_OK;
        rc = move_batches(g, plan);
        if (rc != EGG_OK) return rc;
    }
    return gfail(g, EGG_ERR_INTERNAL, "egg_group: batch hand-over did not settle");
}

int group_step(egg_group *g, double delta, int S, int C) {  // one _step (L:1722) on every device
    const size_t n = g->h.size();
    int rc = sync_budget(g);
    if (rc != EGG_OK) return rc;
    if (n == 1) {
        GTRY(g, 0, egg_step(g->h[0], delta, S, C));
        return EGG_OK;
    }
    for (size_t k = 0; k < n; ++k) GTRY(g, k, egg_step_begin(g->h[k], delta, S, C));
    auto discard = [&]() {
        for (size_t k = 0; k < n; ++k) (void)egg_step_end(g->h[k], 0);
    };
    std::vector<Claim> cl;
    double cell[2];
    rc = gather_claims(g, cl, cell);
    if (rc != EGG_OK) {
        discard();
        return rc;
    }
    std::map<int64_t, int> plan;
    int conflicts = 0;
    plan_moves(g, cl, cell, plan, conflicts);
    // the budget guard: what the step in flight visited, summed over the devices, against the global budget
    int64_t visits[2] = {g->committed_visits[0], g->committed_visits[1]};
    double budget[2] = {0, 0};
    {
        int64_t sum[2] = {0, 0};
        for (size_t k = 0; k < n; ++k) {
            int64_t v[2] = {0, 0};
            double b[2] = {0, 0};
            const int prc = egg_step_peek_visits(g->h[k], v, b);
            if (prc < 0) {
                discard();
                return gfail(g, prc, "device %d: %s", (int)k, egg_last_error(g->h[k]));
            }
            for (int w = 0; w < 2; ++w) {
                sum[w] += v[w];
                budget[w] = std::max(budget[w], b[w]);
            }
        }
        for (int w = 0; w < 2; ++w) visits[w] = std::max(visits[w], sum[w]);
    }
    for (int w = 0; w < 2; ++w)
        if ((double)visits[w] > std::max(1.0, std::ceil(budget[w]))) {
            discard();
            return gfail(g, EGG_ERR_UNSUPPORTED,
                         "collision budget may bind across devices (type %d: up to %lld visits in a pass, budget %.2f): "
                         "exact-budget mode needs all particles of the type on one device", w, (long long)visits[w], budget[w]);
        }
    auto note_committed = [&]() {
        g->committed_visits[0] = g->committed_visits[1] = 0;
        for (size_t k = 0; k < n; ++k) {
            egg_stats st;
            if (egg_get_stats(g->h[k], &st) == EGG_OK)
                for (int w = 0; w < 2; ++w) g->committed_visits[w] += st.max_pass_visits[w];
        }
    };
    if (conflicts == 0) {
        for (size_t k = 0; k < n; ++k) GTRY(g, k, egg_step_end(g->h[k], 1));
        note_committed();
        if (!plan.empty()) {  // strayed islands: handed to the slab they are in before the next step
            rc = rebalance(g, delta, S, C);
            if (rc != EGG_OK) return rc;
        }
        return EGG_OK;
    }
    discard();
    g->discarded_steps++;
    rc = rebalance(g, delta, S, C);
    if (rc != EGG_OK) return rc;
    for (size_t k = 0; k < n; ++k) GTRY(g, k, egg_step_begin(g->h[k], delta, S, C));
    for (size_t k = 0; k < n; ++k) GTRY(g, k, egg_step_end(g->h[k], 1));
    note_committed();
    return EGG_OK;
}

}  // namespace

extern "C" {

int egg_group_create(const egg_config *white, const egg_config *yolk, int32_t n_devices, const int32_t *devices,
                     const double *cuts, egg_group **out) {
    if (!white || !out || n_devices < 1 || !devices || (n_devices > 1 && !cuts)) return EGG_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    egg_group *g = new egg_group();
    for (int k = 0; k <= n_devices; ++k) g->cuts.push_back(cuts ? cuts[k] : (k == 0 ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity()));
    for (int k = 0; k < n_devices; ++k)
        if (!(g->cuts[(size_t)k] < g->cuts[(size_t)k + 1])) {
            delete g;
            return EGG_ERR_INVALID_ARGUMENT;
        }
    for (int k = 0; k < n_devices; ++k) {
        egg_handle *h = nullptr;
        const int rc = egg_create(white, yolk, devices[k], &h);
        if (rc != EGG_OK) {
            for (egg_handle *q : g->h) egg_destroy(q);
            delete g;
            return rc;  // (egg_last_error(NULL) holds the reason)
        }
        g->h.push_back(h);
    }
    *out = g;
    return EGG_OK;
}

void egg_group_destroy(egg_group *g) {
    if (!g) return;
    for (egg_handle *h : g->h) egg_destroy(h);
    delete g;
}

const char *egg_group_last_error(const egg_group *g) { return g ? g->error.c_str() : egg_last_error(nullptr); }

int32_t egg_group_n_devices(const egg_group *g) { return g ? (int32_t)g->h.size() : 0; }

egg_handle *egg_group_handle(egg_group *g, int32_t k) { return (g && k >= 0 && k < (int32_t)g->h.size()) ? g->h[(size_t)k] : nullptr; }

int egg_group_set_halo(egg_group *g, double halo_px) {
    if (!g || !(halo_px >= 0)) return EGG_ERR_INVALID_ARGUMENT;
    g->halo = halo_px;
    return EGG_OK;
}

int egg_group_add(egg_group *g, double x, double y, double white_radius, double yolk_radius, int64_t white_n, int64_t yolk_n,
                  int64_t *out_id) {
    if (!g) return EGG_ERR_INVALID_ARGUMENT;
    const int64_t gid = (int64_t)g->batch.size() + 1;
    const int k = slab_of(g, x);
    int64_t lid = 0;
    const int rc = egg_add_many_keyed(g->h[(size_t)k], 1, &x, &y, white_radius, yolk_radius, white_n, yolk_n, &gid, &lid);
    if (rc < 0) return gfail(g, rc, "device %d: %s", k, egg_last_error(g->h[(size_t)k]));
    egg_group::Rec r;
    r.owner = k;
    r.local = lid;
    r.alive = true;
    g->batch.push_back(r);
    g->budget_stale = true;
    if (out_id) *out_id = gid;
    return rc;
}

int egg_group_remove(egg_group *g, int64_t id) {
    if (!g) return EGG_ERR_INVALID_ARGUMENT;
    if (id < 1 || id > (int64_t)g->batch.size() || !g->batch[(size_t)id - 1].alive) {
        gfail(g, EGG_WARN_UNKNOWN_ID, "egg_group_remove: no batch with id `%lld`", (long long)id);
        return EGG_WARN_UNKNOWN_ID;  // the reference warns and carries on (L:145)
    }
    egg_group::Rec &r = g->batch[(size_t)id - 1];
    GTRY(g, r.owner, egg_remove(g->h[(size_t)r.owner], r.local));
    r.alive = false;
    g->budget_stale = true;
    return EGG_OK;
}

int egg_group_set_target(egg_group *g, int64_t id, double x, double y) {
    if (!g) return EGG_ERR_INVALID_ARGUMENT;
    if (id < 1 || id > (int64_t)g->batch.size() || !g->batch[(size_t)id - 1].alive) {
        gfail(g, EGG_WARN_UNKNOWN_ID, "egg_group_set_target: no batch with id `%lld`", (long long)id);
        return EGG_WARN_UNKNOWN_ID;  // L:259
    }
    const egg_group::Rec &r = g->batch[(size_t)id - 1];
    return egg_set_target(g->h[(size_t)r.owner], r.local, x, y);
}

int egg_group_get_position(egg_group *g, int64_t id, double *x, double *y) {
    if (!g || !x || !y) return EGG_ERR_INVALID_ARGUMENT;
    if (id < 1 || id > (int64_t)g->batch.size() || !g->batch[(size_t)id - 1].alive)
        return gfail(g, EGG_ERR_UNKNOWN_ID, "egg_group_get_position: no batch with id `%lld`", (long long)id);  // L:286
    const egg_group::Rec &r = g->batch[(size_t)id - 1];
    GTRY(g, r.owner, egg_get_position(g->h[(size_t)r.owner], r.local, x, y));
    return EGG_OK;
}

int egg_group_owner(const egg_group *g, int64_t id, int32_t *device_index, int64_t *local_id) {
    if (!g || id < 1 || id > (int64_t)g->batch.size() || !g->batch[(size_t)id - 1].alive) return EGG_ERR_UNKNOWN_ID;
    if (device_index) *device_index = g->batch[(size_t)id - 1].owner;
    if (local_id) *local_id = g->batch[(size_t)id - 1].local;
    return EGG_OK;
}

int egg_group_step(egg_group *g, double delta, int32_t n_substeps, int32_t n_collision_steps) {
    if (!g || n_substeps < 1 || n_collision_steps < 1) return EGG_ERR_INVALID_ARGUMENT;
    return group_step(g, delta, n_substeps, n_collision_steps);
}

int egg_group_update(egg_group *g, double delta, double step_delta, int32_t n_substeps, int32_t n_collision_steps,
                     int32_t *out_n_steps) {  // the reference's accumulator (L:199-216) around the group's _step
    if (!g) return EGG_ERR_INVALID_ARGUMENT;
    if (!(step_delta > 0) || n_substeps < 1 || n_collision_steps < 1 || std::isnan(delta))
        return gfail(g, EGG_ERR_INVALID_ARGUMENT, "egg_group_update: invalid delta / step_delta / n_substeps / n_collision_steps");
    g->elapsed = g->elapsed + delta;
    int n_steps = 0;
    const double max_n_steps = std::max(4.0, 4 * std::ceil((1.0 / 60.0) / step_delta));
    while (g->elapsed >= step_delta) {
        const int rc = group_step(g, step_delta, n_substeps, n_collision_steps);
        if (rc != EGG_OK) return rc;
        g->elapsed = g->elapsed - step_delta;
        n_steps += 1;
        if (n_steps > max_n_steps) {  // death-spiral guard, L:208-213
            g->elapsed = 0;
            break;
        }
    }
    g->alpha = std::min(std::max(g->elapsed / step_delta, 0.0), 1.0);
    if (out_n_steps) *out_n_steps = n_steps;
    return EGG_OK;
}

int egg_group_get_counters(const egg_group *g, int64_t *migrations, int64_t *discarded_steps) {
    if (!g) return EGG_ERR_INVALID_ARGUMENT;
    if (migrations) *migrations = g->migrations;
    if (discarded_steps) *discarded_steps = g->discarded_steps;
    return EGG_OK;
}

}  // extern "C"
