/*
 * eggsim_oracle.c -- sequential CPU restatement of the reference's particle
 * step.  See eggsim_oracle.h for the role of this file (test infrastructure,
 * "parity unpinned").  Build with -O2 -ffp-contract=off: every arithmetic
 * expression below is written in the reference's evaluation order and must not
 * be contracted into FMAs (LuaJIT on x86-64 evaluates in plain SSE2 doubles).
 *
 * Citations: L = /root/reference/simulation_handler.lua, M = /root/reference/math.lua.
 */
#include "eggsim_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EGG_EPS 1e-8 /* M:2 */
#define EGG_PI 3.14159265358979323846

/* ------------------------------------------------------------------ helpers */

static double egg_clamp(double x, double lo, double hi) { /* M:16-26 */
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

static double egg_mix(double lo, double hi, double t) { /* M:33-35 */
    return lo * (1 - t) + hi * t;
}

static void egg_normalize(double x, double y, double *ox, double *oy) { /* M:53-60 */
    double magnitude = sqrt(x * x + y * y);
    if (magnitude < EGG_EPS) {
        *ox = 0;
        *oy = 0;
    } else {
        *ox = x / magnitude;
        *oy = y / magnitude;
    }
}

static double egg_magnitude(double x, double y) { /* M:66-68 */
    return sqrt(x * x + y * y);
}

static double egg_distance(double x1, double y1, double x2, double y2) { /* M:96-100 */
    double dx = x2 - x1;
    double dy = y2 - y1;
    return egg_magnitude(dx, dy);
}

static double egg_squared_distance(double x1, double y1, double x2, double y2) { /* M:108-112 */
    double dx = x2 - x1;
    double dy = y2 - y1;
    return dx * dx + dy * dy;
}

/* ------------------------------------------------------------ int containers */

typedef struct {
    int *v;
    int n, cap;
} ilist;

static void ilist_push(ilist *l, int x) {
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 8;
        l->v = (int *)realloc(l->v, sizeof(int) * (size_t)l->cap);
    }
    l->v[l->n++] = x;
}

/* spatial_hash: Table<cell key, Table<particle index>> (L:1351, L:1486-1511).
 * The reference keys cells by Szudzik's pairing of the sign-folded cell
 * coordinates (L:1474-1483), a bijection Z^2 -> N, so keying by the integer
 * pair itself gives the same table semantics. */
typedef struct {
    uint64_t *keys;
    int *vals; /* list id + 1, 0 = empty slot */
    size_t cap, n;
    ilist *lists;
    int n_lists, cap_lists;
} cellmap;

static uint64_t mix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static uint64_t cell_key(double cx, double cy) {
    return ((uint64_t)(uint32_t)(int32_t)(int64_t)cx << 32) | (uint64_t)(uint32_t)(int32_t)(int64_t)cy;
}

static void cellmap_clear(cellmap *m) { /* table.clear(env.spatial_hash) */
    if (m->n) memset(m->vals, 0, sizeof(int) * m->cap);
    m->n = 0;
    for (int i = 0; i < m->n_lists; ++i) m->lists[i].n = 0;
    m->n_lists = 0;
}

static void cellmap_grow(cellmap *m) {
    size_t ocap = m->cap;
    uint64_t *okeys = m->keys;
    int *ovals = m->vals;
    m->cap = ocap ? ocap * 2 : 1024;
    m->keys = (uint64_t *)malloc(sizeof(uint64_t) * m->cap);
    m->vals = (int *)calloc(m->cap, sizeof(int));
    for (size_t i = 0; i < ocap; ++i) {
        if (!ovals[i]) continue;
        size_t h = mix64(okeys[i]) & (m->cap - 1);
        while (m->vals[h]) h = (h + 1) & (m->cap - 1);
        m->keys[h] = okeys[i];
        m->vals[h] = ovals[i];
    }
    free(okeys);
    free(ovals);
}

static ilist *cellmap_find(cellmap *m, uint64_t key) {
    if (!m->cap) return NULL;
    size_t h = mix64(key) & (m->cap - 1);
    while (m->vals[h]) {
        if (m->keys[h] == key) return &m->lists[m->vals[h] - 1];
        h = (h + 1) & (m->cap - 1);
    }
    return NULL;
}

static ilist *cellmap_find_or_insert(cellmap *m, uint64_t key) {
    if ((m->n + 1) * 2 > m->cap) cellmap_grow(m);
    size_t h = mix64(key) & (m->cap - 1);
    while (m->vals[h]) {
        if (m->keys[h] == key) return &m->lists[m->vals[h] - 1];
        h = (h + 1) & (m->cap - 1);
    }
    if (m->n_lists == m->cap_lists) {
        int ncap = m->cap_lists ? m->cap_lists * 2 : 256;
        m->lists = (ilist *)realloc(m->lists, sizeof(ilist) * (size_t)ncap);
        memset(m->lists + m->cap_lists, 0, sizeof(ilist) * (size_t)(ncap - m->cap_lists));
        m->cap_lists = ncap;
    }
    m->keys[h] = key;
    m->vals[h] = ++m->n_lists;
    m->n++;
    return &m->lists[m->n_lists - 1];
}

static void cellmap_free(cellmap *m) {
    for (int i = 0; i < m->cap_lists; ++i) free(m->lists[i].v);
    free(m->lists);
    free(m->keys);
    free(m->vals);
}

/* collided: Set<pair key> (L:1349, L:1584-1590).  The reference key is
 * Szudzik(min, max) of the two particle indices; (min, max) itself is used. */
typedef struct {
    uint64_t *keys; /* key + 1, 0 = empty */
    size_t cap, n;
} pairset;

static void pairset_clear(pairset *s) { /* table.clear(env.collided) */
    if (s->n) memset(s->keys, 0, sizeof(uint64_t) * s->cap);
    s->n = 0;
}

static void pairset_grow(pairset *s) {
    size_t ocap = s->cap;
    uint64_t *okeys = s->keys;
    s->cap = ocap ? ocap * 2 : 4096;
    s->keys = (uint64_t *)calloc(s->cap, sizeof(uint64_t));
    for (size_t i = 0; i < ocap; ++i) {
        if (!okeys[i]) continue;
        size_t h = mix64(okeys[i]) & (s->cap - 1);
        while (s->keys[h]) h = (h + 1) & (s->cap - 1);
        s->keys[h] = okeys[i];
    }
    free(okeys);
}

/* returns 1 if the key was already present, else inserts it and returns 0 */
static int pairset_test_and_set(pairset *s, uint64_t key) {
    if ((s->n + 1) * 2 > s->cap) pairset_grow(s);
    key += 1;
    size_t h = mix64(key) & (s->cap - 1);
    while (s->keys[h]) {
        if (s->keys[h] == key) return 1;
        h = (h + 1) & (s->cap - 1);
    }
    s->keys[h] = key;
    s->n++;
    return 0;
}

/* ------------------------------------------------------------------- state */

typedef struct { /* one particle system + its environment (L:1344-1390) */
    int64_t n, cap;
    double *f[EGG_ORACLE_N_FIELDS]; /* the live fields of the AoS record (L:713-740) as SoA */
    egg_oracle_config cfg;
    /* env */
    int has_env;
    double env_min_mass, env_max_mass, env_min_radius, env_max_radius;
    int should_update_mass, should_update_radius;
    double damping, follow_compliance, collision_compliance, cohesion_compliance;
    double max_n_collisions, cell_radius;
    double min_x, min_y, max_x, max_y, centroid_x, centroid_y, max_radius, max_velocity;
    double last_centroid_x, last_centroid_y;
    cellmap hash;
    pairset collided;
} psys;

typedef struct {
    int64_t id;
    int alive;
    double target_x, target_y;
    double white_radius, yolk_radius;
    double sqrt_white_radius, sqrt_yolk_radius; /* env.batch_id_to_radius (L:1789-1792) */
    ilist idx[2];                               /* particle indices, 0-based */
} batch;

struct egg_oracle {
    psys sys[2];
    batch *batches;
    int64_t n_batches_total, cap_batches, n_alive;
    int64_t current_batch_id;
    double elapsed, interpolation_alpha;
    double mass_distribution_variance; /* L:447 */
    double max_collision_fraction;     /* L:448 */
    int64_t budget_n[2]; /* >= 0: the N of max_n_collisions (see egg_oracle_set_budget_particles); -1: the system's own count */
    /* stats */
    egg_oracle_pass_stat *stats;
    int n_stats, cap_stats;
    int64_t total_visited, total_steps;
    int trace_on;
    egg_oracle_pair *trace;
    int64_t n_trace, cap_trace;
};

static void psys_reserve(psys *s, int64_t need) {
    if (need <= s->cap) return;
    int64_t ncap = s->cap ? s->cap : 256;
    while (ncap < need) ncap *= 2;
    for (int k = 0; k < EGG_ORACLE_N_FIELDS; ++k)
        s->f[k] = (double *)realloc(s->f[k], sizeof(double) * (size_t)ncap);
    s->cap = ncap;
}

static batch *find_batch(egg_oracle *o, int64_t id) {
    if (id < 1 || id > o->n_batches_total) return NULL;
    batch *b = &o->batches[id - 1];
    return b->alive ? b : NULL;
}

/* ----------------------------------------------------------- step helpers */

static double strength_to_compliance(double strength, double sub_step_delta) { /* L:1337-1341 */
    double alpha = 1 - egg_clamp(strength, 0, 1);
    double alpha_per_substep = alpha / (sub_step_delta * sub_step_delta);
    return alpha_per_substep;
}

static void update_environment(egg_oracle *o, psys *s, double sub_delta) { /* L:1726-1774 */
    const egg_oracle_config *config = &s->cfg;
    if (!s->has_env) { /* _create_environment(nil), L:1345-1369 */
        s->should_update_mass = 1;
        s->should_update_radius = 1;
        s->has_env = 1;
    } else { /* L:1370-1389 */
        cellmap_clear(&s->hash);
        pairset_clear(&s->collided);
        s->min_x = INFINITY;
        s->min_y = INFINITY;
        s->max_x = -INFINITY;
        s->max_y = -INFINITY;
        s->centroid_x = 0;
        s->centroid_y = 0;
        s->should_update_mass = config->min_mass != s->env_min_mass || config->max_mass != s->env_max_mass;
        s->should_update_radius = config->min_radius != s->env_min_radius || config->max_radius != s->env_max_radius;
    }
    s->env_min_mass = config->min_mass;
    s->env_max_mass = config->max_mass;
    s->env_min_radius = config->min_radius;
    s->env_max_radius = config->max_radius;

    /* N = every particle of the type in the handler (L:1752-1753).  A test that steps a CHUNK of a larger scene's
     * independent islands sets N to the whole scene's count, so that the chunk sees the scene's budget. */
    const int which = (int)(s - o->sys);
    double n = (double)(o->budget_n[which] >= 0 ? o->budget_n[which] : s->n);
    s->max_n_collisions = o->max_collision_fraction * (n * n); /* L:1752-1753 */

    double max_factor = fmax(config->collision_overlap_factor, config->cohesion_interaction_distance_factor);
    s->cell_radius = fmax(1, config->max_radius * max_factor); /* L:1756-1760 */

    s->damping = 1 - egg_clamp(config->damping, 0, 1); /* L:1768 */
    s->follow_compliance = strength_to_compliance(config->follow_strength, sub_delta);
    s->collision_compliance = strength_to_compliance(config->collision_strength, sub_delta);
    s->cohesion_compliance = strength_to_compliance(config->cohesion_strength, sub_delta);
}

static void update_last_positions(psys *s) { /* L:1795-1815 */
    double sum_x = 0, sum_y = 0;
    for (int64_t i = 0; i < s->n; ++i) {
        double x = s->f[EGG_ORACLE_X][i];
        double y = s->f[EGG_ORACLE_Y][i];
        s->f[EGG_ORACLE_LAST_X][i] = x;
        s->f[EGG_ORACLE_LAST_Y][i] = y;
        sum_x = sum_x + x;
        sum_y = sum_y + y;
    }
    if (s->n > 0) {
        s->last_centroid_x = sum_x / (double)s->n;
        s->last_centroid_y = sum_y / (double)s->n;
    } else {
        s->last_centroid_x = 0;
        s->last_centroid_y = 0;
    }
}

static void pre_solve(psys *s, double delta) { /* L:1393-1432 */
    double *X = s->f[EGG_ORACLE_X], *Y = s->f[EGG_ORACLE_Y];
    double *VX = s->f[EGG_ORACLE_VX], *VY = s->f[EGG_ORACLE_VY];
    double damping = s->damping;
    for (int64_t i = 0; i < s->n; ++i) {
        double x = X[i], y = Y[i];
        s->f[EGG_ORACLE_PREV_X][i] = x;
        s->f[EGG_ORACLE_PREV_Y][i] = y;

        double velocity_x = VX[i] * damping;
        double velocity_y = VY[i] * damping;
        VX[i] = velocity_x;
        VY[i] = velocity_y;

        X[i] = x + delta * velocity_x;
        Y[i] = y + delta * velocity_y;

        double mass_t = s->f[EGG_ORACLE_MASS_T][i];
        if (s->should_update_mass) {
            double mass = egg_mix(s->env_min_mass, s->env_max_mass, mass_t);
            s->f[EGG_ORACLE_MASS][i] = mass;
            s->f[EGG_ORACLE_INV_MASS][i] = 1 / mass;
        }
        if (s->should_update_radius) {
            s->f[EGG_ORACLE_RADIUS][i] = egg_mix(s->env_min_radius, s->env_max_radius, mass_t);
        }
    }
}

static void solve_follow_constraint(egg_oracle *o, psys *s, int which) { /* L:1435-1471 */
    double *X = s->f[EGG_ORACLE_X], *Y = s->f[EGG_ORACLE_Y];
    double compliance = s->follow_compliance;
    for (int64_t i = 0; i < s->n; ++i) {
        int64_t batch_id = (int64_t)s->f[EGG_ORACLE_BATCH_ID][i];
        const batch *b = &o->batches[batch_id - 1];
        double follow_x = b->target_x;
        double follow_y = b->target_y;

        double x = X[i], y = Y[i];
        double current_distance = egg_distance(x, y, follow_x, follow_y);
        double target_distance = 2 * (which == EGG_ORACLE_WHITE ? b->sqrt_white_radius : b->sqrt_yolk_radius);

        double inverse_mass = s->f[EGG_ORACLE_INV_MASS][i];
        if (inverse_mass > EGG_EPS && current_distance > target_distance) {
            double dx, dy;
            egg_normalize(follow_x - x, follow_y - y, &dx, &dy);

            double constraint_violation = current_distance - target_distance;
            double delta_lambda = constraint_violation / (inverse_mass + compliance);

            double x_correction = dx * delta_lambda * inverse_mass;
            double y_correction = dy * delta_lambda * inverse_mass;

            X[i] = X[i] + x_correction;
            Y[i] = Y[i] + y_correction;
        }
    }
}

static void rebuild_spatial_hash(psys *s) { /* L:1486-1511 */
    double r = s->cell_radius;
    for (int64_t i = 0; i < s->n; ++i) {
        double cell_x = floor(s->f[EGG_ORACLE_X][i] / r);
        double cell_y = floor(s->f[EGG_ORACLE_Y][i] / r);
        s->f[EGG_ORACLE_CELL_X][i] = cell_x;
        s->f[EGG_ORACLE_CELL_Y][i] = cell_y;
        ilist *entry = cellmap_find_or_insert(&s->hash, cell_key(cell_x, cell_y));
        ilist_push(entry, (int)i); /* table.insert(entry, particle_i): append, never de-duplicated */
    }
}

static void enforce_distance(double ax, double ay, double bx, double by, double inverse_mass_a,
                             double inverse_mass_b, double target_distance, double compliance,
                             double out[4]) { /* L:1514-1545 */
    double dx = bx - ax;
    double dy = by - ay;

    double current_distance = egg_magnitude(dx, dy);
    egg_normalize(dx, dy, &dx, &dy);

    double constraint_violation = current_distance - target_distance;
    double mass_sum = inverse_mass_a + inverse_mass_b;
    double divisor = (mass_sum + compliance);
    if (divisor < EGG_EPS) {
        out[0] = out[1] = out[2] = out[3] = 0;
        return;
    }

    double correction = -constraint_violation / divisor;
    double max_correction = fabs(constraint_violation);
    correction = egg_clamp(correction, -max_correction, max_correction);

    out[0] = -dx * correction * inverse_mass_a;
    out[1] = -dy * correction * inverse_mass_a;
    out[2] = dx * correction * inverse_mass_b;
    out[3] = dy * correction * inverse_mass_b;
}

static void trace_push(egg_oracle *o, int self_i, int other_i, int active) {
    if (o->n_trace == o->cap_trace) {
        o->cap_trace = o->cap_trace ? o->cap_trace * 2 : 4096;
        o->trace = (egg_oracle_pair *)realloc(o->trace, sizeof(egg_oracle_pair) * (size_t)o->cap_trace);
    }
    egg_oracle_pair *p = &o->trace[o->n_trace++];
    p->self_i = self_i;
    p->other_i = other_i;
    p->pass_seq = o->n_stats - 1;
    p->active = active;
}

static void solve_collision(egg_oracle *o, psys *s, egg_oracle_pass_stat *st) { /* L:1548-1666 */
    double *X = s->f[EGG_ORACLE_X], *Y = s->f[EGG_ORACLE_Y];
    const double *W = s->f[EGG_ORACLE_INV_MASS], *R = s->f[EGG_ORACLE_RADIUS];
    const double *B = s->f[EGG_ORACLE_BATCH_ID];
    const double collision_overlap_factor = s->cfg.collision_overlap_factor;
    const double cohesion_interaction_distance_factor = s->cfg.cohesion_interaction_distance_factor;
    const double max_n_collisions = s->max_n_collisions;
    double n_collided = 0;
    double c[4];

    for (int64_t self_i = 0; self_i < s->n; ++self_i) {
        double self_inverse_mass = W[self_i];
        double self_radius = R[self_i];
        double self_batch_id = B[self_i];
        double cell_x = s->f[EGG_ORACLE_CELL_X][self_i];
        double cell_y = s->f[EGG_ORACLE_CELL_Y][self_i];

        for (int x_offset = -1; x_offset <= 1; ++x_offset) {
            for (int y_offset = -1; y_offset <= 1; ++y_offset) {
                ilist *entry = cellmap_find(&s->hash, cell_key(cell_x + x_offset, cell_y + y_offset));
                if (entry == NULL) continue;

                /* ipairs(entry): the list is not modified while iterating */
                for (int k = 0; k < entry->n; ++k) {
                    int64_t other_i = entry->v[k];
                    if (self_i == other_i) continue;

                    uint64_t lo = (uint64_t)(self_i < other_i ? self_i : other_i);
                    uint64_t hi = (uint64_t)(self_i < other_i ? other_i : self_i);
                    if (pairset_test_and_set(&s->collided, (lo << 32) | hi)) continue;

                    double other_inverse_mass = W[other_i];
                    double other_radius = R[other_i];
                    double other_batch_id = B[other_i];

                    if (self_inverse_mass + other_inverse_mass < EGG_EPS) continue; /* L:1601 */

                    { /* cohesion (L:1603-1630); numerically a no-op, kept literal */
                        double self_x = X[self_i], self_y = Y[self_i];
                        double other_x = X[other_i], other_y = Y[other_i];
                        double interaction_distance;
                        if (self_batch_id == other_batch_id)
                            interaction_distance = 0;
                        else
                            interaction_distance = cohesion_interaction_distance_factor * (self_radius + other_radius);

                        if (self_batch_id == other_batch_id &&
                            egg_squared_distance(self_x, self_y, other_x, other_y) <=
                                interaction_distance * interaction_distance) {
                            enforce_distance(self_x, self_y, other_x, other_y, self_inverse_mass,
                                             other_inverse_mass, interaction_distance, s->cohesion_compliance, c);
                            X[self_i] = self_x + c[0];
                            Y[self_i] = self_y + c[1];
                            X[other_i] = other_x + c[2];
                            Y[other_i] = other_y + c[3];
                        }
                    }

                    int active = 0;
                    { /* collision (L:1632-1654) */
                        double min_distance = collision_overlap_factor * (self_radius + other_radius);
                        double self_x = X[self_i], self_y = Y[self_i];
                        double other_x = X[other_i], other_y = Y[other_i];
                        double distance = egg_squared_distance(self_x, self_y, other_x, other_y);
                        if (distance <= min_distance * min_distance) {
                            enforce_distance(self_x, self_y, other_x, other_y, self_inverse_mass,
                                             other_inverse_mass, min_distance, s->collision_compliance, c);
                            X[self_i] = self_x + c[0];
                            Y[self_i] = self_y + c[1];
                            X[other_i] = other_x + c[2];
                            Y[other_i] = other_y + c[3];
                            active = 1;
                        }
                    }

                    n_collided = n_collided + 1;
                    st->n_visited++;
                    st->n_active += active;
                    if (o->trace_on) trace_push(o, (int)self_i, (int)other_i, active);
                    if (n_collided >= max_n_collisions) { /* L:1657-1658 */
                        st->cut = 1;
                        return;
                    }
                }
            }
        }
    }
}

static void post_solve(psys *s, double delta) { /* L:1669-1718 */
    double min_x = INFINITY, min_y = INFINITY, max_x = -INFINITY, max_y = -INFINITY;
    double centroid_x = 0, centroid_y = 0;
    double max_velocity = 0, max_radius = 0;
    for (int64_t i = 0; i < s->n; ++i) {
        double x = s->f[EGG_ORACLE_X][i];
        double y = s->f[EGG_ORACLE_Y][i];
        double velocity_x = (x - s->f[EGG_ORACLE_PREV_X][i]) / delta;
        double velocity_y = (y - s->f[EGG_ORACLE_PREV_Y][i]) / delta;
        s->f[EGG_ORACLE_VX][i] = velocity_x;
        s->f[EGG_ORACLE_VY][i] = velocity_y;

        double velocity_magnitude = egg_magnitude(velocity_x, velocity_y);
        if (velocity_magnitude > max_velocity) max_velocity = velocity_magnitude;

        centroid_x = centroid_x + x;
        centroid_y = centroid_y + y;

        double r = s->f[EGG_ORACLE_RADIUS][i];
        if (r > max_radius) max_radius = r;
        min_x = fmin(min_x, x - r);
        min_y = fmin(min_y, y - r);
        max_x = fmax(max_x, x + r);
        max_y = fmax(max_y, y + r);
    }
    if (s->n > 0) {
        centroid_x = centroid_x / (double)s->n;
        centroid_y = centroid_y / (double)s->n;
    }
    s->min_x = min_x;
    s->min_y = min_y;
    s->max_x = max_x;
    s->max_y = max_y;
    s->centroid_x = centroid_x;
    s->centroid_y = centroid_y;
    s->max_radius = max_radius;
    s->max_velocity = max_velocity;
}

static egg_oracle_pass_stat *new_stat(egg_oracle *o, int which, int sub_step, int pass) {
    if (o->n_stats == o->cap_stats) {
        o->cap_stats = o->cap_stats ? o->cap_stats * 2 : 32;
        o->stats = (egg_oracle_pass_stat *)realloc(o->stats, sizeof(egg_oracle_pass_stat) * (size_t)o->cap_stats);
    }
    egg_oracle_pass_stat *st = &o->stats[o->n_stats++];
    memset(st, 0, sizeof(*st));
    st->which = which;
    st->sub_step = sub_step;
    st->pass = pass;
    return st;
}

void egg_oracle_step(egg_oracle *o, double delta, int n_sub_steps, int n_collision_steps) { /* L:1722-1989 */
    double sub_delta = fmax(delta / n_sub_steps, EGG_EPS);
    psys *white = &o->sys[EGG_ORACLE_WHITE], *yolk = &o->sys[EGG_ORACLE_YOLK];

    o->n_stats = 0;
    o->n_trace = 0;

    update_environment(o, white, sub_delta);
    update_environment(o, yolk, sub_delta);

    for (int64_t k = 0; k < o->n_batches_total; ++k) { /* L:1789-1792 */
        batch *b = &o->batches[k];
        if (!b->alive) continue;
        b->sqrt_white_radius = sqrt(b->white_radius);
        b->sqrt_yolk_radius = sqrt(b->yolk_radius);
    }

    update_last_positions(white);
    update_last_positions(yolk);

    for (int sub_step_i = 0; sub_step_i < n_sub_steps; ++sub_step_i) {
        pre_solve(white, sub_delta);
        pre_solve(yolk, sub_delta);

        solve_follow_constraint(o, white, EGG_ORACLE_WHITE);
        solve_follow_constraint(o, yolk, EGG_ORACLE_YOLK);

        for (int collision_i = 0; collision_i < n_collision_steps; ++collision_i) {
            rebuild_spatial_hash(white);
            rebuild_spatial_hash(yolk);

            egg_oracle_pass_stat *st = new_stat(o, EGG_ORACLE_WHITE, sub_step_i, collision_i);
            solve_collision(o, white, st);
            o->total_visited += st->n_visited;
            st = new_stat(o, EGG_ORACLE_YOLK, sub_step_i, collision_i);
            solve_collision(o, yolk, st);
            o->total_visited += st->n_visited;

            if (collision_i + 1 < n_collision_steps) { /* L:1905-1912: NOT after the last pass */
                cellmap_clear(&white->hash);
                pairset_clear(&white->collided);
                cellmap_clear(&yolk->hash);
                pairset_clear(&yolk->collided);
            }
        }

        post_solve(white, sub_delta);
        post_solve(yolk, sub_delta);
    }
    o->total_steps++;
}

int egg_oracle_update(egg_oracle *o, double delta, double step_delta, int n_sub_steps,
                      int n_collision_steps) { /* L:199-216 */
    o->elapsed = o->elapsed + delta;
    double step = step_delta;
    int n_steps = 0;
    double max_n_steps = fmax(4, 4 * ceil((1.0 / 60) / step_delta));
    while (o->elapsed >= step) {
        egg_oracle_step(o, step, n_sub_steps, n_collision_steps);
        o->elapsed = o->elapsed - step;
        n_steps = n_steps + 1;
        if (n_steps > max_n_steps) {
            o->elapsed = 0;
            break;
        }
    }
    o->interpolation_alpha = egg_clamp(o->elapsed / step, 0, 1);
    return n_steps;
}

/* ------------------------------------------------------------ construction */

egg_oracle *egg_oracle_create(const egg_oracle_config *white, const egg_oracle_config *yolk) { /* L:425-459 */
    egg_oracle *o = (egg_oracle *)calloc(1, sizeof(egg_oracle));
    o->sys[0].cfg = *white;
    o->sys[1].cfg = yolk ? *yolk : *white;
    o->mass_distribution_variance = 4;
    o->max_collision_fraction = 0.05;
    o->budget_n[0] = o->budget_n[1] = -1;
    o->current_batch_id = 1;
    egg_oracle_step(o, 0, 1, 1); /* L:562, "step once to init environments" */
    o->total_steps = 0;
    return o;
}

void egg_oracle_destroy(egg_oracle *o) {
    if (!o) return;
    for (int w = 0; w < 2; ++w) {
        for (int k = 0; k < EGG_ORACLE_N_FIELDS; ++k) free(o->sys[w].f[k]);
        cellmap_free(&o->sys[w].hash);
        free(o->sys[w].collided.keys);
    }
    for (int64_t k = 0; k < o->n_batches_total; ++k) {
        free(o->batches[k].idx[0].v);
        free(o->batches[k].idx[1].v);
    }
    free(o->batches);
    free(o->stats);
    free(o->trace);
    free(o);
}

void egg_oracle_set_config(egg_oracle *o, int which, const egg_oracle_config *cfg) { o->sys[which].cfg = *cfg; }

static double butterworth(double variance, double t) { /* L:923-925; x^4 as repeated squaring */
    double u = variance * (t - 0.5);
    double u2 = u * u;
    return 1 / (1 + u2 * u2);
}

static void add_particle(egg_oracle *o, int which, double center_x, double center_y, double x_radius,
                         double y_radius, int64_t particle_i, int64_t n_particles, int64_t batch_id,
                         batch *b) { /* L:907-997 */
    psys *s = &o->sys[which];
    const egg_oracle_config *config = &s->cfg;
    double n = (double)n_particles, pi = (double)particle_i;

    /* fibonacci_spiral (L:907-918) */
    double golden_ratio = (1 + sqrt(5)) / 2;
    double golden_angle = 2 * EGG_PI / (golden_ratio * golden_ratio);
    double r = sqrt((pi - 1) / n);
    double theta = pi * golden_angle;
    double dx = r * x_radius * cos(theta);
    double dy = r * y_radius * sin(theta);
    double x = center_x + dx;
    double y = center_y + dy;

    /* get_mass (L:921-938) */
    double variance = o->mass_distribution_variance;
    double left = (pi - 0.5) / n;
    double right = (pi + 0.5) / n;
    double center = 0.5 * (left + right);
    double half_width = 0.5 * (right - left);
    double t1 = center - half_width / sqrt(3);
    double t2 = center + half_width / sqrt(3);
    double t = 0.5 * (butterworth(variance, t1) + butterworth(variance, t2));

    double mass = egg_mix(config->min_mass, config->max_mass, t);
    double radius = egg_mix(config->min_radius, config->max_radius, t);

    psys_reserve(s, s->n + 1);
    int64_t i = s->n++;
    s->f[EGG_ORACLE_X][i] = x;
    s->f[EGG_ORACLE_Y][i] = y;
    s->f[EGG_ORACLE_VX][i] = 0;
    s->f[EGG_ORACLE_VY][i] = 0;
    s->f[EGG_ORACLE_PREV_X][i] = x;
    s->f[EGG_ORACLE_PREV_Y][i] = y;
    s->f[EGG_ORACLE_RADIUS][i] = radius;
    s->f[EGG_ORACLE_MASS_T][i] = t;
    s->f[EGG_ORACLE_MASS][i] = mass;
    s->f[EGG_ORACLE_INV_MASS][i] = 1 / mass;
    s->f[EGG_ORACLE_CELL_X][i] = -INFINITY;
    s->f[EGG_ORACLE_CELL_Y][i] = -INFINITY;
    s->f[EGG_ORACLE_BATCH_ID][i] = (double)batch_id;
    s->f[EGG_ORACLE_LAST_X][i] = x;
    s->f[EGG_ORACLE_LAST_Y][i] = y;
    ilist_push(&b->idx[which], (int)i);
}

int64_t egg_oracle_add(egg_oracle *o, double x, double y, double white_radius, double yolk_radius,
                       int64_t white_n, int64_t yolk_n) { /* L:27-135, L:881-1033 */
    const egg_oracle_config *wc = &o->sys[0].cfg, *yc = &o->sys[1].cfg;
    double white_particle_radius = egg_mix(wc->min_radius, wc->max_radius, 0.5);
    double yolk_particle_radius = egg_mix(yc->min_radius, yc->max_radius, 0.5);
    if (white_n <= 0)
        white_n = (int64_t)ceil((EGG_PI * (white_radius * white_radius)) /
                                (EGG_PI * (white_particle_radius * white_particle_radius)));
    if (yolk_n <= 0)
        yolk_n = (int64_t)ceil((EGG_PI * (yolk_radius * yolk_radius)) /
                               (EGG_PI * (yolk_particle_radius * yolk_particle_radius)));

    if (o->n_batches_total == o->cap_batches) {
        o->cap_batches = o->cap_batches ? o->cap_batches * 2 : 64;
        o->batches = (batch *)realloc(o->batches, sizeof(batch) * (size_t)o->cap_batches);
    }
    batch *b = &o->batches[o->n_batches_total++];
    memset(b, 0, sizeof(*b));
    int64_t batch_id = o->current_batch_id;
    o->current_batch_id = o->current_batch_id + 1;
    b->id = batch_id;
    b->alive = 1;
    b->white_radius = white_radius; /* math.max(x_radius, y_radius), both equal (L:889-890) */
    b->yolk_radius = yolk_radius;
    b->target_x = x;
    b->target_y = y;

    for (int64_t i = 1; i <= white_n; ++i)
        add_particle(o, EGG_ORACLE_WHITE, x, y, white_radius, white_radius, i, white_n, batch_id, b);
    for (int64_t i = 1; i <= yolk_n; ++i)
        add_particle(o, EGG_ORACLE_YOLK, x, y, yolk_radius, yolk_radius, i, yolk_n, batch_id, b);
    o->n_alive++;
    return batch_id;
}

static void remove_particles(egg_oracle *o, int which, const ilist *indices) { /* L:1038-1097 */
    psys *s = &o->sys[which];
    if (indices->n == 0) return;
    int64_t total = s->n;
    char *remove = (char *)calloc((size_t)total, 1);
    for (int k = 0; k < indices->n; ++k) remove[indices->v[k]] = 1;
    int64_t *new_index = (int64_t *)malloc(sizeof(int64_t) * (size_t)total);
    int64_t write = 0;
    for (int64_t read = 0; read < total; ++read) new_index[read] = remove[read] ? -1 : write++;
    for (int64_t read = 0; read < total; ++read) {
        int64_t w = new_index[read];
        if (w >= 0 && w != read)
            for (int k = 0; k < EGG_ORACLE_N_FIELDS; ++k) s->f[k][w] = s->f[k][read];
    }
    s->n = write;
    for (int64_t bi = 0; bi < o->n_batches_total; ++bi) {
        batch *b = &o->batches[bi];
        if (!b->alive) continue;
        ilist *l = &b->idx[which];
        int wp = 0;
        for (int rp = 0; rp < l->n; ++rp) {
            int64_t ni = new_index[l->v[rp]];
            if (ni >= 0) l->v[wp++] = (int)ni;
        }
        l->n = wp;
    }
    free(remove);
    free(new_index);
}

int egg_oracle_remove(egg_oracle *o, int64_t id) { /* L:140-155 */
    batch *b = find_batch(o, id);
    if (!b) return 1;
    b->alive = 0; /* self._batch_id_to_batch[batch_id] = nil, before _remove */
    o->n_alive--;
    remove_particles(o, EGG_ORACLE_WHITE, &b->idx[0]);
    remove_particles(o, EGG_ORACLE_YOLK, &b->idx[1]);
    return 0;
}

int egg_oracle_set_target(egg_oracle *o, int64_t id, double x, double y) { /* L:254-264 */
    batch *b = find_batch(o, id);
    if (!b) return 1;
    b->target_x = x;
    b->target_y = y;
    return 0;
}

int egg_oracle_get_target(egg_oracle *o, int64_t id, double *x, double *y) { /* L:268-278 */
    batch *b = find_batch(o, id);
    if (!b) return 1;
    *x = b->target_x;
    *y = b->target_y;
    return 0;
}

int egg_oracle_get_position(egg_oracle *o, int64_t id, double *ox, double *oy) { /* L:281-295, L:1134-1148 */
    batch *b = find_batch(o, id);
    if (!b) return 1;
    double x = 0, y = 0;
    for (int w = 0; w < 2; ++w) {
        const psys *s = &o->sys[w];
        for (int k = 0; k < b->idx[w].n; ++k) {
            x = x + s->f[EGG_ORACLE_X][b->idx[w].v[k]];
            y = y + s->f[EGG_ORACLE_Y][b->idx[w].v[k]];
        }
    }
    double n = (double)(b->idx[0].n + b->idx[1].n);
    *ox = x / n;
    *oy = y / n;
    return 0;
}

/* --------------------------------------------------------------- accessors */

int64_t egg_oracle_n_particles(const egg_oracle *o, int which) { return o->sys[which].n; }
int64_t egg_oracle_n_batches(const egg_oracle *o) { return o->n_alive; }
void egg_oracle_copy_field(const egg_oracle *o, int which, int field, double *dst) {
    memcpy(dst, o->sys[which].f[field], sizeof(double) * (size_t)o->sys[which].n);
}
double egg_oracle_elapsed(const egg_oracle *o) { return o->elapsed; }
double egg_oracle_interpolation_alpha(const egg_oracle *o) { return o->interpolation_alpha; }

void egg_oracle_env(const egg_oracle *o, int which, double out[16]) {
    const psys *s = &o->sys[which];
    out[0] = s->damping;
    out[1] = s->follow_compliance;
    out[2] = s->collision_compliance;
    out[3] = s->cohesion_compliance;
    out[4] = s->max_n_collisions;
    out[5] = s->cell_radius;
    out[6] = s->min_x;
    out[7] = s->min_y;
    out[8] = s->max_x;
    out[9] = s->max_y;
    out[10] = s->centroid_x;
    out[11] = s->centroid_y;
    out[12] = s->max_radius;
    out[13] = s->max_velocity;
    out[14] = s->last_centroid_x;
    out[15] = s->last_centroid_y;
}

int egg_oracle_n_pass_stats(const egg_oracle *o) { return o->n_stats; }
void egg_oracle_pass_stats(const egg_oracle *o, egg_oracle_pass_stat *dst) {
    memcpy(dst, o->stats, sizeof(egg_oracle_pass_stat) * (size_t)o->n_stats);
}
int64_t egg_oracle_total_visited(const egg_oracle *o) { return o->total_visited; }
int64_t egg_oracle_total_steps(const egg_oracle *o) { return o->total_steps; }

void egg_oracle_set_trace(egg_oracle *o, int enabled) { o->trace_on = enabled; }
void egg_oracle_set_budget_particles(egg_oracle *o, int which, int64_t n) { o->budget_n[which] = n; }
int64_t egg_oracle_n_trace(const egg_oracle *o) { return o->n_trace; }
void egg_oracle_trace(const egg_oracle *o, egg_oracle_pair *dst) {
    memcpy(dst, o->trace, sizeof(egg_oracle_pair) * (size_t)o->n_trace);
}
