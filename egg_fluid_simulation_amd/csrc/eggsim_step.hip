// eggsim_step.hip -- gfx950 kernels of the XPBD particle step.
//
// Replaces SimulationHandler:_step's sub-step loop (simulation_handler.lua:1821-1932,
// "L:" below) for one particle type.  One workgroup (one wave64) owns one TILE: a
// set of whole batches whose particles provably cannot meet any other tile's
// particles during this step (each particle must stay inside its atom's claimed
// cell box; the kernel checks that at every hash rebuild and raises
// EggStatus::fail_claim otherwise, and the host then re-tiles and re-runs the step).
//
// The whole step of a tile runs out of LDS:
//   pre-solve + follow (L:1393-1471)  ->  per collision pass: LDS cell hash
//   (L:1486-1511), visit lists, order-preserving DAG execution of the pair
//   projections (L:1514-1666)  ->  post-solve velocities (L:1690-1693).
//
// Exactness: the reference's pair loop is sequential Gauss-Seidel, so results
// depend on the visiting order.  Two pair projections commute iff they share no
// particle; therefore executing every pair when it is the next pending pair of
// BOTH its particles gives bit-identical positions.  Visit lists are built from
// cell adjacency in the reference's attempt order (3x3 cells x-outer/y-inner,
// entries of the previous un-cleared pass before this pass's, ascending index;
// L:1568-1578, L:1509, L:1905-1912), tests/tile_model.py states the same rules
// in Python and is checked against the CPU oracle.
//
// All arithmetic is IEEE double in the reference's evaluation order; this file
// must be compiled with -ffp-contract=off and without fast-math.
#include <hip/hip_runtime.h>
#include "eggsim_device.h"

#define EGG_EMPTY_KEY 0xFFFFFFFFu
#define EGG_NONE 0xFFFFu
#define EGG_SELF 0x8000u
#define EGG_IDX 0x7FFFu

// Diagnostic build only (-DEGG_PROFILE): per-phase cycle sums of tile 0 go to a side buffer that no
// other code reads.  The shipped library is built without it.
#ifdef EGG_PROFILE
__device__ unsigned long long egg_prof[16];
#define PROF_DECL unsigned long long _pt = __builtin_amdgcn_s_memtime(), _pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PROF(k)                                                  \
    {                                                            \
        unsigned long long _n = __builtin_amdgcn_s_memtime();   \
        _pacc[k] += _n - _pt;                                    \
        _pt = _n;                                                \
    }
#define PROF_FLUSH                                                                                    \
    if (lane == 0 && tile == 0) {                                                                    \
        for (int _k = 0; _k < 10; ++_k) atomicAdd(&egg_prof[_k], _pacc[_k]);                           \
        atomicAdd(&egg_prof[10], rounds_total);                                                      \
        atomicAdd(&egg_prof[11], 1ull);                                                              \
    }
#else
#define PROF_DECL
#define PROF(k)
#define PROF_FLUSH
#endif

namespace {

struct Tile {
    // particle state
    double *x, *y, *px, *py, *vx, *vy, *w, *r;
    double *atx, *aty, *afd;
    uint32_t *ckey_b;                 // [2][nmax]      packed tile-relative cell of each particle
    uint32_t *hkeys_b, *hmeta_b;      // [2][ht]        cell hash: key, (start << 16 | count)
    uint32_t *own_off_b;              // [2][nmax + 1]  CSR offsets of the visit lists
    uint32_t *inc_off, *fill;
    uint32_t *queue_b;                // [2][nmax / 2 + 1]
    int s_n, s_h, s_o, s_q, s_l;      // strides of the double-buffered arrays
    // double-buffered arrays are addressed by arithmetic, never through pointer tables, so
    // that a runtime buffer index does not force the struct into scratch memory
    __device__ uint32_t *ckey(int b) const { return ckey_b + b * s_n; }
    __device__ uint32_t *hkeys(int b) const { return hkeys_b + b * s_h; }
    __device__ uint32_t *hmeta(int b) const { return hmeta_b + b * s_h; }
    __device__ uint32_t *own_off(int b) const { return own_off_b + b * s_o; }
    __device__ uint32_t *queue(int b) const { return queue_b + b * s_q; }
    __device__ uint16_t *hitems(int b) const { return hitems_b + b * s_n; }
    __device__ uint16_t *own_ent(int b) const { return own_ent_b + b * s_l; }
    int32_t *aclaim, *aaabb, *aoff, *abatch;
    int32_t *sc;  // scalars: 0,1 queue counts; 2 total; 3 origin x; 4 origin y; 5 misc
    uint16_t *hitems_b;               // [2][nmax]      particles sorted by cell
    uint16_t *pslot, *aslot, *ptr, *nlo, *nxt, *stamp;
    uint16_t *own_ent_b;              // [2][lcap]      visit lists
    uint16_t *inc_ent, *inc_tmp;
    int n, na, ht, lcap;
};

__device__ inline unsigned char *carve(unsigned char *&p, size_t bytes) {
    unsigned char *q = p;
    p += (bytes + 7) & ~(size_t)7;
    return q;
}

__device__ inline int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive prefix sum of cnt[0..n) into off[0..n], off[n] = total; one wave
__device__ inline int block_exclusive_scan(const uint32_t *cnt, uint32_t *off, int n, int lane) {
    int carry = 0;
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        int v = (i < n) ? (int)cnt[i] : 0;
        int incl = wave_incl_scan(v, lane);
        if (i < n) off[i] = (uint32_t)(carry + incl - v);
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) off[n] = (uint32_t)carry;
    return carry;
}

__device__ inline int cell_slot(uint32_t ka, uint32_t kb) {
    // slot of cell ka in the 3x3 loop around cell kb (x offset outer, y inner; L:1568-1569), or -1
    int dcx = (int)(ka >> 16) - (int)(kb >> 16);
    int dcy = (int)(ka & 0xFFFFu) - (int)(kb & 0xFFFFu);
    if (dcx < -1 || dcx > 1 || dcy < -1 || dcy > 1) return -1;
    return (dcx + 1) * 3 + (dcy + 1);
}

__device__ inline uint32_t hash_cell(uint32_t key, int ht) { return (key * 2654435761u) >> 7 & (uint32_t)(ht - 1); }

// returns (start << 16 | count) of the cell's item list, or 0 when the cell is empty
__device__ inline uint32_t hash_lookup(const uint32_t *keys, const uint32_t *meta, uint32_t key, int ht) {
    uint32_t h = hash_cell(key, ht);
    for (int probe = 0; probe < ht; ++probe) {
        uint32_t k = keys[h];
        if (k == key) return meta[h];
        if (k == EGG_EMPTY_KEY) return 0;
        h = (h + 1) & (uint32_t)(ht - 1);
    }
    return 0;
}

// XPBD distance projection between two particles, L:1514-1545 with the collision
// caller L:1632-1654 (and the numerically dead cohesion block L:1603-1630).
__device__ inline void solve_pair(const Tile &t, int a, int b, double overlap, double compliance, double eps) {
    double wa = t.w[a], wb = t.w[b];
    if (wa + wb < eps) return;  // L:1601
    double ra = t.r[a], rb = t.r[b];
    double ax = t.x[a], ay = t.y[a], bx = t.x[b], by = t.y[b];
    double dx = bx - ax, dy = by - ay;
    double d2 = dx * dx + dy * dy;
    if (d2 <= 0.0 && t.abatch[t.aslot[a]] == t.abatch[t.aslot[b]]) {
        // cohesion fires only for coincident same-batch particles (interaction distance 0,
        // L:1608-1616); its corrections are +0 for self and -0 for other, so the only
        // observable effect is x_self + 0.0 (turns -0.0 into +0.0)
        ax = ax + 0.0;
        ay = ay + 0.0;
        t.x[a] = ax;
        t.y[a] = ay;
    }
    double min_distance = overlap * (ra + rb);
    if (d2 <= min_distance * min_distance) {
        double current = sqrt(d2);
        double nx, ny;
        if (current < eps) {  // math.normalize, math.lua:53-60
            nx = 0.0;
            ny = 0.0;
        } else {
            nx = dx / current;
            ny = dy / current;
        }
        double violation = current - min_distance;
        double divisor = (wa + wb) + compliance;
        double cax, cay, cbx, cby;
        if (divisor < eps) {
            cax = cay = cbx = cby = 0.0;
        } else {
            double correction = -violation / divisor;
            double max_correction = fabs(violation);
            if (correction < -max_correction) correction = -max_correction;
            if (correction > max_correction) correction = max_correction;
            cax = -nx * correction * wa;
            cay = -ny * correction * wa;
            cbx = nx * correction * wb;
            cby = ny * correction * wb;
        }
        t.x[a] = ax + cax;
        t.y[a] = ay + cay;
        t.x[b] = bx + cbx;
        t.y[b] = by + cby;
    }
}

// k-th element of particle i's pair sequence: pairs where i is `other` visited by
// smaller selves, then i's own visits, then pairs visited by larger selves (stale pass only)
__device__ inline uint32_t seq_entry(const Tile &t, int cur, int i, int k) {
    int nl = t.nlo[i];
    int o0 = (int)t.own_off(cur)[i], no = (int)t.own_off(cur)[i + 1] - o0;
    int i0 = (int)t.inc_off[i], ni = (int)t.inc_off[i + 1] - i0;
    if (k < nl) return t.inc_ent[i0 + k];
    if (k < nl + no) return (uint32_t)t.own_ent(cur)[o0 + k - nl] | EGG_SELF;
    if (k < no + ni) return t.inc_ent[i0 + k - no];
    return EGG_NONE;
}

struct PassCtx {
    int cur;          // which own_off/own_ent/ckey/hash buffer is "this pass"
    int stale;        // previous pass's hash lists and collided set are still alive (Q3)
    int prev_uncut;   // previous pass visited every adjacent pair (no budget cut)
};

// is the unordered pair {i, j} in `collided` from the previous pass?
__device__ inline bool in_prev(const Tile &t, const PassCtx &c, int i, int j) {
    int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t *ko = t.ckey(c.cur ^ 1);
    if (cell_slot(ko[hi], ko[lo]) < 0) return false;  // never met in the previous (fresh) pass
    if (c.prev_uncut) return true;
    const uint32_t *off = t.own_off(c.cur ^ 1);
    const uint16_t *ent = t.own_ent(c.cur ^ 1);
    for (uint32_t e = off[lo]; e < off[lo + 1]; ++e)
        if (ent[e] == (uint16_t)hi) return true;
    return false;
}

// does self i visit j when it meets it at 3x3 slot s through j's old (isnew=0) or new cell?
__device__ inline bool accept(const Tile &t, const PassCtx &c, int i, int j, int s, int isnew) {
    if (j == i) return false;
    if (!c.stale) return j > i;  // fresh pass: adjacency is symmetric, the smaller index visits
    const uint32_t *kn = t.ckey(c.cur), *ko = t.ckey(c.cur ^ 1);
    uint32_t ki = kn[i];
    // j can sit in i's attempt order twice (old cell and new cell): first occurrence wins
    if (!isnew) {
        int s2 = cell_slot(kn[j], ki);
        if (s2 >= 0 && s2 < s) return false;
    } else {
        int s2 = cell_slot(ko[j], ki);
        if (s2 >= 0 && s2 <= s) return false;
    }
    if (in_prev(t, c, i, j)) return false;
    if (j < i) {  // j's loop ran first: did it meet i?
        uint32_t kj = kn[j];
        if (cell_slot(ki, kj) >= 0 || cell_slot(ko[i], kj) >= 0) return false;
    }
    return true;
}

template <bool FILL>
__device__ inline int enumerate_visits(const Tile &t, const PassCtx &c, int i, uint16_t *dst) {
    const uint32_t ki = t.ckey(c.cur)[i];
    int count = 0;
    for (int s = 0; s < 9; ++s) {
        int dx = s / 3 - 1, dy = s % 3 - 1;
        uint32_t nk = (uint32_t)((int)ki + dx * 65536 + dy);
        if (c.stale) {
            uint32_t m = hash_lookup(t.hkeys(c.cur ^ 1), t.hmeta(c.cur ^ 1), nk, t.ht);
            int st = (int)(m >> 16), cn = (int)(m & 0xFFFFu);
            for (int e = 0; e < cn; ++e) {
                int j = t.hitems(c.cur ^ 1)[st + e];
                if (accept(t, c, i, j, s, 0)) {
                    if (FILL) dst[count] = (uint16_t)j;
                    ++count;
                }
            }
        }
        uint32_t m = hash_lookup(t.hkeys(c.cur), t.hmeta(c.cur), nk, t.ht);
        int st = (int)(m >> 16), cn = (int)(m & 0xFFFFu);
        for (int e = 0; e < cn; ++e) {
            int j = t.hitems(c.cur)[st + e];
            if (accept(t, c, i, j, s, 1)) {
                if (FILL) dst[count] = (uint16_t)j;
                ++count;
            }
        }
    }
    return count;
}

}  // namespace

extern "C" __global__ void __launch_bounds__(EGG_WAVE) egg_step_kernel(EggStepArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    if (tile >= A.n_tiles) return;
    PROF_DECL

    // ---------------------------------------------------------------- LDS carve
    Tile t;
    {
        unsigned char *p = smem;
        size_t n = (size_t)A.nmax, a = (size_t)A.amax, h = (size_t)A.ht, l = (size_t)A.lcap;
        t.x = (double *)carve(p, n * 8);
        t.y = (double *)carve(p, n * 8);
        t.px = (double *)carve(p, n * 8);
        t.py = (double *)carve(p, n * 8);
        t.vx = (double *)carve(p, n * 8);
        t.vy = (double *)carve(p, n * 8);
        t.w = (double *)carve(p, n * 8);
        t.r = (double *)carve(p, n * 8);
        t.atx = (double *)carve(p, a * 8);
        t.aty = (double *)carve(p, a * 8);
        t.afd = (double *)carve(p, a * 8);
        t.ckey_b = (uint32_t *)carve(p, 2 * n * 4);
        t.hkeys_b = (uint32_t *)carve(p, 2 * h * 4);
        t.hmeta_b = (uint32_t *)carve(p, 2 * h * 4);
        t.own_off_b = (uint32_t *)carve(p, 2 * (n + 1) * 4);
        t.inc_off = (uint32_t *)carve(p, (n + 1) * 4);
        t.fill = (uint32_t *)carve(p, n * 4);
        t.queue_b = (uint32_t *)carve(p, (n / 2 + 1) * 2 * 4);
        t.s_n = (int)n;
        t.s_h = (int)h;
        t.s_o = (int)n + 1;
        t.s_q = (int)n / 2 + 1;
        t.s_l = (int)l;
        t.aclaim = (int32_t *)carve(p, a * 4 * 4);
        t.aaabb = (int32_t *)carve(p, a * 4 * 4);
        t.aoff = (int32_t *)carve(p, (a + 1) * 4);
        t.abatch = (int32_t *)carve(p, a * 4);
        t.sc = (int32_t *)carve(p, 16 * 4);
        t.hitems_b = (uint16_t *)carve(p, 2 * n * 2);
        t.pslot = (uint16_t *)carve(p, n * 2);
        t.aslot = (uint16_t *)carve(p, n * 2);
        t.ptr = (uint16_t *)carve(p, n * 2);
        t.nlo = (uint16_t *)carve(p, n * 2);
        t.nxt = (uint16_t *)carve(p, n * 2);
        t.stamp = (uint16_t *)carve(p, n * 2);
        t.own_ent_b = (uint16_t *)carve(p, 2 * l * 2);
        t.inc_ent = (uint16_t *)carve(p, l * 2);
        t.inc_tmp = (uint16_t *)carve(p, l * 2);
        t.ht = A.ht;
        t.lcap = A.lcap;
    }

    // -------------------------------------------------------------- load tile
    const int a_begin = A.tile_atom_begin[tile];
    const int na = A.tile_atom_begin[tile + 1] - a_begin;
    t.na = na;
    if (lane == 0) {
        int off = 0;
        int ox = 0x7FFFFFFF, oy = 0x7FFFFFFF, hx = -0x7FFFFFFF, hy = -0x7FFFFFFF;
        for (int k = 0; k < na; ++k) {
            int atom = A.tile_atoms[a_begin + k];
            t.aoff[k] = off;
            off += A.atom_count[atom];
            for (int q = 0; q < 4; ++q) t.aclaim[4 * k + q] = A.atom_claim[4 * atom + q];
            ox = min(ox, t.aclaim[4 * k + 0]);
            oy = min(oy, t.aclaim[4 * k + 1]);
            hx = max(hx, t.aclaim[4 * k + 2]);
            hy = max(hy, t.aclaim[4 * k + 3]);
            t.aaabb[4 * k + 0] = 0x7FFFFFFF;
            t.aaabb[4 * k + 1] = 0x7FFFFFFF;
            t.aaabb[4 * k + 2] = -0x7FFFFFFF;
            t.aaabb[4 * k + 3] = -0x7FFFFFFF;
            t.atx[k] = A.atom_tx[atom];
            t.aty[k] = A.atom_ty[atom];
            t.afd[k] = A.atom_fd[atom];
            t.abatch[k] = A.atom_batch[atom];
        }
        t.aoff[na] = off;
        t.sc[2] = off;
        t.sc[3] = ox - 2;  // packed cell coordinates are relative to this origin and stay >= 1
        t.sc[4] = oy - 2;
        // 3x3 lookups reach one cell beyond the claim box on both sides
        if ((long long)hx - (ox - 2) + 2 > 65534ll || (long long)hy - (oy - 2) + 2 > 65534ll) {
            atomicExch(&A.status->fail_range, 1);
        }
        if (off > A.nmax || na > A.amax) atomicExch(&A.status->fail_overflow, 1);
    }
    __syncthreads();
    const int n = min(t.sc[2], A.nmax);
    t.n = n;
    const int org_x = t.sc[3], org_y = t.sc[4];

    for (int k = 0; k < na; ++k) {
        int atom = A.tile_atoms[a_begin + k];
        int g0 = A.atom_offset[atom];
        int l0 = t.aoff[k], cnt = t.aoff[k + 1] - l0;
        for (int q = lane; q < cnt; q += EGG_WAVE) {
            int i = l0 + q;
            if (i >= n) break;
            int g = g0 + q;
            t.x[i] = A.x_in[g];
            t.y[i] = A.y_in[g];
            t.vx[i] = A.vx_in[g];
            t.vy[i] = A.vy_in[g];
            t.w[i] = A.inv_mass[g];
            t.r[i] = A.radius[g];
            t.aslot[i] = (uint16_t)k;
        }
    }
    __syncthreads();

    PROF(0)  // carve + load
    const double sub_delta = A.sub_delta, eps = A.eps;
    // visits allowed per pass before the budget return: n_collided >= budget after the increment
    long long budget_m = (long long)ceil(A.budget);
    if (budget_m < 1) budget_m = 1;

    int cur = 0;
    int have_prev = 0;   // buffers cur^1 hold an un-cleared previous pass
    int prev_uncut = 1;
    int pass_seq = 0;
    unsigned long long rounds_total = 0;
    unsigned int max_list = 0;
    bool bad = false;

    for (int s = 0; s < A.n_substeps; ++s) {
        // ------------------------------------------------ pre-solve, L:1393-1432
        for (int i = lane; i < n; i += EGG_WAVE) {
            double x = t.x[i], y = t.y[i];
            t.px[i] = x;
            t.py[i] = y;
            double vx = t.vx[i] * A.damping;
            double vy = t.vy[i] * A.damping;
            t.vx[i] = vx;
            t.vy[i] = vy;
            x = x + sub_delta * vx;
            y = y + sub_delta * vy;
            // ------------------------------------- follow constraint, L:1435-1471
            int k = t.aslot[i];
            double fx = t.atx[k], fy = t.aty[k];
            double dx = fx - x, dy = fy - y;
            double current = sqrt(dx * dx + dy * dy);
            double target = t.afd[k];
            double im = t.w[i];
            if (im > eps && current > target) {
                double nx, ny;
                if (current < eps) {
                    nx = 0.0;
                    ny = 0.0;
                } else {
                    nx = dx / current;
                    ny = dy / current;
                }
                double violation = current - target;
                double lambda = violation / (im + A.follow_compliance);
                x = x + nx * lambda * im;
                y = y + ny * lambda * im;
            }
            t.x[i] = x;
            t.y[i] = y;
        }
        __syncthreads();
        PROF(1)  // pre-solve + follow

        for (int c = 0; c < A.n_collision_steps; ++c, ++pass_seq) {
            PassCtx ctx;
            ctx.cur = cur;
            ctx.stale = have_prev;
            ctx.prev_uncut = prev_uncut;

            // ----------------------------------- rebuild spatial hash, L:1486-1511
            for (int h = lane; h < t.ht; h += EGG_WAVE) {
                t.hkeys(cur)[h] = EGG_EMPTY_KEY;
                t.hmeta(cur)[h] = 0;
            }
            __syncthreads();
            for (int i = lane; i < n; i += EGG_WAVE) {
                double fcx = floor(t.x[i] / A.cell_size);
                double fcy = floor(t.y[i] / A.cell_size);
                const int32_t *cl = &t.aclaim[4 * t.aslot[i]];
                // NaN / out-of-range coordinates fail the box test below
                int cx = (fcx >= -2.0e9 && fcx <= 2.0e9) ? (int)fcx : 0x7FFFFFF0;
                int cy = (fcy >= -2.0e9 && fcy <= 2.0e9) ? (int)fcy : 0x7FFFFFF0;
                if (cx < cl[0] || cx > cl[2] || cy < cl[1] || cy > cl[3]) {
                    bad = true;
                    cx = cl[0];  // keep the data structures in range; the step is discarded anyway
                    cy = cl[1];
                }
                uint32_t key = ((uint32_t)(cx - org_x) << 16) | (uint32_t)(cy - org_y);
                t.ckey(cur)[i] = key;
                uint32_t h = hash_cell(key, t.ht);
                for (;;) {
                    uint32_t old = atomicCAS(&t.hkeys(cur)[h], EGG_EMPTY_KEY, key);
                    if (old == EGG_EMPTY_KEY || old == key) break;
                    h = (h + 1) & (uint32_t)(t.ht - 1);
                }
                t.pslot[i] = (uint16_t)h;
                atomicAdd(&t.hmeta(cur)[h], 1u);
            }
            __syncthreads();
            {   // cell start offsets: exclusive scan of the per-slot counts
                int carry = 0;
                for (int base = 0; base < t.ht; base += EGG_WAVE) {
                    int h = base + lane;
                    int v = (int)t.hmeta(cur)[h];
                    int incl = wave_incl_scan(v, lane);
                    t.hmeta(cur)[h] = ((uint32_t)(carry + incl - v) << 16) | (uint32_t)v;
                    carry += __shfl(incl, 63, 64);
                }
            }
            for (int i = lane; i < n; i += EGG_WAVE) t.fill[i] = 0;
            __syncthreads();
            // unordered scatter, then rank inside the cell so that each cell's items ascend (L:1509)
            for (int i = lane; i < n; i += EGG_WAVE) {
                uint32_t m = t.hmeta(cur)[t.pslot[i]];
                uint32_t pos = atomicAdd(&t.fill[m >> 16], 1u);
                t.inc_tmp[(m >> 16) + pos] = (uint16_t)i;
            }
            __syncthreads();
            for (int i = lane; i < n; i += EGG_WAVE) {
                uint32_t m = t.hmeta(cur)[t.pslot[i]];
                int st = (int)(m >> 16), cn = (int)(m & 0xFFFFu);
                int rank = 0;
                for (int e = 0; e < cn; ++e) rank += (t.inc_tmp[st + e] < (uint16_t)i) ? 1 : 0;
                t.hitems(cur)[st + rank] = (uint16_t)i;
            }
            __syncthreads();
            PROF(2)  // cell hash

            // ---------------------------------------- visit lists (count, scan, fill)
            for (int i = lane; i < n; i += EGG_WAVE)
                t.fill[i] = (uint32_t)enumerate_visits<false>(t, ctx, i, nullptr);
            __syncthreads();
            PROF(3)  // visit count
            int total = block_exclusive_scan(t.fill, t.own_off(cur), n, lane);
            __syncthreads();
            max_list = max(max_list, (unsigned int)total);  // what this pass needs, even when it does not fit
            if (total > t.lcap) {
                if (lane == 0) atomicExch(&A.status->fail_overflow, 1);
                bad = true;
                // truncate so that nothing below indexes out of bounds; results are discarded
                for (int i = lane; i <= n; i += EGG_WAVE) t.own_off(cur)[i] = min(t.own_off(cur)[i], (uint32_t)t.lcap);
                __syncthreads();
            }
            if (total <= t.lcap) {
                for (int i = lane; i < n; i += EGG_WAVE)
                    enumerate_visits<true>(t, ctx, i, &t.own_ent(cur)[t.own_off(cur)[i]]);
            } else {
                for (int e = lane; e < t.lcap; e += EGG_WAVE) t.own_ent(cur)[e] = 0;
                total = t.lcap;
            }
            __syncthreads();

            PROF(4)  // visit fill
            // ------------------------------------------- budget cut, L:1657-1658
            int this_cut = 0;
            if (A.single_tile) {
                // pairs failing the mass guard (L:1601) are marked but not counted; count them out
                long long counted = 0;
                long long cut_pos = -1;
                // (serial in lane 0 only when such pairs can exist; otherwise position = count)
                bool guard_possible = false;
                for (int i = lane; i < n; i += EGG_WAVE) guard_possible |= (t.w[i] * 2.0 < eps) || !(t.w[i] == t.w[i]);
                guard_possible = __any(guard_possible);
                if (!guard_possible) {
                    if ((long long)total > budget_m) cut_pos = budget_m;  // keep entries [0, budget_m)
                } else {
                    if (lane == 0) {
                        long long cp = -1;
                        for (int i = 0; i < n && cp < 0; ++i)
                            for (uint32_t e = t.own_off(cur)[i]; e < t.own_off(cur)[i + 1]; ++e) {
                                if (t.w[i] + t.w[t.own_ent(cur)[e]] < eps) continue;
                                if (++counted >= budget_m) {
                                    cp = (long long)e + 1;
                                    break;
                                }
                            }
                        t.sc[5] = (cp >= 0 && cp < total) ? (int)cp : -1;
                    }
                    __syncthreads();
                    cut_pos = t.sc[5];
                }
                if (cut_pos >= 0 && cut_pos < total) {
                    this_cut = 1;
                    for (int i = lane; i <= n; i += EGG_WAVE)
                        t.own_off(cur)[i] = min(t.own_off(cur)[i], (uint32_t)cut_pos);
                    total = (int)cut_pos;
                    __syncthreads();
                }
            }
            if (lane == 0) {
                atomicAdd(&A.status->visits[min(pass_seq, EGG_MAX_PASSES - 1)], (unsigned long long)total);
                if (this_cut) atomicExch(&A.status->was_cut, 1);
            }

            PROF(5)  // budget
            // ---------------- incoming lists: transpose of the visit lists, ascending in self
            for (int i = lane; i < n; i += EGG_WAVE) t.fill[i] = 0;
            __syncthreads();
            for (int i = lane; i < n; i += EGG_WAVE)
                for (uint32_t e = t.own_off(cur)[i]; e < t.own_off(cur)[i + 1]; ++e)
                    atomicAdd(&t.fill[t.own_ent(cur)[e]], 1u);
            __syncthreads();
            block_exclusive_scan(t.fill, t.inc_off, n, lane);
            __syncthreads();
            for (int i = lane; i < n; i += EGG_WAVE) t.fill[i] = 0;
            __syncthreads();
            for (int i = lane; i < n; i += EGG_WAVE)
                for (uint32_t e = t.own_off(cur)[i]; e < t.own_off(cur)[i + 1]; ++e) {
                    int j = t.own_ent(cur)[e];
                    uint32_t pos = atomicAdd(&t.fill[j], 1u);
                    t.inc_tmp[t.inc_off[j] + pos] = (uint16_t)i;
                }
            __syncthreads();
            for (int i = lane; i < n; i += EGG_WAVE) {
                int st = (int)t.inc_off[i], cn = (int)t.inc_off[i + 1] - st;
                int nl = 0;
                for (int e = 0; e < cn; ++e) {
                    uint16_t cself = t.inc_tmp[st + e];
                    int rank = 0;
                    for (int f = 0; f < cn; ++f) rank += (t.inc_tmp[st + f] < cself) ? 1 : 0;
                    t.inc_ent[st + rank] = cself;
                    nl += (cself < (uint16_t)i) ? 1 : 0;
                }
                t.nlo[i] = (uint16_t)nl;
                t.ptr[i] = 0;
                t.stamp[i] = EGG_NONE;
            }
            __syncthreads();

            PROF(6)  // transpose
            // -------------------------------------- DAG execution of the pair solves
            for (int i = lane; i < n; i += EGG_WAVE) t.nxt[i] = (uint16_t)seq_entry(t, cur, i, 0);
            if (lane == 0) {
                t.sc[0] = 0;
                t.sc[1] = 0;
            }
            __syncthreads();
            for (int i = lane; i < n; i += EGG_WAVE) {
                uint32_t e = t.nxt[i];
                if (e != EGG_NONE && (e & EGG_SELF)) {
                    int p = (int)(e & EGG_IDX);
                    if (t.nxt[p] == (uint16_t)i) {
                        int pos = atomicAdd(&t.sc[0], 1);
                        t.queue(0)[pos] = (uint32_t)i | ((uint32_t)p << 16);
                    }
                }
            }
            __syncthreads();
            int qc = 0;
            unsigned int round = 0;
            int done_pairs = 0;
            for (;;) {
                const int qn = t.sc[qc];
                if (qn == 0) break;
                if (round > (unsigned int)total) break;  // cannot happen; guards against a hang
                __syncthreads();
                if (lane == 0) t.sc[qc ^ 1] = 0;
                for (int base = 0; base < qn; base += EGG_WAVE) {
                    int q = base + lane;
                    if (q < qn) {
                        uint32_t pr = t.queue(qc)[q];
                        int a = (int)(pr & 0xFFFFu), b = (int)(pr >> 16);
                        solve_pair(t, a, b, A.overlap_factor, A.collision_compliance, eps);
                        int pa = t.ptr[a] + 1, pb = t.ptr[b] + 1;
                        t.ptr[a] = (uint16_t)pa;
                        t.ptr[b] = (uint16_t)pb;
                        t.nxt[a] = (uint16_t)seq_entry(t, cur, a, pa);
                        t.nxt[b] = (uint16_t)seq_entry(t, cur, b, pb);
                        t.stamp[a] = (uint16_t)round;
                        t.stamp[b] = (uint16_t)round;
                    }
                }
                done_pairs += qn;
                __syncthreads();
                for (int base = 0; base < qn; base += EGG_WAVE) {
                    int q = base + lane;
                    if (q < qn) {
                        uint32_t pr = t.queue(qc)[q];
#pragma unroll
                        for (int side = 0; side < 2; ++side) {
                            int p = side == 0 ? (int)(pr & 0xFFFFu) : (int)(pr >> 16);
                            uint32_t e = t.nxt[p];
                            if (e == EGG_NONE) continue;
                            int o = (int)(e & EGG_IDX);
                            uint32_t eo = t.nxt[o];
                            if (eo == EGG_NONE || (int)(eo & EGG_IDX) != p) continue;
                            bool p_self = (e & EGG_SELF) != 0;
                            if (t.stamp[o] == (uint16_t)round && !p_self) continue;  // o's lane pushes it
                            int pos = atomicAdd(&t.sc[qc ^ 1], 1);
                            t.queue(qc ^ 1)[pos] = p_self ? ((uint32_t)p | ((uint32_t)o << 16))
                                                          : ((uint32_t)o | ((uint32_t)p << 16));
                        }
                    }
                }
                __syncthreads();
                qc ^= 1;
                ++round;
            }
            rounds_total += round;
            PROF(7)  // DAG
            if (done_pairs != total && !__any(bad)) {
                // lists of a discarded (bad) step may be inconsistent; otherwise this is a bug
                if (lane == 0) atomicExch(&A.status->fail_stall, 1);
            }
            __syncthreads();

            // ------------------------------ clear policy between passes, L:1905-1912
            if (c + 1 < A.n_collision_steps) {
                have_prev = 0;
                prev_uncut = 1;
            } else {
                have_prev = 1;  // hash lists and collided survive into the next sub-step (Q3)
                prev_uncut = !this_cut;
                cur ^= 1;
            }
        }

        // ------------------------------------------------ post-solve, L:1690-1693
        for (int i = lane; i < n; i += EGG_WAVE) {
            t.vx[i] = (t.x[i] - t.px[i]) / sub_delta;
            t.vy[i] = (t.y[i] - t.py[i]) / sub_delta;
        }
        __syncthreads();
    }

    PROF(8)  // post-solve
    // ------------------------------------------------------------- write back
    int slack = 0x7FFFFFFF;
    for (int k = 0; k < na; ++k) {
        int atom = A.tile_atoms[a_begin + k];
        int g0 = A.atom_offset[atom];
        int l0 = t.aoff[k], cnt = t.aoff[k + 1] - l0;
        const int32_t *cl = &t.aclaim[4 * k];
        int lo_x = 0x7FFFFFFF, lo_y = 0x7FFFFFFF, hi_x = -0x7FFFFFFF, hi_y = -0x7FFFFFFF;
        for (int q = lane; q < cnt; q += EGG_WAVE) {
            int i = l0 + q;
            if (i >= n) break;
            int g = g0 + q;
            double x = t.x[i], y = t.y[i];
            A.x_out[g] = x;
            A.y_out[g] = y;
            A.vx_out[g] = t.vx[i];
            A.vy_out[g] = t.vy[i];
            double fcx = floor(x / A.cell_size), fcy = floor(y / A.cell_size);
            int cx = (fcx >= -2.0e9 && fcx <= 2.0e9) ? (int)fcx : 0x7FFFFFF0;
            int cy = (fcy >= -2.0e9 && fcy <= 2.0e9) ? (int)fcy : 0x7FFFFFF0;
            lo_x = min(lo_x, cx);
            lo_y = min(lo_y, cy);
            hi_x = max(hi_x, cx);
            hi_y = max(hi_y, cy);
            slack = min(slack, min(min(cx - cl[0], cl[2] - cx), min(cy - cl[1], cl[3] - cy)));
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo_x = min(lo_x, __shfl_xor(lo_x, d, 64));
            lo_y = min(lo_y, __shfl_xor(lo_y, d, 64));
            hi_x = max(hi_x, __shfl_xor(hi_x, d, 64));
            hi_y = max(hi_y, __shfl_xor(hi_y, d, 64));
        }
        if (lane == 0) {
            A.atom_aabb_out[4 * atom + 0] = lo_x;
            A.atom_aabb_out[4 * atom + 1] = lo_y;
            A.atom_aabb_out[4 * atom + 2] = hi_x;
            A.atom_aabb_out[4 * atom + 3] = hi_y;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) slack = min(slack, __shfl_xor(slack, d, 64));
    bad = __any(bad);
    if (lane == 0) {
        atomicMin(&A.status->min_slack, slack);
        if (bad) atomicExch(&A.status->fail_claim, 1);
        atomicMax(&A.status->max_list, (unsigned long long)max_list);
        atomicAdd(&A.status->rounds, rounds_total);
    }
    PROF(9)  // write back
    PROF_FLUSH
}

#ifdef EGG_PROFILE
extern "C" void egg_prof_read(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(egg_prof), sizeof(egg_prof));
}
extern "C" void egg_prof_reset() {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(egg_prof), z, sizeof(z));
}
#endif

// ---------------------------------------------------------------------------
// small kernels around the step

// cells occupied by each atom at the current positions (used when tiles are formed)
extern "C" __global__ void egg_atom_bounds_kernel(const double *x, const double *y, const int32_t *atom_offset,
                                                   const int32_t *atom_count, int n_atoms, double cell_size,
                                                   int32_t *aabb_out) {
    int atom = blockIdx.x;
    if (atom >= n_atoms) return;
    int lane = threadIdx.x;
    int g0 = atom_offset[atom], cnt = atom_count[atom];
    int lo_x = 0x7FFFFFFF, lo_y = 0x7FFFFFFF, hi_x = -0x7FFFFFFF, hi_y = -0x7FFFFFFF;
    for (int q = lane; q < cnt; q += EGG_WAVE) {
        double fcx = floor(x[g0 + q] / cell_size), fcy = floor(y[g0 + q] / cell_size);
        int cx = (fcx >= -2.0e9 && fcx <= 2.0e9) ? (int)fcx : 0x7FFFFFF0;
        int cy = (fcy >= -2.0e9 && fcy <= 2.0e9) ? (int)fcy : 0x7FFFFFF0;
        lo_x = min(lo_x, cx);
        lo_y = min(lo_y, cy);
        hi_x = max(hi_x, cx);
        hi_y = max(hi_y, cy);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo_x = min(lo_x, __shfl_xor(lo_x, d, 64));
        lo_y = min(lo_y, __shfl_xor(lo_y, d, 64));
        hi_x = max(hi_x, __shfl_xor(hi_x, d, 64));
        hi_y = max(hi_y, __shfl_xor(hi_y, d, 64));
    }
    if (lane == 0) {
        aabb_out[4 * atom + 0] = lo_x;
        aabb_out[4 * atom + 1] = lo_y;
        aabb_out[4 * atom + 2] = hi_x;
        aabb_out[4 * atom + 3] = hi_y;
    }
}

// mass / radius re-derivation after a config change (L:1420-1430): mix(min, max, mass_t)
extern "C" __global__ void egg_rederive_kernel(const double *mass_t, double *inv_mass, double *radius, int n,
                                                int do_mass, double min_mass, double max_mass, int do_radius,
                                                double min_radius, double max_radius) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double tt = mass_t[i];
    if (do_mass) {
        double mass = min_mass * (1 - tt) + max_mass * tt;
        inv_mass[i] = 1 / mass;
    }
    if (do_radius) radius[i] = min_radius * (1 - tt) + max_radius * tt;
}

// get_position (L:281-295, L:1134-1148): sequential sum over the batch's white then yolk
// particles, in index order, then one division -- one thread per requested batch so the
// floating-point summation order is the reference's.
extern "C" __global__ void egg_centroid_kernel(const double *wx, const double *wy, const double *yx,
                                                const double *yy, const int32_t *w_off, const int32_t *w_cnt,
                                                const int32_t *y_off, const int32_t *y_cnt, int n, double *out_x,
                                                double *out_y) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    double x = 0, y = 0;
    int o = w_off[b], c = w_cnt[b];
    for (int k = 0; k < c; ++k) {
        x = x + wx[o + k];
        y = y + wy[o + k];
    }
    o = y_off[b];
    int c2 = y_cnt[b];
    for (int k = 0; k < c2; ++k) {
        x = x + yx[o + k];
        y = y + yy[o + k];
    }
    double cnt = (double)(c + c2);
    out_x[b] = x / cnt;
    out_y[b] = y / cnt;
}
