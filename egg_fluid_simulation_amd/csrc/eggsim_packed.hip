// eggsim_packed.hip -- the packed pipeline: gfx950 kernels of the XPBD particle step for the throughput
// regime (many tiles per CU).
//
// The fused step kernel (eggsim_step.hip) runs a tile's pair projections with a dataflow scheduler: a
// pair runs when it is the next pending pair of both its particles.  That is exact, but a wave-instruction
// of the projection then carries the one or two pairs of ONE island that happen to be ready (a default
// 157-particle blob has ~6 ready pairs at any time), so a full chip issues at ~2 % lane utilisation.
//
// Here the same dependency order is made static.  Replacing SimulationHandler:_step's sub-step loop
// (simulation_handler.lua:1821-1932, "L:") for the tiles of a launch class, per collision pass:
//
//   egg_pk_lists   one workgroup per tile.  Spatial hash (L:1486-1511) and every particle's visit list in the
//                  reference's attempt order (L:1568-1590; the rules of eggsim_tile.h), written to global
//                  memory in (self, position) order -- which IS the reference's sequential pair order.
//   egg_pk_levels  one wave per GROUP of tiles.  Walks each tile's pair sequence in order and gives every
//                  pair its level: 1 + the larger level of the previous pair of either particle (the
//                  longest path of the pair-dependency DAG).  The walk is sequential per tile, but cheap
//                  (integers), vectorised over the run of one `self` (a max-plus prefix scan) and over
//                  the tiles of the group.  Then the group's pairs are counting-sorted by level.
//   egg_pk_exec    one wave per group, the group's positions in LDS.  Level after level, 64 pairs per
//                  wave-instruction: two pairs of one level share no particle and every predecessor of
//                  a pair lies in a lower level, so any order inside a level gives the sequential result
//                  bit for bit; one wave's LDS operations execute in issue order, so no barrier is needed.
//
// Around the passes: egg_pk_begin (gather into packed order + pre-solve + follow of the first sub-step,
// L:1393-1471), egg_pk_mid (post-solve of a sub-step + pre-solve + follow of the next), egg_pk_end
// (post-solve, L:1690-1693, scatter back to particle order, per-atom cell boxes / travel), egg_pk_reduce
// (per-pass visit counts and slack: one atomic per launch instead of one per tile).
//
// All arithmetic is IEEE double in the reference's evaluation order: compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include "eggsim_device.h"

#include "eggsim_tile.h"

namespace {

// ------------------------------------------------------------------ visit lists with a visitor
// (the same rules as enum_fresh / enum_stale in eggsim_tile.h; emit(position, other) is called in attempt order)

// The packed path lays its dense cell grid out x-major (cell index = x * height + y): the three cells of one
// column of the 3x3 loop (x offset outer, y inner, L:1568-1569) are then neighbours in memory, and -- the items being
// sorted by cell -- their items form ONE contiguous run, already in the reference's order.
__device__ inline uint32_t pk_cell_meta(const Tile &t, int gh, int buf, uint32_t key) {
    if (t.use_grid) return t.cell(buf)[(int)(key >> 16) * gh + (int)(key & 0xFFFFu)];
    return cell_meta(t, buf, key);
}

template <class F>
__device__ inline int pk_visit_fresh(const Tile &t, int gh, int cur, int i, F emit) {
    const uint32_t ki = t.ckey(cur)[i];
    const uint16_t *items = t.hitems(cur);
    int count = 0;
    if (t.use_grid) {
        const uint32_t *cells = t.cell(cur);
        const int h1 = (int)(ki >> 16) * gh + (int)(ki & 0xFFFFu);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int h0 = h1 + (p - 1) * gh - 1;  // cell (x + p - 1, y - 1); claimed cells have a ring of empty cells around
            const uint32_t m0 = cells[h0], m2 = cells[h0 + 2];
            const int st = (int)(m0 >> 16), en = (int)(m2 >> 16) + (int)(m2 & 0xFFFFu);
            for (int e = st; e < en; ++e) {
                const int j = items[e];
                if (j > i) {
                    emit(count, j);
                    ++count;
                }
            }
        }
        return count;
    }
    for (int p = 0; p < 3; ++p) {
        uint32_t m[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) m[r] = cell_meta(t, cur, (uint32_t)((int)ki + (p - 1) * 65536 + (r - 1)));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int st = (int)(m[r] >> 16), cn = (int)(m[r] & 0xFFFFu);
            for (int e = 0; e < cn; ++e) {
                const int j = items[st + e];
                if (j > i) {
                    emit(count, j);
                    ++count;
                }
            }
        }
    }
    return count;
}

// Packed cell keys are (x << 16 | y), both fields < 2^15.  With kbm = kb - 0x00010001 the difference t = ka - kbm holds
// (dx + 1, dy + 1) in its two halves, so: ka lies in the 3x3 cells around kb <=> both halves are 0, 1 or 2, and then t
// ORDERS the nine cells like the reference's loop (x offset outer, y inner, L:1568-1569): slot(ka) < slot(kc) <=>
// t(ka) < t(kc).  (A borrow out of the low half leaves 0xFFFF.. there, which fails the test as it should.)
__device__ __forceinline__ bool pk_adjacent(uint32_t t) { return ((t | (t + 0x00010001u)) & 0xFFFCFFFCu) == 0u; }

// The stale pass (Q3: first pass of a later sub-step; the cell lists still hold the previous pass's entries in front of
// this pass's, and `collided` still holds the previous pass's pairs).  The rules of accept_stale (eggsim_tile.h), taken
// apart by what an entry's origin already decides -- the previous pass was not cut by the budget here (prev_uncut), so
// "in collided" == "the two OLD cells are adjacent":
//   * an OLD entry of cell c (j's old cell is c): the pair was visited before iff c is adjacent to i's old cell -- one
//     test per CELL, and a self that kept its cell skips all nine old lists.  Otherwise j counts unless its new cell comes
//     earlier in i's loop (first occurrence wins), or j < i and j's own loop met i (through i's new or old cell);
//   * a NEW entry of cell c (j's new cell is c, adjacent to i's): j < i met i in its own loop, so only j > i counts,
//     unless j's old cell comes earlier in i's loop or in the same cell (old entries stand in front), or the old cells
//     are adjacent (visited before).
template <class F>
__device__ inline int pk_visit_stale(const Tile &t, int gh, const PassCtx &c, int i, F emit) {
    const uint32_t *kn = t.ckey(c.cur), *ko = t.ckey(c.prev);
    const uint32_t kni = kn[i], koi = ko[i];
    const uint32_t knim = kni - 0x00010001u, koim = koi - 0x00010001u;
    const uint16_t *items_o = t.hitems(c.prev), *items_n = t.hitems(c.cur);
    int count = 0;
    for (int s = 0; s < 9; ++s) {
        const uint32_t ts = (uint32_t)(s / 3) << 16 | (uint32_t)(s % 3);  // this cell's place in the loop, as a key difference
        const uint32_t nk = knim + ts;                                    // = kni + (s / 3 - 1, s % 3 - 1)
        const uint32_t mn = pk_cell_meta(t, gh, c.cur, nk);
        if (!pk_adjacent(nk - koim)) {  // (else every old entry of the cell was i's partner in the previous pass)
            const uint32_t mo = pk_cell_meta(t, gh, c.prev, nk);
            const int st = (int)(mo >> 16), cn = (int)(mo & 0xFFFFu);
            for (int e = 0; e < cn; ++e) {
                const int j = (int)items_o[st + e];
                const uint32_t knj = kn[j];
                const uint32_t tn = knj - knim;
                const bool near_new = pk_adjacent(tn);
                const bool earlier = near_new && tn < ts;
                const bool met_by_j = j < i && (near_new || pk_adjacent(knj - koim));
                if (!earlier && !met_by_j) {
                    emit(count, j);
                    ++count;
                }
            }
        }
        const int st = (int)(mn >> 16), cn = (int)(mn & 0xFFFFu);
        for (int e = 0; e < cn; ++e) {
            const int j = (int)items_n[st + e];
            if (j > i) {
                const uint32_t koj = ko[j];
                const uint32_t to = koj - knim;
                if (!(pk_adjacent(to) && to <= ts) && !pk_adjacent(koj - koim)) {
                    emit(count, j);
                    ++count;
                }
            }
        }
    }
    return count;
}

// Cell records (start << 16 | count) and per-cell item lists, ascending particle index inside a cell
// (L:1509), of generation `buf`, from the packed cells in t.ckey(buf).  tmp: n words of scratch.
__device__ inline void pk_build_grid(const Tile &t, int gh, int buf, uint32_t *tmp, int tid, int nthreads, uint32_t *wtot) {
    const int n = t.n;
    for (int h = tid; h < t.ncell; h += nthreads) t.cell(buf)[h] = 0;
    if (!t.use_grid)
        for (int h = tid; h < t.ncell; h += nthreads) t.hkeys(buf)[h] = EGG_EMPTY_KEY;
    for (int i = tid; i < n; i += nthreads) t.fill[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += nthreads) {
        const uint32_t key = t.ckey(buf)[i];
        uint32_t h;
        if (t.use_grid) {
            h = (key >> 16) * (uint32_t)gh + (key & 0xFFFFu);  // x-major, see pk_cell_meta
            if (h >= (uint32_t)t.ncell) h = 0;  // only after a range / claim failure
        } else {
            h = hash_cell(key, t.ccap);
            for (;;) {
                const uint32_t old = atomicCAS(&t.hkeys(buf)[h], EGG_EMPTY_KEY, key);
                if (old == EGG_EMPTY_KEY || old == key) break;
                h = (h + 1) & (uint32_t)(t.ccap - 1);
            }
        }
        t.pslot[i] = (uint16_t)h;
        atomicAdd(&t.cell(buf)[h], 1u);
    }
    __syncthreads();
    block_exclusive_scan<true>(t.cell(buf), t.cell(buf), t.ncell, tid, nthreads, wtot);
    __syncthreads();
    for (int i = tid; i < n; i += nthreads) {
        const uint32_t m = t.cell(buf)[t.pslot[i]];
        const uint32_t pos = atomicAdd(&t.fill[m >> 16], 1u);
        tmp[(m >> 16) + pos] = (uint32_t)i;
    }
    __syncthreads();
    for (int i = tid; i < n; i += nthreads) {
        const uint32_t m = t.cell(buf)[t.pslot[i]];
        const int st = (int)(m >> 16), cn = (int)(m & 0xFFFFu);
        int rank = 0;
        for (int e = 0; e < cn; ++e) rank += (tmp[st + e] < (uint32_t)i) ? 1 : 0;
        t.hitems(buf)[st + rank] = (uint16_t)i;
    }
    __syncthreads();
}

// pre-solve (L:1393-1432) + follow constraint (L:1435-1471) of one particle, exactly as in egg_step_body
__device__ __forceinline__ void pk_pre_follow(const EggPackedArgs &A, double2 ps, double2 &v, double im, double fx, double fy,
                                              double target, double2 &out) {
    v.x = v.x * A.damping;
    v.y = v.y * A.damping;
    double x = ps.x + A.sub_delta * v.x;
    double y = ps.y + A.sub_delta * v.y;
    const double dx = fx - x, dy = fy - y;
    const double current = sqrt(dx * dx + dy * dy);
    if (im > A.eps && current > target) {
        double nx, ny;
        if (current < A.eps) {
            nx = 0.0;
            ny = 0.0;
        } else {
            nx = dx / current;
            ny = dy / current;
        }
        const double violation = current - target;
        const double lambda = violation / (im + A.follow_compliance);
        x = x + nx * lambda * im;
        y = y + ny * lambda * im;
    }
    out = make_double2(x, y);
}

#define EGG_NEG_LEVEL (-0x40000000)

}  // namespace

// ------------------------------------------------------------------------------------------------
// Packed order: which particle / atom each packed slot holds.  One workgroup per tile, run when tiles are formed.
extern "C" __global__ void __launch_bounds__(64) egg_pk_plan_kernel(EggPackedArgs A) {
    const int tile = blockIdx.x;
    if (tile >= A.n_tiles) return;
    const int4 geo = ((const int4 *)A.tile_geo)[2 * tile];
    int p = geo.x;
    for (int k = 0; k < geo.w; ++k) {
        const int atom = A.tile_atoms[geo.z + k];
        const int g0 = A.atom_offset[atom], cnt = A.atom_count[atom];
        for (int q = threadIdx.x; q < cnt; q += 64) {
            A.pk_src[p + q] = g0 + q;
            A.pk_atom[p + q] = atom;
            A.pk_aslot[p + q] = (uint16_t)k;
        }
        p += cnt;
    }
}

// Start of a step: gather into packed order, pre-solve + follow of the first sub-step.
extern "C" __global__ void __launch_bounds__(256) egg_pk_begin_kernel(EggPackedArgs A) {
    if (blockIdx.x == 0 && A.status_next) {
        // nothing touches the other status block while this step runs: give it its initial state
        unsigned long long *q = (unsigned long long *)A.status_next;
        for (int w = threadIdx.x; w < (int)(sizeof(EggStatus) / 8); w += 256) q[w] = 0ull;
        __syncthreads();
        if (threadIdx.x == 0) A.status_next->min_slack = 0x7FFFFFFF;
    }
    const int p = A.p_begin + (int)(blockIdx.x * 256 + threadIdx.x);
    if (p >= A.p_end) return;
    const int g = A.pk_src[p], atom = A.pk_atom[p];
    const double2 ps = make_double2(A.x_in[g], A.y_in[g]);
    double2 v = make_double2(A.vx_in[g], A.vy_in[g]);
    const double2 wr = make_double2(A.inv_mass[g], A.radius[g]);
    double2 out;
    pk_pre_follow(A, ps, v, wr.x, A.atom_tx[atom], A.atom_ty[atom], A.atom_fd[atom], out);
    ((double2 *)A.pk_prev)[p] = ps;
    ((double2 *)A.pk_pos)[p] = out;
    ((double2 *)A.pk_wr)[p] = wr;
}

// Between two sub-steps: post-solve of the one (L:1690-1693), pre-solve + follow of the next.
extern "C" __global__ void __launch_bounds__(256) egg_pk_mid_kernel(EggPackedArgs A) {
    const int p = A.p_begin + (int)(blockIdx.x * 256 + threadIdx.x);
    if (p >= A.p_end) return;
    const int atom = A.pk_atom[p];
    const double2 ps = ((const double2 *)A.pk_pos)[p], pv = ((const double2 *)A.pk_prev)[p];
    double2 v = make_double2((ps.x - pv.x) / A.sub_delta, (ps.y - pv.y) / A.sub_delta);
    const double im = ((const double2 *)A.pk_wr)[p].x;
    double2 out;
    pk_pre_follow(A, ps, v, im, A.atom_tx[atom], A.atom_ty[atom], A.atom_fd[atom], out);
    ((double2 *)A.pk_prev)[p] = ps;  // (the velocity itself is never stored: post-solve recomputes it from the two positions)
    ((double2 *)A.pk_pos)[p] = out;
}

// ------------------------------------------------------------------------------------------------
// One collision pass, phase 1: spatial hash + visit lists of a tile.
template <bool STALE>
__device__ __forceinline__ void egg_pk_lists_body(const EggPackedArgs &A, const int tile) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    if (tile >= A.n_tiles) return;
    Tile t;
    uint32_t *own_off, *tmp;
    uint16_t *stage;  // [n][stage_cap] partners found by the counting pass, so that the fill does not enumerate again
    {
        const size_t n = (size_t)A.nmax, a = (size_t)A.amax, cc = (size_t)A.ccap;
        unsigned char *p = smem;
        t.pos = nullptr;  // (positions are only needed for the cells: each thread keeps its first particle's in registers)
        t.wr = nullptr;  // (inverse mass, radius) stay in global memory: only the rare slow-pair test reads them per pair
        constexpr size_t G = STALE ? 2 : 1;  // (a fresh pass holds one generation of cells: 6-7 KB less for a dense tile, a fourth tile per CU)
        t.ckey_b = (uint32_t *)carve(p, G * n * 4);
        t.cell_b = (uint32_t *)carve(p, G * cc * 4);
        t.hkeys_b = (uint32_t *)carve(p, A.use_grid ? 0 : G * cc * 4);
        own_off = (uint32_t *)carve(p, (n + 1) * 4);
        t.fill = (uint32_t *)carve(p, n * 4);
        t.aclaim = (int32_t *)carve(p, a * 4 * 4);
        t.aoff = (int32_t *)carve(p, (a + 1) * 4);
        t.sc = (int32_t *)carve(p, 16 * 4);
        t.hitems_b = (uint16_t *)carve(p, G * n * 2);
        t.aslot = (uint16_t *)carve(p, n * 2);
        // (the grid builder's scratch -- a word and a 16-bit slot per particle -- shares its place with the staged partners,
        // which are first written when the grids are done: 3.8 KB less per dense tile, so that four of them leave a CU room
        // for the other stream's workgroups)
        {
            const size_t stage_bytes = egg_align16(((size_t)A.stage_cap + 1) * (n + 1) * 2), tmp_bytes = egg_align16(n * 4) + egg_align16(n * 2);
            unsigned char *q = (unsigned char *)carve(p, stage_bytes > tmp_bytes ? stage_bytes : tmp_bytes);
            stage = (uint16_t *)q;
            tmp = (uint32_t *)q;
            t.pslot = (uint16_t *)(q + egg_align16(n * 4));
        }
        t.s_n = (int)n;
        t.s_c = (int)cc;
        t.s_o = 0;
        t.s_l = 0;
        t.ccap = A.ccap;
        t.lcap = A.lcap;
        t.use_grid = A.use_grid;
        t.own_off_b = own_off;
        t.own_ent_b = nullptr;  // never read: the previous pass is never cut here (prev_uncut)
        t.inc_tmp = tmp;
    }
    __shared__ uint32_t wtot[16];

    // everything about the tile in one 32-byte record (a chain of dependent loads here -- tile -> atoms -> counts ->
    // claims -- cost five memory round trips with the whole workgroup waiting)
    // (wave-uniform: through the constant address space it arrives by ONE scalar load, beside the vector memory
    // pipeline; the record was written by the host or by an earlier launch)
    typedef const __attribute__((address_space(4))) int32_t egg_const_i32;
    egg_const_i32 *geo = (egg_const_i32 *)(uintptr_t)(A.tile_geo + 8 * (size_t)tile);
    const int p0 = geo[0], n = min(geo[1], A.nmax);
    const int a_begin = geo[2], na = min(geo[3], A.amax);
    const int org_x = geo[4], org_y = geo[5];
    const int geo_w = geo[6], geo_h = geo[7];
    t.na = na;
    t.n = n;
    t.gw = geo_w;
    const int gh = geo_h;
    t.ncell = A.use_grid ? (int)min((long long)geo_w * geo_h, (long long)A.ccap) : A.ccap;
    for (int k = tid; k < na; k += nthreads) ((int4 *)t.aclaim)[k] = ((const int4 *)A.tile_claims)[a_begin + k];
    const int cur = 0, prev = 1;  // LDS generation buffers of this launch
    uint32_t *g_ckey_cur = A.pk_ckey + (size_t)(A.substep & 1) * A.pk_stride + p0;
    const uint32_t *g_ckey_prev = A.pk_ckey + (size_t)((A.substep + 1) & 1) * A.pk_stride + p0;
    const double2 first_pos = ((const double2 *)A.pk_pos)[p0 + min(tid, max(n - 1, 0))];  // (requested now, used after the next barrier)
    for (int i = tid; i < n; i += nthreads) {
        t.aslot[i] = A.pk_aslot[p0 + i];
        if (STALE) t.ckey(prev)[i] = g_ckey_prev[i];
    }
    // Can every pair of this tile take the hand-expanded arithmetic (pair_needs_reference false for all of them)?
    // Sufficient per particle: eps / 2 <= w <= 2^298 and |overlap * r| <= 2^298, with the compliance <= 2^298: then
    // eps <= w_i + w_j, eps <= divisor <= 2^300 and min_distance^2 <= 2^600 for every pair (floating-point addition is
    // monotonic; NaN fails the comparisons).  Nearly always true; the per-pair test then drops out of the fill.
    // (masses and radii do not change inside a step: the first list pass of the step decides, the later ones read the
    // answer -- the test costs a 16-byte load per particle, as much as the positions)
    bool all_fast;
    if (A.pass_seq == 0) {
        bool mine_fast = A.collision_compliance <= 0x1p298;
        for (int i = tid; i < n; i += nthreads) {
            const double2 w = ((const double2 *)A.pk_wr)[p0 + i];
            mine_fast = mine_fast && (w.x >= A.eps * 0.5) && (w.x <= 0x1p298) && (fabs(A.overlap_factor * w.y) <= 0x1p298);
        }
        all_fast = __syncthreads_and(mine_fast) != 0;
        if (tid == 0) A.tile_fast[tile] = all_fast ? 1 : 0;
    } else {
        all_fast = A.tile_fast[tile] != 0;
        __syncthreads();  // t.aclaim is read below
    }

    // ----------------------------------- spatial hash of this pass, L:1486-1511 (claim check as in egg_step_body)
    bool bad = false;
    for (int i = tid; i < n; i += nthreads) {
        const double2 ps = i == tid ? first_pos : ((const double2 *)A.pk_pos)[p0 + i];
        const double fcx = floor(ps.x / A.cell_size);
        const double fcy = floor(ps.y / A.cell_size);
        const int32_t *cl = &t.aclaim[4 * t.aslot[i]];
        int cx = (fcx >= -2.0e9 && fcx <= 2.0e9) ? (int)fcx : 0x7FFFFFF0;
        int cy = (fcy >= -2.0e9 && fcy <= 2.0e9) ? (int)fcy : 0x7FFFFFF0;
        if (cx < cl[0] || cx > cl[2] || cy < cl[1] || cy > cl[3]) {
            bad = true;
            A.atom_fail[A.tile_atoms[a_begin + t.aslot[i]]] = 1;
            cx = cl[0];
            cy = cl[1];
        }
        const uint32_t key = ((uint32_t)(cx - org_x) << 16) | (uint32_t)(cy - org_y);
        t.ckey(cur)[i] = key;
        g_ckey_cur[i] = key;
    }
    const bool any_bad = __syncthreads_or(bad) != 0;
    int total = 0;
    uint32_t guarded = 0;
    if (!any_bad) {
        pk_build_grid(t, gh, cur, tmp, tid, nthreads, wtot);
        if (STALE) pk_build_grid(t, gh, prev, tmp, tid, nthreads, wtot);
        PassCtx ctx;
        ctx.cur = cur;
        ctx.prev = prev;
        ctx.live = STALE ? 1 : 0;
        ctx.G = 2;
        ctx.stale = STALE ? 1 : 0;
        ctx.prev_uncut = 1;  // the budget never cuts a multi-tile pass (the host checks the visit counts afterwards)
        ctx.cut_mask = 0;

        // ---------------------------------------- visit lists: count (keeping what is found), offsets
        // What the counting pass finds is kept, so that the lists are enumerated once.  A particle visits only partners with
        // a larger index, so particle i has ~all its neighbours to keep and particle n - 1 - i ~none: the two SHARE 2 x
        // stage_cap slots, i filling them from the front and its mirror image from the back (with stage_cap slots each,
        // every fifth particle of a dense tile overflowed and 63 % of the waves had such a lane, which then enumerates again
        // with the whole wave waiting).  Whether the two met in the middle is known when both are counted.
        const int SC = A.stage_cap;
        for (int i = tid; i < n; i += nthreads) {
            const int m = n - 1 - i, pair = min(i, m);
            // (a pair's slots are 2 SC + 2 words apart: with 2 SC = 32 words = 64 bytes the lanes of a wave -- consecutive
            // pairs -- would all write into two of the 32 LDS banks)
            uint16_t *slots = stage + (size_t)pair * (2 * SC + 2);
            const int cap = m == i ? SC : 2 * SC, at = i <= m ? 0 : 2 * SC - 1, dir = i <= m ? 1 : -1;
            auto keep = [&](int k, int j) {
                if (k < cap) slots[at + dir * k] = (uint16_t)j;
            };
            t.fill[i] = STALE ? (uint32_t)pk_visit_stale(t, gh, ctx, i, keep) : (uint32_t)pk_visit_fresh(t, gh, cur, i, keep);
        }
        __syncthreads();
        block_exclusive_scan<false>(t.fill, own_off, n, tid, nthreads, wtot);
        __syncthreads();
        total = (int)own_off[n];
    }
    const bool fits = total <= A.lcap;
    if (tid == 0) {
        if (any_bad) atomicExch(&A.status->fail_claim, 1);
        if (!fits) atomicExch(&A.status->fail_overflow, 1);
        const size_t slot = (size_t)min(A.pass_seq, EGG_PK_MAX_PASSES - 1) * A.n_tiles + tile;
        A.tile_need[slot] = total;  // what the pass needs, even when it does not fit
        // a tile that failed a check gets an empty stream: the later phases then leave it alone (the step is re-run)
        A.tile_total[tile] = (fits && !any_bad) ? total : 0;
    }
    // ---------------------------------------- the tile's pair stream: the visit entries of every self, in order
    if (!any_bad && fits) {
        uint32_t *gstream = A.lists + (size_t)tile * A.scap;
        const int SC = A.stage_cap;
        for (int i = tid; i < n; i += nthreads) {
            const int cnt = (int)t.fill[i];
#ifdef EGG_PROFILE
            if (!STALE && tile < 64) {
                const bool lost = (n - 1 - i == i) ? cnt > SC : cnt + (int)t.fill[n - 1 - i] > 2 * SC;
                if (lost) atomicAdd(&A.status->visits[16], 1ull);
                if (__builtin_amdgcn_ballot_w64(lost) != 0ull && (tid & 63) == 0) atomicAdd(&A.status->visits[17], 1ull);
                if ((tid & 63) == 0) atomicAdd(&A.status->visits[18], 1ull);
                atomicMax(&A.status->visits[19], (unsigned long long)cnt);
            }
#endif
            uint32_t *dst = gstream + own_off[i];
            const double2 wi = all_fast ? make_double2(0.0, 0.0) : ((const double2 *)A.pk_wr)[p0 + i];
            auto emit = [&](int k, int j) {
                if (all_fast) {  // (no pair of the tile can need the reference path or fail the mass guard: nothing to test)
                    dst[k] = (uint32_t)i | ((uint32_t)j << 16);
                    return;
                }
                const double2 wj = ((const double2 *)A.pk_wr)[p0 + j];
                const bool slow = pair_needs_reference(wi, wj, A.overlap_factor, A.collision_compliance, A.eps);
                // a pair failing the mass guard (L:1601) is marked in `collided` but leaves n_collided alone
                if (slow && wi.x + wj.x < A.eps) ++guarded;
                dst[k] = (uint32_t)i | (slow ? 0x8000u : 0u) | ((uint32_t)j << 16);
            };
            const int m = n - 1 - i, pair = min(i, m);
            const uint16_t *slots = stage + (size_t)pair * (2 * SC + 2);
            const int at = i <= m ? 0 : 2 * SC - 1, dir = i <= m ? 1 : -1;
            const bool kept = m == i ? cnt <= SC : cnt + (int)t.fill[m] <= 2 * SC;  // (the two did not meet in the middle)
            if (kept && all_fast) {
                // (a scatter costs the compute unit's vector memory path about a cycle per LANE and store instruction, whatever
                // its width: four entries per store, the address only dword-aligned -- which the compiler will not do)
                int k = 0;
                for (; k + 4 <= cnt; k += 4) {
                    typedef uint32_t egg_u4 __attribute__((ext_vector_type(4)));
                    egg_u4 v;
                    v.x = (uint32_t)i | ((uint32_t)slots[at + dir * k] << 16);
                    v.y = (uint32_t)i | ((uint32_t)slots[at + dir * (k + 1)] << 16);
                    v.z = (uint32_t)i | ((uint32_t)slots[at + dir * (k + 2)] << 16);
                    v.w = (uint32_t)i | ((uint32_t)slots[at + dir * (k + 3)] << 16);
                    uint32_t *const where = dst + k;
                    __asm__ volatile("global_store_dwordx4 %0, %1, off" : : "v"(where), "v"(v) : "memory");
                }
                for (; k < cnt; ++k) dst[k] = (uint32_t)i | ((uint32_t)slots[at + dir * k] << 16);
            } else if (kept) {
                for (int k = 0; k < cnt; ++k) emit(k, (int)slots[at + dir * k]);
            } else if (STALE) {
                PassCtx ctx;
                ctx.cur = cur;
                ctx.prev = prev;
                ctx.live = 1;
                ctx.G = 2;
                ctx.stale = 1;
                ctx.prev_uncut = 1;
                ctx.cut_mask = 0;
                pk_visit_stale(t, gh, ctx, i, emit);
            } else {
                pk_visit_fresh(t, gh, cur, i, emit);
            }
        }
    }
    // n_collided of this tile and pass: visited pairs that passed the mass guard
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) guarded += __shfl_xor(guarded, d, 64);
    if ((tid & 63) == 0) wtot[tid >> 6] = guarded;
    __syncthreads();
    if (tid == 0) {
        uint32_t gsum = 0;
        for (int w = 0; w < (nthreads + 63) / 64; ++w) gsum += wtot[w];
        A.tile_visits[(size_t)min(A.pass_seq, EGG_PK_MAX_PASSES - 1) * A.n_tiles + tile] =
            (fits && !any_bad) ? total - (int)gsum : 0;
    }
}
extern "C" __global__ void __launch_bounds__(1024) egg_pk_lists_fresh_kernel(EggPackedArgs A) { egg_pk_lists_body<false>(A, blockIdx.x); }
extern "C" __global__ void __launch_bounds__(1024) egg_pk_lists_stale_kernel(EggPackedArgs A) { egg_pk_lists_body<true>(A, blockIdx.x); }

// ------------------------------------------------------------------------------------------------
// Phase 2: levels.
//
// Sequential definition: for the pairs e = (a, b) in the reference's order, level(e) = 1 + max(last[a], last[b]),
// then last[a] = last[b] = level(e).  A run of one self a (partners b_0 .. b_m-1, all different, none equal to
// a) gives  l_k = max(l_{k-1}, last[b_k]) + 1  with l_{-1} = last[a], i.e.
// l_k = k + 1 + max(last[a], max_{j <= k}(last[b_j] - j)): a prefix maximum over the lanes that hold the run.

// End of a level pass (one wave): the deepest level goes to the status block, the level histogram becomes the first
// slot of every level in the group's sorted list (level 0 is empty) and the executor's work list: chunks of at most
// 64 pairs, each inside one level, levels ascending.  An overflowed group is left alone by the later phases (the
// step is re-run with larger tables).
// cursor (may be hist itself): receives the first slot of every level as well, for a sort inside the same kernel.
// Returns whether the group is usable (no table overflowed).
__device__ __forceinline__ bool pk_levels_finish(const EggPackedArgs &A, int g, const uint32_t *hist, int maxlev, bool over, int lane, uint32_t *cursor = nullptr) {
    const int lev_cap = A.lev_cap;
    if (over && lane == 0) atomicExch(&A.status->fail_levels, 1);
    // the deepest chain of the step (egg_stats.max_levels); most groups see a value that is already larger
    if (lane == 0 && maxlev > __hip_atomic_load(&A.status->max_level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(&A.status->max_level, maxlev);
    const int nlev = min(maxlev, lev_cap);
    uint32_t *lstart = A.lev_start + (size_t)g * (lev_cap + 2);
    uint32_t *chunks = A.chunks + (size_t)g * A.chunk_cap;
    uint32_t carry = 0, ccarry = 0;
    for (int b0 = 1; b0 <= nlev; b0 += 64) {
        const int L = b0 + lane;
        const uint32_t v = (L <= nlev) ? hist[L] : 0u;
        const uint32_t nch = (v + 63u) >> 6;
        const uint32_t incl = (uint32_t)wave_incl_scan((int)v, lane), cincl = (uint32_t)wave_incl_scan((int)nch, lane);
        const uint32_t start = carry + incl - v, cb = ccarry + cincl - nch;
        if (L <= nlev) {
            lstart[L] = start;
            if (cursor) cursor[L] = start;
            for (uint32_t c = 0; c < nch; ++c)
                if (cb + c < (uint32_t)A.chunk_cap) chunks[cb + c] = (start + 64u * c) | ((min(64u, v - 64u * c) - 1u) << 26);
        }
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        ccarry += (uint32_t)__builtin_amdgcn_readlane((int)cincl, 63);
    }
    const bool usable = !over && ccarry <= (uint32_t)A.chunk_cap && carry <= (uint32_t)A.sort_cap;
    if (lane == 0) {
        lstart[nlev + 1] = carry;
        A.grp_nchunks[g] = usable ? (int)ccarry : 0;
        A.grp_nlev[g] = usable ? nlev : 0;
    }
    return usable;
}

// The in-order walk (throughput regime: more groups than SIMDs, every instruction counts).  One workgroup per group,
// a sub-wave of 16 lanes per tile at a time; the stream is read through a small LDS window per sub-wave (one coalesced
// refill per EGG_PK_WINDOW words), so the walk itself never waits for global memory.  A turn takes the next 16 entries
// whatever selves they belong to; runs of different selves may be levelled side by side when they touch no common
// particle, which is found out on the spot: every entry stamps its two particles with the index of its run inside the
// turn (LDS minimum: the earliest run wins), reads the stamps back, and a run that finds an earlier run's stamp on one
// of its particles ends the turn in front of it (the first run never conflicts, so every turn makes progress).  The
// particles of consecutive selves are rarely the same -- neighbours in index are not neighbours in space (the batches
// are Fibonacci spirals) -- so a turn carries two to four runs.  Inside a turn each run is a segment of the
// prefix-maximum scan.
extern "C" __global__ void __launch_bounds__(256) egg_pk_levels_mr16_kernel(EggPackedArgs A) {
    constexpr int WD = 16;
    extern __shared__ __align__(16) unsigned char smem[];
    const int g = blockIdx.x;
    if (g >= A.n_groups) return;
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int4 gg = ((const int4 *)A.grp_geo)[g];
    const int t0 = __builtin_amdgcn_readfirstlane(gg.x), t1 = __builtin_amdgcn_readfirstlane(gg.y);
    const int p0 = __builtin_amdgcn_readfirstlane(gg.z), np = __builtin_amdgcn_readfirstlane(gg.w);
    const int lev_cap = A.lev_cap;
    constexpr int W = EGG_PK_WINDOW;
    const int nsubs = nthreads / WD;
    uint32_t *hist = (uint32_t *)smem;  // [lev_cap + 2] pairs per level
    unsigned char *q8 = smem + egg_align16((size_t)(lev_cap + 2) * 4);
    uint16_t *last = (uint16_t *)q8;  // [np] level of each particle's last pair
    q8 += egg_align16((size_t)np * 2);
    uint32_t *stamp = (uint32_t *)q8;  // [np] earliest run of the turn in progress that touches the particle
    q8 += egg_align16((size_t)np * 4);
    uint32_t *win_all = (uint32_t *)q8;
    __shared__ int wave_max[16], wave_over[16];
    for (int i = tid; i <= lev_cap + 1; i += nthreads) hist[i] = 0;
    for (int i = tid; i < np; i += nthreads) {
        last[i] = 0;
        stamp[i] = 0xFFFFFFFFu;
    }
    __syncthreads();
    const int subg = tid / WD, sl = tid % WD, sub = lane / WD;  // sub-wave in the workgroup / lane in it / sub-wave in the wave
    uint32_t *win = win_all + subg * W;
    int maxlev = 0;
    bool over = false;
    for (int ti = t0 + subg; ti < t1; ti += nsubs) {
        const int base = ((const int4 *)A.tile_geo)[2 * ti].x - p0;
        const int slen = A.tile_total[ti];
        const uint32_t *stream = A.lists + (size_t)ti * A.scap;
        uint16_t *lv = A.lvl + (size_t)ti * A.scap;
        // The window is refilled from registers: the words any next window can need ([q + W - WD, q + 2 W)) are
        // requested as soon as a window is in place, a whole window's walk ahead of their use.
        constexpr int PF = (W + WD) / WD;
        uint32_t pf[PF];
        int pf_base = 0;
#pragma unroll
        for (int u = 0; u < PF; ++u) pf[u] = stream[min(sl + WD * u, max(slen - 1, 0))];
        int q = 0, wlen = 0, r = 0;
        while (q + r < slen) {
            if (r + WD > wlen && q + wlen < slen) {  // fewer than WD entries left in the window and more in the stream
                q += r;
                r = 0;
                wlen = min(W, slen - q);
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int pos = pf_base + sl + WD * u - q;
                    if (pos >= 0 && pos < W) win[pos] = pf[u];
                }
                pf_base = q + W - WD;
#pragma unroll
                for (int u = 0; u < PF; ++u) pf[u] = stream[min(pf_base + sl + WD * u, slen - 1)];
            }
            const int avail = min(wlen - r, WD);  // >= 1
            const bool have = sl < avail;
            const uint32_t rec = have ? win[r + sl] : 0u;
            const int a = (int)(rec & 0x7FFFu), b = (int)((rec >> 16) & 0x7FFFu);
            // run structure of the turn: a run starts where the self changes
            const int a_prev = __builtin_amdgcn_update_dpp(-1, a, 0x111, 0xf, 0xf, false);  // row_shr:1
            const bool head = have && (sl == 0 || a != a_prev);
            const unsigned long long heads64 = __ballot(head);
            const uint32_t heads = (uint32_t)(heads64 >> (sub * WD)) & 0xFFFFu;
            const uint32_t upto = heads & (0xFFFFFFFFu >> (31 - sl));          // heads at lanes <= sl
            const int run_idx = __builtin_popcount(upto) - 1;
            const int run_start = 31 - __builtin_clz(upto | 1u);
            const int pos_in_run = sl - run_start;
            // the earliest run of the turn to touch a particle leaves its index on it
            if (have) {
                atomicMin(&stamp[base + b], (uint32_t)run_idx);
                atomicMin(&stamp[base + a], (uint32_t)run_idx);
            }
            const uint32_t sb = stamp[base + b], sa = stamp[base + a];
            const int c_raw = (int)last[base + b];
            const int xa = (int)last[base + a];
            if (have) {  // the stamps are the next turn's again
                stamp[base + b] = 0xFFFFFFFFu;
                stamp[base + a] = 0xFFFFFFFFu;
            }
            const bool clash = have && (sb < (uint32_t)run_idx || sa < (uint32_t)run_idx);
            const unsigned long long clash64 = __ballot(clash);
            const uint32_t clashes = (uint32_t)(clash64 >> (sub * WD)) & 0xFFFFu;
            // the turn ends in front of the first run that clashes with an earlier one
            int m = avail;
            if (clashes) {
                const int f = __builtin_ctz(clashes);
                m = 31 - __builtin_clz((heads & (0xFFFFFFFFu >> (31 - f))) | 1u);  // start of the run lane f is in
            }
            const bool valid = sl < m;
            // segmented inclusive prefix maximum of (level of the partner's last pair - position in the run)
            int v = valid ? c_raw - pos_in_run : EGG_NEG_LEVEL;
            {
                int t;
                t = __builtin_amdgcn_update_dpp(EGG_NEG_LEVEL, v, 0x111, 0xf, 0xf, false);  // row_shr:1
                if (pos_in_run >= 1 && (sl & 15) >= 1) v = max(v, t);
                t = __builtin_amdgcn_update_dpp(EGG_NEG_LEVEL, v, 0x112, 0xf, 0xf, false);  // row_shr:2
                if (pos_in_run >= 2 && (sl & 15) >= 2) v = max(v, t);
                t = __builtin_amdgcn_update_dpp(EGG_NEG_LEVEL, v, 0x114, 0xf, 0xf, false);  // row_shr:4
                if (pos_in_run >= 4 && (sl & 15) >= 4) v = max(v, t);
                t = __builtin_amdgcn_update_dpp(EGG_NEG_LEVEL, v, 0x118, 0xf, 0xf, false);  // row_shr:8
                if (pos_in_run >= 8 && (sl & 15) >= 8) v = max(v, t);
            }
            int l = pos_in_run + 1 + max(xa, v);
            if (valid && l > lev_cap) {  // deeper than the level table: report what is needed, keep the tables in range
                over = true;
                maxlev = max(maxlev, l);
                l = lev_cap;
            }
            if (valid) {
                const bool run_ends = (sl == m - 1) || ((heads >> (sl + 1)) & 1u);
                last[base + b] = (uint16_t)l;
                lv[q + r + sl] = (uint16_t)l;
                atomicAdd(&hist[l], 1u);
                maxlev = max(maxlev, l);
                if (run_ends) last[base + a] = (uint16_t)l;
            }
            r += m;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxlev = max(maxlev, __shfl_xor(maxlev, d, 64));
    over = __any(over);
    if (lane == 0) {
        wave_max[tid >> 6] = maxlev;
        wave_over[tid >> 6] = over ? 1 : 0;
    }
    __syncthreads();
    if (tid >= 64) return;
    for (int w = 0; w < nthreads / 64; ++w) {
        maxlev = max(maxlev, wave_max[w]);
        over = over || wave_over[w];
    }
    pk_levels_finish(A, g, hist, maxlev, over, lane);
}

// Diagnostic build (make prof, -DEGG_PROFILE): cycle stamps of the phases of egg_pk_levels_ooo_kernel, group 0 and the
// slowest group, in the unused tail of EggStatus::visits (printed by the host with EGGSIM_DEBUG=1).  Never in the product.
#ifdef EGG_PROFILE
#define EGG_STAMP(name) const unsigned long long name = __builtin_amdgcn_s_memtime()
#else
#define EGG_STAMP(name)
#endif

// The out-of-order walk (latency regime: no more groups than SIMDs, the chip waits for the longest chain).
//
// The in-order walk above is serial per tile: a dense 628-particle island with 6,500 pairs takes ~540 turns.  Here the
// stream is cut into BATCHES of 64 entries (four rows of 16 lanes); the waves of the workgroup take the batches round
// robin and level the runs of a batch in WHATEVER order their inputs become final.  That needs, per entry, the
// number of earlier stream entries that involve its self and its partner (`expect`): an entry may be levelled exactly
// when the completed-pair counters of both particles have reached those numbers.  In reference order the pairs of a
// particle p are [p visited by selves < p] [p's own run] [p visited by selves > p (stale passes only)], so three LDS
// atomic adds per batch, issued by ONE wave in stream order, hand out the numbers: partners larger than the self,
// selves, partners smaller than the self.  Inside one ds_add_rtn_u32 the lanes that hit the same address are served in
// ascending lane order on gfx950 (not an architectural promise: egg_pk_probe_lds_order_kernel checks it when a handle is
// created, and the host only chooses this kernel where it holds), which is the stream order of a batch.
//   word[p] = completed pairs of p << 16 | level of its last pair     (one 32-bit word: read and written whole)
// A run (or the piece of it inside a row) whose first pending entry finds word[self] at its expected count levels the
// longest prefix whose partners are at theirs -- the same prefix-maximum formula as above -- and publishes the new
// words; other waves see them on their next poll.  Deadlock-free: the oldest unfinished entry of the stream depends
// only on finished ones.
// (nthreads: the threads of the workgroup that take part -- all of them, or the first four waves of a fused pass)
__device__ __forceinline__ void pk_levels_ooo_body(const EggPackedArgs &A, const int g, const int nthreads) {
    extern __shared__ __align__(16) unsigned char smem[];
    EGG_STAMP(TS);
    if (g >= A.n_groups) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int4 gg = ((const int4 *)A.grp_geo)[g];
    const int t0 = __builtin_amdgcn_readfirstlane(gg.x), t1 = __builtin_amdgcn_readfirstlane(gg.y);
    const int p0 = __builtin_amdgcn_readfirstlane(gg.z), np = __builtin_amdgcn_readfirstlane(gg.w);
    const int nt = t1 - t0;  // <= 64 tiles per group
    const int lev_cap = A.lev_cap;
    uint32_t *hist = (uint32_t *)smem;  // [lev_cap + 2] pairs per level
    unsigned char *q8 = smem + egg_align16((size_t)(lev_cap + 2) * 4);
    uint32_t *word = (uint32_t *)q8;  // [np]
    q8 += egg_align16((size_t)np * 4);
    uint32_t *cnt = (uint32_t *)q8;   // [np + 64] entries seen so far per particle (the ranking pass)
    q8 += egg_align16((size_t)(np + 64) * 4);
    uint32_t *cnt2 = (uint32_t *)q8;  // [np + 64] the same for the second half of a stream, when two waves rank a tile
    q8 += egg_align16((size_t)(np + 64) * 4);
    // [nt][lev_lds_cap] the level of every stream entry.  In LDS because a level stored to global memory inside the walk
    // would make every wait for the next batch's entries a wait for those stores as well (loads and stores share one
    // counter, and the number of stores in between is not known at compile time): ~1,500 cycles per batch.
    uint16_t *llv = (uint16_t *)q8;
    const int lcap_lds = A.lev_lds_cap;
    __shared__ int tile_base[64], tile_len[64], tile_half[64];
    __shared__ int wave_max[16], wave_over[16];
    for (int i = tid; i <= lev_cap + 1; i += nthreads) hist[i] = 0;
    for (int i = tid; i < np + 64; i += nthreads) {  // (64 spare counters behind the particles: see the ranking pass)
        if (i < np) word[i] = 0;
        cnt[i] = 0;
        cnt2[i] = 0;
    }
    // With two waves to a tile the ranking is split: one wave counts the first half of the stream, the other the second
    // half from zero in counters of its own; the second half's numbers lack the first half's totals per particle, which
    // are simply what `cnt` holds when the pass is over -- the walk adds them when it sets a batch up.
    const bool split_rank = 2 * nt <= nwaves;
    if (tid < nt) {  // the group's tiles: first particle (group-local), stream length
        tile_base[tid] = ((const int4 *)A.tile_geo)[2 * (t0 + tid)].x - p0;
        int slen = A.tile_total[t0 + tid];
        if (slen > lcap_lds) {  // does not fit this launch's LDS: the step is re-run with a larger array
            atomicExch(&A.status->fail_levlds, 1);
            slen = 0;
        }
        tile_len[tid] = slen;
        tile_half[tid] = split_rank ? min(slen, ((slen >> 1) + 255) & ~255) : slen;  // (whole groups of four batches)
    }
    __syncthreads();
    EGG_STAMP(T0);
    // ---- ranking: one wave per tile, stream order.  expect(self) | expect(partner) << 16 per entry, into A.rank.
    // Every lane takes part in all three adds (a branch around an atomic costs a wait for its result): a lane whose
    // add does not apply adds 0 to a spare counter of its own.
    for (int job = wave; job < (split_rank ? 2 * nt : nt); job += nwaves) {
        const int t = split_rank ? job % nt : job, second = split_rank ? job / nt : 0;
        const int base = tile_base[t], slen = tile_len[t];
        const int ebeg = second ? tile_half[t] : 0, eend = second ? slen : tile_half[t];
        uint32_t *const cn = second ? cnt2 : cnt;
        const uint32_t *stream = A.lists + (size_t)(t0 + t) * A.scap;
        uint32_t *rank = A.rank + (size_t)(t0 + t) * A.scap;
        const int spare = np + lane;
        uint32_t nxt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) nxt[u] = stream[min(ebeg + 64 * u + lane, max(slen - 1, 0))];
        for (int e0 = ebeg; e0 < eend; e0 += 256) {
            uint32_t rec[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) rec[u] = nxt[u];
#pragma unroll
            for (int u = 0; u < 4; ++u) nxt[u] = stream[min(e0 + 256 + 64 * u + lane, slen - 1)];  // (the next four batches are on their way)
            uint32_t xs[4], xo[4];  // (all adds go out before the first result is waited for)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool valid = e0 + 64 * u + lane < eend;
                const int a = (int)(rec[u] & 0x7FFFu), b = (int)((rec[u] >> 16) & 0x7FFFu);
                const bool up = valid && b > a, down = valid && b < a;
                xo[u] = __hip_atomic_fetch_add(&cn[up ? base + b : spare], up ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __asm__ volatile("" ::: "memory");
                // (the lanes of a run hit one address and are served in lane order: consecutive numbers)
                xs[u] = __hip_atomic_fetch_add(&cn[valid ? base + a : spare], valid ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __asm__ volatile("" ::: "memory");
                if (__any(down)) {  // stale passes only
                    const uint32_t x3 = __hip_atomic_fetch_add(&cn[down ? base + b : spare], down ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __asm__ volatile("" ::: "memory");
                    xo[u] = up ? xo[u] : x3;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 64 * u + lane;
                if (e < eend) rank[e] = (xs[u] & 0xFFFFu) | (xo[u] << 16);
            }
        }
    }
    __threadfence_block();
    EGG_STAMP(T1);
    __syncthreads();
    EGG_STAMP(T2);
    // ---- the walk.  Fewer tiles than waves: the waves of a tile take its batches round robin; otherwise a wave
    // walks whole tiles alone.
    int maxlev = 0;
#ifdef EGG_PROFILE
    unsigned long long acc_read = 0, acc_dec = 0, acc_pro = 0;
    unsigned int idle_turns = 0;
#endif
    unsigned int turns = 0;
    bool over = false;
    const int sl = lane & 15, row = lane >> 4;
    const int wpt = min(2, max(1, nwaves / nt)), conc = min(nt, nwaves);  // waves per tile (one: no overlap; four slow each other's polls down -- both measured), tiles walked side by side
    const int sub = wave / conc;
    for (int t = wave % conc; t < nt && sub < wpt; t += conc) {
        const int base = tile_base[t], slen = tile_len[t];
        const uint32_t *stream = A.lists + (size_t)(t0 + t) * A.scap;
        const uint32_t *rank = A.rank + (size_t)(t0 + t) * A.scap;
        uint16_t *lv = llv + (size_t)t * lcap_lds;
        const int half = tile_half[t];
        const int nb = (slen + 63) >> 6;
        // The entries (and their ranks) of a batch are requested TWO batches of this wave ahead of their use (a batch is
        // walked in ~1,500 cycles, less than a trip to memory): two register sets, the loop handles two batches per turn.
        auto fetch = [&](int q, uint32_t &r, uint32_t &k) {
            const int e = min(q * 64 + lane, max(slen - 1, 0));
            r = stream[e];
            k = rank[e];
        };
        uint32_t recA = 0, rkA = 0, recB = 0, rkB = 0;
        if (sub < nb) fetch(sub, recA, rkA);
        if (sub + wpt < nb) fetch(sub + wpt, recB, rkB);
        auto walk_batch = [&](const uint32_t rec, const uint32_t rk, const int q) __attribute__((always_inline)) {
#ifdef EGG_PROFILE
            const unsigned long long p0 = __builtin_amdgcn_s_memtime();
            __asm__ volatile("v_mov_b32 %0, %1\n v_mov_b32 %0, %2" : "=&v"(idle_turns) : "v"(rec), "v"(rk));
            __asm__ volatile("s_nop 0" ::: "memory");
            const unsigned long long p1 = __builtin_amdgcn_s_memtime();
            acc_read += p1 - p0;
#endif
            const int e = q * 64 + lane;
            const bool valid = e < slen;
            const int a = base + (int)(rec & 0x7FFFu), b = base + (int)((rec >> 16) & 0x7FFFu);
            uint32_t ea = rk & 0xFFFFu, eb = rk >> 16;
            if (q * 64 >= half) {  // (ranked from zero by the second wave: add what the first half of the stream held)
                ea += cnt[a];
                eb += cnt[b];
            }
            // run pieces inside the row of 16 lanes
            const int a_prev = __builtin_amdgcn_update_dpp(-1, a, 0x111, 0xf, 0xf, false);  // row_shr:1
            const bool head = valid && (sl == 0 || a != a_prev);
            const uint32_t heads = (uint32_t)(__ballot(head) >> (row * 16)) & 0xFFFFu;
            const uint32_t upto = heads & (0xFFFFu >> (15 - sl));  // heads at lanes <= sl
            const int seg_start = 31 - __builtin_clz(upto | 1u);
            // (later pieces of the row get a larger offset: an UNsegmented prefix maximum of value + offset then never
            // takes its result from an earlier piece, and the scan needs no per-step segment test)
            const int seg_off = __builtin_popcount(upto) << 20;
            const uint32_t above = heads >> (sl + 1);
            const uint32_t seg_end = above ? (uint32_t)(sl + 1 + __builtin_ctz(above)) : 16u;
            const uint32_t segmask = valid ? (0xFFFFu >> (16u - seg_end)) & ~((1u << seg_start) - 1u) : 0u;
            // f: the first lane of my piece that is not levelled yet (the pending lanes are its tail [f, seg_end)); the
            // expected counts of a piece's lanes are consecutive, so every lane knows the count its piece is waiting for
            uint32_t f = valid ? (uint32_t)seg_start : 17u;
            const uint32_t eas = ea - (uint32_t)sl;                   // + f: what word[self] must hold for lane f to go
            const uint32_t ebp = (eb + 1u) << 16, eap = (ea + 1u) << 16;  // the counts this entry leaves behind
            uint16_t *lve = lv + e;
#ifdef EGG_PROFILE
            acc_pro += __builtin_amdgcn_s_memtime() - p1;
#endif
            const uint32_t seg_end_v = valid ? seg_end : 0u;  // (no pending lanes where there is no entry)
            // (ballot != 0 instead of __any: the latter materialises the lane condition as 0 / 1 and compares again)
            while (__builtin_amdgcn_ballot_w64((uint32_t)sl >= f && f < seg_end_v) != 0ull) {
                ++turns;
#ifdef EGG_PROFILE
                const unsigned long long q0 = __builtin_amdgcn_s_memtime();
#endif
                const uint32_t wa = __hip_atomic_load(&word[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t wb = __hip_atomic_load(&word[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef EGG_PROFILE
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long q1 = __builtin_amdgcn_s_memtime();
#endif
                const uint32_t R = (uint32_t)(__ballot((wb >> 16) == eb) >> (row * 16));  // lanes of my row whose partner is ready
                const uint32_t nr = segmask & ~R & ~((1u << f) - 1u);
                const uint32_t stop = min((nr ? (uint32_t)__builtin_ctz(nr) : 0xFFFFFFFFu), seg_end);
                const bool ok = (wa >> 16) == eas + f;
                const uint32_t pos = (uint32_t)sl - f;
                const bool fire = valid && ok && pos < stop - f;
#ifdef EGG_PROFILE
                if (!__any(fire)) ++idle_turns;
#endif
                if (__builtin_amdgcn_ballot_w64(fire) == 0ull) {
                    if (turns >= EGG_PK_SPIN_LIMIT) break;  // (never in a consistent stream: give up -- reported below -- instead of hanging the GPU; looked at on idle turns only)
                    __builtin_amdgcn_s_sleep(1);  // (2, 4 and 8 were slower, none at all too: the polls of four waves crowd the LDS)
                    continue;
                }
                int v = fire ? (int)(wb & 0xFFFFu) - (int)pos + seg_off : EGG_NEG_LEVEL;
                // (one instruction per step -- lanes without a source keep their own value; written by hand because the
                // compiler spells max(v, dpp(v)) as copy + move-with-DPP + max.  s_nop 1: a VGPR written by a VALU
                // instruction may be read through DPP two wait states later)
                __asm__ volatile(
                    "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                    "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                    "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
                    "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
                    : "+v"(v));
                v -= seg_off;
                if (fire) {
                    const int lraw = (int)pos + 1 + max((int)(wa & 0xFFFFu), v);
                    maxlev = max(maxlev, lraw);  // deeper than the level table: reported below, the tables stay in range
                    const uint32_t l = (uint32_t)min(lraw, lev_cap);
                    __hip_atomic_store(&word[b], ebp | l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((uint32_t)sl == stop - 1u) __hip_atomic_store(&word[a], eap | l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    *lve = (uint16_t)l;
                    atomicAdd(&hist[l], 1u);
                }
                f = (ok && f < stop) ? stop : f;
#ifdef EGG_PROFILE
                acc_dec += __builtin_amdgcn_s_memtime() - q0;
#endif
            }
        };
        for (int q = sub; q < nb; q += 2 * wpt) {
            {
                const uint32_t r = recA, k = rkA;
                if (q + 2 * wpt < nb) fetch(q + 2 * wpt, recA, rkA);
                walk_batch(r, k, q);
            }
            if (q + wpt < nb) {
                const uint32_t r = recB, k = rkB;
                if (q + 3 * wpt < nb) fetch(q + 3 * wpt, recB, rkB);
                walk_batch(r, k, q + wpt);
            }
        }
    }
    over = maxlev > lev_cap || turns >= EGG_PK_SPIN_LIMIT;  // (a wave that gave up leaves levels undefined: nothing of this group is used)
    if (turns >= EGG_PK_SPIN_LIMIT && lane == 0) atomicExch(&A.status->fail_stall, 4);
    EGG_STAMP(T3);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxlev = max(maxlev, __shfl_xor(maxlev, d, 64));
    over = __any(over);
    if (lane == 0) {
        wave_max[wave] = maxlev;
        wave_over[wave] = over ? 1 : 0;
        atomicMax(&A.status->rounds, (unsigned long long)turns);  // (diagnostic: the most turns any wave of the step's walks took)
    }
    __syncthreads();
    EGG_STAMP(T3b);
    __shared__ int usable_s;
    if (tid < 64) {
        for (int w = 0; w < nwaves; ++w) {
            maxlev = max(maxlev, wave_max[w]);
            over = over || wave_over[w];
        }
        // (hist becomes the sort's cursors in place: every lane reads its level's count before it writes the start)
        bool lds_ok = true;
        for (int t = 0; t < nt; ++t) lds_ok = lds_ok && A.tile_total[t0 + t] <= lcap_lds;
        const bool usable = pk_levels_finish(A, g, hist, maxlev, over || !lds_ok, lane, hist) && lds_ok;
        if (lane == 0) usable_s = usable ? 1 : 0;
    }
    __syncthreads();
    EGG_STAMP(T4);
    if (!usable_s) return;
    // ---- counting sort of the group's pairs by level (what egg_pk_sort_direct_kernel does for the in-order walk):
    // indices become group-local, bit 31 marks a pair
    uint32_t *sorted = A.sorted + (size_t)g * A.sort_cap;
    for (int t = 0; t < nt; ++t) {
        const uint32_t base = (uint32_t)tile_base[t];
        const int slen = tile_len[t];
        const uint32_t *stream = A.lists + (size_t)(t0 + t) * A.scap;
        const uint16_t *lv = llv + (size_t)t * lcap_lds;
        // (32 entries per lane requested before the first is used: a dense tile's stream in one or two memory round trips
        // instead of five)
        for (int e0 = 0; e0 < slen; e0 += 32 * nthreads) {
            uint32_t rec[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) rec[u] = stream[min(e0 + u * nthreads + tid, slen - 1)];
            uint32_t pos[32];  // (every cursor is asked for before the first one is used; lanes beyond the stream add 0)
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int e = e0 + u * nthreads + tid;
                const bool valid = e < slen;
                const uint32_t l = (uint32_t)lv[min(e, slen - 1)];
                pos[u] = __hip_atomic_fetch_add(&hist[l], valid ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (e0 + u * nthreads + tid < slen)
                    sorted[pos[u]] = 0x80000000u | ((rec[u] & 0x7FFFu) + base) | (rec[u] & 0x8000u) | ((((rec[u] >> 16) & 0x7FFFu) + base) << 16);
        }
    }
#ifdef EGG_PROFILE
    __syncthreads();
    EGG_STAMP(T5);
    if (tid == 0) {
        atomicMax(&A.status->visits[62], T5 - TS);
        atomicMax(&A.status->visits[61], T3 - T2);
        atomicAdd(&A.status->visits[40 + min(14ull, (T5 - TS) / 25000ull)], 1ull);  // histogram of the groups' total cycles, 25,000 per bucket
        if (g == 0) {
            const unsigned long long v[5] = {T0 - TS, T1 - T0, T3 - T2, T4 - T3b, T5 - T4};
            for (int k = 0; k < 5; ++k) A.status->visits[55 + k] = v[k];
            A.status->visits[60] = T5 - TS;
            A.status->visits[36] = acc_read;
            A.status->visits[37] = acc_dec;
            A.status->visits[38] = turns;
            A.status->visits[39] = acc_pro;
        }
    }
#endif
}

extern "C" __global__ void __launch_bounds__(1024) egg_pk_levels_ooo_kernel(EggPackedArgs A) { pk_levels_ooo_body(A, blockIdx.x, blockDim.x); }

// Does one ds_add_rtn_u32 serve the lanes that hit the same LDS address in ascending lane order?  (What
// egg_pk_levels_ooo_kernel's ranking pass relies on.)  Pseudo-random keys from key spaces of 1 .. 1000 values; every
// mismatch against the exact count of earlier lanes with the same key is counted.
extern "C" __global__ void __launch_bounds__(64) egg_pk_probe_lds_order_kernel(int trials, unsigned long long *bad) {
    __shared__ uint32_t cnt[1024];
    const int lane = threadIdx.x;
    const int nkeys = 1 + (int)((blockIdx.x * 37u) % 1000u);
    uint32_t s = 0x9E3779B9u * (blockIdx.x * 64u + lane + 1u);
    unsigned long long mism = 0;
    for (int t = 0; t < trials; ++t) {
        for (int k = lane; k < nkeys; k += 64) cnt[k] = 0;
        __syncthreads();
        s = s * 1664525u + 1013904223u;
        const uint32_t key = (((s >> 8) & 0xFFFFu) * (uint32_t)nkeys) >> 16;  // in [0, nkeys)
        const uint32_t got = __hip_atomic_fetch_add(&cnt[key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t want = 0;
        for (int j = 0; j < 64; ++j) {
            const uint32_t kj = (uint32_t)__shfl((int)key, j, 64);
            want += (j < lane && kj == key) ? 1u : 0u;
        }
        mism += got != want ? 1ull : 0ull;
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mism += __shfl_xor(mism, d, 64);
    if (lane == 0 && mism) atomicAdd(bad, mism);
}

// ------------------------------------------------------------------------------------------------
// Phase 2b: counting sort of a group's pairs by level (one workgroup per group: the walk above is one wave, this
// part has no dependencies and wants many loads in flight).  Indices become group-local; bit 31 marks a pair.
// IN_LDS: the sorted list is assembled in LDS and written out in whole lines (a scatter straight to global memory
// leaves every 64-byte line half a dozen partial writes apart in time: several times the bytes at the memory side).
template <bool IN_LDS>
__device__ __forceinline__ void egg_pk_sort_body(const EggPackedArgs &A) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *cursor = (uint32_t *)smem;  // [nlev + 2] next free slot of every level
    uint32_t *sbuf = (uint32_t *)(smem + egg_align16((size_t)(A.lev_cap + 2) * 4));
    const int g = blockIdx.x;
    if (g >= A.n_groups) return;
    const int tid = threadIdx.x;
    const int nlev = A.grp_nlev[g];
    if (nlev <= 0) return;
    const int4 gg = ((const int4 *)A.grp_geo)[g];
    const int t0 = gg.x, t1 = gg.y, p0 = gg.z;
    const uint32_t *lstart = A.lev_start + (size_t)g * (A.lev_cap + 2);
    for (int L = 1 + tid; L <= nlev + 1; L += 256) cursor[L] = lstart[L];
    __syncthreads();
    const uint32_t total = cursor[nlev + 1];
    uint32_t *sorted = A.sorted + (size_t)g * A.sort_cap;
    uint32_t *dst = IN_LDS ? sbuf : sorted;
    // a wave per tile, twelve entries per lane requested before the first is used: a sparse tile's whole stream in one
    // memory round trip
    const int lane = tid & 63;
    for (int ti = t0 + (tid >> 6); ti < t1; ti += 4) {
        const uint32_t base = (uint32_t)(((const int4 *)A.tile_geo)[2 * ti].x - p0);
        const uint32_t *stream = A.lists + (size_t)ti * A.scap;
        const uint16_t *lv = A.lvl + (size_t)ti * A.scap;
        const int slen = A.tile_total[ti];
        for (int e0 = 0; e0 < slen; e0 += 768) {
            uint32_t rec[12], l[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int e = min(e0 + 64 * u + lane, slen - 1);
                rec[u] = stream[e];
                l[u] = (uint32_t)lv[e];
            }
#pragma unroll
            for (int u = 0; u < 12; ++u)
                if (e0 + 64 * u + lane < slen) {
                    const uint32_t pos = atomicAdd(&cursor[l[u]], 1u);
                    dst[pos] = 0x80000000u | ((rec[u] & 0x7FFFu) + base) | (rec[u] & 0x8000u) | ((((rec[u] >> 16) & 0x7FFFu) + base) << 16);
                }
        }
    }
    if (IN_LDS) {
        __syncthreads();
        for (uint32_t x = (uint32_t)tid; x < total; x += 256) sorted[x] = sbuf[x];
    }
}
extern "C" __global__ void __launch_bounds__(256) egg_pk_sort_kernel(EggPackedArgs A) { egg_pk_sort_body<true>(A); }
extern "C" __global__ void __launch_bounds__(256) egg_pk_sort_direct_kernel(EggPackedArgs A) { egg_pk_sort_body<false>(A); }

// ------------------------------------------------------------------------------------------------
// Phase 3: the pair projections (L:1514-1545, L:1632-1654), chunk after chunk: at most 64 pairs of ONE level per
// wave-instruction.  The work list is static, so everything but the positions runs ahead of the arithmetic: the
// chunk descriptors come by scalar loads, the entries are requested three chunks ahead, the (inverse mass, radius)
// records of both particles two chunks ahead; the positions are read when they are needed -- from LDS.
// PREDICATED: the projection without branches on its chain (project_pair_predicated) -- for launches whose waves are
// alone on their SIMDs (time = levels x chain latency); on a full chip the branches skip work that the lanes of other
// waves could use, and win by 2 %.
template <bool PREDICATED>
__device__ __forceinline__ void egg_pk_exec_body(const EggPackedArgs &A, const int g) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (g >= A.n_groups) return;
    const int lane = threadIdx.x & 63;
    const int nch = A.grp_nchunks[g];
    if (nch <= 0) return;  // nothing to do: positions stay as they are
    const int4 gg = ((const int4 *)A.grp_geo)[g];
    const int p0 = __builtin_amdgcn_readfirstlane(gg.z), np = __builtin_amdgcn_readfirstlane(gg.w);
    double2 *lpos = (double2 *)smem;
    double2 *gpos = (double2 *)A.pk_pos + p0;
    const double2 *gwr = (const double2 *)A.pk_wr + p0;
    const uint32_t *sorted = A.sorted + (size_t)g * A.sort_cap;
    const uint32_t *chunks = A.chunks + (size_t)g * A.chunk_cap;
    // chunk descriptors are wave-uniform: read through the constant address space they arrive by scalar loads, beside
    // the vector memory pipeline (written by the previous launch; the scalar cache starts every launch clean)
    typedef const __attribute__((address_space(4))) uint32_t egg_const_u32;
    egg_const_u32 *kchunks = (egg_const_u32 *)(uintptr_t)chunks;
    // the group's positions into LDS: eight loads in flight per lane (a load-wait-store loop pays one memory
    // round trip per 64 particles)
    for (int i0 = 0; i0 < np; i0 += 512) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gpos[min(i0 + 64 * u + lane, np - 1)];
        // (unconditional stores: beyond the end the last particle is stored again -- a conditional store would
        // pull each load into its own branch and serialise the round trips)
#pragma unroll
        for (int u = 0; u < 8; ++u) lpos[min(i0 + 64 * u + lane, np - 1)] = v[u];
    }
    const double overlap = A.overlap_factor, compliance = A.collision_compliance, eps = A.eps;
    auto load_desc = [&](int c) -> uint32_t { return kchunks[min(c, nch - 1)]; };
    auto load_rec = [&](int c, uint32_t desc) -> uint32_t {  // the entries of chunk c; 0 for the lanes beyond it and the list
        const uint32_t start = desc & 0x3FFFFFFu, cnt = (c < nch) ? (desc >> 26) + 1u : 0u;
        // (unconditional load: the lanes beyond the chunk read the words behind it -- the list has 64 spare words -- and
        // drop them; a load under a lane mask costs a branch per chunk)
        const uint32_t w = sorted[start + (uint32_t)lane];
        return ((uint32_t)lane < cnt) ? w : 0u;
    };
    // Software pipeline over the static work list.  A chunk of a dense island is one level of its dependency chain --
    // a few hundred cycles -- and every stage of the pipeline is a memory round trip of about that length: descriptor
    // (scalar) -> entries -> the pairs' particle constants (gather) -> the position-independent part of the projection.
    // Each stage therefore runs TWO chunks ahead of its consumer (one chunk ahead, the chunk in progress waited for
    // the stage before it nearly every time).  The rings of eight are indexed with compile-time constants (the loop
    // is unrolled by eight), so nothing is copied and no load is waited for before its chunk is due.
    constexpr int R = 8, D_DSC = 7, D_REC = 5, D_WR = 3, D_PC = 1;
    uint32_t dsc[R], rec[R];
    double2 wa[R], wb[R], pc[R];  // pc: what a pair's projection needs that does not depend on positions -- the refined
                                  // reciprocal of its divisor and its minimum distance -- computed ahead, off the
                                  // dependent chain of the chunk in progress
    auto pair_constants = [&](double2 a, double2 b) {
        return make_double2(egg_rcp_refined((a.x + b.x) + compliance), overlap * (a.y + b.y));
    };
#pragma unroll
    for (int u = 0; u < D_DSC; ++u) dsc[u] = load_desc(u);
#pragma unroll
    for (int u = 0; u < D_REC; ++u) rec[u] = load_rec(u, dsc[u]);
#pragma unroll
    for (int u = 0; u < D_WR; ++u) {
        wa[u] = gwr[rec[u] & 0x7FFFu];
        wb[u] = gwr[(rec[u] >> 16) & 0x7FFFu];
    }
#pragma unroll
    for (int u = 0; u < D_PC; ++u) pc[u] = pair_constants(wa[u], wb[u]);
    if (PREDICATED) {
        // One wave alone on its SIMD issues one instruction at a time, and every instruction between the last add of a
        // level and the first subtraction of the next is time on the island's dependent chain.  The order of a turn is
        // therefore fixed by hand (scheduling barriers; left alone the compiler put the ~35 instructions of the look-ahead
        // stages between a level's last add and its stores):
        //   stores of chunk c-1 | loads of chunk c | the look-ahead stages, in the shadow of that LDS turn-around |
        //   the projection of chunk c (the descriptor's scalar load behind its first instruction: a scalar load in
        //   flight would be waited for together with the positions)
        // (a value that passed through EGG_PIN is computed neither earlier nor later than that point: pure arithmetic is
        // otherwise free to move across a scheduling barrier, and to sink into the next turn)
#define EGG_PIN(x) __asm__ volatile("" : "+v"(x))
        typedef double egg_v2d __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) egg_v2d egg_lds_d2;  // (32-bit LDS addresses: a select between two of them is one instruction)
        egg_lds_d2 *const lp = (egg_lds_d2 *)lpos;
        egg_lds_d2 *const spare = lp + np + lane;  // lanes with nothing to store write their own spare slot
        double md2s[R];
#pragma unroll
        for (int u = 0; u < D_PC; ++u) md2s[u] = pc[u].y * pc[u].y;
        egg_lds_d2 *qa = lp + (rec[0] & 0x7FFFu), *qb = lp + ((rec[0] >> 16) & 0x7FFFu);
        uint32_t flags = rec[0] & 0x80008000u;  // bit 31: the lane has a pair, bit 15: the pair takes the reference path
        // (every load of the prologue is waited for HERE: with loads pending on entry the compiler's wait-count pass cannot
        // tell how many are in flight at the loop head and waits for all of them -- including the ones a turn old -- once
        // per trip of the unrolled loop)
#pragma unroll
        for (int u = 0; u < D_REC; ++u) EGG_PIN(rec[u]);
#pragma unroll
        for (int u = 0; u < D_WR; ++u) {
            EGG_PIN(wa[u].x);
            EGG_PIN(wa[u].y);
            EGG_PIN(wb[u].x);
            EGG_PIN(wb[u].y);
        }
        for (int c0 = 0; c0 < nch; c0 += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int c = c0 + u;
                // (lanes without a pair read particle 0 of the group and write nothing)
                const egg_v2d va = *qa, vb = *qb;
                double2 pa = make_double2(va.x, va.y), pb = make_double2(vb.x, vb.y);
                __builtin_amdgcn_sched_barrier(0);
                uint32_t r1 = rec[(u + 1) & (R - 1)];
                double2 a1 = wa[(u + D_PC) & (R - 1)], b1 = wb[(u + D_PC) & (R - 1)];
                EGG_PIN(r1);
                EGG_PIN(a1.x);
                EGG_PIN(a1.y);
                EGG_PIN(b1.x);
                EGG_PIN(b1.y);
                rec[(u + D_REC) & (R - 1)] = load_rec(c + D_REC, dsc[(u + D_REC) & (R - 1)]);
                wa[(u + D_WR) & (R - 1)] = gwr[rec[(u + D_WR) & (R - 1)] & 0x7FFFu];
                wb[(u + D_WR) & (R - 1)] = gwr[(rec[(u + D_WR) & (R - 1)] >> 16) & 0x7FFFu];
                wa[(u + D_PC) & (R - 1)] = a1;
                wb[(u + D_PC) & (R - 1)] = b1;
                double2 pc1 = pair_constants(a1, b1);
                double md21 = pc1.y * pc1.y;
                egg_lds_d2 *qa1 = lp + (r1 & 0x7FFFu), *qb1 = lp + ((r1 >> 16) & 0x7FFFu);
                uint32_t flags1 = r1 & 0x80008000u;
                EGG_PIN(pc1.x);
                EGG_PIN(pc1.y);
                EGG_PIN(md21);
                EGG_PIN(qa1);
                EGG_PIN(qb1);
                EGG_PIN(flags1);
                pc[(u + D_PC) & (R - 1)] = pc1;
                md2s[(u + D_PC) & (R - 1)] = md21;
                __builtin_amdgcn_sched_barrier(0);
                const bool store = project_pair_predicated(
                    [&]() { return A.atom_batch[A.pk_atom[p0 + (int)(qa - lp)]] == A.atom_batch[A.pk_atom[p0 + (int)(qb - lp)]]; },
                    [&]() {
                        __builtin_amdgcn_sched_barrier(0);
                        dsc[(u + D_DSC) & (R - 1)] = load_desc(c + D_DSC);
                    },
                    [&](double2 &wra, double2 &wrb) {
                        wra = wa[u];
                        wrb = wb[u];
                    },
                    (flags >> 31) != 0, (flags & 0x8000u) != 0, pa, pb, wa[u].x, wb[u].x, (wa[u].x + wb[u].x) + compliance, pc[u], md2s[u], overlap,
                    compliance, eps);
                {
                    egg_v2d oa, ob;
                    oa.x = pa.x;
                    oa.y = pa.y;
                    ob.x = pb.x;
                    ob.y = pb.y;
                    *(store ? qa : spare) = oa;
                    *(store ? qb : spare) = ob;
                }
                __builtin_amdgcn_sched_barrier(0);
                qa = qa1;
                qb = qb1;
                flags = flags1;
            }
        }
#undef EGG_PIN
    } else {
        for (int c0 = 0; c0 < nch; c0 += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int c = c0 + u;
                rec[(u + D_REC) & (R - 1)] = load_rec(c + D_REC, dsc[(u + D_REC) & (R - 1)]);
                dsc[(u + D_DSC) & (R - 1)] = load_desc(c + D_DSC);
                wa[(u + D_WR) & (R - 1)] = gwr[rec[(u + D_WR) & (R - 1)] & 0x7FFFu];
                wb[(u + D_WR) & (R - 1)] = gwr[(rec[(u + D_WR) & (R - 1)] >> 16) & 0x7FFFu];
                pc[(u + D_PC) & (R - 1)] = pair_constants(wa[(u + D_PC) & (R - 1)], wb[(u + D_PC) & (R - 1)]);
                const uint32_t r0 = rec[u];
                if (r0 >> 31) {
                    const int ga = (int)(r0 & 0x7FFFu), gb = (int)((r0 >> 16) & 0x7FFFu);
                    double2 pa = lpos[ga], pb = lpos[gb];
                    project_pair<true>([&]() { return A.atom_batch[A.pk_atom[p0 + ga]] == A.atom_batch[A.pk_atom[p0 + gb]]; },
                                       (r0 & 0x8000u) != 0, pa, pb, wa[u], wb[u], pc[u], overlap, compliance, eps);
                    lpos[ga] = pa;
                    lpos[gb] = pb;
                }
            }
        }
    }
    for (int i = lane; i < np; i += 64) gpos[i] = lpos[i];
}
extern "C" __global__ void __launch_bounds__(64) egg_pk_exec_kernel(EggPackedArgs A) { egg_pk_exec_body<false>(A, blockIdx.x); }
extern "C" __global__ void __launch_bounds__(64) egg_pk_exec_chain_kernel(EggPackedArgs A) { egg_pk_exec_body<true>(A, blockIdx.x); }

// ------------------------------------------------------------------------------------------------
// The executor of the fused pass, split over TWO waves on two SIMDs.
//
// A lone wave issues one instruction every ~4.5-5.4 cycles whatever it is, so on the island's dependent chain every
// instruction counts -- and a third of egg_pk_exec_body's turn (37 of 99 instructions: chunk descriptors, entries,
// the gathers of the inverse masses, the pairs' reciprocals and minimum distances, LDS addresses) has nothing to do with
// the chain.  Here a HELPER wave does all of that, a few chunks ahead, and leaves per lane and chunk a ready-made record
// in an LDS ring: the two LDS addresses (+ flags), (inverse masses), (reciprocal of the divisor, minimum distance),
// (divisor, minimum distance squared).  The EXECUTOR's turn is then: two stores, two position loads, four record loads
// for the next chunk in the shadow of that LDS turn-around, the projection.  Measured on the bare chain
// (scripts/micro/chain_floor.hip): 420 cycles per level for the chain alone, 475-515 as this consumer, 635 for
// egg_pk_exec_body in place.
// Flow control through two LDS words: `produced` (chunks the helper has finished; it writes the records, then the
// word -- one wave's LDS operations execute in order) and `consumed` (records the executor has read, published every
// fourth chunk); the helper never runs more than EGG_PK_RING chunks ahead of `consumed`.
typedef double egg_v2d __attribute__((ext_vector_type(2)));
typedef uint32_t egg_v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) egg_v2d egg_lds_v2d;
typedef __attribute__((address_space(3))) uint32_t egg_lds_u32;
struct PkRing {
    egg_v2d *a, *c, *d;  // [EGG_PK_RING][64] (wa, wb) | (1 / divisor refined, minimum distance) | (divisor, minimum distance^2)
    egg_v2u *q;          // [EGG_PK_RING][64] LDS address of a | has pair | slow << 1, LDS address of b
    // (LDS pointers by type: through a generic pointer a volatile access becomes a FLAT instruction, which reaches the LDS
    // by another path than the ds_ instructions around it -- and the records must land before the word that announces them)
    volatile egg_lds_u32 *produced, *consumed;
};
__device__ __forceinline__ PkRing pk_ring_at(unsigned char *base, uint32_t *produced, uint32_t *consumed) {
    PkRing r;
    r.a = (egg_v2d *)base;
    r.c = r.a + EGG_PK_RING * 64;
    r.d = r.c + EGG_PK_RING * 64;
    r.q = (egg_v2u *)(r.d + EGG_PK_RING * 64);
    r.produced = (volatile egg_lds_u32 *)produced;
    r.consumed = (volatile egg_lds_u32 *)consumed;
    return r;
}

__device__ __forceinline__ void egg_pk_exec_helper(const EggPackedArgs &A, const int g, const PkRing ring) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int nch = A.grp_nchunks[g];
    if (nch <= 0) return;
    const int nch_pad = (nch + 7) & ~7;  // (the executor's loop is unrolled by eight: the chunks behind the list are empty ones)
    const int4 gg = ((const int4 *)A.grp_geo)[g];
    const int p0 = __builtin_amdgcn_readfirstlane(gg.z);
    egg_lds_v2d *const lp = (egg_lds_v2d *)smem;
    const double2 *gwr = (const double2 *)A.pk_wr + p0;
    const uint32_t *sorted = A.sorted + (size_t)g * A.sort_cap;
    typedef const __attribute__((address_space(4))) uint32_t egg_const_u32;
    egg_const_u32 *kchunks = (egg_const_u32 *)(uintptr_t)(A.chunks + (size_t)g * A.chunk_cap);
    const double overlap = A.overlap_factor, compliance = A.collision_compliance;
    auto load_desc = [&](int c) -> uint32_t { return kchunks[min(c, nch - 1)]; };
    auto load_rec = [&](int c, uint32_t desc) -> uint32_t {
        const uint32_t start = desc & 0x3FFFFFFu, cnt = (c < nch) ? (desc >> 26) + 1u : 0u;
        const uint32_t w = sorted[start + (uint32_t)lane];
        return ((uint32_t)lane < cnt) ? w : 0u;
    };
    constexpr int R = 8, D_DSC = 6, D_REC = 4, D_WR = 2;  // every stage two chunks ahead of its consumer, as in egg_pk_exec_body
    uint32_t dsc[R], rec[R];
    double2 wa[R], wb[R];
#pragma unroll
    for (int u = 0; u < D_DSC; ++u) dsc[u] = load_desc(u);
#pragma unroll
    for (int u = 0; u < D_REC; ++u) rec[u] = load_rec(u, dsc[u]);
#pragma unroll
    for (int u = 0; u < D_WR; ++u) {
        wa[u] = gwr[rec[u] & 0x7FFFu];
        wb[u] = gwr[(rec[u] >> 16) & 0x7FFFu];
    }
    int room = EGG_PK_RING;  // chunks below this index may be written: consumed + EGG_PK_RING
    unsigned spins = 0;       // polls spent waiting for the executor, over the whole list (a guard against hanging the GPU)
    for (int k0 = 0; k0 < nch_pad; k0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int k = k0 + u;
            rec[(u + D_REC) & (R - 1)] = load_rec(k + D_REC, dsc[(u + D_REC) & (R - 1)]);
            dsc[(u + D_DSC) & (R - 1)] = load_desc(k + D_DSC);
            wa[(u + D_WR) & (R - 1)] = gwr[rec[(u + D_WR) & (R - 1)] & 0x7FFFu];
            wb[(u + D_WR) & (R - 1)] = gwr[(rec[(u + D_WR) & (R - 1)] >> 16) & 0x7FFFu];
            const uint32_t r0 = rec[u];
            const double2 a = wa[u], b = wb[u];
            egg_v2d ra, rc, rd;
            ra.x = a.x;
            ra.y = b.x;
            const double divisor = (a.x + b.x) + compliance;
            rc.x = egg_rcp_refined(divisor);
            rc.y = overlap * (a.y + b.y);
            rd.x = divisor;
            rd.y = rc.y * rc.y;
            egg_v2u rq;  // (lanes without a pair: particle 0 of the group, read and never stored)
            rq.x = (uint32_t)(uintptr_t)(lp + (r0 & 0x7FFFu)) | (r0 >> 31) | ((r0 >> 14) & 2u);
            rq.y = (uint32_t)(uintptr_t)(lp + ((r0 >> 16) & 0x7FFFu));
            if (k >= room) {  // the ring is full: wait for the executor
                for (;;) {
                    room = (int)__builtin_amdgcn_readfirstlane(*ring.consumed) + EGG_PK_RING;
                    if (k < room || ++spins >= EGG_PK_SPIN_LIMIT) break;  // (the second: the executor is gone -- reported below)
                    __builtin_amdgcn_s_sleep(1);  // (4, 16, 32 were slower: the executor runs dry when the helper reacts late)
                }
            }
            const int slot = (u & (EGG_PK_RING - 1)) * 64 + lane;
            ring.a[slot] = ra;
            ring.c[slot] = rc;
            ring.d[slot] = rd;
            ring.q[slot] = rq;
            __asm__ volatile("" ::: "memory");
            *ring.produced = (uint32_t)(k + 1);
            __asm__ volatile("" ::: "memory");
        }
    }
    if (spins >= EGG_PK_SPIN_LIMIT && lane == 0) atomicExch(&A.status->fail_stall, 4);
}

__device__ __forceinline__ void egg_pk_exec_consumer(const EggPackedArgs &A, const int g, const PkRing ring) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int nch = A.grp_nchunks[g];
    if (nch <= 0) return;  // nothing to do: positions stay as they are
    const int nch_pad = (nch + 7) & ~7;
    const int4 gg = ((const int4 *)A.grp_geo)[g];
    const int p0 = __builtin_amdgcn_readfirstlane(gg.z), np = __builtin_amdgcn_readfirstlane(gg.w);
    double2 *lpos = (double2 *)smem;
    double2 *gpos = (double2 *)A.pk_pos + p0;
    for (int i0 = 0; i0 < np; i0 += 512) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gpos[min(i0 + 64 * u + lane, np - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) lpos[min(i0 + 64 * u + lane, np - 1)] = v[u];
    }
    const double overlap = A.overlap_factor, compliance = A.collision_compliance, eps = A.eps;
    egg_lds_v2d *const lp = (egg_lds_v2d *)smem;
    egg_lds_v2d *const spare = lp + np + lane;  // lanes with nothing to store write their own spare slot
    int seen = 0;  // chunks the helper is known to have finished
#ifdef EGG_PROFILE
    unsigned long long waited = 0, polls = 0;
#endif
    auto wait_for = [&](int chunk) {
        if (__builtin_expect(seen <= chunk, 0)) {
#ifdef EGG_PROFILE
            const unsigned long long w0 = __builtin_amdgcn_s_memtime();
            ++polls;
#endif
            for (unsigned spins = 0;; ++spins) {
                seen = (int)__builtin_amdgcn_readfirstlane(*ring.produced);
                if (seen > chunk) break;
                if (spins >= EGG_PK_SPIN_LIMIT) {  // (the helper is gone: give up loudly -- the step is an error then; the rest of
                    if (lane == 0) atomicExch(&A.status->fail_stall, 4);  // the loop runs through on whatever the ring holds)
                    seen = 0x7FFFFFFF;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#ifdef EGG_PROFILE
            waited += __builtin_amdgcn_s_memtime() - w0;
#endif
        }
        __asm__ volatile("" ::: "memory");
    };
    wait_for(0);
    egg_v2d ra = ring.a[lane], rc = ring.c[lane], rd = ring.d[lane];
    egg_v2u rq = ring.q[lane];
    // (the helper's progress is read with every chunk's records and looked at a chunk later, when it has long arrived: a
    // look of its own is an LDS round trip on the executor's path -- 117 of them per 432 chunks, 8 % of its time)
    uint32_t peek = 0;
    for (int c0 = 0; c0 < nch_pad; c0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            egg_lds_v2d *const qa = (egg_lds_v2d *)(uintptr_t)(rq.x & ~15u), *const qb = (egg_lds_v2d *)(uintptr_t)rq.y;
            const bool has_pair = (rq.x & 1u) != 0, slow = (rq.x & 2u) != 0;
            const egg_v2d va = *qa, vb = *qb;
            __builtin_amdgcn_sched_barrier(0);
            // the next chunk's record, in the shadow of the loads above
            egg_v2d ra1 = ra, rc1 = rc, rd1 = rd;
            egg_v2u rq1 = rq;
            if (u < 7 || c + 1 < nch_pad) {
                seen = max(seen, (int)__builtin_amdgcn_readfirstlane(peek));
                wait_for(c + 1);
                peek = *ring.produced;
                const int slot = ((u + 1) & (EGG_PK_RING - 1)) * 64 + lane;
                ra1 = ring.a[slot];
                rc1 = ring.c[slot];
                rd1 = ring.d[slot];
                rq1 = ring.q[slot];
            }
            __builtin_amdgcn_sched_barrier(0);
            double2 pa = make_double2(va.x, va.y), pb = make_double2(vb.x, vb.y);
            const bool store = project_pair_predicated(
                [&]() { return A.atom_batch[A.pk_atom[p0 + (int)(qa - lp)]] == A.atom_batch[A.pk_atom[p0 + (int)(qb - lp)]]; }, []() {},
                [&](double2 &wra, double2 &wrb) {  // (the rare pair on the reference path fetches its radii itself)
                    wra = ((const double2 *)A.pk_wr)[p0 + (int)(qa - lp)];
                    wrb = ((const double2 *)A.pk_wr)[p0 + (int)(qb - lp)];
                },
                has_pair, slow, pa, pb, ra.x, ra.y, rd.x, make_double2(rc.x, rc.y), rd.y, overlap, compliance, eps);
            {
                egg_v2d oa, ob;
                oa.x = pa.x;
                oa.y = pa.y;
                ob.x = pb.x;
                ob.y = pb.y;
                *(store ? qa : spare) = oa;
                *(store ? qb : spare) = ob;
            }
            if (u == 3 || u == 7) {  // records up to chunk c + 1 are in registers: their slots may be written again
                __asm__ volatile("" ::: "memory");
                *ring.consumed = (uint32_t)(c + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            ra = ra1;
            rc = rc1;
            rd = rd1;
            rq = rq1;
        }
    }
    for (int i = lane; i < np; i += 64) gpos[i] = lpos[i];
#ifdef EGG_PROFILE
    if (g == 0 && lane == 0) {
        A.status->visits[27] = waited;
        A.status->visits[26] = polls;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Levels, sort and executor of a group in ONE launch (the latency regime: dense islands, groups no more than SIMDs).
// Against two launches this saves a launch boundary per pass and -- more -- the executor's wait for the slowest group
// of the walk: a group's projections start the moment ITS levels are sorted.  The phases reuse the dynamic LDS from its
// base; the waves the executor does not need exit (a finished wave no longer counts at the workgroup's barriers).
// (The list kernel stays a launch of its own: its 640-thread workgroups at 42 registers fit three per CU; with this
// kernel's registers only two would, and the second half of the islands would wait for a whole pass of the first --
// measured: 3.4 ms per step against 2.4.)
extern "C" __global__ void __launch_bounds__(512) egg_pk_levexec_kernel(EggPackedArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int g = blockIdx.x;
    if (g >= A.n_groups) return;
    pk_levels_ooo_body(A, g, (int)blockDim.x);  // four waves for up to two tiles, eight for up to four
    __threadfence_block();  // (the sorted list and the chunk descriptors just stored are read back below)
    // Which waves become the executor and its helper?  Two groups share a compute unit, each with a wave on every SIMD.
    // An executor wave is bound by its own instruction issue, so two of them on one SIMD run at half speed each -- with
    // "wave 0" for everybody that happened on a quarter of the compute units (780-830 cycles per level there against
    // 640).  So the executor and helper waves of a compute unit claim their SIMDs in a per-unit word in global memory:
    // a group takes SIMDs no other group's executor or helper is on, and gives them back when it is done.
    __shared__ uint32_t simd_of_wave[8];
    __shared__ int exec_wave_s, help_wave_s;
    __shared__ uint32_t claim_key_s, claim_exec_s, claim_help_s;
    __shared__ uint32_t ring_produced, ring_consumed;
    const int wave = threadIdx.x >> 6;
    const uint32_t hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // HW_REG_HW_ID: SIMD_ID 5:4, CU_ID 11:8, SH_ID 12, SE_ID 15:13
    if ((threadIdx.x & 63) == 0) simd_of_wave[wave] = (hw_id >> 4) & 3u;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID: XCC_ID 3:0
        const uint32_t key = ((hw_id >> 8) & 0xFFu) | (xcc << 8);
        const int nwaves = (int)(blockDim.x >> 6);
        uint32_t present = 0;
        for (int w = 0; w < nwaves; ++w) present |= 1u << simd_of_wave[w];
        uint32_t got[2] = {0, 0};
        uint32_t seen = __hip_atomic_load(&A.simd_claims[key], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int role = 0; role < 2; ++role)
            for (int tries = 0; tries < 8; ++tries) {
                const uint32_t free_simds = present & ~seen & ~got[0] & 0xFu;
                if (!free_simds) break;  // every SIMD we have a wave on is taken (more than two groups on the unit): share one, unclaimed
                const uint32_t want = free_simds & (0u - free_simds);
                const uint32_t prev = atomicOr(&A.simd_claims[key], want);
                seen = prev | want;
                if (!(prev & want)) {
                    got[role] = want;
                    break;
                }
            }
        int ew = -1, hw = -1;
        for (int w = 0; w < nwaves; ++w) {
            if (got[0] && ew < 0 && (1u << simd_of_wave[w]) == got[0]) ew = w;
            if (got[1] && hw < 0 && (1u << simd_of_wave[w]) == got[1]) hw = w;
        }
        if (ew < 0) ew = 0;
        if (hw < 0 || hw == ew) hw = (ew + 1) % nwaves;
        exec_wave_s = ew;
        help_wave_s = hw;
        claim_key_s = key;
        claim_exec_s = got[0];
        claim_help_s = got[1];
        ring_produced = 0;
        ring_consumed = 0;
    }
    __syncthreads();
    const int np = __builtin_amdgcn_readfirstlane(((const int4 *)A.grp_geo)[g].w);
    const PkRing ring = pk_ring_at(smem + egg_align16((size_t)(np + 64) * 16), &ring_produced, &ring_consumed);
    if (wave == help_wave_s) {
        __builtin_amdgcn_s_dcache_inv();  // (the chunk descriptors come by scalar loads: written by this launch, a moment ago)
        egg_pk_exec_helper(A, g, ring);
        if ((threadIdx.x & 63) == 0 && claim_help_s) atomicAnd(&A.simd_claims[claim_key_s], ~claim_help_s);
        return;
    }
    if (wave != exec_wave_s) return;
    EGG_STAMP(E0);
#ifdef EGG_PROFILE
    const unsigned long long R0 = __builtin_amdgcn_s_memrealtime();
#endif
    __builtin_amdgcn_s_setprio(3);  // (the executor's instructions first wherever its SIMD is shared: -1 % of a late step)
    egg_pk_exec_consumer(A, g, ring);
    if ((threadIdx.x & 63) == 0 && claim_exec_s) atomicAnd(&A.simd_claims[claim_key_s], ~claim_exec_s);
#ifdef EGG_PROFILE
    __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    EGG_STAMP(E1);
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long nch = (unsigned long long)max(A.grp_nchunks[g], 1);
        if (g == 0) {
            A.status->visits[30] = E1 - E0;
            A.status->visits[31] = nch;
            A.status->visits[28] = __builtin_amdgcn_s_memrealtime() - R0;  // 100 MHz ticks of the same interval
        }
        atomicMax(&A.status->visits[32], (E1 - E0) * 16ull / nch);  // slowest group, cycles per chunk x 16
        const uint32_t simd = (hw_id >> 4) & 3u;
        if (simd < 3) atomicAdd(&A.status->visits[33 + simd], 1ull);  // executor waves per SIMD (the rest sit on SIMD 3)
        if (!claim_exec_s) atomicAdd(&A.status->visits[29], 1ull);     // executor waves that found no free SIMD
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// End of a step: post-solve of the last sub-step (L:1690-1693), scatter to particle order, per-atom cell boxes
// and last-sub-step travel (what the host sizes the next step's claims with), slack to the claim edges.
extern "C" __global__ void __launch_bounds__(256) egg_pk_end_kernel(EggPackedArgs A) {
    const int tile = blockIdx.x;
    if (tile >= A.n_tiles) return;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int4 geo = ((const int4 *)A.tile_geo)[2 * tile];
    const int a_begin = geo.z, na = geo.w;
    __shared__ int32_t red[8];
    __shared__ int32_t wslack[4];
    int slack = 0x7FFFFFFF;
    int p = geo.x;
    for (int k = 0; k < na; ++k) {
        const int atom = A.tile_atoms[a_begin + k];
        const int cnt = A.atom_count[atom];
        __syncthreads();
        if (tid < 8) red[tid] = (tid < 2) ? 0x7FFFFFFF : (tid < 4) ? -0x7FFFFFFF : 0;
        __syncthreads();
        const int4 clv = ((const int4 *)A.tile_claims)[a_begin + k];
        const int32_t cl[4] = {clv.x, clv.y, clv.z, clv.w};
        int lo_x = 0x7FFFFFFF, lo_y = 0x7FFFFFFF, hi_x = -0x7FFFFFFF, hi_y = -0x7FFFFFFF;
        int d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        for (int q = tid; q < cnt; q += nthreads) {
            const int g = A.pk_src[p + q];
            const double2 ps = ((const double2 *)A.pk_pos)[p + q], pv = ((const double2 *)A.pk_prev)[p + q];
            const double2 v = make_double2((ps.x - pv.x) / A.sub_delta, (ps.y - pv.y) / A.sub_delta);
            A.x_out[g] = ps.x;
            A.y_out[g] = ps.y;
            A.vx_out[g] = v.x;
            A.vy_out[g] = v.y;
            const double fcx = floor(ps.x / A.cell_size), fcy = floor(ps.y / A.cell_size);
            const int cx = (fcx >= -2.0e9 && fcx <= 2.0e9) ? (int)fcx : 0x7FFFFFF0;
            const int cy = (fcy >= -2.0e9 && fcy <= 2.0e9) ? (int)fcy : 0x7FFFFFF0;
            lo_x = min(lo_x, cx);
            lo_y = min(lo_y, cy);
            hi_x = max(hi_x, cx);
            hi_y = max(hi_y, cy);
            const double ddx = ps.x - pv.x, ddy = ps.y - pv.y;
            const int qx = (int)fmin(fmax(ddx * 16.0, -2.0e9), 2.0e9), qy = (int)fmin(fmax(ddy * 16.0, -2.0e9), 2.0e9);
            if (qx >= 0) d0 = max(d0, qx); else d1 = max(d1, -qx);
            if (qy >= 0) d2 = max(d2, qy); else d3 = max(d3, -qy);
            slack = min(slack, min(min(cx - cl[0], cl[2] - cx), min(cy - cl[1], cl[3] - cy)));
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo_x = min(lo_x, __shfl_xor(lo_x, d, 64));
            lo_y = min(lo_y, __shfl_xor(lo_y, d, 64));
            hi_x = max(hi_x, __shfl_xor(hi_x, d, 64));
            hi_y = max(hi_y, __shfl_xor(hi_y, d, 64));
            d0 = max(d0, __shfl_xor(d0, d, 64));
            d1 = max(d1, __shfl_xor(d1, d, 64));
            d2 = max(d2, __shfl_xor(d2, d, 64));
            d3 = max(d3, __shfl_xor(d3, d, 64));
        }
        if ((tid & 63) == 0) {
            atomicMin(&red[0], lo_x);
            atomicMin(&red[1], lo_y);
            atomicMax(&red[2], hi_x);
            atomicMax(&red[3], hi_y);
            atomicMax(&red[4], d0);
            atomicMax(&red[5], d1);
            atomicMax(&red[6], d2);
            atomicMax(&red[7], d3);
        }
        __syncthreads();
        if (tid < 4) A.atom_aabb_out[4 * atom + tid] = red[tid];
        else if (tid < 8) A.atom_disp_out[4 * atom + tid - 4] = red[tid];
        p += cnt;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) slack = min(slack, __shfl_xor(slack, d, 64));
    if ((tid & 63) == 0) wslack[tid >> 6] = slack;
    __syncthreads();
    if (tid == 0) {
        int s = wslack[0];
        for (int w = 1; w < (nthreads + 63) / 64; ++w) s = min(s, wslack[w]);
        A.tile_slack[tile] = s;
    }
}

// The per-tile counters of the class become the step's status entries: one atomic per workgroup instead of one per
// tile (thousands of same-address atomics would serialise at ~12 ns apiece).  Block b < n_passes sums the visits of
// pass b; the last block reduces slack and list sizes.
extern "C" __global__ void __launch_bounds__(1024) egg_pk_reduce_kernel(EggPackedArgs A, int n_passes) {
    __shared__ long long wsum[16];
    __shared__ int wmin[16], wmax[16];
    const int tid = threadIdx.x, ps = blockIdx.x;
    if (ps < n_passes) {
        long long s = 0;
        const int32_t *v = A.tile_visits + (size_t)ps * A.n_tiles;
        for (int t0 = 0; t0 < A.n_tiles; t0 += 4096) {
            int x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 1024 * u + tid;
                x[u] = (t < A.n_tiles) ? v[t] : 0;
            }
            s += (long long)x[0] + x[1] + x[2] + x[3];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        if ((tid & 63) == 0) wsum[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            long long tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            atomicAdd(&A.status->visits[ps], (unsigned long long)tot);
        }
        return;
    }
    int slack = 0x7FFFFFFF, ml = 0;
    for (int t = tid; t < A.n_tiles; t += 1024) slack = min(slack, A.tile_slack[t]);
    for (size_t x = tid; x < (size_t)n_passes * A.n_tiles; x += 1024) ml = max(ml, A.tile_need[x]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        slack = min(slack, __shfl_xor(slack, d, 64));
        ml = max(ml, __shfl_xor(ml, d, 64));
    }
    if ((tid & 63) == 0) {
        wmin[tid >> 6] = slack;
        wmax[tid >> 6] = ml;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) {
            slack = min(slack, wmin[w]);
            ml = max(ml, wmax[w]);
        }
        atomicMin(&A.status->min_slack, min(slack, wmin[0]));
        atomicMax(&A.status->max_list, (unsigned long long)max(ml, wmax[0]));
    }
}
