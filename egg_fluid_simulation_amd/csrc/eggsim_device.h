// eggsim_device.h -- structs shared by the host side (eggsim_host.cpp) and the
// gfx950 kernels (eggsim_step.hip).  Not part of the public ABI.
#pragma once
#include <stdint.h>

#define EGG_MAX_PASSES 64     // per-pass counters kept for the first 64 collision passes of a step
#define EGG_WAVE 64

// Written by the step kernel, read back by the host after every step.
struct EggStatus {
    int32_t fail_claim;     // a particle left its atom's claimed cell box: tiles were not provably independent
    int32_t fail_overflow;  // a tile needed more visit-list entries than the launch provided
    int32_t fail_stall;     // the DAG executor did not drain (internal error guard)
    int32_t fail_range;     // a cell coordinate does not fit the packed tile-relative form
    int32_t min_slack;      // min over particles of the distance (cells) to their claim box edge at step end
    int32_t was_cut;        // single-tile mode: the collision budget cut some pass (L:1657-1658)
    int32_t pad0, pad1;
    unsigned long long visits[EGG_MAX_PASSES]; // visited pairs per collision pass, summed over tiles
    unsigned long long max_list;               // largest visit-list length any tile had in one pass
    unsigned long long rounds;                 // DAG rounds, summed over tiles and passes
};

// One particle type (white or yolk) of one handler.  All pointers are device pointers.
struct EggStepArgs {
    // particle state, SoA, particle-index order (= the reference's array order, L:964-993)
    const double *x_in, *y_in, *vx_in, *vy_in;
    double *x_out, *y_out, *vx_out, *vy_out;
    const double *inv_mass, *radius;
    // atoms: the particles of one batch of this type, contiguous
    const int32_t *atom_offset;  // first particle
    const int32_t *atom_count;
    const int32_t *atom_batch;   // batch slot (equality = "same batch", L:1609)
    const double *atom_tx, *atom_ty;  // follow target (L:1763-1766)
    const double *atom_fd;            // 2 * sqrt(batch radius) (L:1454, L:1790)
    const int32_t *atom_claim;   // int4 {lo_x, lo_y, hi_x, hi_y}: cells the atom's particles may occupy this step
    int32_t *atom_aabb_out;      // int4: cells occupied at the end of the step
    // tiles: independent groups of atoms, one workgroup each
    const int32_t *tile_atom_begin;  // [n_tiles + 1] into tile_atoms
    const int32_t *tile_atoms;       // atom ids, ascending inside a tile
    int32_t n_tiles;
    // scalars of the environment (L:1726-1774)
    double sub_delta, damping, follow_compliance, collision_compliance;
    double overlap_factor, cell_size, eps;
    double budget;             // max_n_collisions = fraction * N^2 (L:1752-1753)
    int32_t single_tile;       // 1: this launch is one tile holding every particle -> exact budget handling
    int32_t n_substeps, n_collision_steps;
    // LDS geometry of this launch
    int32_t nmax;   // particles per tile (capacity)
    int32_t amax;   // atoms per tile (capacity)
    int32_t ht;     // cell hash table size, power of two
    int32_t lcap;   // visit-list entries per pass (capacity)
    EggStatus *status;
};

static inline size_t egg_align8(size_t v) { return (v + 7) & ~(size_t)7; }

// dynamic LDS bytes the step kernel carves for the geometry above (must match eggsim_step.hip)
static inline size_t egg_step_lds_bytes(int nmax, int amax, int ht, int lcap) {
    size_t n = (size_t)nmax, a = (size_t)amax, h = (size_t)ht, l = (size_t)lcap;
    size_t b = 0;
    b += 8 * n * 8;                 // x y px py vx vy w r
    b += a * 3 * 8;                 // atx aty afd
    b += egg_align8(2 * n * 4);     // ckey[2]
    b += egg_align8(2 * h * 4);     // hkeys[2]
    b += egg_align8(2 * h * 4);     // hmeta[2]
    b += egg_align8(2 * (n + 1) * 4); // own_off[2]
    b += egg_align8((n + 1) * 4);   // inc_off
    b += egg_align8(n * 4);         // fill
    b += egg_align8((n / 2 + 1) * 2 * 4); // queue[2]
    b += egg_align8(a * 4 * 4);     // aclaim
    b += egg_align8(a * 4 * 4);     // aaabb
    b += egg_align8((a + 1) * 4);   // aoff
    b += egg_align8(a * 4);         // abatch
    b += egg_align8(16 * 4);        // scalars
    b += egg_align8(2 * n * 2);     // hitems[2]
    b += egg_align8(n * 2) * 6;     // pslot aslot ptr nlo nxt stamp
    b += egg_align8(2 * l * 2);     // own_ent[2]
    b += egg_align8(l * 2) * 2;     // inc_ent inc_tmp
    return b;
}
