// eggsim_device.h -- structs shared by the host side (eggsim_host_*.hip) and the
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
    int32_t fail_levels;    // packed pipeline: a group's pair-dependency DAG is deeper than the level table of the launch
    int32_t max_level;      // deepest level any group reached (what the level table must hold)
    int32_t fail_levlds;    // packed pipeline, out-of-order walk: a tile's pair stream is longer than the LDS level array of the launch
    int32_t reserved0;
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
    int32_t *atom_fail;          // cleared when the atom's tile starts, set to 1 when a particle of the atom left its claim
    int32_t *atom_disp_out;      // int4: max particle travel in the LAST sub-step towards +x, -x, +y, -y (1/16 px)
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
    int32_t nmax;      // particles per tile (capacity)
    int32_t amax;      // atoms per tile (capacity)
    int32_t ccap;      // cells: dense grid capacity (use_grid) or hash slots (power of two)
    int32_t use_grid;  // 1: cells are a dense grid over the tile's claim box, 0: open-addressing hash
    int32_t lcap;      // visit-list entries per pass (capacity)
    int32_t spin_sleep;  // 1: idle waves of the pair dataflow sleep between polls (many tiles per CU)
    int32_t threads;     // workgroup size this type's tiles want (a shared launch may bring more: the surplus waves exit)
    int32_t gens;        // hash generations kept alive (2; n_substeps when there is one collision pass per sub-step)
    int32_t pair_cache;  // 1: LDS holds lcap more 16-byte records (per-pair projection terms, see Tile::pinv)
    EggStatus *status;
    EggStatus *status_next;  // the other status block: re-initialised by this launch for the next one
    unsigned char *scratch;  // egg_step_kernel_gl / _gs: n_tiles slices of scratch_stride bytes
    unsigned long long scratch_stride;
};

#if defined(__HIPCC__)
__host__ __device__
#endif
static inline size_t egg_align16(size_t v) { return (v + 15) & ~(size_t)15; }

// bytes of one tile's slice of EggStepArgs::scratch (visit lists in global memory: own_pack 4 B, inc_tmp 8 B per entry)
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline size_t egg_step_scratch_bytes(int lcap, int single_tile, int gens = 2) {
    size_t l = (size_t)lcap, g = (size_t)(gens < 2 ? 2 : gens);
    return ((l * 4 + 15) & ~(size_t)15) + ((l * 8 + 15) & ~(size_t)15) + ((((single_tile ? g : 0) * l * 2) + 15) & ~(size_t)15);
}

// dynamic LDS bytes the step kernel carves for the geometry above (must match eggsim_step.hip)
static inline size_t egg_step_lds_bytes(int nmax, int amax, int ccap, int use_grid, int lcap, int single_tile,
                                        int global_lists, int threads, int pair_cache = 0, int gens = 2) {
    size_t n = (size_t)nmax, a = (size_t)amax, c = (size_t)ccap, l = (size_t)lcap, g = (size_t)(gens < 2 ? 2 : gens);
    size_t b = 0;
    b += 2 * egg_align16(n * 16);            // pos wr
    b += 2 * egg_align16((nmax > threads || threads >= 3 * nmax) ? n * 16 : 0);  // prev vel (registers otherwise; wide tiles: LDS)
    b += 3 * egg_align16(a * 8);             // atx aty afd
    b += egg_align16(g * n * 4);             // ckey[gens]
    b += egg_align16(g * c * 4);             // cell[gens]
    b += egg_align16(use_grid ? 0 : g * c * 4);  // hkeys[gens]
    b += egg_align16((single_tile ? g : 1) * (n + 1) * 4);  // own_off
    b += egg_align16((n + 1) * 4);           // inc_off
    b += 2 * egg_align16(n * 4);             // fill done
    if (!global_lists) b += 2 * egg_align16(l * 4);  // own_pack inc_tmp
    if (!global_lists && pair_cache) b += egg_align16(l * 16);  // pinv
    b += egg_align16(a * 4 * 4);             // aclaim
    b += egg_align16((a + 1) * 4);           // aoff
    b += 2 * egg_align16(a * 4);             // abatch aglob
    b += 2 * egg_align16(a * 4 * 4);         // aaabb adisp
    b += egg_align16(16 * 4);                // scalars
    b += egg_align16(g * n * 2);             // hitems[gens]
    b += 3 * egg_align16(n * 2);             // pslot aslot nlo
    if (!global_lists) b += egg_align16(single_tile ? g * l * 2 : 0);  // own_ent (exact-budget mode)
    return b;
}

// threads of the workgroup that runs a tile of at most nmax particles (one thread per particle)
static inline int egg_step_threads(int nmax, int spread) {
    int t = (nmax * spread + EGG_WAVE - 1) / EGG_WAVE * EGG_WAVE;
    return t < EGG_WAVE ? EGG_WAVE : (t > 1024 ? 1024 : t);
}

// ---------------------------------------------------------------------------------------------
// Packed pipeline (eggsim_packed.hip), the throughput path for many tiles per CU: one launch per phase
// instead of one fused launch per step.  The particles of the participating tiles are kept in PACKED order
// (tile after tile, atoms in tile order) in scratch arrays for the duration of a step; a collision pass is
//   egg_pk_lists   (one workgroup per tile)  cell grid + visit lists in the reference's order -> global memory
//   egg_pk_levels  (one workgroup per group) longest-path level of every pair (in-order or out-of-order walk)
//   egg_pk_sort    (one workgroup per group) the group's pairs sorted by level
//   egg_pk_exec    (one wave per group)      the group's positions in LDS; level by level, 64 pairs at a time
// A GROUP is a run of consecutive tiles whose particles one wave keeps in LDS.  Pairs of one level share no
// particle and all their predecessors lie in lower levels, so running the levels in order, each level's
// pairs in any order, is the reference's sequential result bit for bit.
struct EggPackedArgs {
    // particle-order state of the type, atoms, claims: as in EggStepArgs
    const double *x_in, *y_in, *vx_in, *vy_in;
    double *x_out, *y_out, *vx_out, *vy_out;
    const double *inv_mass, *radius;
    const int32_t *atom_offset, *atom_count, *atom_batch;
    const double *atom_tx, *atom_ty, *atom_fd;
    const int32_t *atom_claim;
    int32_t *atom_aabb_out, *atom_fail, *atom_disp_out;
    const int32_t *tile_atoms;  // atom ids of all tiles of the type (tile_geo gives a tile's range)
    int32_t n_tiles, n_groups;
    // Geometry of this class, computed by the host when tiles are formed, so that a kernel learns everything
    // about its tile or group from ONE load (every dependent global load costs a full memory round trip):
    const int32_t *tile_geo;    // [n_tiles][8]: first packed particle, particles, first entry in tile_atoms / tile_claims,
                                //               atoms, cell origin x, y (claims' minimum - 2), grid width, height
    const int32_t *tile_claims; // [4] per entry of tile_atoms: the atom's claim box {lo_x, lo_y, hi_x, hi_y}
    const int32_t *grp_geo;     // [n_groups][4]: first tile, end tile, first packed particle, particles
    int32_t p_begin, p_end;     // packed range of the class
    // packed per-particle arrays of the type (indexed by packed index)
    double *pk_pos, *pk_prev, *pk_wr;  // double2 each: position, position at the start of the sub-step, (inverse mass, radius)
    int32_t *pk_src;           // particle index in the particle-order arrays
    int32_t *pk_atom;          // atom id
    uint16_t *pk_aslot;        // atom slot inside the tile
    uint32_t *pk_ckey;         // [2][pk_stride] packed cell of the last pass of each sub-step parity
    int32_t pk_stride;
    // per tile (class-relative), `scap` words each: the tile's pair STREAM in the reference's order -- the visit
    // entries self | slow << 15 | other << 16 (tile-local indices) of every particle as `self`, ascending
    uint32_t *lists;
    uint16_t *lvl;             // level of each stream entry
    uint32_t *rank;            // per stream entry (out-of-order level walk): earlier entries involving the self | the partner << 16
    // per group, `sort_cap` words each: the group's pairs sorted by level (bit 31 set, indices group-local), and its
    // work list of `chunk_cap` words: chunk c = first word | (pairs - 1) << 26, at most 64 pairs of ONE level, levels ascending
    uint32_t *sorted;
    uint32_t *chunks;
    int32_t *grp_nchunks;      // [n_groups]
    uint32_t *lev_start;       // [n_groups][lev_cap + 2] first slot of every level in the group's sorted list
    int32_t *grp_nlev;         // [n_groups]
    int32_t *tile_total;       // [n_tiles] visit entries of the current pass (0: the tile failed a check)
    int32_t *tile_visits;      // [EGG_PK_MAX_PASSES][n_tiles] n_collided of each pass (L:1657)
    int32_t *tile_need;        // [EGG_PK_MAX_PASSES][n_tiles] visit entries each pass needed (list capacity check)
    int32_t *tile_slack;       // [n_tiles]
    int32_t *tile_fast;        // [n_tiles] 1: every pair of the tile may take the hand-expanded arithmetic (found by the step's first list pass)
    int32_t lcap, scap, lev_cap, sort_cap, chunk_cap;
    // LDS geometry of egg_pk_lists
    int32_t nmax, amax, ccap, use_grid, stage_cap;
    // environment
    double sub_delta, damping, follow_compliance, collision_compliance, overlap_factor, cell_size, eps;
    int32_t n_substeps, n_collision_steps;
    int32_t pass_seq, substep, stale;  // of the launch
    int32_t tune;              // developer switches (EGGSIM_TUNE), 0 in normal operation; unused by the kernels at present
    uint32_t *simd_claims;     // [4096] per compute unit: SIMDs taken by executor waves of egg_pk_levexec_kernel right now
    int32_t lev_lds_cap;       // out-of-order walk: entries of a tile's stream whose levels fit the LDS array of the launch
    EggStatus *status, *status_next;
};
#define EGG_PK_MAX_PASSES 64
#define EGG_PK_WINDOW 128  // stream words a sub-wave of egg_pk_levels holds in LDS at a time

// dynamic LDS of egg_pk_lists for the geometry above (must match eggsim_packed.hip)
// (gens: cell generations the launch holds -- 1 for a fresh pass, 2 for the stale pass)
static inline size_t egg_pk_lists_lds_bytes(int nmax, int amax, int ccap, int use_grid, int stage_cap, int gens) {
    size_t n = (size_t)nmax, a = (size_t)amax, c = (size_t)ccap, b = 0, G = (size_t)gens;
    b += egg_align16(G * n * 4);                  // ckey[gens]
    b += egg_align16(G * c * 4);                  // cell[gens]
    b += egg_align16(use_grid ? 0 : G * c * 4);   // hkeys[gens]
    b += egg_align16((n + 1) * 4);                // own_off
    b += egg_align16(n * 4);                      // fill
    b += egg_align16(a * 4 * 4);                  // aclaim
    b += egg_align16((a + 1) * 4);                // aoff
    b += egg_align16(16 * 4);                     // scalars
    b += egg_align16(G * n * 2);                  // hitems[gens]
    b += egg_align16(n * 2);                      // aslot
    {   // the partners kept by the counting pass; the grid builder's scratch (tmp, pslot) lives there before
        const size_t stage = egg_align16(((size_t)stage_cap + 1) * (n + 1) * 2), tmp = egg_align16(n * 4) + egg_align16(n * 2);  // ((n + 1) / 2 pairs of 2 stage_cap + 2 words)
        b += stage > tmp ? stage : tmp;
    }
    return b;
}
// egg_pk_levexec_kernel: the ring of ready-made pair records between its helper wave and its executor wave:
// EGG_PK_RING chunks x 64 lanes x (8 B addresses + 3 x 16 B constants)
#define EGG_PK_SPIN_LIMIT (1u << 24)  // polls a wave of the packed pipeline waits for another wave before it gives up (fail_stall = 4): seconds
#define EGG_PK_RING 8
#define EGG_PK_RING_BYTES (EGG_PK_RING * 64 * 56)
// dynamic LDS of egg_pk_levels_mr16 (the in-order walk): level histogram, last level and a stamp word per particle of the
// group, one stream window per sub-wave of 16 lanes
static inline size_t egg_pk_levels_mr_lds_bytes(int lev_cap, int group_particles, int threads) {
    return egg_align16((size_t)(lev_cap + 2) * 4) + egg_align16((size_t)group_particles * 2) +
           egg_align16((size_t)group_particles * 4) + egg_align16((size_t)(threads / 16) * EGG_PK_WINDOW * 4);
}
// dynamic LDS of egg_pk_levels_ooo (the out-of-order walk): level histogram, (completed pairs | last level) and the
// ranking pass's entry counter per particle of the group
// (+ a spare counter per lane), the levels of every stream entry of the group's tiles
static inline size_t egg_pk_levels_ooo_lds_bytes(int lev_cap, int group_particles, int tiles, int lev_lds_cap) {
    return egg_align16((size_t)(lev_cap + 2) * 4) + egg_align16((size_t)group_particles * 4) + 2 * egg_align16((size_t)(group_particles + 64) * 4) +
           egg_align16((size_t)tiles * (size_t)lev_lds_cap * 2);
}

// ---- headless renderer (eggsim_render.hip) ----
#define EGG_RENDER_TILE 16    // canvas tile edge in px: one workgroup of 256 threads per tile
#define EGG_RENDER_STAGE 128  // particles a tile stages in LDS at a time
struct EggRenderArgs {  // pass 1 of one type: particles -> canvas (_update_canvases, L:1995-2113)
    const double *x, *y, *last_x, *last_y, *vx, *vy, *radius;
    const int32_t *atom_offset;  // first particle of every atom (= batch), ascending
    const float4 *atom_color;    // the batch's particle colour (L:1110-1129)
    int32_t n, n_atoms;
    float t, tx, ty;             // interpolation alpha; translation world -> canvas px
    float texture_scale, motion_blur;
    int32_t premultiply;         // the non-instanced draw loop's setColor(r a, g a, b a, a) (L:2035-2041)
    int32_t cw, ch, tiles_x, tiles_y;
    uint32_t *tile_count, *tile_start, *tile_cursor;  // [tiles], [tiles + 1], [tiles]
    uint32_t *entries;           // the tiles' particle lists
    uint32_t *totals;            // [0] all list entries, [1] the longest list
    const float *texture;        // tsize x tsize density texture
    int32_t tsize;
    float4 *canvas;              // ch x cw
};
// dynamic LDS of egg_render_splat_kernel: bordered texture, staged instances + colours, the tile's list (padded to 2^k)
static inline size_t egg_render_splat_lds_bytes(int tsize, size_t padded_list) {
    return egg_align16((size_t)(tsize + 2) * (tsize + 2) * 4) + (size_t)EGG_RENDER_STAGE * (32 + 16) + padded_list * 4;
}
struct EggCompositeLayer {  // pass 2 of one type (_draw_canvases, L:2117-2175)
    const float4 *canvas;
    int32_t w, h;
    float x0, y0;  // the canvas's top-left corner on the screen
    float4 color, outline_color;
    float outline_thickness, highlight_strength, shadow_strength;
};
struct EggCompositeArgs {
    float4 *screen;
    int32_t screen_w, screen_h, n_layers;
    float threshold, smoothness;
    int32_t use_particle_color, use_lighting;
    EggCompositeLayer layer[2];
};

// the arguments of up to four launch classes sharing one launch (egg_step_kernel_multi*); unused slots have n_tiles = 0
struct EggStepArgs4 {
    EggStepArgs a[4];
};
