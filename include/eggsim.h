/*
 * eggsim.h -- C ABI of libeggsim.so: the MI355X (gfx950) implementation of the
 * XPBD particle step of Clemapfel/egg_fluid_simulation.
 *
 * The reference has no native/FFI layer; its boundary is the Lua class
 * `SimulationHandler` (simulation_handler.lua:9-419).  Every entry point below
 * is what a LuaJIT `ffi.cdef` wrapper of that class binds for the solver path
 * (the wrapper is lua/egg_fluid_simulation/simulation_handler.lua; the stub a
 * maintainer adds is shown in INTEGRATION.md).  Reference citations are
 * file:line into /root/reference/simulation_handler.lua ("L:").
 *
 * Conventions
 *  - plain C types only; scalars are double / int64_t / int32_t;
 *  - every call returns an int status: EGG_OK, a positive "warning class"
 *    status where the reference prints a warning and carries on, or a negative
 *    "error class" status where the reference throws (log.error);
 *  - egg_last_error(h) gives the message of the last non-OK status;
 *  - arrays are caller-owned and copied during the call;
 *  - one handle is used by one thread at a time (the reference is
 *    single-threaded, non-reentrant);
 *  - there is NO CPU fallback: creating a handle without a usable HIP device
 *    fails with EGG_ERR_NO_DEVICE.
 */
#ifndef EGGSIM_H
#define EGGSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGGSIM_ABI_VERSION 2

typedef struct egg_handle egg_handle;

enum {
    EGG_OK = 0,
    EGG_WARN_UNKNOWN_ID = 1,     /* set_target_position / remove on a missing id: warning, no throw (L:145, L:259) */
    EGG_WARN_FEW_PARTICLES = 2,  /* add: white_n < 10 or yolk_n < 5 (L:114-120); the batch IS created */
    EGG_ERR_UNKNOWN_ID = -1,     /* get_position / get_target_position / get_n_particles (L:273, L:286, L:415) */
    EGG_ERR_INVALID_ARGUMENT = -2, /* the reference's log.error paths in add/update (L:71-85, L:184-197) */
    EGG_ERR_NO_DEVICE = -3,
    EGG_ERR_DEVICE = -4,         /* a HIP call failed; message has the HIP error string */
    EGG_ERR_UNSUPPORTED = -5,    /* a configuration the device path does not implement yet */
    EGG_ERR_INTERNAL = -6
};

enum { EGG_WHITE = 0, EGG_YOLK = 1 };

/* The solver-relevant keys of a white/yolk config table (L:1152-1249, defaults in
 * simulation_handler_default_config.lua:10-68) plus the three hidden constants
 * (L:447-448, math.lua:2).  Values are taken as already validated/clamped by the
 * host wrapper (_load_config, L:1253-1320); the library re-applies the clamps the
 * step itself applies (L:1338, L:1768). */
typedef struct {
    double damping;
    double follow_strength;
    double cohesion_strength;
    double cohesion_interaction_distance_factor;
    double collision_strength;
    double collision_overlap_factor;
    double min_mass, max_mass;
    double min_radius, max_radius;
    double max_collision_fraction;     /* 0.05 */
    double mass_distribution_variance; /* 4 */
    double eps;                        /* 1e-8 */
} egg_config;

/* fills *cfg with the reference defaults for `which` */
int egg_default_config(int which, egg_config *cfg);

/* SimulationHandler(white_config, yolk_config) (L:425-459).  yolk == NULL means
 * "same as white" (L:426).  device = HIP device ordinal. */
int egg_create(const egg_config *white, const egg_config *yolk, int device, egg_handle **out);
void egg_destroy(egg_handle *h);
const char *egg_last_error(const egg_handle *h); /* h may be NULL: last create error */

/* set_white_config / set_yolk_config, get_*_config (L:226-248) */
int egg_set_config(egg_handle *h, int which, const egg_config *cfg);
int egg_get_config(const egg_handle *h, int which, egg_config *cfg);

/* add(x, y, white_radius, yolk_radius, ..., white_n, yolk_n) -> id (L:27-135).
 * Pass NaN for a radius and EGG_DEFAULT_COUNT for a count where the caller gave nil: the reference's
 * defaults apply (L:41-58).  Any other count <= 1 -- an explicit 0 or negative one included -- is the
 * reference's "particle count cannot be 1 or negative" error (L:79-85) and creates nothing.
 * Colors are render attributes and stay on the host side. */
#define EGG_DEFAULT_COUNT ((int64_t)-1)
int egg_add(egg_handle *h, double x, double y, double white_radius, double yolk_radius,
            int64_t white_n, int64_t yolk_n, int64_t *out_id);
/* n batches with identical radii in one call (bulk form for 10^4..10^5 batches) */
int egg_add_many(egg_handle *h, int64_t n, const double *xs, const double *ys, double white_radius,
                 double yolk_radius, int64_t white_n, int64_t yolk_n, int64_t *out_ids);
/* Multi-GPU sharding.  The reference keeps every batch of a handler in ONE array in creation order
 * (L:964-993), and the pair solver's results depend on that order.  When the batches are spread over
 * several handlers (one per GPU), each handler must lay its particles out in the global creation
 * order restricted to its batches: `key` is a batch's position in that global order.
 * egg_add_many_keyed appends (keys ascending and larger than every key present); egg_export_batch /
 * egg_import_batch move a batch with its complete particle state between handlers, the import
 * inserting it at its key's place.  State layout: 9 fields x n particles, field-major:
 * x, y, vx, vy, last_x, last_y, inv_mass, radius, mass_t.  The state buffers may be host memory or memory of the
 * handle's device (the copies use hipMemcpyDefault): a multi-GPU host hands a batch over device to device -- e.g.
 * straight into and out of the tensors an RCCL send / receive works on -- without a bounce through the host. */
typedef struct {
    int64_t key;
    double target_x, target_y;
    double white_radius, yolk_radius;
    int64_t n_white, n_yolk;
} egg_batch_info;
int egg_add_many_keyed(egg_handle *h, int64_t n, const double *xs, const double *ys, double white_radius,
                       double yolk_radius, int64_t white_n, int64_t yolk_n, const int64_t *keys, int64_t *out_ids);
int egg_export_batch(egg_handle *h, int64_t id, egg_batch_info *info, double *white_state, double *yolk_state);
int egg_import_batch(egg_handle *h, const egg_batch_info *info, const double *white_state, const double *yolk_state,
                     int64_t *out_id);

/* remove(id) (L:140-155, L:1037-1106) */
int egg_remove(egg_handle *h, int64_t id);

/* set_target_position / get_target_position (L:254-278) */
int egg_set_target(egg_handle *h, int64_t id, double x, double y);
int egg_set_targets_many(egg_handle *h, int64_t n, const int64_t *ids, const double *xs, const double *ys);
int egg_get_target(const egg_handle *h, int64_t id, double *x, double *y);

/* update(delta, step_delta, n_substeps, n_collision_steps) (L:168-222): runs the
 * fixed-step accumulator; *out_n_steps = number of _step calls made. */
int egg_update(egg_handle *h, double delta, double step_delta, int32_t n_substeps,
               int32_t n_collision_steps, int32_t *out_n_steps);
/* _step(delta, n_sub_steps, n_collision_steps) directly (L:1722) */
int egg_step(egg_handle *h, double delta, int32_t n_substeps, int32_t n_collision_steps);
/* Forms the tiles and claims the next _step(step_delta, n_substeps, n_collision_steps) will use, without
 * running it.  Multi-GPU: egg_get_bounds_many then returns each batch's CLAIM for that step, which
 * neighbouring ranks exchange to decide whether two batches on different ranks could interact. */
int egg_prepare_step(egg_handle *h, double step_delta, int32_t n_substeps, int32_t n_collision_steps);
/* A _step in two halves, so that a caller can overlap its own work (the multi-GPU neighbour exchange)
 * with the kernels: egg_step_begin forms the tiles and launches; egg_step_end(commit = 1) waits,
 * validates, re-runs if needed and commits -- begin + end(1) == egg_step; egg_step_end(commit = 0)
 * discards the launched step (the state is double-buffered, nothing was committed). */
int egg_step_begin(egg_handle *h, double delta, int32_t n_substeps, int32_t n_collision_steps);
int egg_step_end(egg_handle *h, int32_t commit);
/* Between egg_step_begin and egg_step_end: waits for the launched step and reports, per type, the most pairs it
 * visited in one collision pass and the budget 0.05 N^2 it is priced at (L:1657-1658, L:1752-1753) -- BEFORE
 * anything is committed.  Multi-GPU: the reference counts the visits of ALL particles against the budget, a rank
 * sees its own; the ranks add these up and discard the step when the sum could have tripped the early return. */
int egg_step_peek_visits(egg_handle *h, int64_t max_pass_visits[2], double budget[2]);
/* blocks until all device work of this handle is finished */
int egg_synchronize(egg_handle *h);

/* get_position(id) -> mean particle position of the batch (L:281-295, L:1134-1148) */
int egg_get_position(egg_handle *h, int64_t id, double *x, double *y);
int egg_get_positions_many(egg_handle *h, int64_t n, const int64_t *ids, double *xs, double *ys);

/* axis-aligned bounds (px) of each batch, white and yolk together: the cells the batch CLAIMS for the
 * upcoming step when the tiles are current (after egg_prepare_step), otherwise the spatial-hash cells
 * its particles occupy (L:1494-1495).  Used by the multi-GPU slab exchange; the
 * reference computes the same per-environment AABB in _post_solve (L:1703-1709). */
int egg_get_bounds_many(egg_handle *h, int64_t n, const int64_t *ids, double *lo_x, double *lo_y,
                        double *hi_x, double *hi_y);

/* per type: boxes[8 * k + 4 * type + {0,1,2,3}] = lo_x, lo_y, hi_x, hi_y (px) of batch k's claim for the
 * upcoming step (occupied cells if the tiles are not current); cell_sizes[type] = that type's hash cell.
 * Batches of different handlers are independent iff, for both types, their boxes are at least one cell
 * of that type apart in x or in y -- the criterion that separates tiles inside one handler. */
int egg_get_claims_many(egg_handle *h, int64_t n, const int64_t *ids, double *boxes, double *cell_sizes);

/* get_n_particles(id) / get_n_particles() with id < 0 (L:409-419) */
int egg_get_n_particles(const egg_handle *h, int64_t id, int64_t *n_white, int64_t *n_yolk);
/* list_ids (L:399-405): ids in creation order; returns the count in *n, copies min(*n, cap) ids */
int egg_list_ids(const egg_handle *h, int64_t cap, int64_t *ids, int64_t *n);
int egg_get_elapsed(const egg_handle *h, double *elapsed, double *interpolation_alpha);

/* particle fields for egg_download_particles */
enum {
    EGG_FIELD_X = 0, EGG_FIELD_Y, EGG_FIELD_VX, EGG_FIELD_VY, EGG_FIELD_LAST_X, EGG_FIELD_LAST_Y,
    EGG_FIELD_RADIUS, EGG_FIELD_INV_MASS, EGG_FIELD_MASS_T, EGG_FIELD_BATCH_ID, EGG_N_FIELDS
};
/* copies one field of every particle of `which`, in particle-index order, into dst
 * (doubles).  x,y,last_x,last_y,vx,vy,radius form the reference's instanced-draw
 * record (L:513-517, L:744-813). */
int egg_download_particles(egg_handle *h, int which, int field, double *dst, int64_t cap);

/* The per-type reductions the reference's _post_solve / update_last_positions keep in its environment
 * (L:1669-1718, L:1795-1815) -- what :draw() sizes and places its canvases with (L:1946-1950, L:2007,
 * L:2132).  Computed on demand from the device arrays, bounds and maxima in parallel, the centroid sums
 * serially in particle order like the reference: after every _step they equal the reference's env fields
 * bit for bit.  (The reference's fields stay stale until the next _step when batches are added or removed
 * in between; these follow the arrays.)  Before the first _step: bounds +-inf, everything else 0. */
typedef struct {
    double min_x, min_y, max_x, max_y;        /* AABB including the particle radius */
    double centroid_x, centroid_y;            /* mean position */
    double max_radius, max_velocity;
    double last_centroid_x, last_centroid_y;  /* mean of the positions at the start of the most recent _step */
} egg_environment;
int egg_get_environment(egg_handle *h, int which, egg_environment *out);

/* ---- headless renderer: SimulationHandler:draw() (L:158-161) into a float32 RGBA image ----
 * The reference draws through LOVE / OpenGL: _update_canvases (L:1995-2113, simulation_handler_instanced_draw.glsl over
 * the texture of simulation_handler_particle_texture.glsl) splats every particle into one canvas per type, _draw_canvases
 * (L:2117-2175, simulation_handler_outline.glsl, simulation_handler_lighting.glsl) composites them.  These entry points run
 * the same passes as HIP kernels, for image-level regression without a window: float32 canvases sampled at pixel centres,
 * no MSAA, instances blended in particle order (what GL guarantees).  Colours are straight rgba in [0, 1]. */
typedef struct {
    float color[4], outline_color[4];  /* config.color / config.outline_color (default_config.lua:22-23, 54-55) */
    double outline_thickness;          /* px; 0 skips the outline pass AND its setColor (L:2137-2142) */
    double highlight_strength, shadow_strength;
    double texture_scale, motion_blur;
} egg_render_config;
int egg_default_render_config(int which, egg_render_config *cfg);
/* the render keys of set_white_config / set_yolk_config (L:226-236): the config gets a NEW colour table, batches created
 * without a colour keep the old one (L:1307-1311) */
int egg_set_render_config(egg_handle *h, int which, const egg_render_config *cfg);
int egg_get_render_config(const egg_handle *h, int which, egg_render_config *cfg);
/* the handler's hidden constants _use_particle_color, _use_lighting (L:448-449).  With use_particle_color == 0 (default)
 * particles are created white (L:985-990) whatever the batch colour is. */
int egg_set_render_flags(egg_handle *h, int32_t use_particle_color, int32_t use_lighting);
/* the white_color / yolk_color argument of add (L:22-23): the batch gets its own colour table; its particles take the
 * colour only when use_particle_color is set (L:978-990).  Call right after egg_add. */
int egg_set_add_color(egg_handle *h, int64_t id, int which, double r, double g, double b, double a);
/* set_white_color / set_yolk_color (L:328-398): components are clamped to [0, 1]; the batch's particles take the colour
 * (L:1110-1129).  A batch created without a colour argument shares the CONFIG's colour table (L:49-50), so the call
 * also changes config.color -- the colour _draw_canvases tints the whole type with; that aliasing is the reference's.
 * Unknown id: EGG_WARN_UNKNOWN_ID. */
int egg_set_color(egg_handle *h, int64_t id, int which, double r, double g, double b, double a);
typedef struct {
    int32_t screen_w, screen_h;      /* render target; world px = screen px + origin (love.graphics.translate) */
    double origin_x, origin_y;
    double interpolation_alpha;      /* NaN: the handle's (egg_update, L:216) */
    double threshold, smoothness;    /* _thresholding_threshold / _smoothness (L:444-445) */
    int32_t use_instancing;          /* 1: instanced_draw.glsl; 0: the draw loop L:2009-2052 (colour premultiplied by its alpha) */
    int32_t canvas_w[2], canvas_h[2];/* 0: from the environment's bounds (L:1945-1954), growing only over the calls */
    float clear[4];                  /* what the screen holds before draw() */
} egg_render_params;
int egg_default_render_params(egg_render_params *p);
/* draw(): clears the screen image to p->clear, runs both passes, copies screen_w * screen_h * 4 floats (row-major, RGBA)
 * into rgba (may be NULL: the image stays on the device).  Nothing is drawn before the first _step or while one of the
 * types has no particles (the reference has no canvas then: L:1997-1999, L:2118). */
int egg_render(egg_handle *h, const egg_render_params *p, float *rgba);
/* the density canvas of `which` as the last egg_render left it: *w x *h RGBA floats, its top-left corner in world px */
int egg_render_canvas(egg_handle *h, int which, float *rgba, int64_t cap_pixels, int32_t *w, int32_t *hgt, double *x0,
                      double *y0);
/* the particle density texture (L:620-682): *size x *size alpha values (all four channels of the texture hold them) */
int egg_render_particle_texture(egg_handle *h, float *alpha, int64_t cap, int32_t *size);

/* kernels of the packed pipeline (csrc/eggsim_packed.hip), for egg_stats.pk_kernel_ms */
enum {
    EGG_PK_KIND_BEGIN = 0, EGG_PK_KIND_MID, EGG_PK_KIND_LISTS_FRESH, EGG_PK_KIND_LISTS_STALE, EGG_PK_KIND_LEVELS,
    EGG_PK_KIND_SORT, EGG_PK_KIND_EXEC, EGG_PK_KIND_END, EGG_PK_KIND_REDUCE,
    EGG_PK_KIND_PASS, /* egg_pk_levexec_kernel: levels + sort + executor of a dense group in one launch */
    EGG_PK_N_KINDS
};

/* egg_stats.pk_variants: the kernel a phase of the packed pipeline runs depends on the regime (csrc/eggsim_packed.hip) */
enum {
    EGG_PK_VARIANT_LEVELS_INORDER = 1, /* egg_pk_levels_mr16_kernel: more groups than SIMDs */
    EGG_PK_VARIANT_LEVELS_OOO = 2,     /* egg_pk_levels_ooo_kernel (levels + sort in one launch): dense islands on a chip that is not full */
    EGG_PK_VARIANT_EXEC = 4,           /* egg_pk_exec_kernel */
    EGG_PK_VARIANT_EXEC_CHAIN = 8,     /* egg_pk_exec_chain_kernel: branch-free projection, executor waves alone on their SIMDs */
    EGG_PK_VARIANT_SORT_LDS = 16,      /* egg_pk_sort_kernel: sorted list assembled in LDS */
    EGG_PK_VARIANT_SORT_DIRECT = 32,   /* egg_pk_sort_direct_kernel */
    EGG_PK_VARIANT_PASS_FUSED = 64     /* egg_pk_levexec_kernel: out-of-order walk, sort and chain executor of a group in one launch */
};

/* counters of the device path, cumulative since creation */
typedef struct {
    int64_t steps;           /* _step calls executed */
    int64_t pair_solves;     /* visited pairs = n_collided increments (L:1657) */
    int64_t follow_solves;   /* follow-constraint evaluations (N * S per step) */
    int64_t kernel_launches;
    int64_t retiles;         /* host re-clusterings of particles into tiles */
    int64_t redo_steps;      /* steps re-run after a failed independence/budget check */
    int64_t n_tiles[2];      /* current tile count per type */
    int64_t max_tile_particles[2];
    double last_step_kernel_ms; /* device time of the step kernels of the most recent _step (HIP events) */
    int64_t single_tile[2];  /* 1 if that type currently runs in exact-budget single-tile mode */
    double kernel_ms[2];     /* device time of that type's step launches in the most recent _step (EGG_OPT_TIMING) */
    double kernel_ms_sum[2]; /* the same, summed over all committed steps since EGG_OPT_TIMING was switched on */
    int64_t timed_steps;
    int64_t max_pass_visits[2]; /* most pairs visited in one collision pass of the most recent _step, per type */
    double budget[2];           /* max_collision_fraction * N^2 of the most recent _step (L:1752-1753), per type */
    int64_t fused_launch;       /* 1 if the most recent _step ran both types' tiles in one launch (kernel_ms[0] == kernel_ms[1] is then that launch) */
    int64_t packed[2];          /* launch classes of that type currently stepped by the packed pipeline (one launch per phase) */
    /* EGG_OPT_TIMING = 2: HIP events around every launch of the packed pipeline, summed per kernel kind and type since the
     * option was set: [type][kind], kind as in EGG_PK_KIND_* below */
    double pk_kernel_ms[2][EGG_PK_N_KINDS];
    int64_t pk_kernel_launches[2][EGG_PK_N_KINDS];
    /* host wall time of _step's phases, summed since creation: [0] tiles and claims (re-clustering, packed plan), [1] uploads
     * and kernel launches, [2] waiting for the status block */
    double host_ms[3];
    /* packed pipeline: the longest chain of dependent pairs in one collision pass of the most recent _step, per type -- the
     * number of levels its executor ran one after the other (the path's latency floor: DESIGN.md section 4) */
    int64_t max_levels[2];
    /* packed pipeline: which kernel variants the classes of that type run (EGG_PK_VARIANT_* bits; several classes may differ) */
    int64_t pk_variants[2];
} egg_stats;
int egg_get_stats(egg_handle *h, egg_stats *out);

/* Diagnostic: the step kernel expands its f64 divisions and square root by hand (see
 * eggsim_step.hip); this runs those expansions against `/` and sqrt() on n random operand pairs on
 * the device and returns the number of results that differ in any bit (must be 0). */
int egg_selftest_arith(egg_handle *h, int64_t n_operand_pairs, uint64_t seed, int64_t *mismatches);

/* tuning knobs (not part of the reference surface) */
enum {
    EGG_OPT_CLAIM_MARGIN_CELLS = 0, /* initial margin around an atom's cells when tiles are formed */
    EGG_OPT_TILE_TARGET_PARTICLES,  /* pack independent islands into tiles up to this size (0 = one island per tile; default 60: small islands share a wave) */
    EGG_OPT_TIMING,                 /* 1: record HIP events around the step kernels; 2: also around every launch of the packed pipeline (profiling runs: ~2 events per launch) */
    EGG_OPT_FORCE_SINGLE_TILE,      /* 1: always run each type as one tile (exact budget path) */
    EGG_OPT_THREADS_PER_PARTICLE,   /* lanes per particle: 0 automatic (3 for tiles that have a CU to themselves: visit lists built column-wise), 1 or 3 forced */
    EGG_OPT_SPIN_SLEEP,             /* -1 auto, 0 never, 1 always: idle dataflow waves sleep between polls */
    EGG_OPT_BUDGET_PARTICLES_WHITE, /* multi-GPU: N of the collision budget 0.05 N^2 (L:1752-1753) = particles of ALL ranks; -1 = local */
    EGG_OPT_BUDGET_PARTICLES_YOLK,
    EGG_OPT_FORCE_GLOBAL_STATE,     /* test hook: 1 = every tile keeps its state in global memory (the large-island fallback) */
    EGG_OPT_FUSE_TYPES,             /* 1 (default): white and yolk tiles share one launch when the chip holds several tiles per CU; 0: one launch per type */
    EGG_OPT_PACKED,                 /* packed pipeline (one launch per phase, pair projections of many islands packed into full waves): -1 automatic (large scenes), 0 never, 1 whenever a launch class is eligible */
    EGG_OPT_GROUP_PARTICLES,        /* packed pipeline: particles whose positions one wave of the pair executor keeps in LDS (0, the default: by scene size, 320..1280) */
    EGG_OPT_LEVEL_WALK              /* packed pipeline, the pass that gives every pair its dependency level: 0 (default) by regime -- out of order for dense islands (> 256 particles) while the groups are no more than the chip's SIMDs, in order otherwise --, 1 always in order, 2 out of order everywhere */
};
int egg_set_option(egg_handle *h, int option, double value);

/* ---- several GPUs in one process (csrc/eggsim_group.cpp) -------------------------------------------------------
 * The multi-device form of the handle for a host that is ONE process (the LuaJIT wrapper): one egg_handle per device
 * behind one egg_group, x-slabs [cuts[k], cuts[k + 1]) of the plane per device (cuts: n_devices + 1 ascending values;
 * NULL with one device), global batch ids.  A batch is stepped by the device whose slab held its position when it was
 * added; batches whose claims for a step come within one spatial-hash cell of each other across devices are handed to
 * ONE device before that step runs (exact Gauss-Seidel order cannot cross a cut), so the results equal a
 * single handle's bit for bit.  A collision budget 0.05 N^2 (L:1752-1753) that could bind across devices is refused with
 * EGG_ERR_UNSUPPORTED.  The same device ordinal may appear more than once (several handles on one GPU: testing).
 * egg_fluid_simulation_amd/sharding.py is the same protocol between processes over RCCL. */
typedef struct egg_group egg_group;
int egg_group_create(const egg_config *white, const egg_config *yolk, int32_t n_devices, const int32_t *devices,
                     const double *cuts, egg_group **out);
void egg_group_destroy(egg_group *g);
const char *egg_group_last_error(const egg_group *g);
int32_t egg_group_n_devices(const egg_group *g);
egg_handle *egg_group_handle(egg_group *g, int32_t k); /* the k-th device's handle, for downloads and statistics */
int egg_group_set_halo(egg_group *g, double halo_px);  /* how far outside its slab an island may idle before it is handed on (64) */
/* add / remove / set_target_position / get_position / update of SimulationHandler, ids global (L:27-135, L:140-155,
 * L:254-264, L:281-295, L:168-222) */
int egg_group_add(egg_group *g, double x, double y, double white_radius, double yolk_radius, int64_t white_n, int64_t yolk_n,
                  int64_t *out_id);
int egg_group_remove(egg_group *g, int64_t id);
int egg_group_set_target(egg_group *g, int64_t id, double x, double y);
int egg_group_get_position(egg_group *g, int64_t id, double *x, double *y);
int egg_group_update(egg_group *g, double delta, double step_delta, int32_t n_substeps, int32_t n_collision_steps,
                     int32_t *out_n_steps);
int egg_group_step(egg_group *g, double delta, int32_t n_substeps, int32_t n_collision_steps); /* _step directly */
/* which device index holds batch `id` now, and under which id of that device's handle */
int egg_group_owner(const egg_group *g, int64_t id, int32_t *device_index, int64_t *local_id);
int egg_group_get_counters(const egg_group *g, int64_t *migrations, int64_t *discarded_steps);

#ifdef __cplusplus
}
#endif
#endif
