// eggsim_host.h -- what the host-side translation units of libeggsim.so share: the handle's state (particle arrays in
// HBM as SoA, batches, tiles, launch classes, the packed pipeline's plan), small helpers and the functions one unit
// calls in another.  Not part of the public ABI (include/eggsim.h).
//
//   eggsim_host_state.hip   particle creation, atoms, device buffers of a particle type
//   eggsim_host_tiling.hip  retile(): claims, islands, tiles, launch classes, the packed pipeline's groups
//   eggsim_host_step.hip    _step: environment scalars, kernel launches, validation / re-run / commit
//   eggsim_host_abi.hip     the extern "C" entry points of include/eggsim.h (except the renderer's)
//   eggsim_host_render.hip  egg_render* : the headless renderer's host side
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/eggsim.h"
#include "eggsim_device.h"

extern "C" __global__ void egg_step_kernel(EggStepArgs A);
extern "C" __global__ void egg_step_kernel_gl(EggStepArgs A);
extern "C" __global__ void egg_step_kernel_occ(EggStepArgs A);
extern "C" __global__ void egg_step_kernel_wide(EggStepArgs A);
extern "C" __global__ void egg_env_bounds_kernel(const double *, const double *, const double *, const double *, const double *, int,
                                                   unsigned long long *);
extern "C" __global__ void egg_env_sums_kernel(const double *, const double *, const double *, const double *, int, double *);
extern "C" __global__ void egg_step_kernel_multi(EggStepArgs4 P);
extern "C" __global__ void egg_step_kernel_multi_occ(EggStepArgs4 P);
extern "C" __global__ void egg_step_kernel_multi_wide(EggStepArgs4 P);
extern "C" __global__ void egg_step_kernel_mg(EggStepArgs A);
extern "C" __global__ void egg_step_kernel_gl_mg(EggStepArgs A);
extern "C" __global__ void egg_step_kernel_gs_mg(EggStepArgs A);
extern "C" __global__ void egg_step_kernel_gs(EggStepArgs A);
extern "C" __global__ void egg_selftest_arith_kernel(unsigned long long, int, unsigned long long *);
extern "C" __global__ void egg_pk_plan_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_begin_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_mid_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_lists_fresh_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_lists_stale_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_levels_mr16_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_levels_ooo_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_probe_lds_order_kernel(int trials, unsigned long long *bad);
extern "C" __global__ void egg_pk_exec_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_exec_chain_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_levexec_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_sort_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_sort_direct_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_end_kernel(EggPackedArgs A);
extern "C" __global__ void egg_pk_reduce_kernel(EggPackedArgs A, int n_passes);
extern "C" __global__ void egg_render_count_kernel(EggRenderArgs A);
extern "C" __global__ void egg_render_fill_kernel(EggRenderArgs A);
extern "C" __global__ void egg_render_scan_kernel(EggRenderArgs A);
extern "C" __global__ void egg_render_splat_kernel(EggRenderArgs A);
extern "C" __global__ void egg_render_composite_kernel(EggCompositeArgs A);
extern "C" __global__ void egg_render_clear_kernel(float4 *, size_t, float4);
extern "C" __global__ void egg_atom_bounds_kernel(const double *, const double *, const int32_t *, const int32_t *,
                                                   int, double, int32_t *);
extern "C" __global__ void egg_rederive_kernel(const double *, double *, double *, int, int, double, double, int,
                                                double, double);
extern "C" __global__ void egg_centroid_kernel(const double *, const double *, const double *, const double *,
                                                const int32_t *, const int32_t *, const int32_t *, const int32_t *,
                                                int, double *, double *);

namespace egghost {


constexpr double kPi = 3.14159265358979323846;
constexpr size_t kLdsMax = 160 * 1024;  // per CU on gfx950; what a workgroup may use is probed at create
constexpr int kMaxTileParticles = 32000;  // 15-bit local indices in the kernel's pair sequences
constexpr int kMaxListEntries = 60000;    // lists in LDS: 16-bit list positions in the transposition's records
constexpr int kMaxGlobalListEntries = 8 << 20;  // lists in global memory (64-bit records): bounded by memory only

extern std::string g_create_error;  // why the last egg_create failed (egg_last_error(NULL))

template <typename T>
struct DevBuf {  // growable device array
    T *p = nullptr;
    size_t cap = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    hipError_t reserve(size_t n, bool keep, hipStream_t s) {
        if (n <= cap) return hipSuccess;
        size_t ncap = std::max<size_t>(n, cap ? cap * 2 : 1024);
        T *q = nullptr;
        hipError_t e = hipMalloc((void **)&q, ncap * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep && p && cap) {
            e = hipMemcpyAsync(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                (void)hipFree(q);
                return e;
            }
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = ncap;
        return hipSuccess;
    }
};

template <typename T>
struct PinnedBuf {  // growable page-locked host array (async copies read/write it without staging)
    T *p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        size_t ncap = std::max<size_t>(n, cap ? cap * 2 : 1024);
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc((void **)&p, ncap * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = ncap;
        return e;
    }
};

struct Batch {
    int64_t id = 0;
    bool alive = false;
    double target_x = 0, target_y = 0;
    double white_radius = 0, yolk_radius = 0;
    int64_t n[2] = {0, 0};
    int64_t key = 0;  // position in the global creation order; particles are laid out in ascending key
    // render attributes (never read by the solver): the rgba its particles carry (L:978-990, L:1110-1129), and whether
    // the batch's colour table is its own (a colour argument of add) or the config's table (L:49-50)
    float pcolor[2][4] = {{1, 1, 1, 1}, {1, 1, 1, 1}};
    bool own_color[2] = {false, false};
};

struct Atom {
    int32_t batch = 0;  // index into Handle::batches
    int32_t offset = 0, count = 0;
};

struct Box {
    int32_t lo_x, lo_y, hi_x, hi_y;
};

struct LaunchClass {  // tiles of similar size share a launch (uniform LDS geometry)
    int first_tile = 0, n_tiles = 0;
    int nmax = 0, amax = 0, ccap = 0, use_grid = 0, lcap = 0;
    int global_lists = 0;  // visit lists in the scratch buffer instead of LDS
    int global_state = 0;  // everything in the scratch buffer (islands too large for LDS)
    int threads = 0;       // workgroup size
    int wide = 0;          // three lanes per particle for the list-building phases (egg_step_kernel_wide)
    int pair_cache = 0;    // per-pair projection terms cached in LDS (16 B per list entry)
    size_t scratch_stride = 0;
    size_t lds = 0, scratch_offset = 0;
    int packed = -1;       // index into System::pk when the class runs through the packed pipeline (eggsim_packed.hip)
};

// A launch class in the packed pipeline: its tiles' particles occupy [p_begin, p_end) of the type's packed arrays,
// consecutive tiles form GROUPS (one wave of egg_pk_levels / egg_pk_exec each).
struct PackedClass {
    int cls = 0;            // index into System::classes
    int n_tiles = 0, n_groups = 0;
    int p_begin = 0, p_end = 0;
    int tile_base = 0;      // first slot of the class in the per-tile arrays
    int group_base = 0;     // first slot in the per-group arrays
    size_t meta_tile_geo = 0, meta_grp_geo = 0;  // offsets (ints) into System::pk_meta
    int max_group_particles = 0;
    int fused_pass = 0;     // 1: levels, sort and executor of a group are ONE launch (egg_pk_levexec_kernel)
    size_t lds_pass = 0;    // its dynamic LDS: the larger of the two phases
    int lev_lds_cap = 0;    // out-of-order walk: stream entries per tile whose levels the LDS of a launch may hold (sized at re-tiling)
    int lev_lds_now = 0;    // ... and holds in this step's launches: no more than the last step's longest stream + 25 % asks for
    size_t lds_levels_now = 0, lds_pass_now = 0;
    int max_tiles_in_group = 0;
    int levels_ooo = 0;     // the level walk of the class: 0 in order (egg_pk_levels_mr16_kernel), 1 out of order (egg_pk_levels_ooo_kernel)
    int levels_threads = 64; // workgroup of the level walk (up to four waves per group)
    int lcap = 0, scap = 0;  // visit entries / stream words (entries + one header per particle) a tile may have
    int stage_cap = 0;       // partners per particle the list kernel's counting pass keeps in LDS
    int sort_cap = 0;        // words of a group's sorted list
    size_t sort_base = 0;
    int chunk_cap = 0;       // chunk descriptors per group
    size_t chunk_base = 0;
    size_t lds_sort = 0;     // 0: the sort kernel scatters straight into global memory
    size_t entry_base = 0;  // first stream word of the class in the per-entry arrays
    size_t lds_lists = 0, lds_lists_stale = 0, lds_levels = 0, lds_exec = 0;
    int threads_lists = 64, threads_lists_stale = 64;
};

struct System {  // one particle type
    egg_config cfg{};
    int64_t n = 0;
    int cur = 0;
    DevBuf<double> x[2], y[2], vx[2], vy[2], inv_mass, radius, mass_t;
    // atoms (host + device mirrors)
    std::vector<Atom> atoms;
    DevBuf<int32_t> d_atom_offset, d_atom_count, d_atom_batch;
    // what a step launch reports, in ONE device buffer so that one copy brings it back: two status blocks
    // (the launch writes one and re-initialises the other for the next launch), the atoms' end-of-step cell
    // boxes, their last-sub-step travel
    DevBuf<int32_t> d_out;
    int parity = 0;  // status block of the most recent launch
    hipStream_t wait_stream = nullptr;  // stream the most recent launch of this type went to (the white one when fused)
    int timing_from = 0;                // type whose events time the most recent launch
    int gens = 2;    // hash generations the kernel keeps (n_substeps when n_collision_steps == 1, see PassCtx)
    std::vector<Box> aabb;  // host copy of the atoms' occupied cells
    bool aabb_valid = false;
    bool aabb_on_device = false;  // d_atom_aabb holds the cells of the CURRENT positions (written by the last step)
    bool atoms_dirty = true, targets_dirty = true, tiling_dirty = true;
    bool claims_stale = false;  // a target moved since the tiles were formed
    std::vector<int32_t> disp;                  // per atom: max particle travel of the last step (+x,-x,+y,-y; 1/16 px)
    bool disp_valid = false;                    // fetched together with the boxes of the current positions
    bool swept = false;                          // some claim was extended along predicted motion
    std::vector<int> extra_margin;               // per batch: extra claim cells after a failed check (decays)
    DevBuf<unsigned long long> d_env;            // egg_get_environment: 6 ordered keys + 4 sums
    DevBuf<int32_t> d_atom_fail;                 // per atom: a particle left the claim in the last launch
    // tiles
    std::vector<int32_t> tile_atom_begin, tile_atoms;
    // Per-step metadata (targets, claims, tiles) goes up in ONE async copy from a pinned staging
    // image; the atoms' end-of-step boxes and travel come back in one async copy behind the kernels.
    std::vector<double> h_tx, h_ty, h_fd;
    std::vector<Box> h_claim;
    bool meta_dirty = true;
    PinnedBuf<unsigned char> stage_up, stage_down;
    DevBuf<unsigned char> d_meta;
    size_t meta_off_ty = 0, meta_off_fd = 0, meta_off_claim = 0, meta_off_tbegin = 0, meta_off_tatoms = 0;
    bool out_copied = false;  // stage_down holds this launch's boxes / travel
    bool targets_moving = false;  // the caller moved targets before the most recent step
    bool eager_boxes = true;  // copy the atoms' boxes / travel back behind every step (the scene re-tiles every step: moving
                              // targets); a scene at rest fetches them only when a tiling needs them
    DevBuf<unsigned char> d_scratch;
    std::vector<LaunchClass> classes;
    int margin = 2;
    bool padded = false;   // the current claims carry motion padding or extra margins (see the commit in do_step)
    int since_tiling = 0;  // committed steps on the current tiling
    int single_tile = 0;  // exact-budget mode: everything in one tile
    int uncut_streak = 0;
    double list_factor = 6.0;  // visit-list capacity per particle, grows on overflow
    size_t list_min = 0;
    // environment of the previous step (L:1731-1744)
    bool has_env = false;
    double env_min_mass = 0, env_max_mass = 0, env_min_radius = 0, env_max_radius = 0;
    double tiled_cell_size = 0;
    // parameters of the step the tiles are being formed for (claims are swept along the follow motion)
    double step_follow_compliance = 57.6, step_damping = 0.9;
    int step_substeps = 2;
    bool pk_allowed = true;  // the (sub-steps, passes) shape of the step fits the packed pipeline's per-pass tables
    // packed pipeline (see PackedClass)
    std::vector<PackedClass> pk;
    std::vector<int32_t> pk_meta_host;       // tile_p0 / grp_tile0 of every packed class
    DevBuf<int32_t> pk_meta, pk_src, pk_atom, pk_tile, pk_nchunks;
    DevBuf<double> pk_pos, pk_prev, pk_wr;
    DevBuf<uint32_t> pk_ckey, pk_lists, pk_rank, pk_sorted, pk_levstart, pk_chunks;
    size_t pk_chunk_words = 0;
    DevBuf<uint16_t> pk_lvl, pk_aslot;
    size_t pk_meta_claims = 0;               // offset (ints) of the tile claims inside pk_meta
    int pk_n = 0, pk_tiles = 0, pk_groups = 0;
    size_t pk_entries = 0;                   // stream words over all packed tiles
    size_t pk_sort_words = 0;                // sorted-list words over all packed groups
    int pk_lev_cap = 255;                    // levels the tables hold; grows when a group's DAG is deeper
    int pk_seen_levels = 0;                  // longest dependency chain (levels) of a pass of the last committed step
    unsigned long long pk_seen_list = 0;     // longest pair stream any packed tile had in one pass of the last committed step
    size_t pk_lev_lds_min = 0;               // out-of-order walk: smallest LDS level array (entries per tile) after a fail_levlds
    bool pk_plan_dirty = true;
    // EGG_OPT_TIMING = 2: one event pair per launch group of the packed pipeline, read back when the step is committed
    struct PkStamp { hipEvent_t a, b; int kind, launches; };
    std::vector<PkStamp> pk_stamps;
    size_t pk_stamps_used = 0;
    EggStatus *h_status = nullptr;  // the most recent launch's status block inside stage_down (pinned)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

constexpr size_t kStatInts = 160;  // one status block, padded to a multiple of 16 bytes
static_assert(sizeof(EggStatus) <= kStatInts * 4, "status block too small");
inline EggStatus *d_stat(System &s, int parity) { return (EggStatus *)(s.d_out.p + parity * kStatInts); }
inline int32_t *d_aabb(System &s) { return s.d_out.p + 2 * kStatInts; }
inline int32_t *d_disp(System &s) { return s.d_out.p + 2 * kStatInts + 4 * s.atoms.size(); }

}  // namespace egghost
using namespace egghost;

struct egg_handle {
    int device = 0;
    System sys[2];
    std::vector<Batch> batches;  // index = id - 1 (ids are never reused, L:999-1000)
    std::vector<int32_t> order;  // indices of the live batches in ascending key = particle layout order
    int64_t next_key = 1;
    int64_t budget_particles[2] = {-1, -1};  // >= 0: particle count of the budget 0.05 N^2 (multi-GPU: global N)
    int64_t n_alive = 0;
    double elapsed = 0, interpolation_alpha = 0;
    egg_stats stats{};
    std::string error;
    int opt_margin = 2;
    int opt_no_fuse = 0;      // 1: never put both types into one launch
    int opt_tile_target = 60;  // islands smaller than a wave share one (a 15-particle yolk blob uses a quarter of its lanes)
    int opt_timing = 0;
    int opt_force_single = 0;
    int opt_spread = 0;  // threads per particle: 0 = automatic (3 for tiles that have a CU to themselves), else 1..4
    int opt_spin_sleep = -1;  // -1 auto
    bool packed_auto = false;  // the automatic choice, made when the white tiles are formed
    int opt_tune = 0;         // EGGSIM_TUNE: developer switches (bit 6: levels, sort and executor as separate launches instead of the fused pass)
    int opt_level_walk = 0;   // packed pipeline, EGG_OPT_LEVEL_WALK: 0 by regime, 1 always in order, 2 out of order wherever the probe allows
    DevBuf<uint32_t> simd_claims;   // egg_pk_levexec_kernel: which SIMDs of a compute unit run an executor wave (zero between launches)
    bool lds_lane_ordered = false;  // one ds_add_rtn serves same-address lanes in ascending lane order (probed at create)
    int opt_packed = -1;      // packed pipeline: -1 automatic (large scenes), 0 never, 1 every eligible class
    int opt_group_particles = 0;  // particles one wave of the packed executor keeps in LDS (16 B each): 0 = by scene size (retile), at most 1280
    int opt_force_global_state = 0;  // test hook: run every tile through the global-memory-state kernel  // threads per particle in the step kernel's workgroups (pair dataflow spreading)
    hipDeviceProp_t prop{};
    size_t lds_limit = 64 * 1024;  // dynamic LDS a step-kernel workgroup may use
    bool in_flight = false;        // egg_step_begin without its egg_step_end
    double flight_delta = 0;
    int flight_s = 0, flight_c = 0;
    // headless renderer (eggsim_render.hip)
    struct Render {
        egg_render_config cfg[2];
        int use_particle_color = 0, use_lighting = 1;  // L:448-449
        int canvas_w[2] = {0, 0}, canvas_h[2] = {0, 0};  // canvases only grow (L:1957-1970)
        double canvas_x0[2] = {0, 0}, canvas_y0[2] = {0, 0};  // world position of the canvases of the last egg_render
        bool canvas_valid = false;
        int last_w[2] = {0, 0}, last_h[2] = {0, 0};  // canvas sizes of the last egg_render
        DevBuf<float4> canvas[2], screen, atom_color;
        DevBuf<float> texture;
        std::vector<float> texture_host;
        int tsize = 0;
        double texture_radius = -1;
        DevBuf<uint32_t> tiles, entries, totals;
    } render;
};
namespace egghost {

int fail(egg_handle *h, int code, const char *fmt, ...);

// state-mutating entry points are refused between egg_step_begin and egg_step_end: the launched step reads the
// arrays and tiles they would change, and egg_step_end validates / commits exactly what was launched
#define REJECT_IN_FLIGHT(h, name)                                                                      \
    do {                                                                                               \
        if ((h)->in_flight) return fail(h, EGG_ERR_INVALID_ARGUMENT, name ": a step is in flight (egg_step_begin without egg_step_end)"); \
    } while (0)

#define HIP_TRY(h, expr)                                                                               \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(h, EGG_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                        __FILE__, __LINE__);                                                           \
    } while (0)


inline double clampd(double x, double lo, double hi) {  // math.lua:16-26
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
inline double mixd(double lo, double hi, double t) { return lo * (1 - t) + hi * t; }  // math.lua:33-35

Batch *find_batch(egg_handle *h, int64_t id);
const Batch *find_batch(const egg_handle *h, int64_t id);

// the per-particle template of a batch: offsets from the centre, mass factor, mass, radius
// (fibonacci_spiral L:907-918, get_mass L:921-938, add_particle L:941-997)
struct ParticleTemplate {
    std::vector<double> dx, dy, t, inv_mass, radius;
};
void make_template(const egg_config &cfg, double batch_radius, int64_t n_particles, ParticleTemplate &out);
int reserve_particles(egg_handle *h, System &s, int64_t need);
int append_particles(egg_handle *h, System &s, const ParticleTemplate &tp, int64_t n_batches, const double *cx, const double *cy);
int reserve_out(egg_handle *h, System &s, size_t na);
hipError_t wait_step(hipStream_t stream);
int upload_atoms(egg_handle *h, int which);
double cell_size_of(const egg_config &c);  // L:1756-1760
int fetch_end_aabb(egg_handle *h, System &s);

// eggsim_host_tiling.hip
extern double g_retile_ms[8];  // developer aid (EGGSIM_HOST_PROFILE=1): wall time of retile()'s sections
int retile(egg_handle *h, int which);

// eggsim_host_step.hip
struct Env {  // scalars of update_environment (L:1726-1774)
    double sub_delta, damping, follow_c, collision_c, budget, cell;
};
Env make_env(const egg_config &c, double sub_delta, int64_t n);
// phase: kWhole = the complete step; kPrepare = tiles/claims only; kBegin = launch the first attempt and
// return (egg_step_begin); kEnd = finish a begun step: validate, re-run if needed, commit (egg_step_end)
enum { kWhole = 0, kPrepare = 1, kBegin = 2, kEnd = 3 };
int do_step(egg_handle *h, double delta, int S, int C, int phase = kWhole);  // L:1722-1989

}  // namespace egghost

