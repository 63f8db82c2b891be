// eggsim_host.hip -- host side of libeggsim.so: handle state, particle creation,
// tiling, launch orchestration and the C ABI of include/eggsim.h.
//
// The reference keeps everything in Lua tables and runs `_step` on the host
// (simulation_handler.lua, "L:").  Here the particle arrays live in HBM as SoA
// (double buffered: the inactive buffer is the reference's last_update_x/y,
// L:1795-1818), the host keeps only batch bookkeeping, and a step is one
// kernel launch per particle type and tile size class.
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

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr size_t kLdsMax = 160 * 1024;  // per CU on gfx950; what a workgroup may use is probed at create
constexpr int kMaxTileParticles = 32000;  // 15-bit local indices in the kernel's pair sequences
constexpr int kMaxListEntries = 60000;    // lists in LDS: 16-bit list positions in the transposition's records
constexpr int kMaxGlobalListEntries = 8 << 20;  // lists in global memory (64-bit records): bounded by memory only

std::string g_create_error;

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
    int lev_lds_cap = 0;    // out-of-order walk: stream entries per tile whose levels the LDS of the launch holds
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
    size_t lds_lists = 0, lds_levels = 0, lds_exec = 0;
    int threads_lists = 64;
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

}  // namespace

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
    int opt_tune = 0;         // EGGSIM_TUNE: developer experiments inside the packed kernels
    int opt_level_walk = 0;   // packed pipeline, EGG_OPT_LEVEL_WALK: 0 by regime, 1 always in order, 2 out of order wherever the probe allows
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

namespace {

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

double clampd(double x, double lo, double hi) {  // math.lua:16-26
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
double mixd(double lo, double hi, double t) { return lo * (1 - t) + hi * t; }  // math.lua:33-35

Batch *find_batch(egg_handle *h, int64_t id) {
    if (id < 1 || id > (int64_t)h->batches.size()) return nullptr;
    Batch *b = &h->batches[(size_t)id - 1];
    return b->alive ? b : nullptr;
}
const Batch *find_batch(const egg_handle *h, int64_t id) { return find_batch(const_cast<egg_handle *>(h), id); }

// ---------------------------------------------------------------- particles

// the per-particle template of a batch: offsets from the centre, mass factor, mass, radius
// (fibonacci_spiral L:907-918, get_mass L:921-938, add_particle L:941-997)
struct ParticleTemplate {
    std::vector<double> dx, dy, t, inv_mass, radius;
};

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

int reserve_out(egg_handle *h, System &s, size_t na);

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

// ------------------------------------------------------------------ tiling

int fetch_end_aabb(egg_handle *h, System &s);

// Groups atoms into tiles.  Atoms whose margin-padded cell boxes come within one cell of
// each other may interact during the step and must share a tile ("islands": connected
// components of that relation); independent islands may additionally be packed into one
// tile to fill a wave.  Each atom's claim box is its padded box: the kernel verifies that
// no particle leaves it, which proves that particles of different tiles never occupy
// adjacent cells (claims of different islands are separated by >= 1 empty cell).
// developer aid (EGGSIM_HOST_PROFILE=1): wall time of retile()'s sections, printed when the handle is destroyed
static double g_retile_ms[8];
struct RetileLap {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(int k) {
        const auto n = std::chrono::steady_clock::now();
        g_retile_ms[k] += std::chrono::duration<double, std::milli>(n - t).count();
        t = n;
    }
};

int retile(egg_handle *h, int which) {
    System &s = h->sys[which];
    RetileLap lap;
    int rc = upload_atoms(h, which);
    if (rc != EGG_OK) return rc;
    const size_t na = s.atoms.size();
    const double cell = cell_size_of(s.cfg);
    s.extra_margin.resize(h->batches.size(), 0);
    s.tile_atom_begin.assign(1, 0);
    s.tile_atoms.clear();
    s.classes.clear();
    h->stats.n_tiles[which] = 0;
    h->stats.max_tile_particles[which] = 0;
    if (na == 0) {
        s.tiling_dirty = false;
        s.tiled_cell_size = cell;
        return EGG_OK;
    }
    if (!s.aabb_valid && s.aabb_on_device && s.tiled_cell_size == cell) {
        rc = fetch_end_aabb(h, s);
        if (rc != EGG_OK) return rc;
    }
    s.tiled_cell_size = cell;
    if (!s.aabb_valid) {
        hipLaunchKernelGGL(egg_atom_bounds_kernel, dim3((unsigned)na), dim3(EGG_WAVE), 0, s.stream, s.x[s.cur].p,
                           s.y[s.cur].p, s.d_atom_offset.p, s.d_atom_count.p, (int)na, cell, d_aabb(s));
        HIP_TRY(h, hipGetLastError());
        s.aabb.resize(na);
        HIP_TRY(h, hipMemcpyAsync(s.aabb.data(), d_aabb(s), na * sizeof(Box), hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(h, hipStreamSynchronize(s.stream));
        s.aabb_valid = true;
        s.disp_valid = false;
        h->stats.kernel_launches++;
    }
    for (const Box &b : s.aabb)
        if (b.lo_x < -2000000000 || b.hi_x > 2000000000 || b.lo_y < -2000000000 || b.hi_y > 2000000000)
            return fail(h, EGG_ERR_UNSUPPORTED, "particle coordinates are not finite or exceed +-2e9 cells");

    lap(0);
    std::vector<Box> claim(na);
    std::vector<int> comp(na);
    const bool single = s.single_tile || h->opt_force_single;
    if (single) {
        Box u{std::numeric_limits<int32_t>::max(), std::numeric_limits<int32_t>::max(),
              std::numeric_limits<int32_t>::min(), std::numeric_limits<int32_t>::min()};
        for (const Box &b : s.aabb) {
            u.lo_x = std::min(u.lo_x, b.lo_x);
            u.lo_y = std::min(u.lo_y, b.lo_y);
            u.hi_x = std::max(u.hi_x, b.hi_x);
            u.hi_y = std::max(u.hi_y, b.hi_y);
        }
        int64_t ext = std::max<int64_t>((int64_t)u.hi_x - u.lo_x, (int64_t)u.hi_y - u.lo_y);
        if (ext > 60000) return fail(h, EGG_ERR_UNSUPPORTED, "single-tile extent of %lld cells is too large", (long long)ext);
        int m = (int)std::min<int64_t>(std::max(s.margin, 64), (60000 - ext) / 2);
        if (m < 1) m = 1;
        u = Box{u.lo_x - m, u.lo_y - m, u.hi_x + m, u.hi_y + m};
        for (size_t k = 0; k < na; ++k) {
            claim[k] = u;
            comp[k] = 0;
        }
    } else {
        // Claim = occupied cells + margin, swept along the displacement the follow constraint is about
        // to cause: per sub-step a particle farther than the slack from its target moves towards it by
        // (d - slack) * w / (w + follow compliance) (L:1461-1468), so a blob far from its target moves
        // many cells in one step in a known direction.  Sweeping the claim that way (with 50 % head room)
        // keeps fast blobs inside their claims without widening every neighbour's margin.
        const int m = s.margin;
        const double w_max = 1.0 / std::max(s.cfg.min_mass, 1e-300);  // lightest particle moves most
        const double pull = w_max / (w_max + s.step_follow_compliance);
        // inertia: x += dt * v carries on what the particles did in the previous step (times the
        // damping); the step kernel reports every atom's largest particle travel per direction
        const bool have_motion = s.disp_valid && s.disp.size() == 4 * na;
        s.swept = false;
        for (size_t k = 0; k < na; ++k) {
            const Box &b = s.aabb[k];
            const Batch &B = h->batches[(size_t)s.atoms[k].batch];
            const double cx = 0.5 * ((double)b.lo_x + b.hi_x + 1.0) * cell, cy = 0.5 * ((double)b.lo_y + b.hi_y + 1.0) * cell;
            const double dx = B.target_x - cx, dy = B.target_y - cy;
            const double dist = std::sqrt(dx * dx + dy * dy);
            const double slack = 2 * std::sqrt(which == EGG_WHITE ? B.white_radius : B.yolk_radius);
            // Predict the step like the solver runs it: per sub-step the displacement is
            // damping * (previous sub-step's displacement) [pre-solve, L:1411-1418] plus the follow pull
            // (remaining distance beyond the slack) * w / (w + compliance) [L:1461-1468], along the
            // direction to the target.  Collisions only redistribute that inside the blob.
            const double ux = dist > 0 ? dx / dist : 0, uy = dist > 0 ? dy / dist : 0;
            const int mk = m + s.extra_margin[(size_t)s.atoms[k].batch];
            int side[4];
            for (int q = 0; q < 4; ++q) {
                const double dir = (q == 0) ? ux : (q == 1) ? -ux : (q == 2) ? uy : -uy;  // +x, -x, +y, -y
                double d = have_motion ? s.disp[4 * k + q] / 16.0 : 0.0;  // last sub-step, towards this side
                double remaining = (dist > slack && std::isfinite(dist)) ? dist - slack : 0.0;
                double travel = 0;
                for (int sub = 0; sub < s.step_substeps; ++sub) {
                    const double p = remaining * pull;
                    remaining -= p;
                    d = s.step_damping * d + std::max(0.0, dir) * p;
                    travel += d;
                }
                side[q] = std::max(mk, (int)std::min(4096.0, std::ceil(1.25 * travel / cell)));
                s.swept |= side[q] > mk;
            }
            claim[k] = Box{b.lo_x - side[1], b.lo_y - side[3], b.hi_x + side[0], b.hi_y + side[2]};
        }
        lap(1);
        // union-find over atoms
        std::vector<int> parent(na);
        std::iota(parent.begin(), parent.end(), 0);
        auto find = [&](int v) {
            while (parent[v] != v) {
                parent[v] = parent[parent[v]];
                v = parent[v];
            }
            return v;
        };
        auto touch = [&](int a, int b) {  // not separated by a full empty cell column or row
            const Box &A = claim[(size_t)a], &B = claim[(size_t)b];
            if ((int64_t)B.lo_x > (int64_t)A.hi_x + 1 || (int64_t)A.lo_x > (int64_t)B.hi_x + 1) return;
            if ((int64_t)B.lo_y > (int64_t)A.hi_y + 1 || (int64_t)A.lo_y > (int64_t)B.hi_y + 1) return;
            int ra = find(a), rb = find(b);
            if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
        };
        // Candidate pairs from a grid of square buckets as wide as an ordinary claim (+ the empty cell): two
        // such claims that touch have their low corners in the same or in adjacent buckets.  The few claims wider
        // than a bucket (a blob flying towards a far target) are tested against everything.
        int64_t ext_sum = 0, ext_max = 0, min_x = std::numeric_limits<int64_t>::max(), min_y = min_x, max_x = -min_x, max_y = -min_x;
        auto ext_of = [](const Box &c) { return std::max<int64_t>((int64_t)c.hi_x - c.lo_x, (int64_t)c.hi_y - c.lo_y) + 2; };
        for (const Box &c : claim) {
            const int64_t e = ext_of(c);
            ext_sum += e;
            ext_max = std::max(ext_max, e);
            min_x = std::min<int64_t>(min_x, c.lo_x);
            min_y = std::min<int64_t>(min_y, c.lo_y);
            max_x = std::max<int64_t>(max_x, c.lo_x);
            max_y = std::max<int64_t>(max_y, c.lo_y);
        }
        const int64_t ext_mean = ext_sum / (int64_t)na + 1;
        const int64_t bucket = ext_max <= 4 * ext_mean ? ext_max : 2 * ext_mean;
        const int64_t nbx = (max_x - min_x) / bucket + 1, nby = (max_y - min_y) / bucket + 1;
        if (nbx * nby <= 8 * (int64_t)na + 1024) {
            std::vector<int32_t> cell_of(na), start((size_t)(nbx * nby) + 1, 0), member(na), wide;
            for (size_t k = 0; k < na; ++k) {
                const Box &c = claim[k];
                if (ext_of(c) > bucket) {
                    cell_of[k] = -1;
                    wide.push_back((int32_t)k);
                    continue;
                }
                cell_of[k] = (int32_t)(((int64_t)c.lo_x - min_x) / bucket * nby + ((int64_t)c.lo_y - min_y) / bucket);
                start[(size_t)cell_of[k] + 1]++;
            }
            for (size_t c = 0; c < (size_t)(nbx * nby); ++c) start[c + 1] += start[c];
            {
                std::vector<int32_t> fill(start.begin(), start.end() - 1);
                for (size_t k = 0; k < na; ++k)
                    if (cell_of[k] >= 0) member[(size_t)fill[(size_t)cell_of[k]]++] = (int32_t)k;
            }
            for (size_t k = 0; k < na; ++k) {
                const int32_t c = cell_of[k];
                if (c < 0) continue;
                const int64_t bx = c / nby, by = c % nby;
                // own bucket (later members) and the four forward neighbours: every pair once
                for (int32_t m = start[(size_t)c]; m < start[(size_t)c + 1]; ++m)
                    if (member[(size_t)m] > (int32_t)k) touch((int)k, member[(size_t)m]);
                const int64_t nb[4][2] = {{bx, by + 1}, {bx + 1, by - 1}, {bx + 1, by}, {bx + 1, by + 1}};
                for (const auto &q : nb) {
                    if (q[0] >= nbx || q[1] < 0 || q[1] >= nby) continue;
                    const size_t c2 = (size_t)(q[0] * nby + q[1]);
                    for (int32_t m = start[c2]; m < start[c2 + 1]; ++m) touch((int)k, member[(size_t)m]);
                }
            }
            for (int32_t wk : wide)
                for (size_t k = 0; k < na; ++k)
                    if ((int32_t)k != wk) touch(wk, (int)k);
        } else {
            // atoms scattered over far more buckets than there are atoms: sweep over lo_x
            std::vector<int> order(na);
            std::iota(order.begin(), order.end(), 0);
            std::sort(order.begin(), order.end(), [&](int a, int b) { return claim[a].lo_x < claim[b].lo_x; });
            for (size_t i = 0; i < na; ++i)
                for (size_t j = i + 1; j < na; ++j) {
                    if ((int64_t)claim[order[j]].lo_x > (int64_t)claim[order[i]].hi_x + 1) break;
                    touch(order[i], order[j]);
                }
        }
        for (size_t k = 0; k < na; ++k) comp[k] = find((int)k);
    }

    lap(2);
    // islands: atoms grouped by component (in the order of each component's first atom), ascending atom index
    // inside (keeps particle order); flat arrays, this runs every step while blobs move
    std::vector<int32_t> island_of(na, -1), isl_begin(1, 0), isl_atoms(na);
    {
        int32_t n_isl = 0;
        std::vector<int32_t> size;
        for (size_t k = 0; k < na; ++k) {
            const int r = comp[k];
            if (island_of[(size_t)r] < 0) {
                island_of[(size_t)r] = n_isl++;
                size.push_back(0);
            }
            size[(size_t)island_of[(size_t)r]]++;
        }
        isl_begin.resize((size_t)n_isl + 1);
        for (int32_t i = 0; i < n_isl; ++i) isl_begin[(size_t)i + 1] = isl_begin[(size_t)i] + size[(size_t)i];
        std::vector<int32_t> fill(isl_begin.begin(), isl_begin.end() - 1);
        for (size_t k = 0; k < na; ++k) isl_atoms[(size_t)fill[(size_t)island_of[(size_t)comp[k]]]++] = (int32_t)k;
    }
    struct TileTmp {  // atoms: isl_atoms[a_begin, a_end) -- islands that share a tile are neighbours in that array
        int32_t a_begin = 0, a_end = 0;
        int64_t particles = 0;
        Box box{std::numeric_limits<int32_t>::max(), std::numeric_limits<int32_t>::max(),
                std::numeric_limits<int32_t>::min(), std::numeric_limits<int32_t>::min()};
    };
    auto grow = [](Box a, const Box &b) {
        a.lo_x = std::min(a.lo_x, b.lo_x);
        a.lo_y = std::min(a.lo_y, b.lo_y);
        a.hi_x = std::max(a.hi_x, b.hi_x);
        a.hi_y = std::max(a.hi_y, b.hi_y);
        return a;
    };
    auto extent = [](const Box &b) { return std::max<int64_t>((int64_t)b.hi_x - b.lo_x, (int64_t)b.hi_y - b.lo_y); };
    std::vector<TileTmp> tiles;
    tiles.reserve(isl_begin.size());
    const int64_t target = single ? 0 : h->opt_tile_target;
    for (size_t i = 0; i + 1 < isl_begin.size(); ++i) {
        TileTmp one;
        one.a_begin = isl_begin[i];
        one.a_end = isl_begin[i + 1];
        for (int32_t k = one.a_begin; k < one.a_end; ++k) {
            const int32_t a = isl_atoms[(size_t)k];
            one.particles += s.atoms[(size_t)a].count;
            one.box = grow(one.box, claim[(size_t)a]);
        }
        // independent islands share a tile only while the joint claim box stays small enough for the dense cell
        // grid: a tile spanning the scene falls back to the hash table, which made such tiles (and with them
        // the whole launch) 7x slower
        auto grid_cells = [](const Box &b) { return ((int64_t)b.hi_x - b.lo_x + 4) * ((int64_t)b.hi_y - b.lo_y + 4); };
        if (target > 0 && !tiles.empty() && tiles.back().particles + one.particles <= target &&
            extent(grow(tiles.back().box, one.box)) <= 60000 && grid_cells(grow(tiles.back().box, one.box)) <= 2048) {
            // independent islands may share a tile (fills the wave's lanes in the pair executor)
            TileTmp &tt = tiles.back();
            tt.a_end = one.a_end;
            tt.particles += one.particles;
            tt.box = grow(tt.box, one.box);
        } else {
            tiles.push_back(one);
        }
    }
    for (auto &tt : tiles) {
        std::sort(isl_atoms.begin() + tt.a_begin, isl_atoms.begin() + tt.a_end);
        if (tt.particles > kMaxTileParticles)
            return fail(h, EGG_ERR_UNSUPPORTED,
                        "%lld particles of one type interact in one island; the LDS tile kernel handles at most %d",
                        (long long)tt.particles, kMaxTileParticles);
        // (tt.box is the union of the tile's claims)
        const int64_t ext_x = (int64_t)tt.box.hi_x - tt.box.lo_x, ext_y = (int64_t)tt.box.hi_y - tt.box.lo_y;
        if (ext_x > 65000 || ext_y > 65000)
            return fail(h, EGG_ERR_UNSUPPORTED, "a tile spans %lld x %lld cells; limit is 65000",
                        (long long)ext_x, (long long)ext_y);
    }
    // big tiles first, and similar sizes adjacent so that they share a launch class
    std::stable_sort(tiles.begin(), tiles.end(),
                     [](const TileTmp &a, const TileTmp &b) { return a.particles > b.particles; });

    s.tile_atoms.reserve(na);
    s.tile_atom_begin.reserve(tiles.size() + 1);
    for (auto &tt : tiles) {
        s.tile_atoms.insert(s.tile_atoms.end(), isl_atoms.begin() + tt.a_begin, isl_atoms.begin() + tt.a_end);
        s.tile_atom_begin.push_back((int32_t)s.tile_atoms.size());
    }
    lap(3);
    // launch classes: consecutive tiles whose particle count is within 2x
    size_t t0 = 0;
    size_t scratch_bytes = 0;
    while (t0 < tiles.size()) {
        size_t t1 = t0;
        int64_t nmax = tiles[t0].particles;
        int amax = 0;
        int64_t max_cells = 0;
        while (t1 < tiles.size() && tiles[t1].particles * 2 > nmax) {
            amax = std::max(amax, (int)(tiles[t1].a_end - tiles[t1].a_begin));
            // dense grid the kernel lays over the tile's claim box (see eggsim_step.hip, load tile)
            const Box &bx = tiles[t1].box;
            max_cells = std::max(max_cells, ((int64_t)bx.hi_x - bx.lo_x + 4) * ((int64_t)bx.hi_y - bx.lo_y + 4));
            ++t1;
        }
        LaunchClass lc;
        lc.first_tile = (int)t0;
        lc.n_tiles = (int)(t1 - t0);
        lc.nmax = (int)((nmax + 7) & ~7ll);
        lc.amax = amax;
        if (max_cells <= std::max<int64_t>(2048, 8 * (int64_t)lc.nmax) && max_cells <= 16384) {
            lc.use_grid = 1;
            lc.ccap = (int)((max_cells + 63) & ~63ll);
        } else {
            lc.use_grid = 0;
            int ht = 64;
            while (ht < lc.nmax + lc.nmax / 2) ht *= 2;
            lc.ccap = ht;
        }
        size_t lcap = std::max<size_t>({(size_t)lc.nmax, (size_t)(s.list_factor * lc.nmax), s.list_min});
        lcap = std::min<size_t>(lcap, kMaxListEntries);
        lcap = (lcap + 7) & ~(size_t)7;
        lc.lcap = (int)lcap;
        int threads = egg_step_threads(lc.nmax, 1);
        lc.lds = egg_step_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, lc.lcap, single ? 1 : 0, 0, threads, 0, s.gens);
        bool want_global_state = h->opt_force_global_state != 0;
        if (!want_global_state && (lc.lds > h->lds_limit || lc.lds > 64 * 1024 || threads > 256)) {
            // dense or large tiles: particle state stays in LDS, the visit lists go to global memory
            lc.global_lists = 1;
            lcap = std::max<size_t>({lcap, (size_t)(32.0 * lc.nmax), s.list_min});
            lcap = std::min<size_t>((lcap + 7) & ~(size_t)7, kMaxGlobalListEntries);
            lc.lcap = (int)lcap;
            lc.lds = egg_step_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, lc.lcap, single ? 1 : 0, 1, threads, 0, s.gens);
            if (lc.lds > h->lds_limit) want_global_state = true;
        }
        if (want_global_state) {
            // the particle state itself does not fit (a very large island): everything goes to the tile's
            // scratch slice, laid out like the LDS image followed by the lists
            lc.global_lists = lc.global_state = 1;
            lcap = std::max<size_t>({lcap, (size_t)(32.0 * lc.nmax), s.list_min});
            lc.lcap = (int)std::min<size_t>((lcap + 7) & ~(size_t)7, kMaxGlobalListEntries);
            const size_t state = egg_step_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, lc.lcap, single ? 1 : 0, 1, threads, 0, s.gens);
            lc.scratch_stride = ((state + 255) & ~(size_t)255) + egg_step_scratch_bytes(lc.lcap, single ? 1 : 0, s.gens) + 256;
            lc.lds = 0;
        } else if (lc.global_lists) {
            lc.scratch_stride = (egg_step_scratch_bytes(lc.lcap, single ? 1 : 0, s.gens) + 255) & ~(size_t)255;
        }
        // A tile that has a CU (almost) to itself leaves most of the CU's issue slots idle: give it three
        // lanes per particle, which the kernel uses to build the visit lists column-wise.  With more
        // tiles than that, one lane per particle keeps the most tiles resident.
        if (!lc.global_lists && !lc.global_state && s.gens <= 2) {
            int spread = h->opt_spread;
            // (two such workgroups do not fit one CU's register file, so: at most one tile per CU)
            // -- and only while the white tiles do not fill the chip anyway (then the types share one launch)
            const bool chip_shared = which == 1 && h->stats.n_tiles[0] > h->prop.multiProcessorCount;
            if (spread <= 0) spread = (lc.n_tiles <= h->prop.multiProcessorCount && !chip_shared) ? 3 : 1;
            const int wide_threads = egg_step_threads(lc.nmax, spread);
            if (spread > 1 && wide_threads <= 512 && wide_threads >= 3 * lc.nmax) {
                // (a wide tile keeps velocities and sub-step start positions in LDS: the size depends on the thread count)
                const size_t wide_lds = egg_step_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, lc.lcap, single ? 1 : 0, 0,
                                                           wide_threads, 0, s.gens);
                if (wide_lds <= h->lds_limit && wide_lds <= 64 * 1024) {
                    threads = wide_threads;
                    lc.wide = 1;
                    lc.lds = wide_lds;
                }
            }
            // room to spare (few tiles per CU): cache the position-independent terms of every pair
            const size_t with_cache = egg_step_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, lc.lcap, single ? 1 : 0, 0,
                                                         threads, 1, s.gens);
            const int tiles_per_cu = (lc.n_tiles + h->prop.multiProcessorCount - 1) / h->prop.multiProcessorCount;
            if (with_cache <= h->lds_limit && with_cache <= 64 * 1024 && (size_t)tiles_per_cu * with_cache <= 96 * 1024) {
                lc.pair_cache = 1;
                lc.lds = with_cache;
            }
        }
        lc.threads = threads;
        if (lc.global_lists) {
            scratch_bytes = (scratch_bytes + 255) & ~(size_t)255;
            lc.scratch_offset = scratch_bytes;
            scratch_bytes += (size_t)lc.n_tiles * lc.scratch_stride;
        }
        s.classes.push_back(lc);
        t0 = t1;
    }

    lap(4);
    // ---- packed pipeline (eggsim_packed.hip): which classes take it, their packed ranges and groups
    s.pk.clear();
    s.pk_meta_host.clear();
    s.pk_n = s.pk_tiles = s.pk_groups = 0;
    s.pk_entries = 0;
    s.pk_sort_words = 0;
    s.pk_chunk_words = 0;
    {
        // automatic: scenes large enough that the chip is full of tiles whatever the kernel (the fused kernel's
        // latency per step is lower while every tile has a CU almost to itself); judged on the white tiles so
        // that both types of a scene take the same path
        // Measured crossover on MI355X (ms per step, fused vs packed): separate 157-particle blobs 1024: 0.58 / 0.68,
        // 1536: 0.78 / 0.73, 2048: 1.06 / 0.79, 3072: 1.38 / 0.86; dense 628-particle islands 192: 1.65 / 2.79, 256: 1.71 /
        // 2.82, 320: 2.99 / 2.85, 512: 3.20 / 2.92, 768: 4.92 / 3.09 -- the packed pipeline has a latency floor per
        // collision pass; the fused kernel's time grows with the tiles per CU, and a dense island fills a CU: from the
        // first CU that gets two, its step takes twice as long.
        if (which == 0) {
            const int64_t cus = std::max(1, h->prop.multiProcessorCount);
            const bool dense = !tiles.empty() && tiles.front().particles > 256;
            h->packed_auto = dense ? (int64_t)tiles.size() > cus : (int64_t)tiles.size() >= 6 * cus;
        }
        const bool want = h->opt_packed > 0 || (h->opt_packed < 0 && h->packed_auto);
        const bool allowed = want && !single && s.gens <= 2 && s.pk_allowed;
        for (size_t ci = 0; allowed && ci < s.classes.size(); ++ci) {
            LaunchClass &lc = s.classes[ci];
            if (lc.nmax > 8192) continue;  // (a class whose FUSED kernel would keep its state in global memory may still fit: the list kernel's LDS need is checked below)
            PackedClass pc;
            pc.cls = (int)ci;
            pc.n_tiles = lc.n_tiles;
            pc.lcap = lc.lcap;
            pc.scap = lc.lcap;
            pc.threads_lists = egg_step_threads(lc.nmax, 1);  // (capping it at 512 to fit a fourth dense tile per CU cost 50 %)
            // the counting pass keeps up to stage_cap partners per particle in LDS as long as that does not cost a
            // resident tile per CU (residency: LDS and the 32-wave limit)
            auto tiles_per_cu = [&](int stage) {
                const size_t lds = egg_pk_lists_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, stage);
                if (lds > h->lds_limit) return (size_t)0;  // (what a workgroup may have is a little less than the CU's 160 KiB)
                return std::min<size_t>(kLdsMax / std::max<size_t>(lds, 1), (size_t)2048 / (size_t)pc.threads_lists);
            };
            for (pc.stage_cap = 16; pc.stage_cap > 0; pc.stage_cap -= 4)
                if (tiles_per_cu(pc.stage_cap) == tiles_per_cu(0)) break;
            pc.lds_lists = egg_pk_lists_lds_bytes(lc.nmax, lc.amax, lc.ccap, lc.use_grid, pc.stage_cap);
            if (pc.lds_lists > h->lds_limit) continue;
            const size_t meta_mark = s.pk_meta_host.size();
            pc.p_begin = s.pk_n;
            pc.entry_base = s.pk_entries;
            pc.sort_base = s.pk_sort_words;
            pc.tile_base = s.pk_tiles;
            pc.group_base = s.pk_groups;
            // tile records: first packed particle, particles, first tile atom, atoms, cell origin, grid extent
            pc.meta_tile_geo = (s.pk_meta_host.size() + 3) & ~(size_t)3;  // 16-byte aligned records
            s.pk_meta_host.resize(pc.meta_tile_geo);
            int pn = s.pk_n;
            std::vector<int> tile_first_particle((size_t)lc.n_tiles + 1);
            for (int t = 0; t < lc.n_tiles; ++t) {
                const size_t gt = (size_t)lc.first_tile + t;
                const int a0 = s.tile_atom_begin[gt], a1 = s.tile_atom_begin[gt + 1];
                int32_t lx = std::numeric_limits<int32_t>::max(), ly = lx, hx = std::numeric_limits<int32_t>::min(), hy = hx;
                for (int k = a0; k < a1; ++k) {
                    const Box &c = claim[(size_t)s.tile_atoms[(size_t)k]];
                    lx = std::min(lx, c.lo_x);
                    ly = std::min(ly, c.lo_y);
                    hx = std::max(hx, c.hi_x);
                    hy = std::max(hy, c.hi_y);
                }
                tile_first_particle[(size_t)t] = pn;
                const int32_t rec[8] = {pn, (int32_t)tiles[gt].particles, a0, a1 - a0, lx - 2, ly - 2,
                                        (int32_t)std::min<int64_t>((int64_t)hx - lx + 4, 65535), (int32_t)std::min<int64_t>((int64_t)hy - ly + 4, 65535)};
                s.pk_meta_host.insert(s.pk_meta_host.end(), rec, rec + 8);
                pn += (int)tiles[gt].particles;
            }
            tile_first_particle[(size_t)lc.n_tiles] = pn;
            pc.p_end = pn;
            // groups: consecutive tiles while one wave's LDS holds their positions (tiles are sorted by size, largest first)
            pc.meta_grp_geo = s.pk_meta_host.size();
            // Particles per executor wave.  1280 fills the lanes best on a full chip (eight 157-particle blobs: ~80 pairs
            // per level).  While the class's groups are fewer than the chip's SIMDs the executor and the level walk are
            // latency bound -- a level costs the same ~0.3 us whether its chunk holds 10 pairs or 64, two chunks of one
            // level cost twice that -- so smaller groups (more waves, one chunk per level) are faster: measured ms per
            // step with 320 / 640 / 1280 particles per group, 2048 blobs: 0.77 / 0.82 / 1.01, 4096: 1.03 / 0.94 / 0.99,
            // 8192: 1.59 / 1.32 / 1.32.  Dense islands (four coincident blobs, 628 particles) keep 1280: 3.32 vs 3.54.
            int64_t gp_auto = 1280;
            if (tiles[(size_t)lc.first_tile].particles <= 256) {
                int64_t class_particles = 0;
                for (int t = 0; t < lc.n_tiles; ++t) class_particles += tiles[(size_t)lc.first_tile + t].particles;
                const int64_t simds = 4 * (int64_t)std::max(1, h->prop.multiProcessorCount);
                gp_auto = std::min<int64_t>(1280, std::max<int64_t>(320, (class_particles / simds + 159) / 160 * 160));
            }
            const int64_t gp_max = std::max<int64_t>(h->opt_group_particles > 0 ? h->opt_group_particles : gp_auto,
                                                     tiles[(size_t)lc.first_tile].particles);
            int max_tiles_in_group = 0;
            for (int t = 0; t < lc.n_tiles;) {
                int64_t in_group = 0;
                int t_end = t;
                while (t_end < lc.n_tiles && t_end - t < 64) {
                    const int64_t np = tiles[(size_t)lc.first_tile + t_end].particles;
                    if (t_end > t && (in_group + np > gp_max || in_group + np > 32767)) break;
                    in_group += np;
                    ++t_end;
                }
                const int32_t rec[4] = {t, t_end, tile_first_particle[(size_t)t], (int32_t)in_group};
                s.pk_meta_host.insert(s.pk_meta_host.end(), rec, rec + 4);
                pc.n_groups++;
                pc.max_group_particles = std::max<int>(pc.max_group_particles, (int)in_group);
                max_tiles_in_group = std::max(max_tiles_in_group, t_end - t);
                t = t_end;
            }
            // The level walk.  Dense islands (more than 256 particles: thousands of pairs per pass in hundreds of levels)
            // on a chip that is not full -- groups no more than SIMDs -- wait for the longest dependency chain of one
            // tile: the out-of-order walk levels whatever runs are ready (a 628-particle island's 6,500 pairs: ~165 turns
            // instead of ~540) and sorts in the same launch.  Its fixed costs (ranking pass, barriers, ~40 us) lose against
            // the in-order walk on sparse 157-particle tiles at every scene size (2,048 blobs: 0.41 vs 0.27 ms per step,
            // 16,384: 1.48 vs 0.54; profiles/r03_walk_sweep.txt), and on a full chip its extra instructions do.
            const int simds = 4 * std::max(1, h->prop.multiProcessorCount);
            const bool dense_tiles = tiles[(size_t)lc.first_tile].particles > 256;
            pc.levels_ooo = h->lds_lane_ordered && h->opt_level_walk != 1 && (h->opt_level_walk == 2 || (pc.n_groups <= simds && dense_tiles));
            if (pc.levels_ooo) {
                // the levels of a tile's stream live in LDS: room for 24 pairs per particle (a dense island's first
                // steps: ~20), more after a launch that needed more, never more than the stream itself can hold
                pc.lev_lds_cap = (int)std::min<size_t>((size_t)pc.scap, std::max<size_t>((size_t)24 * lc.nmax, s.pk_lev_lds_min));
                pc.lev_lds_cap = (pc.lev_lds_cap + 7) & ~7;
                pc.levels_threads = 64 * std::min(16, std::max((h->opt_tune & 8) ? 8 : 4, max_tiles_in_group));  // a wave per tile, at least four per group (eight were 10 % slower)
                pc.lds_levels = egg_pk_levels_ooo_lds_bytes(s.pk_lev_cap, pc.max_group_particles, max_tiles_in_group, pc.lev_lds_cap);
                if (pc.lds_levels > h->lds_limit) pc.levels_ooo = 0;  // (a stream too long for LDS: the in-order walk)
            }
            if (!pc.levels_ooo) {
                pc.levels_threads = std::min(256, (max_tiles_in_group * 16 + 63) / 64 * 64);
                pc.lds_levels = egg_pk_levels_mr_lds_bytes(s.pk_lev_cap, pc.max_group_particles, pc.levels_threads);
            }
            const size_t sort_words = (size_t)max_tiles_in_group * (size_t)pc.scap + 64;
            pc.chunk_cap = (int)std::min<size_t>(sort_words / 64 + (size_t)s.pk_lev_cap + 8, (size_t)1 << 28);
            pc.lds_exec = (size_t)pc.max_group_particles * 16;  // (the chain variant adds a spare slot per lane at launch: 1,280 particles + 64 would cost the eighth wave of a CU)
            pc.lds_sort = egg_align16((size_t)(s.pk_lev_cap + 2) * 4) + sort_words * 4;
            if (pc.lds_sort > 64 * 1024) pc.lds_sort = 0;
            if (pc.lds_exec + 64 * 16 > h->lds_limit || pc.lds_levels > h->lds_limit || sort_words >= ((size_t)1 << 26)) {
                s.pk_meta_host.resize(meta_mark);
                continue;
            }
            pc.sort_cap = (int)sort_words;
            pc.chunk_base = s.pk_chunk_words;
            s.pk_chunk_words += (size_t)pc.n_groups * (size_t)pc.chunk_cap;
            lc.packed = (int)s.pk.size();
            s.pk_n = pn;
            s.pk_tiles += lc.n_tiles;
            s.pk_groups += pc.n_groups;
            s.pk_entries += (size_t)lc.n_tiles * (size_t)pc.scap;
            s.pk_sort_words += (size_t)pc.n_groups * (size_t)pc.sort_cap;
            s.pk.push_back(pc);
        }
        if (!s.pk.empty()) {
            // the claims of every tile atom, in tile order (one load per atom slot in the kernels)
            s.pk_meta_claims = (s.pk_meta_host.size() + 3) & ~(size_t)3;
            s.pk_meta_host.resize(s.pk_meta_claims);
            for (int32_t a : s.tile_atoms) {
                const Box &c = claim[(size_t)a];
                const int32_t rec[4] = {c.lo_x, c.lo_y, c.hi_x, c.hi_y};
                s.pk_meta_host.insert(s.pk_meta_host.end(), rec, rec + 4);
            }
            lap(5);
            const size_t np = (size_t)s.pk_n, nt = (size_t)s.pk_tiles, ng = (size_t)s.pk_groups;
            HIP_TRY(h, s.pk_meta.reserve(s.pk_meta_host.size() + 4, false, s.stream));
            HIP_TRY(h, s.pk_src.reserve(np, false, s.stream));
            HIP_TRY(h, s.pk_atom.reserve(np, false, s.stream));
            HIP_TRY(h, s.pk_aslot.reserve(np + 8, false, s.stream));
            HIP_TRY(h, s.pk_pos.reserve(2 * np, false, s.stream));
            HIP_TRY(h, s.pk_prev.reserve(2 * np, false, s.stream));
            HIP_TRY(h, s.pk_wr.reserve(2 * np, false, s.stream));
            HIP_TRY(h, s.pk_ckey.reserve(2 * np, false, s.stream));
            HIP_TRY(h, s.pk_lists.reserve(s.pk_entries + 64, false, s.stream));
            HIP_TRY(h, s.pk_lvl.reserve(s.pk_entries + 64, false, s.stream));
            bool any_ooo = false;
            for (const PackedClass &pc : s.pk) any_ooo |= pc.levels_ooo != 0;
            if (any_ooo) HIP_TRY(h, s.pk_rank.reserve(s.pk_entries + 64, false, s.stream));
            HIP_TRY(h, s.pk_sorted.reserve(s.pk_sort_words + 64, false, s.stream));
            HIP_TRY(h, s.pk_chunks.reserve(s.pk_chunk_words + 64, false, s.stream));
            HIP_TRY(h, s.pk_nchunks.reserve(2 * ng + 8, false, s.stream));
            HIP_TRY(h, s.pk_levstart.reserve(ng * ((size_t)s.pk_lev_cap + 2) + 64, false, s.stream));
            HIP_TRY(h, s.pk_tile.reserve(nt * (3 + 2 * EGG_PK_MAX_PASSES) + 4, false, s.stream));
        }
        s.pk_plan_dirty = true;
    }

    HIP_TRY(h, s.d_scratch.reserve(scratch_bytes + 16, false, s.stream));
    lap(6);
    s.h_claim = claim;
    s.meta_dirty = true;
    s.tiling_dirty = false;
    h->stats.retiles++;
    h->stats.n_tiles[which] = (int64_t)tiles.size();
    h->stats.max_tile_particles[which] = tiles.empty() ? 0 : tiles.front().particles;
    lap(7);
    return EGG_OK;
}

// ------------------------------------------------------------------- step

struct Env {  // scalars of update_environment (L:1726-1774)
    double sub_delta, damping, follow_c, collision_c, budget, cell;
};

Env make_env(const egg_config &c, double sub_delta, int64_t n) {
    Env e;
    e.sub_delta = sub_delta;
    auto compliance = [&](double strength) {  // L:1337-1341
        double alpha = 1 - clampd(strength, 0, 1);
        return alpha / (sub_delta * sub_delta);
    };
    e.damping = 1 - clampd(c.damping, 0, 1);
    e.follow_c = compliance(c.follow_strength);
    e.collision_c = compliance(c.collision_strength);
    double nn = (double)n;
    e.budget = c.max_collision_fraction * (nn * nn);
    e.cell = cell_size_of(c);
    return e;
}

// Start of a launch of one type on `stream`: next status parity, upload of the tile / target image if it changed.
int launch_prologue(egg_handle *h, int which, hipStream_t stream) {
    System &s = h->sys[which];
    s.parity ^= 1;  // this launch's status block; it re-initialises the other one for the next launch
    s.aabb_on_device = false;  // the launch overwrites the atoms' boxes
    s.out_copied = false;
    s.wait_stream = stream;
    const size_t na = s.atoms.size();
    if (s.meta_dirty) {
        auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const size_t nt = s.tile_atom_begin.size();
        s.meta_off_ty = up16(na * 8);
        s.meta_off_fd = s.meta_off_ty + up16(na * 8);
        s.meta_off_claim = s.meta_off_fd + up16(na * 8);
        s.meta_off_tbegin = s.meta_off_claim + up16(na * sizeof(Box));
        s.meta_off_tatoms = s.meta_off_tbegin + up16(nt * 4);
        const size_t bytes = s.meta_off_tatoms + up16(s.tile_atoms.size() * 4 + 4);
        HIP_TRY(h, s.stage_up.reserve(bytes));
        HIP_TRY(h, s.d_meta.reserve(bytes, false, stream));
        unsigned char *b = s.stage_up.p;
        memcpy(b, s.h_tx.data(), na * 8);
        memcpy(b + s.meta_off_ty, s.h_ty.data(), na * 8);
        memcpy(b + s.meta_off_fd, s.h_fd.data(), na * 8);
        memcpy(b + s.meta_off_claim, s.h_claim.data(), na * sizeof(Box));
        memcpy(b + s.meta_off_tbegin, s.tile_atom_begin.data(), nt * 4);
        memcpy(b + s.meta_off_tatoms, s.tile_atoms.data(), s.tile_atoms.size() * 4);
        // safe to reuse the staging image: every earlier copy out of it has completed (each step ends
        // with a stream synchronise)
        HIP_TRY(h, hipMemcpyAsync(s.d_meta.p, b, bytes, hipMemcpyHostToDevice, stream));
        s.meta_dirty = false;
    }
    return EGG_OK;
}

void fill_args(egg_handle *h, int which, const LaunchClass &lc, const Env &env, int S, int C, EggStepArgs &A) {
    System &s = h->sys[which];
    memset(&A, 0, sizeof A);
    const int in = s.cur, out = s.cur ^ 1;
    A.x_in = s.x[in].p;
    A.y_in = s.y[in].p;
    A.vx_in = s.vx[in].p;
    A.vy_in = s.vy[in].p;
    A.x_out = s.x[out].p;
    A.y_out = s.y[out].p;
    A.vx_out = s.vx[out].p;
    A.vy_out = s.vy[out].p;
    A.inv_mass = s.inv_mass.p;
    A.radius = s.radius.p;
    A.atom_offset = s.d_atom_offset.p;
    A.atom_count = s.d_atom_count.p;
    A.atom_batch = s.d_atom_batch.p;
    A.atom_tx = (const double *)s.d_meta.p;
    A.atom_ty = (const double *)(s.d_meta.p + s.meta_off_ty);
    A.atom_fd = (const double *)(s.d_meta.p + s.meta_off_fd);
    A.atom_claim = (const int32_t *)(s.d_meta.p + s.meta_off_claim);
    A.atom_aabb_out = d_aabb(s);
    A.atom_fail = s.d_atom_fail.p;
    A.atom_disp_out = d_disp(s);
    A.tile_atom_begin = (const int32_t *)(s.d_meta.p + s.meta_off_tbegin) + lc.first_tile;
    A.tile_atoms = (const int32_t *)(s.d_meta.p + s.meta_off_tatoms);
    A.n_tiles = lc.n_tiles;
    A.sub_delta = env.sub_delta;
    A.damping = env.damping;
    A.follow_compliance = env.follow_c;
    A.collision_compliance = env.collision_c;
    A.overlap_factor = s.cfg.collision_overlap_factor;
    A.cell_size = env.cell;
    A.eps = s.cfg.eps;
    A.budget = env.budget;
    A.single_tile = (s.single_tile || h->opt_force_single) ? 1 : 0;
    A.n_substeps = S;
    A.n_collision_steps = C;
    A.nmax = lc.nmax;
    A.amax = lc.amax;
    A.ccap = lc.ccap;
    A.use_grid = lc.use_grid;
    A.lcap = lc.lcap;
    // more tiles than the chip can hold at one per CU: idle waves yield their issue slots
    A.spin_sleep = (h->opt_spin_sleep < 0) ? ((lc.n_tiles > 2 * h->prop.multiProcessorCount || lc.global_state) ? 1 : 0)
                                           : h->opt_spin_sleep;
    A.pair_cache = lc.pair_cache;
    A.threads = lc.threads;
    A.gens = s.gens;
    A.status = d_stat(s, s.parity);
    A.status_next = d_stat(s, s.parity ^ 1);
    A.scratch = s.d_scratch.p + lc.scratch_offset;
    A.scratch_stride = lc.scratch_stride;
}

// End of a launch: one copy brings back the status blocks and, behind them, the atoms' end-of-step cell boxes
// and last-sub-step travel: the next tiling (every step while targets move) then needs no further round trip.
int launch_epilogue(egg_handle *h, int which, hipStream_t stream) {
    System &s = h->sys[which];
    const size_t na = s.atoms.size();
    // 32 B per atom; a scene that is not re-tiling (and a very large one) fetches the boxes when a tiling needs them
    const bool with_boxes = na && na <= (size_t)4 << 20 && s.eager_boxes;
    const size_t bytes = (2 * kStatInts + (with_boxes ? 8 * na : 0)) * sizeof(int32_t);
    HIP_TRY(h, s.stage_down.reserve(bytes));
    HIP_TRY(h, hipMemcpyAsync(s.stage_down.p, s.d_out.p, bytes, hipMemcpyDeviceToHost, stream));
    s.h_status = (EggStatus *)(s.stage_down.p + (size_t)s.parity * kStatInts * sizeof(int32_t));
    s.out_copied = with_boxes;
    return EGG_OK;
}

// ---- packed pipeline: one launch per phase (eggsim_packed.hip) for the classes retile() marked
void fill_packed_args(egg_handle *h, int which, const PackedClass &pc, const Env &env, int S, int C, EggPackedArgs &A) {
    System &s = h->sys[which];
    const LaunchClass &lc = s.classes[(size_t)pc.cls];
    memset(&A, 0, sizeof A);
    const int in = s.cur, out = s.cur ^ 1;
    A.x_in = s.x[in].p;
    A.y_in = s.y[in].p;
    A.vx_in = s.vx[in].p;
    A.vy_in = s.vy[in].p;
    A.x_out = s.x[out].p;
    A.y_out = s.y[out].p;
    A.vx_out = s.vx[out].p;
    A.vy_out = s.vy[out].p;
    A.inv_mass = s.inv_mass.p;
    A.radius = s.radius.p;
    A.atom_offset = s.d_atom_offset.p;
    A.atom_count = s.d_atom_count.p;
    A.atom_batch = s.d_atom_batch.p;
    A.atom_tx = (const double *)s.d_meta.p;
    A.atom_ty = (const double *)(s.d_meta.p + s.meta_off_ty);
    A.atom_fd = (const double *)(s.d_meta.p + s.meta_off_fd);
    A.atom_claim = (const int32_t *)(s.d_meta.p + s.meta_off_claim);
    A.atom_aabb_out = d_aabb(s);
    A.atom_fail = s.d_atom_fail.p;
    A.atom_disp_out = d_disp(s);
    A.tile_atoms = (const int32_t *)(s.d_meta.p + s.meta_off_tatoms);
    A.n_tiles = pc.n_tiles;
    A.n_groups = pc.n_groups;
    A.tile_geo = s.pk_meta.p + pc.meta_tile_geo;
    A.grp_geo = s.pk_meta.p + pc.meta_grp_geo;
    A.tile_claims = s.pk_meta.p + s.pk_meta_claims;
    A.p_begin = pc.p_begin;
    A.p_end = pc.p_end;
    A.pk_pos = s.pk_pos.p;
    A.pk_prev = s.pk_prev.p;
    A.pk_wr = s.pk_wr.p;
    A.pk_src = s.pk_src.p;
    A.pk_atom = s.pk_atom.p;
    A.pk_aslot = s.pk_aslot.p;
    A.pk_ckey = s.pk_ckey.p;
    A.pk_stride = s.pk_n;
    A.lists = s.pk_lists.p + pc.entry_base;
    A.lvl = s.pk_lvl.p + pc.entry_base;
    A.rank = pc.levels_ooo ? s.pk_rank.p + pc.entry_base : nullptr;
    A.sorted = s.pk_sorted.p + pc.sort_base;
    A.chunks = s.pk_chunks.p + pc.chunk_base;
    A.grp_nchunks = s.pk_nchunks.p + pc.group_base;
    A.grp_nlev = s.pk_nchunks.p + s.pk_groups + pc.group_base;
    A.lev_start = s.pk_levstart.p + (size_t)pc.group_base * ((size_t)s.pk_lev_cap + 2);
    int32_t *tb = s.pk_tile.p + (size_t)pc.tile_base * (3 + 2 * EGG_PK_MAX_PASSES);
    A.tile_total = tb;
    A.tile_slack = tb + pc.n_tiles;
    A.tile_fast = tb + 2 * (size_t)pc.n_tiles;
    A.tile_visits = tb + 3 * (size_t)pc.n_tiles;
    A.tile_need = tb + (3 + (size_t)EGG_PK_MAX_PASSES) * (size_t)pc.n_tiles;
    A.lcap = pc.lcap;
    A.scap = pc.scap;
    A.lev_cap = s.pk_lev_cap;
    A.sort_cap = pc.sort_cap;
    A.chunk_cap = pc.chunk_cap;
    A.stage_cap = pc.stage_cap;
    A.nmax = lc.nmax;
    A.amax = lc.amax;
    A.ccap = lc.ccap;
    A.use_grid = lc.use_grid;
    A.sub_delta = env.sub_delta;
    A.damping = env.damping;
    A.follow_compliance = env.follow_c;
    A.collision_compliance = env.collision_c;
    A.overlap_factor = s.cfg.collision_overlap_factor;
    A.cell_size = env.cell;
    A.eps = s.cfg.eps;
    A.n_substeps = S;
    A.n_collision_steps = C;
    A.status = d_stat(s, s.parity);
    A.status_next = d_stat(s, s.parity ^ 1);
    A.tune = h->opt_tune;
    A.lev_lds_cap = pc.lev_lds_cap;
}

int launch_packed(egg_handle *h, int which, const Env &env, int S, int C) {
    System &s = h->sys[which];
    const hipStream_t st = s.stream;
    if (s.pk_plan_dirty) {
        HIP_TRY(h, hipMemcpyAsync(s.pk_meta.p, s.pk_meta_host.data(), s.pk_meta_host.size() * sizeof(int32_t),
                                  hipMemcpyHostToDevice, st));
    }
    std::vector<EggPackedArgs> args(s.pk.size());
    for (size_t k = 0; k < s.pk.size(); ++k) fill_packed_args(h, which, s.pk[k], env, S, C, args[k]);
    const bool stamp = h->opt_timing >= 2;
    s.pk_stamps_used = 0;
    auto stamp_begin = [&](int kind) -> System::PkStamp * {
        if (!stamp) return nullptr;
        if (s.pk_stamps_used == s.pk_stamps.size()) {
            System::PkStamp ps{nullptr, nullptr, 0, 0};
            if (hipEventCreate(&ps.a) != hipSuccess || hipEventCreate(&ps.b) != hipSuccess) return nullptr;
            s.pk_stamps.push_back(ps);
        }
        System::PkStamp *ps = &s.pk_stamps[s.pk_stamps_used++];
        ps->kind = kind;
        ps->launches = (int)s.pk.size();
        (void)hipEventRecord(ps->a, st);
        return ps;
    };
    auto launch_all = [&](int kind, auto kernel_of, auto grid_of, auto block_of, auto lds_of) {
        System::PkStamp *ps = stamp_begin(kind);
        for (size_t k = 0; k < s.pk.size(); ++k) {
            const PackedClass &pc = s.pk[k];
            hipLaunchKernelGGL(kernel_of(pc), dim3((unsigned)grid_of(pc)), dim3((unsigned)block_of(pc)), lds_of(pc), st, args[k]);
            h->stats.kernel_launches++;
        }
        if (ps) (void)hipEventRecord(ps->b, st);
    };
    auto flat_grid = [](const PackedClass &pc) { return (pc.p_end - pc.p_begin + 255) / 256; };
    auto c256 = [](const PackedClass &) { return 256; };
    auto c64 = [](const PackedClass &) { return 64; };
    auto no_lds = [](const PackedClass &) { return (size_t)0; };
    auto tiles_of = [](const PackedClass &pc) { return pc.n_tiles; };
    auto groups_of = [](const PackedClass &pc) { return pc.n_groups; };
    if (s.pk_plan_dirty) {
        launch_all(-1, [](const PackedClass &) { return egg_pk_plan_kernel; }, tiles_of, c64, no_lds);
        s.pk_plan_dirty = false;
    }
    launch_all(EGG_PK_KIND_BEGIN, [](const PackedClass &) { return egg_pk_begin_kernel; }, flat_grid, c256, no_lds);
    int pass_seq = 0;
    for (int sub = 0; sub < S; ++sub) {
        if (sub > 0) launch_all(EGG_PK_KIND_MID, [](const PackedClass &) { return egg_pk_mid_kernel; }, flat_grid, c256, no_lds);
        for (int c = 0; c < C; ++c, ++pass_seq) {
            const bool stale = c == 0 && sub > 0;  // hash lists and `collided` survive a sub-step boundary (L:1905-1912)
            for (EggPackedArgs &A : args) {
                A.pass_seq = pass_seq;
                A.substep = sub;
                A.stale = stale ? 1 : 0;
            }
            launch_all(stale ? EGG_PK_KIND_LISTS_STALE : EGG_PK_KIND_LISTS_FRESH,
                       [&](const PackedClass &) { return stale ? egg_pk_lists_stale_kernel : egg_pk_lists_fresh_kernel; }, tiles_of,
                       [](const PackedClass &pc) { return pc.threads_lists; }, [](const PackedClass &pc) { return pc.lds_lists; });
            launch_all(EGG_PK_KIND_LEVELS, [](const PackedClass &pc) { return pc.levels_ooo ? egg_pk_levels_ooo_kernel : egg_pk_levels_mr16_kernel; },
                       groups_of, [](const PackedClass &pc) { return pc.levels_threads; }, [](const PackedClass &pc) { return pc.lds_levels; });
            {   // (the out-of-order walk sorts inside its own launch)
                System::PkStamp *ps = nullptr;
                for (size_t k = 0; k < s.pk.size(); ++k) {
                    const PackedClass &pc = s.pk[k];
                    if (pc.levels_ooo) continue;
                    if (!ps) ps = stamp_begin(EGG_PK_KIND_SORT);
                    hipLaunchKernelGGL(pc.lds_sort ? egg_pk_sort_kernel : egg_pk_sort_direct_kernel, dim3((unsigned)pc.n_groups), dim3(256),
                                       pc.lds_sort ? pc.lds_sort : egg_align16((size_t)(s.pk_lev_cap + 2) * 4), st, args[k]);
                    h->stats.kernel_launches++;
                }
                if (ps) (void)hipEventRecord(ps->b, st);
            }
            // (fewer groups than SIMDs: every executor wave is alone, its time is levels x chain latency)
            const int simds = 4 * std::max(1, h->prop.multiProcessorCount);
            launch_all(EGG_PK_KIND_EXEC, [&](const PackedClass &pc) { return pc.n_groups <= simds ? egg_pk_exec_chain_kernel : egg_pk_exec_kernel; }, groups_of, c64,
                       [&](const PackedClass &pc) { return pc.lds_exec + (pc.n_groups <= simds ? 64 * 16 : 0); });
        }
    }
    launch_all(EGG_PK_KIND_END, [](const PackedClass &) { return egg_pk_end_kernel; }, tiles_of,
               [](const PackedClass &pc) { return std::min(256, pc.threads_lists); }, no_lds);
    const int n_passes = std::min(S * C, EGG_PK_MAX_PASSES);
    System::PkStamp *rs = stamp_begin(EGG_PK_KIND_REDUCE);
    for (size_t k = 0; k < s.pk.size(); ++k) {
        hipLaunchKernelGGL(egg_pk_reduce_kernel, dim3((unsigned)n_passes + 1), dim3(1024), 0, st, args[k], n_passes);
        h->stats.kernel_launches++;
    }
    if (rs) (void)hipEventRecord(rs->b, st);
    HIP_TRY(h, hipGetLastError());
    return EGG_OK;
}

// the throughput-tuned variant: judged by the white tiles (a yolk wave should not hold 167 registers on a full chip)
bool use_occ_variant(const egg_handle *h, const LaunchClass &lc) {
    return std::max<int64_t>(lc.n_tiles, h->stats.n_tiles[0]) >= 4 * (int64_t)h->prop.multiProcessorCount;
}

int launch_type(egg_handle *h, int which, const Env &env, int S, int C) {
    System &s = h->sys[which];
    if (s.n == 0 || s.classes.empty()) return EGG_OK;
    int rc = launch_prologue(h, which, s.stream);
    if (rc != EGG_OK) return rc;
    s.timing_from = which;
    if (h->opt_timing) HIP_TRY(h, hipEventRecord(s.ev0, s.stream));
    // per-atom "left its claim" flags of the step: the packed pipeline's kernels only ever set them (tiles of the
    // fused kernels reset their own atoms' flags themselves)
    if (!s.pk.empty()) HIP_TRY(h, hipMemsetAsync(s.d_atom_fail.p, 0, s.atoms.size() * sizeof(int32_t), s.stream));
    for (const LaunchClass &lc : s.classes) {
        if (lc.packed >= 0) continue;  // stepped by the packed pipeline below
        EggStepArgs A;
        fill_args(h, which, lc, env, S, C, A);
        const dim3 grid((unsigned)lc.n_tiles), block((unsigned)lc.threads);
        if (s.gens > 2) {  // more than two hash generations: the variants with the general list builder
            if (lc.global_state)
                hipLaunchKernelGGL(egg_step_kernel_gs_mg, grid, block, 64, s.stream, A);
            else if (lc.global_lists)
                hipLaunchKernelGGL(egg_step_kernel_gl_mg, grid, block, lc.lds, s.stream, A);
            else
                hipLaunchKernelGGL(egg_step_kernel_mg, grid, block, lc.lds, s.stream, A);
        } else if (lc.global_state)
            hipLaunchKernelGGL(egg_step_kernel_gs, grid, block, 64, s.stream, A);
        else if (lc.global_lists)
            hipLaunchKernelGGL(egg_step_kernel_gl, grid, block, lc.lds, s.stream, A);
        else if (lc.wide)
            hipLaunchKernelGGL(egg_step_kernel_wide, grid, block, lc.lds, s.stream, A);
        else if (use_occ_variant(h, lc))  // throughput regime: residency over spill-freedom
            hipLaunchKernelGGL(egg_step_kernel_occ, grid, block, lc.lds, s.stream, A);
        else
            hipLaunchKernelGGL(egg_step_kernel, grid, block, lc.lds, s.stream, A);
        HIP_TRY(h, hipGetLastError());
        h->stats.kernel_launches++;
    }
    if (!s.pk.empty()) {
        rc = launch_packed(h, which, env, S, C);
        if (rc != EGG_OK) return rc;
    }
    if (h->opt_timing) HIP_TRY(h, hipEventRecord(s.ev1, s.stream));
    return launch_epilogue(h, which, s.stream);
}

// All launch classes of both types as one grid (egg_step_kernel_multi*) when there are at most four, all of
// LDS tiles, all wide or all narrow; otherwise one launch per class on the type's own stream.
bool can_fuse(const egg_handle *h) {
    if (h->opt_no_fuse) return false;
    // measured (ms per step, one launch vs one per type): 10 batches 0.34 vs 0.51, 512: 0.47 vs 0.66, 1024: 0.59 vs
    // 0.75, 4096: 1.82 vs 1.92; from 8192 on the small tiles' share of the larger LDS allocation costs more than
    // their late finish (3.37 vs 3.33, 16384: 6.24 vs 6.13), hence the upper bound
    if (h->stats.n_tiles[0] > 16 * (int64_t)h->prop.multiProcessorCount) return false;
    size_t n_classes = 0;
    int wide = -1;
    for (int w = 0; w < 2; ++w) {
        const System &s = h->sys[w];
        if (s.n == 0 || s.classes.empty() || s.gens > 2 || !s.pk.empty()) return false;
        for (const LaunchClass &lc : s.classes) {
            if (lc.global_lists || lc.global_state) return false;
            if (wide >= 0 && wide != lc.wide) return false;
            wide = lc.wide;
            ++n_classes;
        }
    }
    return n_classes <= 4;
}

int launch_fused(egg_handle *h, const Env *env, int S, int C) {
    System &W = h->sys[0];
    const hipStream_t stream = W.stream;
    EggStepArgs4 P;
    memset(&P, 0, sizeof P);
    // slot 0: the first white class (the bulk of the work); slots 1..3: every other class, yolk first -- the
    // kernel puts their tiles at the front of the grid, so the small tiles start first
    int k = 1, threads = 0;
    int64_t tiles = 0;
    size_t lds = 0;
    for (int w = 1; w >= 0; --w) {
        int rc = launch_prologue(h, w, stream);
        if (rc != EGG_OK) return rc;
        h->sys[w].timing_from = 0;
        for (size_t c = 0; c < h->sys[w].classes.size(); ++c) {
            const LaunchClass &lc = h->sys[w].classes[c];
            fill_args(h, w, lc, env[w], S, C, (w == 0 && c == 0) ? P.a[0] : P.a[k++]);
            threads = std::max(threads, lc.threads);
            lds = std::max(lds, lc.lds);
            tiles += lc.n_tiles;
        }
    }
    const LaunchClass &lw = W.classes[0];
    const dim3 grid((unsigned)tiles), block((unsigned)threads);
    if (h->opt_timing) HIP_TRY(h, hipEventRecord(W.ev0, stream));
    if (lw.wide)
        hipLaunchKernelGGL(egg_step_kernel_multi_wide, grid, block, lds, stream, P);
    else if (use_occ_variant(h, lw))
        hipLaunchKernelGGL(egg_step_kernel_multi_occ, grid, block, lds, stream, P);
    else
        hipLaunchKernelGGL(egg_step_kernel_multi, grid, block, lds, stream, P);
    HIP_TRY(h, hipGetLastError());
    h->stats.kernel_launches++;
    if (h->opt_timing) HIP_TRY(h, hipEventRecord(W.ev1, stream));
    for (int w = 0; w < 2; ++w) {
        int rc = launch_epilogue(h, w, stream);
        if (rc != EGG_OK) return rc;
    }
    return EGG_OK;
}

// atoms, targets and tiles (with the claims of the upcoming step) of one type up to date on the host side
int prepare_type(egg_handle *h, int w) {
    System &s = h->sys[w];
    int rc = upload_atoms(h, w);
    if (rc != EGG_OK) return rc;
    if (s.claims_stale && !s.tiling_dirty) {
        // a moved target only matters when some blob is now farther from its target than its
        // slack + what the margin absorbs; cheap test on the host copy of the boxes
        if (!s.aabb_valid) {
            s.tiling_dirty = true;
        } else {
            const double cell = cell_size_of(s.cfg);
            for (size_t k = 0; k < s.atoms.size() && !s.tiling_dirty; ++k) {
                const Box &b = s.aabb[k];
                const Batch &B = h->batches[(size_t)s.atoms[k].batch];
                const double cx = 0.5 * ((double)b.lo_x + b.hi_x + 1.0) * cell;
                const double cy = 0.5 * ((double)b.lo_y + b.hi_y + 1.0) * cell;
                const double dist = std::hypot(B.target_x - cx, B.target_y - cy);
                const double reach = 0.5 * cell * std::max(b.hi_x - b.lo_x, b.hi_y - b.lo_y) + 2 * cell * s.margin;
                if (!(dist <= reach + 64.0)) s.tiling_dirty = true;
            }
        }
    }
    s.targets_moving = s.claims_stale;
    s.claims_stale = false;
    if (s.tiling_dirty) {
        rc = retile(h, w);
        if (rc != EGG_OK) return rc;
    }
    return EGG_OK;
}

int prepare_tiles(egg_handle *h) {
    for (int w = 0; w < 2; ++w) {
        int rc = prepare_type(h, w);
        if (rc != EGG_OK) return rc;
    }
    return EGG_OK;
}

// phase: kWhole = the complete step; kPrepare = tiles/claims only; kBegin = launch the first attempt and
// return (egg_step_begin); kEnd = finish a begun step: validate, re-run if needed, commit (egg_step_end)
enum { kWhole = 0, kPrepare = 1, kBegin = 2, kEnd = 3 };

int do_step(egg_handle *h, double delta, int S, int C, int phase = kWhole) {  // L:1722-1989
    const double sub_delta = std::max(delta / S, h->sys[0].cfg.eps);
    // One collision pass per sub-step: the reference never clears its hash lists inside the step, so the
    // kernel keeps one generation of cells per sub-step (PassCtx); the ring is sized for up to 8.
    const int gens = (C == 1 && S >= 3) ? S : 2;
    if (gens > 8)
        return fail(h, EGG_ERR_UNSUPPORTED,
                    "n_collision_steps == 1 with n_substeps > 8 (more than 8 un-cleared hash generations)");
    for (int w = 0; w < 2; ++w)
        if (h->sys[w].gens != gens) {
            h->sys[w].gens = gens;
            h->sys[w].tiling_dirty = true;  // LDS geometry of the launch classes depends on it
        }
    Env env[2];
    for (int w = 0; w < 2; ++w) {
        System &s = h->sys[w];
        env[w] = make_env(s.cfg, sub_delta, h->budget_particles[w] >= 0 ? h->budget_particles[w] : s.n);
        // mass / radius follow a config change at the next step (L:1731-1744, L:1420-1430)
        bool upd_mass = !s.has_env || s.cfg.min_mass != s.env_min_mass || s.cfg.max_mass != s.env_max_mass;
        bool upd_radius = !s.has_env || s.cfg.min_radius != s.env_min_radius || s.cfg.max_radius != s.env_max_radius;
        if (phase != kEnd && s.has_env && (upd_mass || upd_radius) && s.n > 0) {
            const int threads = 256;
            hipLaunchKernelGGL(egg_rederive_kernel, dim3((unsigned)((s.n + threads - 1) / threads)), dim3(threads), 0,
                               s.stream, s.mass_t.p, s.inv_mass.p, s.radius.p, (int)s.n, upd_mass ? 1 : 0,
                               s.cfg.min_mass, s.cfg.max_mass, upd_radius ? 1 : 0, s.cfg.min_radius, s.cfg.max_radius);
            HIP_TRY(h, hipGetLastError());
            // the step that reads these arrays may be launched on the OTHER type's stream (one fused launch
            // for both types): finish here -- config changes are rare, the wait costs nothing that matters
            HIP_TRY(h, hipStreamSynchronize(s.stream));
            h->stats.kernel_launches++;
        }
        s.has_env = true;
        s.env_min_mass = s.cfg.min_mass;
        s.env_max_mass = s.cfg.max_mass;
        s.env_min_radius = s.cfg.min_radius;
        s.env_max_radius = s.cfg.max_radius;
        if (env[w].cell != s.tiled_cell_size) {
            s.tiling_dirty = true;
            s.aabb_valid = false;
        }
        s.step_follow_compliance = env[w].follow_c;
        s.step_damping = env[w].damping;
        s.step_substeps = S;
        const bool pk_ok = S * C <= EGG_PK_MAX_PASSES;  // the packed pipeline keeps one visit counter per pass
        if (pk_ok != s.pk_allowed) {
            s.pk_allowed = pk_ok;
            s.tiling_dirty = true;
        }
    }
    if (phase == kPrepare) return prepare_tiles(h);

    for (int attempt = 0;; ++attempt) {
        if (attempt > 24) return fail(h, EGG_ERR_INTERNAL, "step did not validate after %d attempts", attempt);
        if (!(phase == kEnd && attempt == 0)) {  // kEnd: the first attempt is already in flight
            auto now = [] { return std::chrono::steady_clock::now(); };
            auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
            auto t0 = now();
            int rc = prepare_type(h, 0);
            if (rc != EGG_OK) return rc;
            bool white_packed = false;
            for (const LaunchClass &lc : h->sys[0].classes) white_packed |= lc.packed >= 0;
            if (white_packed) {
                // the packed pipeline never shares a launch with the other type: the white launches go out now and
                // run while the host clusters the yolk atoms (white first: it is the critical path)
                h->stats.host_ms[0] += ms_since(t0);
                t0 = now();
                h->stats.fused_launch = 0;
                rc = launch_type(h, 0, env[0], S, C);
                if (rc != EGG_OK) return rc;
                h->stats.host_ms[1] += ms_since(t0);
                t0 = now();
                rc = prepare_type(h, 1);
                if (rc != EGG_OK) return rc;
                h->stats.host_ms[0] += ms_since(t0);
                t0 = now();
                rc = launch_type(h, 1, env[1], S, C);
                if (rc != EGG_OK) return rc;
            } else {
                rc = prepare_type(h, 1);
                if (rc != EGG_OK) return rc;
                h->stats.host_ms[0] += ms_since(t0);
                t0 = now();
                h->stats.fused_launch = can_fuse(h) ? 1 : 0;
                if (h->stats.fused_launch) {
                    rc = launch_fused(h, env, S, C);
                    if (rc != EGG_OK) return rc;
                } else {
                    // white first: with at most a few tiles per CU it is the critical path
                    for (int w = 0; w < 2; ++w) {
                        rc = launch_type(h, w, env[w], S, C);
                        if (rc != EGG_OK) return rc;
                    }
                }
            }
            h->stats.host_ms[1] += ms_since(t0);
            if (phase == kBegin) return EGG_OK;
        }
        bool redo = false;
        double ms = 0;
        for (int w = 0; w < 2; ++w) {
            System &s = h->sys[w];
            if (s.n == 0 || s.classes.empty()) continue;
            const auto t_wait = std::chrono::steady_clock::now();
            HIP_TRY(h, wait_step(s.wait_stream ? s.wait_stream : s.stream));
            h->stats.host_ms[2] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wait).count();
            if (h->opt_timing) {
                float t = 0;
                const System &ts = h->sys[s.timing_from];
                HIP_TRY(h, hipEventElapsedTime(&t, ts.ev0, ts.ev1));
                ms = std::max(ms, (double)t);
                h->stats.kernel_ms[w] = (double)t;
            }
            const EggStatus &st = *s.h_status;
            if (st.fail_overflow) {
                // more visited pairs than the launch had list room for: grow and re-run
                s.list_min = std::max<size_t>(s.list_min, (size_t)(st.max_list * 5 / 4 + 64));
                s.list_factor *= 1.5;
                if (s.list_min > (size_t)kMaxGlobalListEntries)
                    return fail(h, EGG_ERR_UNSUPPORTED, "a tile visits %llu pairs in one pass; limit is %d",
                                (unsigned long long)st.max_list, kMaxGlobalListEntries);
                s.tiling_dirty = true;
                redo = true;
                continue;
            }
            if (st.fail_levlds) {
                // a tile's pair stream outgrew the LDS level array of the out-of-order walk: size it for the longest list seen
                s.pk_lev_lds_min = std::max<size_t>(2 * s.pk_lev_lds_min, (size_t)(st.max_list * 5 / 4 + 64));
                s.tiling_dirty = true;
                redo = true;
                continue;
            }
            if (st.fail_levels) {
                // a group's pair-dependency DAG is deeper than the packed pipeline's level tables: grow and re-run
                s.pk_lev_cap = std::max(2 * s.pk_lev_cap + 1, st.max_level + 64);
                if (s.pk_lev_cap > 65535)
                    return fail(h, EGG_ERR_UNSUPPORTED, "a pair-dependency chain of %d levels in one pass; limit is 65535", st.max_level);
                s.tiling_dirty = true;
                redo = true;
                continue;
            }
            if (st.fail_claim || st.fail_range) {
                // a particle left its claimed cells: tiles were not provably independent.  Widen the
                // claims of the atoms it happened to (wide claims merge neighbours into one tile, so
                // only theirs) and re-run the step.
                std::vector<int32_t> failed(s.atoms.size());
                if (!failed.empty()) {
                    HIP_TRY(h, hipMemcpyAsync(failed.data(), s.d_atom_fail.p, failed.size() * sizeof(int32_t),
                                              hipMemcpyDeviceToHost, s.stream));
                    HIP_TRY(h, hipStreamSynchronize(s.stream));
                }
                s.extra_margin.resize(h->batches.size(), 0);
                bool any = false;
                for (size_t k = 0; k < failed.size(); ++k)
                    if (failed[k]) {
                        int &e = s.extra_margin[(size_t)s.atoms[k].batch];
                        e = std::min(4096, std::max(2, 2 * e));
                        any = true;
                    }
                if (!any) s.margin = std::min(s.margin + std::max(2, s.margin / 2), 4096);  // range failure
                s.tiling_dirty = true;
                s.aabb_valid = false;
                redo = true;
                continue;
            }
            if (st.fail_stall)
                return fail(h, EGG_ERR_INTERNAL, "pair scheduler stalled (type %d, %s)", w,
                            st.fail_stall == 2 ? "time limit reached"
                            : st.fail_stall == 3 ? "workgroup narrower than the wide kernel needs"
                                                 : "a particle's pair sequence did not finish");
            const bool single = s.single_tile || h->opt_force_single;
            if (!single) {
                // budget check (L:1657-1658): the return can only fire if some pass visits more than
                // ceil(budget) pairs; then the visiting order across tiles matters -> exact mode
                const double m = std::max(1.0, std::ceil(env[w].budget));
                const int np = std::min(S * C, EGG_MAX_PASSES);
                for (int p = 0; p < np; ++p)
                    if ((double)st.visits[p] > m) {
                        s.single_tile = 1;
                        s.uncut_streak = 0;
                        s.tiling_dirty = true;
                        redo = true;
                        break;
                    }
            }
        }
        if (redo) {
            h->stats.redo_steps++;
            continue;
        }
        // commit
        for (int w = 0; w < 2; ++w) {
            System &s = h->sys[w];
            if (s.n == 0 || s.classes.empty()) continue;
            const EggStatus &st = *s.h_status;
            s.cur ^= 1;
            const int np = std::min(S * C, EGG_MAX_PASSES);
            int64_t most = 0;
            for (int p = 0; p < np; ++p) {
                h->stats.pair_solves += (int64_t)st.visits[p];
                most = std::max(most, (int64_t)st.visits[p]);
            }
            h->stats.max_pass_visits[w] = most;
            h->stats.max_levels[w] = s.pk.empty() ? 0 : st.max_level;
#ifdef EGG_PROFILE
            if (getenv("EGGSIM_DEBUG") && !s.pk.empty())
                fprintf(stderr, "eggsim: type %d step %lld, last pass, egg_pk_levels_ooo cycles: group 0 init %llu rank %llu walk %llu finish %llu sort %llu total %llu | slowest group: walk %llu total %llu | turns %llu levels %d\n",
                        w, (long long)h->stats.steps, st.visits[55], st.visits[56], st.visits[57], st.visits[58], st.visits[59], st.visits[60], st.visits[61], st.visits[62], st.rounds, st.max_level);
            if (getenv("EGGSIM_DEBUG") && !s.pk.empty()) {
                fprintf(stderr, "   group 0 wave 0: %llu turns; cycles waiting for the batch's entries %llu, in the batch set-up %llu, in the turns %llu\n", st.visits[38], st.visits[36], st.visits[39], st.visits[37]);
                fprintf(stderr, "   groups by total cycles (25k buckets):");
                for (int k = 0; k < 15; ++k) fprintf(stderr, " %llu", st.visits[40 + k]);
                fprintf(stderr, "\n");
            }
#endif
            h->stats.budget[w] = env[w].budget;
            h->stats.follow_solves += s.n * S;
            s.aabb_on_device = true;  // d_atom_aabb now holds end-of-step cells
            if (s.out_copied) {
                const size_t na = s.atoms.size();
                s.aabb.resize(na);
                s.disp.resize(4 * na);
                const unsigned char *boxes = s.stage_down.p + 2 * kStatInts * sizeof(int32_t);
                memcpy(s.aabb.data(), boxes, na * 16);
                memcpy(s.disp.data(), boxes + na * 16, na * 16);
                s.aabb_valid = s.disp_valid = true;
            } else {
                s.aabb_valid = false;  // fetched on demand
                s.disp_valid = false;
            }
            if (st.min_slack < s.margin || s.swept) {
                // some particle has used part of its margin, or blobs are flying: re-tile around the
                // new positions
                s.tiling_dirty = true;
            }
            if (s.margin > h->opt_margin) {  // widened after a failed check: relax again
                s.margin -= 1;
                s.tiling_dirty = true;
            }
            for (int &e : s.extra_margin)
                if (e > 0) {
                    e -= 1;
                    s.tiling_dirty = true;
                }
            if (s.single_tile && !h->opt_force_single) {
                // leave exact mode once the budget has not cut for a while and cannot bind by size
                s.uncut_streak = st.was_cut ? 0 : s.uncut_streak + 1;
                if (s.uncut_streak >= 8) {
                    s.single_tile = 0;
                    s.tiling_dirty = true;
                }
            }
            h->stats.single_tile[w] = s.single_tile;
            // the next step re-tiles, or targets are being moved step by step: the next tiling will want the boxes
            s.eager_boxes = s.tiling_dirty || s.claims_stale || s.targets_moving;
        }
        h->stats.last_step_kernel_ms = ms;
        for (int w = 0; w < 2; ++w) {
            System &s = h->sys[w];
            h->stats.packed[w] = (int64_t)s.pk.size();
            h->stats.pk_variants[w] = 0;
            for (const PackedClass &pc : s.pk)
                h->stats.pk_variants[w] |= (pc.levels_ooo ? EGG_PK_VARIANT_LEVELS_OOO : EGG_PK_VARIANT_LEVELS_INORDER) |
                                           (pc.n_groups <= 4 * std::max(1, h->prop.multiProcessorCount) ? EGG_PK_VARIANT_EXEC_CHAIN : EGG_PK_VARIANT_EXEC) |
                                           (pc.levels_ooo ? 0 : pc.lds_sort ? EGG_PK_VARIANT_SORT_LDS : EGG_PK_VARIANT_SORT_DIRECT);  // (the out-of-order walk sorts in its own launch)
            for (size_t k = 0; h->opt_timing >= 2 && k < s.pk_stamps_used; ++k) {
                const System::PkStamp &ps = s.pk_stamps[k];
                float t = 0;
                if (ps.kind >= 0 && hipEventElapsedTime(&t, ps.a, ps.b) == hipSuccess) {
                    h->stats.pk_kernel_ms[w][ps.kind] += (double)t;
                    h->stats.pk_kernel_launches[w][ps.kind] += ps.launches;
                }
            }
            s.pk_stamps_used = 0;
        }
        if (h->opt_timing) {
            for (int w = 0; w < 2; ++w) h->stats.kernel_ms_sum[w] += h->stats.kernel_ms[w];
            h->stats.timed_steps++;
        }
        h->stats.steps++;
        return EGG_OK;
    }
}

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

}  // namespace

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

// ------------------------------------------------------------------ headless renderer (eggsim_render.hip)

int egg_default_render_config(int which, egg_render_config *cfg) {  // simulation_handler_default_config.lua:22-36, 54-68
    if (!cfg || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    const float white[2][4] = {{0.961f, 0.961f, 0.953f, 1.0f}, {0.973f, 0.796f, 0.529f, 1.0f}};
    const float yolk[2][4] = {{0.969f, 0.682f, 0.141f, 1.0f}, {0.984f, 0.522f, 0.271f, 1.0f}};
    memcpy(cfg->color, which == EGG_WHITE ? white[0] : yolk[0], sizeof cfg->color);
    memcpy(cfg->outline_color, which == EGG_WHITE ? white[1] : yolk[1], sizeof cfg->outline_color);
    cfg->outline_thickness = 1;
    cfg->highlight_strength = which == EGG_WHITE ? 0 : 1;
    cfg->shadow_strength = which == EGG_WHITE ? 1 : 0;
    cfg->texture_scale = 12;
    cfg->motion_blur = 0.0003;
    return EGG_OK;
}

int egg_set_render_config(egg_handle *h, int which, const egg_render_config *cfg) {
    if (!h || !cfg || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    if (!(cfg->outline_thickness >= 0) || !(cfg->texture_scale > 0) || !std::isfinite(cfg->motion_blur) ||
        !std::isfinite(cfg->highlight_strength) || !std::isfinite(cfg->shadow_strength) || !(cfg->outline_thickness <= 256))
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_set_render_config: value out of range");
    h->render.cfg[which] = *cfg;
    // config.color is a new table now -- whatever its values: set_*_config deep-copies (L:1307-1311) --, so batches that
    // shared the old one keep it for themselves.  (Call this where the reference calls set_*_config, not once per frame.)
    for (Batch &b : h->batches) b.own_color[which] = true;
    return EGG_OK;
}

int egg_get_render_config(const egg_handle *h, int which, egg_render_config *cfg) {
    if (!h || !cfg || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    *cfg = h->render.cfg[which];
    return EGG_OK;
}

int egg_set_render_flags(egg_handle *h, int32_t use_particle_color, int32_t use_lighting) {
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    h->render.use_particle_color = use_particle_color != 0;
    h->render.use_lighting = use_lighting != 0;
    return EGG_OK;
}

static float clamp01(double v) { return (float)clampd(v, 0, 1); }

int egg_set_add_color(egg_handle *h, int64_t id, int which, double r, double g, double b, double a) {
    if (!h || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    Batch *B = find_batch(h, id);
    if (!B) return fail(h, EGG_ERR_UNKNOWN_ID, "egg_set_add_color: no batch with id `%lld`", (long long)id);
    if (std::isnan(r) || std::isnan(g) || std::isnan(b) || std::isnan(a))  // L:87-103
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "In SimulationHandler.add: %s color component is not a number", which == EGG_WHITE ? "white" : "yolk");
    B->own_color[which] = true;
    if (h->render.use_particle_color) {  // add does not clamp (L:978-984)
        const float c[4] = {(float)r, (float)g, (float)b, (float)a};
        memcpy(B->pcolor[which], c, sizeof c);
    }
    return EGG_OK;
}

int egg_set_color(egg_handle *h, int64_t id, int which, double r, double g, double b, double a) {
    if (!h || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    if (std::isnan(r) || std::isnan(g) || std::isnan(b) || std::isnan(a))
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_set_color: a colour component is not a number");
    Batch *B = find_batch(h, id);
    if (!B)
        return fail(h, EGG_WARN_UNKNOWN_ID, "In SimulationHandler.%s: no batch with id `%lld`",
                    which == EGG_WHITE ? "set_white_color" : "set_egg_yolk_color", (long long)id);
    const float c[4] = {clamp01(r), clamp01(g), clamp01(b), clamp01(a)};  // _assert_color (L:300-319)
    memcpy(B->pcolor[which], c, sizeof c);
    if (!B->own_color[which]) memcpy(h->render.cfg[which].color, c, sizeof c);  // the shared table (L:49-50, L:349-350)
    return EGG_OK;
}

int egg_default_render_params(egg_render_params *p) {
    if (!p) return EGG_ERR_INVALID_ARGUMENT;
    memset(p, 0, sizeof *p);
    p->screen_w = 800;
    p->screen_h = 600;
    p->interpolation_alpha = std::numeric_limits<double>::quiet_NaN();
    p->threshold = 0.3;    // L:444
    p->smoothness = 0.01;  // L:445
    p->use_instancing = 1;
    return EGG_OK;
}

namespace {

// the density texture: simulation_handler_particle_texture.glsl drawn by _initialize_particle_texture (L:620-682)
int render_texture(egg_handle *h) {
    egg_handle::Render &R = h->render;
    const double radius = std::max(h->sys[0].cfg.max_radius, h->sys[1].cfg.max_radius) * 4;  // L:626-629, L:455
    if (radius == R.texture_radius && R.tsize > 0) return EGG_OK;
    const double padding = 3;  // L:454
    const double size_d = (radius + padding) * 2;
    if (!(radius > 0) || !(size_d <= 1024))
        return fail(h, EGG_ERR_UNSUPPORTED, "particle texture of %g px", size_d);
    const int size = (int)size_d;
    R.texture_host.assign((size_t)size * size, 0.0f);
    for (int j = 0; j < size; ++j)
        for (int i = 0; i < size; ++i) {
            // the quad of 2 radius x 2 radius px sits in the middle of the canvas; uv at the pixel centre
            const double u = (i + 0.5 - (size - 2 * radius) / 2) / (2 * radius), v = (j + 0.5 - (size - 2 * radius) / 2) / (2 * radius);
            if (!(u >= 0 && u < 1 && v >= 0 && v < 1)) continue;
            const double q = 2.0 * std::sqrt((u - 0.5) * (u - 0.5) + (v - 0.5) * (v - 0.5));  // 1 - dist
            R.texture_host[(size_t)j * size + i] = (float)std::exp((-4.0 * kPi / 3.0) * q * q);
        }
    hipStream_t st = h->sys[0].stream;
    HIP_TRY(h, R.texture.reserve((size_t)size * size, false, st));
    HIP_TRY(h, hipMemcpyAsync(R.texture.p, R.texture_host.data(), (size_t)size * size * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    R.tsize = size;
    R.texture_radius = radius;
    return EGG_OK;
}

// pass 1 of one type into R.canvas[which] (cw x ch), centred on the interpolated centroid
int render_splat(egg_handle *h, int which, const egg_environment &env, double t, int cw, int ch, int use_instancing) {
    egg_handle::Render &R = h->render;
    System &s = h->sys[which];
    hipStream_t st = h->sys[0].stream;
    int rc = upload_atoms(h, which);
    if (rc != EGG_OK) return rc;
    const size_t na = s.atoms.size();
    std::vector<float> colors(4 * na);
    for (size_t k = 0; k < na; ++k) memcpy(&colors[4 * k], h->batches[(size_t)s.atoms[k].batch].pcolor[which], 16);
    HIP_TRY(h, R.atom_color.reserve(na, false, st));
    HIP_TRY(h, hipMemcpyAsync(R.atom_color.p, colors.data(), na * 16, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipStreamSynchronize(st));  // (`colors` is pageable and goes out of scope)

    EggRenderArgs A;
    memset(&A, 0, sizeof A);
    A.x = s.x[s.cur].p;
    A.y = s.y[s.cur].p;
    A.last_x = s.x[s.cur ^ 1].p;  // positions at the start of the most recent _step (L:1795-1815)
    A.last_y = s.y[s.cur ^ 1].p;
    A.vx = s.vx[s.cur].p;
    A.vy = s.vy[s.cur].p;
    A.radius = s.radius.p;
    A.atom_offset = s.d_atom_offset.p;
    A.atom_color = R.atom_color.p;
    A.n = (int32_t)s.n;
    A.n_atoms = (int32_t)na;
    A.t = (float)t;
    // frame interpolation of the centroid in doubles like the Lua (L:2057-2058); the transform reaches the GPU as floats
    const double pcx = env.last_centroid_x * (1 - t) + env.centroid_x * t, pcy = env.last_centroid_y * (1 - t) + env.centroid_y * t;
    A.tx = (float)(cw / 2.0 - pcx);
    A.ty = (float)(ch / 2.0 - pcy);
    A.texture_scale = (float)R.cfg[which].texture_scale;
    A.motion_blur = (float)R.cfg[which].motion_blur;
    A.premultiply = use_instancing ? 0 : 1;
    A.cw = cw;
    A.ch = ch;
    A.tiles_x = (cw + EGG_RENDER_TILE - 1) / EGG_RENDER_TILE;
    A.tiles_y = (ch + EGG_RENDER_TILE - 1) / EGG_RENDER_TILE;
    const size_t nt = (size_t)A.tiles_x * A.tiles_y;
    HIP_TRY(h, R.tiles.reserve(3 * nt + 8, false, st));
    HIP_TRY(h, R.totals.reserve(4, false, st));
    A.tile_count = R.tiles.p;
    A.tile_start = R.tiles.p + nt;
    A.tile_cursor = R.tiles.p + 2 * nt + 1;
    A.totals = R.totals.p;
    A.texture = R.texture.p;
    A.tsize = R.tsize;
    HIP_TRY(h, R.canvas[which].reserve((size_t)cw * ch, false, st));
    A.canvas = R.canvas[which].p;
    HIP_TRY(h, hipMemsetAsync(A.tile_count, 0, nt * 4, st));
    const dim3 pgrid((unsigned)((s.n + 255) / 256)), pblock(256);
    hipLaunchKernelGGL(egg_render_count_kernel, pgrid, pblock, 0, st, A);
    hipLaunchKernelGGL(egg_render_scan_kernel, dim3(1), dim3(1024), 0, st, A);
    HIP_TRY(h, hipGetLastError());
    uint32_t totals[2] = {0, 0};
    HIP_TRY(h, hipMemcpyAsync(totals, A.totals, sizeof totals, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    size_t padded = 1;
    while (padded < totals[1]) padded <<= 1;
    const size_t lds = egg_render_splat_lds_bytes(R.tsize, padded);
    if (lds > h->lds_limit)
        return fail(h, EGG_ERR_UNSUPPORTED, "%u particles overlap one %d x %d px canvas tile; the renderer sorts at most %zu in LDS",
                    totals[1], EGG_RENDER_TILE, EGG_RENDER_TILE, (h->lds_limit - egg_render_splat_lds_bytes(R.tsize, 0)) / 8);
    HIP_TRY(h, R.entries.reserve((size_t)totals[0] + 8, false, st));
    A.entries = R.entries.p;
    hipLaunchKernelGGL(egg_render_fill_kernel, pgrid, pblock, 0, st, A);
    hipLaunchKernelGGL(egg_render_splat_kernel, dim3((unsigned)nt), dim3(256), lds, st, A);
    HIP_TRY(h, hipGetLastError());
    h->stats.kernel_launches += 4;
    return EGG_OK;
}

}  // namespace

int egg_render(egg_handle *h, const egg_render_params *p, float *rgba) {
    if (!h || !p) return EGG_ERR_INVALID_ARGUMENT;
    REJECT_IN_FLIGHT(h, "egg_render");
    if (p->screen_w <= 0 || p->screen_h <= 0 || (int64_t)p->screen_w * p->screen_h > ((int64_t)1 << 28))
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render: screen of %d x %d px", p->screen_w, p->screen_h);
    if (!(p->threshold >= 0 && p->threshold <= 1) || !(p->smoothness >= 0) || !std::isfinite(p->origin_x) || !std::isfinite(p->origin_y))
        return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render: threshold / smoothness / origin out of range");
    for (int w = 0; w < 2; ++w)
        if (p->canvas_w[w] < 0 || p->canvas_h[w] < 0 || p->canvas_w[w] > 16384 || p->canvas_h[w] > 16384)
            return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render: canvas size out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    egg_handle::Render &R = h->render;
    hipStream_t st = h->sys[0].stream;
    HIP_TRY(h, hipStreamSynchronize(h->sys[1].stream));
    const size_t npx = (size_t)p->screen_w * p->screen_h;
    HIP_TRY(h, R.screen.reserve(npx, false, st));
    hipLaunchKernelGGL(egg_render_clear_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, R.screen.p, npx,
                       make_float4(p->clear[0], p->clear[1], p->clear[2], p->clear[3]));
    HIP_TRY(h, hipGetLastError());
    h->stats.kernel_launches++;
    // the canvases exist from the first _step on and only while both types have particles (L:1936-1938, L:1997-1999, L:2118)
    const bool drawable = h->stats.steps > 0 && h->sys[0].n > 0 && h->sys[1].n > 0;
    R.canvas_valid = false;
    if (drawable) {
        const double t = std::isnan(p->interpolation_alpha) ? h->interpolation_alpha : clampd(p->interpolation_alpha, 0, 1);
        int rc = render_texture(h);
        if (rc != EGG_OK) return rc;
        egg_environment env[2];
        EggCompositeArgs C;
        memset(&C, 0, sizeof C);
        for (int w = 0; w < 2; ++w) {
            rc = egg_get_environment(h, w, &env[w]);
            if (rc != EGG_OK) return rc;
            const egg_render_config &cfg = R.cfg[w];
            int cw = p->canvas_w[w], ch = p->canvas_h[w];
            if (cw == 0 || ch == 0) {  // resize_canvas_maybe (L:1935-1975)
                const double padding = env[w].max_radius * cfg.texture_scale * (1 + std::max(1.0, env[w].max_velocity) * cfg.motion_blur);
                const double nw = std::min(std::ceil((env[w].max_x - env[w].min_x) + 2 * padding), 2560.0);
                const double nh = std::min(std::ceil((env[w].max_y - env[w].min_y) + 2 * padding), 2560.0);
                if (!(nw >= 1) || !(nh >= 1)) return fail(h, EGG_ERR_UNSUPPORTED, "egg_render: particle bounds are not finite");
                R.canvas_w[w] = std::max(R.canvas_w[w], (int)nw);
                R.canvas_h[w] = std::max(R.canvas_h[w], (int)nh);
                if (cw == 0) cw = R.canvas_w[w];
                if (ch == 0) ch = R.canvas_h[w];
            }
            rc = render_splat(h, w, env[w], t, cw, ch, p->use_instancing);
            if (rc != EGG_OK) return rc;
            // _draw_canvases places the canvas around the CURRENT centroid (L:2131-2132), pass 1 drew around the interpolated one
            R.canvas_x0[w] = env[w].centroid_x - 0.5 * cw;
            R.canvas_y0[w] = env[w].centroid_y - 0.5 * ch;
            EggCompositeLayer &L = C.layer[w];
            L.canvas = R.canvas[w].p;
            L.w = cw;
            L.h = ch;
            L.x0 = (float)(R.canvas_x0[w] - p->origin_x);
            L.y0 = (float)(R.canvas_y0[w] - p->origin_y);
            L.color = make_float4(cfg.color[0], cfg.color[1], cfg.color[2], cfg.color[3]);
            L.outline_color = make_float4(cfg.outline_color[0], cfg.outline_color[1], cfg.outline_color[2], cfg.outline_color[3]);
            L.outline_thickness = (float)cfg.outline_thickness;
            L.highlight_strength = (float)cfg.highlight_strength;
            L.shadow_strength = (float)cfg.shadow_strength;
        }
        C.screen = R.screen.p;
        C.screen_w = p->screen_w;
        C.screen_h = p->screen_h;
        C.n_layers = 2;
        C.threshold = (float)p->threshold;
        C.smoothness = (float)p->smoothness;
        C.use_particle_color = R.use_particle_color;
        C.use_lighting = R.use_lighting;
        hipLaunchKernelGGL(egg_render_composite_kernel, dim3((unsigned)((p->screen_w + 15) / 16), (unsigned)((p->screen_h + 15) / 16)),
                           dim3(256), 0, st, C);
        HIP_TRY(h, hipGetLastError());
        h->stats.kernel_launches++;
        R.canvas_valid = true;
        R.last_w[0] = C.layer[0].w;
        R.last_h[0] = C.layer[0].h;
        R.last_w[1] = C.layer[1].w;
        R.last_h[1] = C.layer[1].h;
    }
    if (rgba) HIP_TRY(h, hipMemcpyAsync(rgba, R.screen.p, npx * 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return EGG_OK;
}

int egg_render_canvas(egg_handle *h, int which, float *rgba, int64_t cap_pixels, int32_t *w, int32_t *hgt, double *x0, double *y0) {
    if (!h || (which != EGG_WHITE && which != EGG_YOLK)) return EGG_ERR_INVALID_ARGUMENT;
    egg_handle::Render &R = h->render;
    if (!R.canvas_valid) return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render_canvas: no canvas (egg_render has not drawn anything)");
    HIP_TRY(h, hipSetDevice(h->device));
    const int cw = R.last_w[which], ch = R.last_h[which];
    if (w) *w = cw;
    if (hgt) *hgt = ch;
    if (x0) *x0 = R.canvas_x0[which];
    if (y0) *y0 = R.canvas_y0[which];
    if (rgba) {
        if (cap_pixels < (int64_t)cw * ch)
            return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render_canvas: buffer holds %lld of %lld pixels", (long long)cap_pixels, (long long)cw * ch);
        HIP_TRY(h, hipMemcpy(rgba, R.canvas[which].p, (size_t)cw * ch * 16, hipMemcpyDeviceToHost));
    }
    return EGG_OK;
}

int egg_render_particle_texture(egg_handle *h, float *alpha, int64_t cap, int32_t *size) {
    if (!h) return EGG_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = render_texture(h);
    if (rc != EGG_OK) return rc;
    const int n = h->render.tsize;
    if (size) *size = n;
    if (alpha) {
        if (cap < (int64_t)n * n) return fail(h, EGG_ERR_INVALID_ARGUMENT, "egg_render_particle_texture: buffer too small");
        HIP_TRY(h, hipMemcpy(alpha, h->render.texture.p, (size_t)n * n * 4, hipMemcpyDeviceToHost));  // what the kernels sample
    }
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
