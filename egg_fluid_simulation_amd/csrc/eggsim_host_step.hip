// eggsim_host_step.hip -- SimulationHandler:_step (simulation_handler.lua:1722-1989) on the host side: environment
// scalars, the launches of the fused step kernels and of the packed pipeline, validation / re-run / commit.  See eggsim_host.h.
#include "eggsim_host.h"

namespace egghost {

// ------------------------------------------------------------------- step

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
    A.lev_lds_cap = pc.lev_lds_now;
    A.simd_claims = h->simd_claims.p;
}

int launch_packed(egg_handle *h, int which, const Env &env, int S, int C) {
    System &s = h->sys[which];
    const hipStream_t st = s.stream;
    if (s.pk_plan_dirty) {
        HIP_TRY(h, hipMemcpyAsync(s.pk_meta.p, s.pk_meta_host.data(), s.pk_meta_host.size() * sizeof(int32_t),
                                  hipMemcpyHostToDevice, st));
    }
    // The LDS level array of the out-of-order walk, per step: what the last step's longest pair stream asks for plus a
    // quarter, never more than the re-tiling allowed.  (Sized at re-tiling only it stayed at the first steps' 77-80 KB per
    // group for a whole run of config 3 -- two groups then fill a CU's 160 KB to the last 3 KB, and any workgroup of the
    // yolk's kernels that holds LDS on the unit keeps the second group waiting.  A stream that outgrows the array fails
    // the launch's check and the step is re-run, as before.)
    for (PackedClass &pc : s.pk) {
        if (!pc.levels_ooo) continue;
        int now = pc.lev_lds_cap;
        if (s.pk_seen_list) now = (int)std::min<size_t>((size_t)now, std::max<size_t>((size_t)(s.pk_seen_list + s.pk_seen_list / 4 + 256), s.pk_lev_lds_min));
        now = std::min(pc.lev_lds_cap, (now + 7) & ~7);
        pc.lev_lds_now = now;
        pc.lds_levels_now = egg_pk_levels_ooo_lds_bytes(s.pk_lev_cap, pc.max_group_particles, pc.max_tiles_in_group, now);
        pc.lds_pass_now = std::max(pc.lds_levels_now, pc.lds_exec + 64 * 16 + (size_t)EGG_PK_RING_BYTES);
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
    // one launch per class that `take` selects (0: every class, 1: the classes with one launch per phase, 2: the classes
    // whose levels, sort and executor are ONE launch, egg_pk_levexec_kernel)
    auto launch_some = [&](int take, int kind, auto kernel_of, auto grid_of, auto block_of, auto lds_of) {
        System::PkStamp *ps = nullptr;
        for (size_t k = 0; k < s.pk.size(); ++k) {
            const PackedClass &pc = s.pk[k];
            if ((take == 1 && pc.fused_pass) || (take == 2 && !pc.fused_pass)) continue;
            if (!ps) ps = stamp_begin(kind);
            hipLaunchKernelGGL(kernel_of(pc), dim3((unsigned)grid_of(pc)), dim3((unsigned)block_of(pc)), lds_of(pc), st, args[k]);
            h->stats.kernel_launches++;
        }
        if (ps) (void)hipEventRecord(ps->b, st);
    };
    auto launch_all = [&](int kind, auto kernel_of, auto grid_of, auto block_of, auto lds_of) { launch_some(0, kind, kernel_of, grid_of, block_of, lds_of); };
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
            launch_some(0, stale ? EGG_PK_KIND_LISTS_STALE : EGG_PK_KIND_LISTS_FRESH,
                        [&](const PackedClass &) { return stale ? egg_pk_lists_stale_kernel : egg_pk_lists_fresh_kernel; }, tiles_of,
                        [stale](const PackedClass &pc) { return stale ? pc.threads_lists_stale : pc.threads_lists; },
                        [stale](const PackedClass &pc) { return stale ? pc.lds_lists_stale : pc.lds_lists; });
            launch_some(1, EGG_PK_KIND_LEVELS, [](const PackedClass &pc) { return pc.levels_ooo ? egg_pk_levels_ooo_kernel : egg_pk_levels_mr16_kernel; },
                        groups_of, [](const PackedClass &pc) { return pc.levels_threads; }, [](const PackedClass &pc) { return pc.levels_ooo ? pc.lds_levels_now : pc.lds_levels; });
            {   // (the out-of-order walk sorts inside its own launch)
                System::PkStamp *ps = nullptr;
                for (size_t k = 0; k < s.pk.size(); ++k) {
                    const PackedClass &pc = s.pk[k];
                    if (pc.levels_ooo || pc.fused_pass) continue;
                    if (!ps) ps = stamp_begin(EGG_PK_KIND_SORT);
                    hipLaunchKernelGGL(pc.lds_sort ? egg_pk_sort_kernel : egg_pk_sort_direct_kernel, dim3((unsigned)pc.n_groups), dim3(256),
                                       pc.lds_sort ? pc.lds_sort : egg_align16((size_t)(s.pk_lev_cap + 2) * 4), st, args[k]);
                    h->stats.kernel_launches++;
                }
                if (ps) (void)hipEventRecord(ps->b, st);
            }
            // (fewer groups than SIMDs: every executor wave is alone, its time is levels x chain latency)
            const int simds = 4 * std::max(1, h->prop.multiProcessorCount);
            launch_some(1, EGG_PK_KIND_EXEC, [&](const PackedClass &pc) { return pc.n_groups <= simds ? egg_pk_exec_chain_kernel : egg_pk_exec_kernel; }, groups_of, c64,
                        [&](const PackedClass &pc) { return pc.lds_exec + (pc.n_groups <= simds ? 64 * 16 : 0); });
            // dense islands on a chip that is not full: levels, sort and executor of a group in one launch
            launch_some(2, EGG_PK_KIND_PASS, [](const PackedClass &) { return egg_pk_levexec_kernel; }, groups_of, [](const PackedClass &pc) { return pc.levels_threads; },
                        [](const PackedClass &pc) { return pc.lds_pass_now; });
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

int do_step(egg_handle *h, double delta, int S, int C, int phase) {  // L:1722-1989
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
            if (st.fail_stall)
                return fail(h, EGG_ERR_INTERNAL, "pair scheduler stalled (type %d, %s)", w,
                            st.fail_stall == 2 ? "time limit reached"
                            : st.fail_stall == 3 ? "workgroup narrower than the wide kernel needs"
                            : st.fail_stall == 4 ? "a wait between the waves of the packed pipeline did not end"
                                                 : "a particle's pair sequence did not finish");
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
                s.pk_lev_lds_min = std::max<size_t>(s.pk_lev_lds_min, (size_t)(st.max_list * 5 / 4 + 256));
                s.pk_seen_list = std::max(s.pk_seen_list, st.max_list);
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
            if (!s.pk.empty()) {
                s.pk_seen_list = st.max_list;
                s.pk_seen_levels = st.max_level;
                s.pk_lev_lds_min = 0;  // (a committed step: the stream lengths it saw size the next one)
            }
#ifdef EGG_PROFILE
            if (getenv("EGGSIM_DEBUG") && !s.pk.empty())
                fprintf(stderr, "eggsim: type %d step %lld, last pass, egg_pk_levels_ooo cycles: group 0 init %llu rank %llu walk %llu finish %llu sort %llu total %llu | slowest group: walk %llu total %llu | turns %llu levels %d\n",
                        w, (long long)h->stats.steps, st.visits[55], st.visits[56], st.visits[57], st.visits[58], st.visits[59], st.visits[60], st.visits[61], st.visits[62], st.rounds, st.max_level);
            if (getenv("EGGSIM_DEBUG") && !s.pk.empty()) {
                fprintf(stderr, "   group 0 wave 0: %llu turns; cycles waiting for the batch's entries %llu, in the batch set-up %llu, in the turns %llu\n", st.visits[38], st.visits[36], st.visits[39], st.visits[37]);
                fprintf(stderr, "   executor of group 0: %llu cycles for %llu chunks = %.1f per chunk; slowest group %.1f per chunk\n", st.visits[30], st.visits[31],
                        (double)st.visits[30] / (double)std::max<unsigned long long>(st.visits[31], 1), (double)st.visits[32] / 16.0);
                fprintf(stderr, "   executor of group 0: %llu ticks of the 100 MHz clock, i.e. the cycle counter ran at %.0f MHz\n", st.visits[28], st.visits[28] ? 100.0 * (double)st.visits[30] / (double)st.visits[28] : 0.0);
                fprintf(stderr, "   executor of group 0 looked at the helper's progress %llu times and spent %llu cycles there\n", st.visits[26], st.visits[27]);
                fprintf(stderr, "   executor waves on SIMD 0 / 1 / 2 (all passes of the step): %llu %llu %llu; without a SIMD of their own: %llu\n", st.visits[33], st.visits[34], st.visits[35], st.visits[29]);
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
            // Re-tile around the new positions when some particle stands in the OUTERMOST cell of its claim (or blobs are
            // flying: their claims follow the motion).  (Re-tiling as soon as any particle had used part of its margin meant
            // re-tiling before every step while blobs spread a pixel per step -- the minimum over thousands of atoms dips
            // below the margin at once, and claims rebuilt tightly dip again: 0.3 ms of host time per step at 4,096 batches.
            // A particle one cell inside its claim needs more than a whole cell (>= 8 px) in ONE step to leave it; if it
            // does, the claim check fails and the step is re-run with wider claims, as ever.)
            if (st.min_slack < std::max(1, s.margin - 1) || s.swept) s.tiling_dirty = true;
            // Claims padded for motion or after a failed check are wider than a calm scene needs (a larger cell grid per tile:
            // config 3 steps 10 % slower on the claims of its first, fast steps): such tiles are rebuilt every eight steps
            // until a tiling without padding stands.
            if (s.padded && ++s.since_tiling >= 8) s.tiling_dirty = true;
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
                                           (pc.levels_ooo ? 0 : pc.lds_sort ? EGG_PK_VARIANT_SORT_LDS : EGG_PK_VARIANT_SORT_DIRECT) |  // (the out-of-order walk sorts in its own launch)
                                           (pc.fused_pass ? EGG_PK_VARIANT_PASS_FUSED : 0);
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

}  // namespace egghost
