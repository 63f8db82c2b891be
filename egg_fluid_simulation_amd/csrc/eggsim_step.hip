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

#include "eggsim_tile.h"


// Diagnostic build only (-DEGG_PROFILE): per-phase cycle sums of tile 0 go to a side buffer that no
// other code reads.  The shipped library is built without it.
#ifdef EGG_PROFILE
__device__ unsigned long long egg_prof[32];
#define PROF_DECL unsigned long long _pt = __builtin_amdgcn_s_memtime(), _pacc[24] = {0};
#define PROF(k)                                                  \
    {                                                            \
        unsigned long long _n = __builtin_amdgcn_s_memtime();   \
        _pacc[k] += _n - _pt;                                    \
        _pt = _n;                                                \
    }
#define PROF_FLUSH                                                                                    \
    if (tid == 0 && tile == 0 && n > 64) {                                                          \
        for (int _k = 0; _k < 10; ++_k) atomicAdd(&egg_prof[_k], _pacc[_k]);                           \
        for (int _k = 10; _k < 24; ++_k) atomicAdd(&egg_prof[_k + 2], _pacc[_k]);                           \
        atomicAdd(&egg_prof[10], rounds_total);                                                      \
        atomicAdd(&egg_prof[11], 1ull);                                                              \
    }
#else
#define PROF_DECL
#define PROF(k)
#define PROF_FLUSH
#endif

namespace {

// -------------------------------------------------------------- pair scheduling
//
// Every particle has an ordered sequence of pairs: pairs where it is `other`, visited by smaller
// selves (ascending), then its own visits, then pairs visited by larger selves (stale pass only).
// A pair may run when it is the next pending pair of BOTH particles; running every pair under
// that rule reproduces the sequential loop bit for bit.
//
// t.done[p] counts the finished pairs of particle p.  The k-th own pair (a -> b) of a is ready when
// done[a] == nlo[a] + k and done[b] == rank of that pair in b's sequence (precomputed, own_pack).
// The thread of `self` runs the projection, stores both positions, THEN both counters.  A reader
// loads the counters THEN the positions.  One wave's LDS operations execute in issue order, so a
// reader that sees the new counter also sees the new position; no barrier is needed and waves
// of the workgroup run ahead of each other freely.
#define EGG_COMPILER_BARRIER() __asm__ volatile("" ::: "memory")

// Addresses of the tile state the scheduler touches: 32-bit LDS pointers (so that they can be kept in
// one VGPR each across the spin loop), generic pointers when the state lives in global memory.
typedef double egg_d2 __attribute__((ext_vector_type(2)));  // double2 without member functions
template <bool GLOBAL_STATE, class T>
struct StatePtr {
    using type = __attribute__((address_space(3))) T *;
};
template <class T>
struct StatePtr<true, T> {
    using type = T *;
};

template <bool GLOBAL_STATE, bool CACHED>
__device__ inline int execute_dataflow(const Tile &t, int cur, int n, int tid, int nthreads, double overlap,
                                       double compliance, double eps, int total, int spin_sleep,
                                       unsigned int &spins_out) {
    using P32 = typename StatePtr<GLOBAL_STATE, uint32_t>::type;
    using PD2 = typename StatePtr<GLOBAL_STATE, egg_d2>::type;
    int solved = 0;
    unsigned int spins = 0;
    // A thread walks its particles in ascending order (one particle when n <= nthreads).  That
    // cannot deadlock: the reference's sequential order, pairs sorted by (self index, position), is
    // a topological order of the dependencies, so a pair only ever waits for pairs of selves <= its
    // own self, and those belong to the same or an earlier slice [base * nthreads, ...) which
    // every thread finishes first.
    // Threads beyond the particle count (wide workgroups) have nothing to schedule and wait at the barrier.
    const int per = (n + nthreads - 1) / nthreads;
    // Hang guard (cannot trigger with consistent lists): wall-clock based, because the number of turns a
    // waiting wave takes says nothing about progress -- it depends on how fast a turn is relative to a
    // projection (sixteen spinning waves against global memory outran a turn-count limit).  s_memrealtime
    // ticks at 100 MHz; it is read every 1024th turn, on the scalar unit.
    const unsigned long long t_start = wall_clock64();
    const unsigned int t_limit_2p24 = 60u;  // ~10 s for one pass of one tile, in units of 2^24 ticks of the 100 MHz clock
    (void)total;
    bool timed_out = false;
    for (int base = 0; base < per && !timed_out; ++base) {
        const int a = tid + base * nthreads;
        const bool has = a < n;
        const int as = has ? a : 0;
        const int o0 = has ? (int)t.own_off(cur)[a] : 0;
        const int no = has ? (int)t.own_off(cur)[a + 1] - o0 : 0;
        const uint32_t nl = t.nlo[as];
        const double2 wra = t.wr[as];
        const P32 p_da = (P32)&t.done[as];
        const PD2 p_pa = (PD2)&t.pos[as];
        int k = 0;
        uint32_t ent = t.own_pack[min(o0, t.lcap - 1)];
        // Everything that depends only on WHICH pair is next is computed when the thread moves on to that
        // pair, not in every turn of the spin loop: a turn is six LDS loads and two compares.  A thread
        // without a pending pair waits for a counter value that never comes.
        uint32_t want_a, want_b;
        P32 p_db;
        PD2 p_pb, p_wb, p_pc = (PD2)t.pinv;
        int i_next;
        auto aim = [&]() {
            const bool live = k < no;
            const int b = live ? (int)(ent & EGG_IDX) : 0;
            want_a = live ? nl + (uint32_t)k : 0xFFFFFFFFu;
            want_b = ent >> 16;
            p_db = (P32)&t.done[b];
            p_pb = (PD2)&t.pos[b];
            p_wb = (PD2)&t.wr[b];
            i_next = min(o0 + k + 1, t.lcap - 1);
            if (CACHED) p_pc = (PD2)&t.pinv[min(o0 + k, t.lcap - 1)];
            if (!GLOBAL_STATE)  // keep them in registers; do not re-derive them from `ent` in the loop
                __asm__ volatile("" : "+v"(want_a), "+v"(want_b), "+v"(p_db), "+v"(p_pb), "+v"(p_wb), "+v"(i_next), "+v"(p_pc));
        };
        int no_live = no;  // == no unless the hang guard fires
        aim();
        [[clang::code_align(64)]] while (__any(k < no_live)) {
            // relaxed workgroup-scope atomics keep these plain ds_read_b32 / ds_write_b32 (a volatile
            // access would go through the flat path); ordering is by issue order, see above
            const uint32_t da = __hip_atomic_load(p_da, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t db = __hip_atomic_load(p_db, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            EGG_COMPILER_BARRIER();  // counters first, then the data they guard
            // global memory: the counter loads go to L2 while a position load may hit in L1 and be performed
            // first -> the counters must have RETURNED before the data they guard is requested (a
            // workgroup-scope acquire fence alone emits no wait on this target)
            if (GLOBAL_STATE) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const egg_d2 va = *p_pa, vb = *p_pb, vw = *p_wb;
            egg_d2 vc = {0.0, 0.0};
            if (CACHED) vc = *p_pc;
            double2 pa = make_double2(va.x, va.y);
            double2 pb = make_double2(vb.x, vb.y);
            const double2 wrb = make_double2(vw.x, vw.y);
            const double2 pc = make_double2(vc.x, vc.y);
            const uint32_t ent_next = t.own_pack[i_next];
            // consume the speculative loads here so that they are issued back to back with the
            // counters instead of being sunk behind the readiness branch (one LDS latency, not four)
            __asm__ volatile("" ::"v"(pa.x), "v"(pa.y), "v"(pb.x), "v"(pb.y), "v"(wrb.x), "v"(wrb.y), "v"(ent_next), "v"(pc.x),
                             "v"(pc.y));
            const bool ready = ((int)(da == want_a) & (int)(db == want_b)) != 0;
            if (ready) {
                const int b_idx = (int)(ent & EGG_IDX);
                project_pair<CACHED>([&]() { return t.abatch[t.aslot[a]] == t.abatch[t.aslot[b_idx]]; }, (ent & 0x8000u) != 0, pa,
                                     pb, wra, wrb, pc, overlap, compliance, eps);
                *p_pa = (egg_d2){pa.x, pa.y};
                *p_pb = (egg_d2){pb.x, pb.y};
                EGG_COMPILER_BARRIER();  // data first, then the counters that publish it
                if (GLOBAL_STATE) {  // positions acknowledged by memory before the counters announce them
                    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                }
                __hip_atomic_store(p_da, da + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(p_db, db + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ++k;
                ent = ent_next;
                ++solved;
                aim();
            }
            // a wave with nothing ready only burns issue slots its SIMD neighbours could use; park it
            // briefly when several tiles share the CU (costs ~3 % when a tile has the CU to itself)
            if (__builtin_expect(spin_sleep != 0, 0) && !__any(ready)) {
                if (GLOBAL_STATE)
                    __builtin_amdgcn_s_sleep(16);  // keep the waiting waves out of the working wave's memory pipeline
                else
                    __builtin_amdgcn_s_sleep(2);
            }
            ++spins;
            if (__builtin_expect((spins & 1023u) == 0u, 0)) {
                if ((unsigned int)((wall_clock64() - t_start) >> 24) > t_limit_2p24) {  // give up: no lane has a pending pair any more
                    timed_out = true;
                    no_live = 0;
                }
            }
        }
    }
    spins_out = timed_out ? 0xFFFFFFFFu : spins;
    return solved;
}

// GLOBAL_LISTS = false: everything in LDS.  GLOBAL_LISTS = true: the visit lists (own_pack, inc_tmp,
// own_ent) live in a per-tile slice of A.scratch; they are written once per pass and read back
// sequentially, and the pair scheduler prefetches its next entry, so HBM/L2 latency stays off
// the critical path.  Dense tiles (several coincident batches) need this: their lists alone
// exceed the LDS a workgroup may have.
// GLOBAL_STATE = true (implies GLOBAL_LISTS): EVERYTHING except a few scalars lives in the tile's slice of
// A.scratch.  This is the fallback for islands whose particle state does not fit into LDS: same
// algorithm, HBM/L2 latency on every access, workgroup-scope release/acquire around the dataflow
// counters (global memory gives no issue-order guarantee).  Slow, but any island up to the index
// limit (32766 particles) is stepped exactly.
// (Args: EggStepArgs, or the same struct in the constant address space when the launch carries several)
template <bool GLOBAL_LISTS, bool GLOBAL_STATE, bool WIDE = false, bool MULTIGEN = false, class Args = EggStepArgs>
__device__ __forceinline__ void egg_step_body(const Args &A, const int tile = (int)blockIdx.x) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    // In a launch shared with the other particle type the workgroup is sized for the larger tiles: the waves
    // this type's tiles do not need leave at once (a finished wave no longer takes part in s_barrier).
    const int nthreads = (A.threads > 0 && A.threads < (int)blockDim.x) ? A.threads : (int)blockDim.x;
    const int lane = tid & 63;
    if (tile >= A.n_tiles || tid >= nthreads) return;
    PROF_DECL

    // ---------------------------------------------------------------- LDS carve
    Tile t;
    {
        size_t n = (size_t)A.nmax, a = (size_t)A.amax, cc = (size_t)A.ccap, l = (size_t)A.lcap;
        unsigned char *p = GLOBAL_STATE ? A.scratch + (size_t)tile * A.scratch_stride : smem;
        t.pos = (double2 *)carve(p, n * 16);
        t.wr = (double2 *)carve(p, n * 16);
        const bool state_in_lds = WIDE || A.nmax > nthreads;
        t.prev = (double2 *)carve(p, state_in_lds ? n * 16 : 0);
        t.vel = (double2 *)carve(p, state_in_lds ? n * 16 : 0);
        t.atx = (double *)carve(p, a * 8);
        t.aty = (double *)carve(p, a * 8);
        t.afd = (double *)carve(p, a * 8);
        const size_t gn = MULTIGEN ? (size_t)max(2, A.gens) : 2;  // hash generations kept (PassCtx)
        t.ckey_b = (uint32_t *)carve(p, gn * n * 4);
        t.cell_b = (uint32_t *)carve(p, gn * cc * 4);
        t.hkeys_b = (uint32_t *)carve(p, A.use_grid ? 0 : gn * cc * 4);
        t.own_off_b = (uint32_t *)carve(p, (A.single_tile ? gn : 1) * (n + 1) * 4);
        t.inc_off = (uint32_t *)carve(p, (n + 1) * 4);
        t.fill = (uint32_t *)carve(p, n * 4);
        t.done = (uint32_t *)carve(p, n * 4);
        if (!GLOBAL_LISTS) {
            t.own_pack = (uint32_t *)carve(p, l * 4);
            t.inc_tmp = (uint32_t *)carve(p, l * 4);
        }
        t.pinv = (!GLOBAL_LISTS && A.pair_cache) ? (double2 *)carve(p, l * 16) : nullptr;
        t.aclaim = (int32_t *)carve(p, a * 4 * 4);
        t.aoff = (int32_t *)carve(p, (a + 1) * 4);
        t.abatch = (int32_t *)carve(p, a * 4);
        t.aglob = (int32_t *)carve(p, a * 4);
        t.aaabb = (int32_t *)carve(p, a * 4 * 4);
        t.adisp = (int32_t *)carve(p, a * 4 * 4);
        __shared__ int32_t sc_lds[16];
        if (GLOBAL_STATE) {
            t.sc = sc_lds;
        } else {
            t.sc = (int32_t *)carve(p, 16 * 4);
        }
        t.hitems_b = (uint16_t *)carve(p, gn * n * 2);
        t.pslot = (uint16_t *)carve(p, n * 2);
        t.aslot = (uint16_t *)carve(p, n * 2);
        t.nlo = (uint16_t *)carve(p, n * 2);
        if (!GLOBAL_LISTS) {
            t.own_ent_b = (uint16_t *)carve(p, A.single_tile ? gn * l * 2 : 0);
        } else {
            unsigned char *g = GLOBAL_STATE ? p : A.scratch + (size_t)tile * A.scratch_stride;
            t.own_pack = (uint32_t *)g;
            t.inc_tmp = (uint32_t *)(g + egg_align16(l * 4));  // l 64-bit entries
            t.own_ent_b = (uint16_t *)(g + egg_align16(l * 4) + egg_align16(l * 8));
        }
        t.s_n = (int)n;
        t.s_c = (int)cc;
        // without exact-budget mode the previous pass's lists are never read (in_prev uses the cell
        // predicate), so both "buffers" may be the same memory
        t.s_o = A.single_tile ? (int)n + 1 : 0;
        t.s_l = (int)l;
        t.ccap = A.ccap;
        t.lcap = A.lcap;
        t.use_grid = A.use_grid;
    }

    // -------------------------------------------------------------- load tile
    const int a_begin = A.tile_atom_begin[tile];
    const int na = A.tile_atom_begin[tile + 1] - a_begin;
    t.na = na;
    if (tile == 0 && A.status_next) {
        // nothing touches the other status block while this launch runs (the host has read it, the next
        // launch has not started): give it its initial state, so that the host uploads nothing per step
        unsigned long long *q = (unsigned long long *)A.status_next;
        for (int w = tid; w < (int)(sizeof(EggStatus) / 8); w += nthreads) q[w] = 0ull;
        __syncthreads();
        if (tid == 0) A.status_next->min_slack = 0x7FFFFFFF;
    }
    if (tid == 0) {
        int off = 0;
        int ox = 0x7FFFFFFF, oy = 0x7FFFFFFF, hx = -0x7FFFFFFF, hy = -0x7FFFFFFF;
        for (int k = 0; k < na; ++k) {
            int atom = A.tile_atoms[a_begin + k];
            A.atom_fail[atom] = 0;
            t.aoff[k] = off;
            off += A.atom_count[atom];
            for (int q = 0; q < 4; ++q) t.aclaim[4 * k + q] = A.atom_claim[4 * atom + q];
            ox = min(ox, t.aclaim[4 * k + 0]);
            oy = min(oy, t.aclaim[4 * k + 1]);
            hx = max(hx, t.aclaim[4 * k + 2]);
            hy = max(hy, t.aclaim[4 * k + 3]);
            t.atx[k] = A.atom_tx[atom];
            t.aty[k] = A.atom_ty[atom];
            t.afd[k] = A.atom_fd[atom];
            t.abatch[k] = A.atom_batch[atom];
            t.aglob[k] = A.atom_offset[atom];
            t.aaabb[4 * k + 0] = 0x7FFFFFFF;
            t.aaabb[4 * k + 1] = 0x7FFFFFFF;
            t.aaabb[4 * k + 2] = -0x7FFFFFFF;
            t.aaabb[4 * k + 3] = -0x7FFFFFFF;
            for (int q = 0; q < 4; ++q) t.adisp[4 * k + q] = 0;
        }
        t.aoff[na] = off;
        t.sc[2] = off;
        // packed cell coordinates are relative to (ox - 2, oy - 2): claimed cells map to
        // [2, ext + 2], their 3x3 neighbours to [1, ext + 3], so a grid of ext + 4 covers them
        t.sc[3] = ox - 2;
        t.sc[4] = oy - 2;
        long long gw = (long long)hx - ox + 4, gh = (long long)hy - oy + 4;
        t.sc[6] = (int)min(gw, 65535ll);
        t.sc[7] = (int)min(gh, 65535ll);
        if (gw > 65534ll || gh > 65534ll) atomicExch(&A.status->fail_range, 1);
        if (A.use_grid && gw * gh > (long long)A.ccap) atomicExch(&A.status->fail_overflow, 1);
        if (off > A.nmax || na > A.amax) atomicExch(&A.status->fail_overflow, 1);
    }
    __syncthreads();
    // (workgroup-uniform values read back from LDS: tell the compiler, so that they live in scalar registers)
    const int n = __builtin_amdgcn_readfirstlane(min(t.sc[2], A.nmax));
    t.n = n;
    const int org_x = __builtin_amdgcn_readfirstlane(t.sc[3]), org_y = __builtin_amdgcn_readfirstlane(t.sc[4]);
    t.gw = __builtin_amdgcn_readfirstlane(t.sc[6]);
    t.ncell = A.use_grid ? __builtin_amdgcn_readfirstlane((int)min((long long)t.sc[6] * t.sc[7], (long long)A.ccap)) : A.ccap;

    for (int k = 0; k < na; ++k) {
        int atom = A.tile_atoms[a_begin + k];
        int g0 = A.atom_offset[atom];
        int l0 = t.aoff[k], cnt = t.aoff[k + 1] - l0;
        for (int q = tid; q < cnt; q += nthreads) {
            int i = l0 + q;
            if (i >= n) break;
            int g = g0 + q;
            t.pos[i] = make_double2(A.x_in[g], A.y_in[g]);
            t.wr[i] = make_double2(A.inv_mass[g], A.radius[g]);
            t.aslot[i] = (uint16_t)k;
        }
    }
    __syncthreads();
    // can any pair of this tile fail the mass guard w_i + w_j < eps (L:1601)?  (NaN compares false there: not guarded)
    bool tile_guard_mine = false;
    for (int i = tid; i < n; i += nthreads) tile_guard_mine |= t.wr[i].x * 2.0 < A.eps;
    const bool tile_guard = __syncthreads_or(tile_guard_mine) != 0;
    // Velocity and sub-step start position are private to a particle.  With one thread per
    // particle (n <= threads) they stay in that thread's registers for the whole step; only tiles
    // with more particles than threads keep them in LDS.
    // (wide tiles have LDS to spare and registers to save: there the two live in LDS as well)
    const bool regs = !WIDE && n <= nthreads;
    double2 rvel = make_double2(0.0, 0.0), rprev = rvel;
    auto global_index = [&](int i) {
        const int k = t.aslot[i];
        return t.aglob[k] + (i - t.aoff[k]);
    };
    if (regs) {
        if (tid < n) {
            const int g = global_index(tid);
            rvel = make_double2(A.vx_in[g], A.vy_in[g]);
        }
    } else {
        for (int i = tid; i < n; i += nthreads) {
            const int g = global_index(i);
            t.vel[i] = make_double2(A.vx_in[g], A.vy_in[g]);
        }
    }

    PROF(0)  // carve + load
    const double sub_delta = A.sub_delta, eps = A.eps;
    // visits allowed per pass before the budget return: n_collided >= budget after the increment
    long long budget_m = (long long)ceil(A.budget);
    if (budget_m < 1) budget_m = 1;

    // a workgroup with at least three lanes per particle builds the visit lists column-wise (launch
    // classes that have a CU almost to themselves get such workgroups, see the host's launch_type)
    // (a compile-time property of the kernel variant: each variant carries only its own list builder)
    const int parts = WIDE ? 3 : 1;
    if (WIDE && nthreads < 3 * n) {  // the host sizes wide workgroups as 3 lanes per particle
        if (tid == 0) atomicExch(&A.status->fail_stall, 3);
        return;
    }
    __shared__ uint32_t wtot[16];  // per-wave totals of the workgroup-wide prefix sums

    // Ring of hash-generation buffers (more than 2 only for C == 1, S >= 3).  Which buffer a pass uses and
    // how many older generations it sees follow from the loop counters alone (the buffer advances once per
    // sub-step; only the first pass of a sub-step is stale), so no state is carried across the pair
    // scheduler's loop in registers; the budget-cut history lives in t.sc[10] (bit d: the pass of age d was cut).
    // The code for more than two generations is compiled into separate kernel variants (MULTIGEN): inlined
    // into the common ones it cost them their scalar registers (spill reloads inside the scheduler's loop).
    const int G = MULTIGEN ? max(2, A.gens) : 2;
    if (tid == 0) t.sc[10] = 0;
    int pass_seq = 0;
    unsigned long long spins_total = 0;
    unsigned int max_list = 0;
    bool bad = false;

    for (int s = 0; s < A.n_substeps; ++s) {
        // ------------------------- pre-solve (L:1393-1432) + follow constraint (L:1435-1471)
        for (int i = tid; i < n; i += nthreads) {
            double2 ps = t.pos[i];
            double2 v = regs ? rvel : t.vel[i];
            if (regs)
                rprev = ps;
            else
                t.prev[i] = ps;
            v.x = v.x * A.damping;
            v.y = v.y * A.damping;
            if (regs)
                rvel = v;
            else
                t.vel[i] = v;
            double x = ps.x + sub_delta * v.x;
            double y = ps.y + sub_delta * v.y;
            int k = t.aslot[i];
            double fx = t.atx[k], fy = t.aty[k];
            double dx = fx - x, dy = fy - y;
            double current = sqrt(dx * dx + dy * dy);
            double target = t.afd[k];
            double im = t.wr[i].x;
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
            t.pos[i] = make_double2(x, y);
        }
        __syncthreads();
        PROF(1)  // pre-solve + follow

        for (int c = 0; c < A.n_collision_steps; ++c, ++pass_seq) {
            PassCtx ctx;
            const int cur = s % G;
            const int live = (c > 0) ? 0 : (A.n_collision_steps == 1 ? min(s, G - 1) : min(s, 1));
            ctx.cur = cur;
            ctx.G = G;
            ctx.prev = (cur + G - 1) % G;
            ctx.live = live;
            ctx.stale = live > 0;
            ctx.cut_mask = 0;  // read from t.sc[10] once the cell grid's barriers have passed
            ctx.prev_uncut = 1;

            // ----------------------------------- rebuild spatial hash, L:1486-1511
            for (int h = tid; h < t.ncell; h += nthreads) t.cell(cur)[h] = 0;
            if (!t.use_grid)
                for (int h = tid; h < t.ncell; h += nthreads) t.hkeys(cur)[h] = EGG_EMPTY_KEY;
            for (int i = tid; i < n; i += nthreads) {
                t.fill[i] = 0;
                t.done[i] = 0;
            }
            __syncthreads();
            PROF(10)  // hash: clear
            for (int i = tid; i < n; i += nthreads) {
                double2 ps = t.pos[i];
                double fcx = floor(ps.x / A.cell_size);
                double fcy = floor(ps.y / A.cell_size);
                const int32_t *cl = &t.aclaim[4 * t.aslot[i]];
                // NaN / out-of-range coordinates fail the box test below
                int cx = (fcx >= -2.0e9 && fcx <= 2.0e9) ? (int)fcx : 0x7FFFFFF0;
                int cy = (fcy >= -2.0e9 && fcy <= 2.0e9) ? (int)fcy : 0x7FFFFFF0;
                if (cx < cl[0] || cx > cl[2] || cy < cl[1] || cy > cl[3]) {
                    bad = true;
                    A.atom_fail[A.tile_atoms[a_begin + t.aslot[i]]] = 1;  // tells the host whose claim to widen
                    cx = cl[0];  // keep the data structures in range; the step is discarded anyway
                    cy = cl[1];
                }
                uint32_t key = ((uint32_t)(cx - org_x) << 16) | (uint32_t)(cy - org_y);
                t.ckey(cur)[i] = key;
                uint32_t h;
                if (t.use_grid) {
                    h = (uint32_t)((cy - org_y) * t.gw + (cx - org_x));
                    if (h >= (uint32_t)t.ncell) h = 0;  // only after a range/overflow failure
                } else {
                    h = hash_cell(key, t.ccap);
                    for (;;) {
                        uint32_t old = atomicCAS(&t.hkeys(cur)[h], EGG_EMPTY_KEY, key);
                        if (old == EGG_EMPTY_KEY || old == key) break;
                        h = (h + 1) & (uint32_t)(t.ccap - 1);
                    }
                }
                t.pslot[i] = (uint16_t)h;
                atomicAdd(&t.cell(cur)[h], 1u);
            }
            if (__syncthreads_or(bad)) {
                // a particle left its claim: this step will be discarded and re-run with new tiles,
                // so stop here (the pair scheduler must not run on inconsistent cells)
                if (tid == 0) atomicExch(&A.status->fail_claim, 1);
                return;
            }
            PROF(11)  // hash: keys + count
            // cell start offsets: exclusive scan of the per-cell counts, in place (start << 16 | count)
            block_exclusive_scan<true>(t.cell(cur), t.cell(cur), t.ncell, tid, nthreads, wtot);
            __syncthreads();
            PROF(12)  // hash: cell scan
            // unordered scatter, then rank inside the cell so that each cell's items ascend (L:1509)
            for (int i = tid; i < n; i += nthreads) {
                uint32_t m = t.cell(cur)[t.pslot[i]];
                uint32_t pos = atomicAdd(&t.fill[m >> 16], 1u);
                t.inc_tmp[(m >> 16) + pos] = (uint32_t)i;
            }
            __syncthreads();
            PROF(13)  // hash: scatter
            for (int i = tid; i < n; i += nthreads) {
                uint32_t m = t.cell(cur)[t.pslot[i]];
                int st = (int)(m >> 16), cn = (int)(m & 0xFFFFu);
                int rank = 0;
                for (int e = 0; e < cn; ++e) rank += (t.inc_tmp[st + e] < (uint32_t)i) ? 1 : 0;
                t.hitems(cur)[st + rank] = (uint16_t)i;
            }
            __syncthreads();
            PROF(2)  // cell hash

            ctx.cut_mask = (unsigned int)t.sc[10];
            ctx.prev_uncut = !((ctx.cut_mask >> 1) & 1u);
            // ---------------------------------------- visit lists (count, scan, fill)
            // parts == 3: three lanes per particle, one per cell column; the per-column counts are scanned
            // in `sub` (the transposition's scratch, unused until the lists are complete)
            uint32_t *sub = t.inc_tmp;
            // The counting pass keeps what it finds (up to stage_cap partners per lane) in the pair cache's
            // memory, which is idle until the rank pass: the fill then copies instead of enumerating again.
            uint16_t *stage = (parts == 3 && t.pinv && !(MULTIGEN && ctx.live > 1)) ? (uint16_t *)t.pinv : nullptr;
            const int stage_cap = stage ? min(16, (t.lcap * 8) / max(1, 3 * n)) & ~1 : 0;
            if (stage_cap < 4) stage = nullptr;
            if (parts == 3) {
                for (int w = tid; w < 3 * n; w += nthreads) {
                    const int i = w / 3, p = w - 3 * i;
                    uint32_t *slots = stage ? (uint32_t *)(stage + (size_t)w * stage_cap) : nullptr;
                    sub[w] = (MULTIGEN && ctx.live > 1) ? (uint32_t)enum_stale_multi<false>(t, ctx, i, nullptr, 3 * p, 3 * p + 3)
                             : ctx.stale ? (stage ? (uint32_t)enum_stale<2>(t, ctx, i, slots, 3 * p, 3 * p + 3, stage_cap)
                                                  : (uint32_t)enum_stale<0>(t, ctx, i, nullptr, 3 * p, 3 * p + 3))
                                         : (stage ? (uint32_t)enum_fresh_column<2>(t, cur, i, p, slots, stage_cap)
                                                  : (uint32_t)enum_fresh_column<0>(t, cur, i, p, nullptr));
                }
            } else if (MULTIGEN && ctx.live > 1) {
                for (int i = tid; i < n; i += nthreads) t.fill[i] = (uint32_t)enum_stale_multi<false>(t, ctx, i, nullptr);
            } else if (ctx.stale) {
                for (int i = tid; i < n; i += nthreads) t.fill[i] = (uint32_t)enum_stale<0>(t, ctx, i, nullptr);
            } else {
                for (int i = tid; i < n; i += nthreads)
                    t.fill[i] = (uint32_t)enum_fresh<false>(t, cur, i, nullptr);
            }
            __syncthreads();
            PROF(3)  // visit count
            if (parts == 3) {
                block_exclusive_scan<false>(sub, sub, 3 * n, tid, nthreads, wtot);
                __syncthreads();
                for (int i = tid; i <= n; i += nthreads) t.own_off(cur)[i] = sub[3 * i];
            } else {
                block_exclusive_scan<false>(t.fill, t.own_off(cur), n, tid, nthreads, wtot);
            }
            __syncthreads();
            PROF(16)  // list offsets scan
            int total = (int)t.own_off(cur)[n];
            max_list = max(max_list, (unsigned int)total);  // what this pass needs, even when it does not fit
            if (total > t.lcap) {
                // the launch has too little list room: report what is needed and stop; the host
                // re-runs the step with larger lists
                if (tid == 0) {
                    atomicExch(&A.status->fail_overflow, 1);
                    atomicMax(&A.status->max_list, (unsigned long long)total);
                }
                return;
            } else if (parts == 3) {
                for (int w = tid; w < 3 * n; w += nthreads) {
                    const int i = w / 3, p = w - 3 * i;
                    const uint32_t o = sub[w];
                    const int cnt = (int)(sub[w + 1] - o);
                    if (stage && cnt <= stage_cap) {  // staged by the counting pass
                        const uint16_t *slots = stage + (size_t)w * stage_cap;
                        for (int q = 0; q < cnt; ++q) {
                            const uint32_t j = slots[q];
                            t.own_pack[o + q] = j | ((uint32_t)i << 16);
                            atomicAdd(&t.done[j], 1u);
                        }
                    } else if (MULTIGEN && ctx.live > 1)
                        enum_stale_multi<true>(t, ctx, i, &t.own_pack[o], 3 * p, 3 * p + 3);
                    else if (ctx.stale)
                        enum_stale<1>(t, ctx, i, &t.own_pack[o], 3 * p, 3 * p + 3);
                    else
                        enum_fresh_column<1>(t, cur, i, p, &t.own_pack[o]);
                }
            } else if (MULTIGEN && ctx.live > 1) {
                for (int i = tid; i < n; i += nthreads)
                    enum_stale_multi<true>(t, ctx, i, &t.own_pack[t.own_off(cur)[i]]);
            } else if (ctx.stale) {
                for (int i = tid; i < n; i += nthreads)
                    enum_stale<1>(t, ctx, i, &t.own_pack[t.own_off(cur)[i]]);
            } else {
                for (int i = tid; i < n; i += nthreads)
                    enum_fresh<true>(t, cur, i, &t.own_pack[t.own_off(cur)[i]]);
            }
            __syncthreads();

            PROF(4)  // visit fill
            // ------------------------------------------- budget cut, L:1657-1658
            int this_cut = 0;
            if (A.single_tile) {
                // pairs failing the mass guard (L:1601) are marked but not counted; count them out
                long long cut_pos = -1;
                // (serial in thread 0 only when such pairs can exist; otherwise position = count)
                bool guard_possible = false;
                for (int i = tid; i < n; i += nthreads) {
                    double w = t.wr[i].x;
                    guard_possible |= (w * 2.0 < eps) || !(w == w);
                }
                guard_possible = __syncthreads_or(guard_possible);
                if (!guard_possible) {
                    if ((long long)total > budget_m) cut_pos = budget_m;  // keep entries [0, budget_m)
                } else {
                    if (tid == 0) {
                        long long cp = -1, counted = 0;
                        for (int i = 0; i < n && cp < 0; ++i)
                            for (uint32_t e = t.own_off(cur)[i]; e < t.own_off(cur)[i + 1]; ++e) {
                                if (t.wr[i].x + t.wr[t.own_pack[e] & 0xFFFFu].x < eps) continue;
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
                    __syncthreads();
                    for (int i = tid; i <= n; i += nthreads)
                        t.own_off(cur)[i] = min(t.own_off(cur)[i], (uint32_t)cut_pos);
                    for (int i = tid; i < n; i += nthreads) t.done[i] = 0;
                    total = (int)cut_pos;
                    __syncthreads();
                    // the incoming counts gathered while filling included the entries just cut off
                    for (int i = tid; i < n; i += nthreads)
                        for (uint32_t e = t.own_off(cur)[i]; e < t.own_off(cur)[i + 1]; ++e)
                            atomicAdd(&t.done[t.own_pack[e] & 0xFFFFu], 1u);
                    __syncthreads();
                }
            }
            if (A.single_tile) {
                // exact-budget mode: the next pass may have to look pairs up in these (cut) lists
                for (int e = tid; e < total; e += nthreads) t.own_ent(cur)[e] = (uint16_t)t.own_pack[e];
            }
            if (tid == 0) {
                atomicAdd(&A.status->visits[min(pass_seq, EGG_MAX_PASSES - 1)], (unsigned long long)total);
                if (this_cut) atomicExch(&A.status->was_cut, 1);
            }

            PROF(5)  // budget
            // ---- ranks: transpose the visit lists (incoming pairs per particle), rank every incoming
            //      pair inside its particle's sequence, and store the rank next to the visit entry
            // t.done holds the incoming-pair count of every particle (gathered by the fill pass)
            block_exclusive_scan<false>(t.done, t.inc_off, n, tid, nthreads, wtot);
            for (int i = tid; i < n; i += nthreads) t.fill[i] = 0;
            __syncthreads();
            PROF(14)  // transpose: scan
            // one lane per visit entry (other | self << 16, as the fill pass left it): append it to the
            // incoming list of `other`
            for (int e = tid; e < total; e += nthreads) {
                const uint32_t rec = t.own_pack[e];
                const uint32_t j = rec & 0xFFFFu;
                const uint32_t pos = atomicAdd(&t.fill[j], 1u);
                if (GLOBAL_LISTS)
                    ((unsigned long long *)t.inc_tmp)[t.inc_off[j] + pos] = (unsigned long long)(rec >> 16) | ((unsigned long long)e << 32);
                else
                    t.inc_tmp[t.inc_off[j] + pos] = (rec >> 16) | ((uint32_t)e << 16);
            }
            for (int i = tid; i < n; i += nthreads) t.done[i] = 0;  // from here on: the scheduler's progress counter
            __syncthreads();
            PROF(15)  // transpose: scatter
            // one lane per incoming entry: its rank in `other`'s pair sequence = incoming pairs from smaller
            // selves, then other's own visits, then incoming pairs from larger selves (stale pass only)
            for (int x = tid; x < total; x += nthreads) {
                uint32_t cself, e;
                if (GLOBAL_LISTS) {
                    const unsigned long long rec = ((const unsigned long long *)t.inc_tmp)[x];
                    cself = (uint32_t)rec & 0xFFFFu;
                    e = (uint32_t)(rec >> 32);
                } else {
                    const uint32_t rec = t.inc_tmp[x];
                    cself = rec & 0xFFFFu;
                    e = rec >> 16;
                }
                auto inc_self = [&](int q) -> uint32_t {
                    return GLOBAL_LISTS ? (uint32_t)((const unsigned long long *)t.inc_tmp)[q] & 0xFFFFu : t.inc_tmp[q] & 0xFFFFu;
                };
                const int i = (int)(t.own_pack[e] & 0xFFFFu);  // the `other` of entry e: whose incoming list x is in
                const int st = (int)t.inc_off[i], cn = (int)t.inc_off[i + 1] - st;
                uint32_t rank = 0;
                for (int f = 0; f < cn; ++f) rank += (inc_self(st + f) < cself) ? 1u : 0u;
                if (cself > (uint32_t)i) rank += t.own_off(cur)[i + 1] - t.own_off(cur)[i];
                // bit 15: this pair must take the reference path (position-independent part of the test)
                const uint32_t slow = pair_needs_reference(t.wr[cself], t.wr[i], A.overlap_factor, A.collision_compliance, eps)
                                          ? 0x8000u : 0u;
                t.own_pack[e] = (uint32_t)i | slow | (rank << 16);
                if (t.pinv) {
                    const double2 ws = t.wr[cself], wo = t.wr[i];
                    t.pinv[e] = make_double2(egg_rcp_refined((ws.x + wo.x) + A.collision_compliance),
                                             A.overlap_factor * (ws.y + wo.y));
                }
            }
            // pairs in which a particle is `other` of a smaller self come before its own visits
            for (int i = tid; i < n; i += nthreads) {
                const int st = (int)t.inc_off[i], cn = (int)t.inc_off[i + 1] - st;
                uint32_t nl = 0;
                for (int f = 0; f < cn; ++f) {
                    const uint32_t sf = GLOBAL_LISTS ? (uint32_t)((const unsigned long long *)t.inc_tmp)[st + f] & 0xFFFFu
                                                     : t.inc_tmp[st + f] & 0xFFFFu;
                    nl += (sf < (uint32_t)i) ? 1u : 0u;
                }
                t.nlo[i] = (uint16_t)nl;
            }
            __syncthreads();
            // n_collided of this pass (L:1657) counts the visited pairs that passed the mass guard (L:1601): take the
            // others off again.  There are none unless some inverse mass is below eps / 2 (tile_guard, found once per
            // step when the tile is loaded), so the common case pays nothing here -- the hot variants of this kernel
            // sit at their register caps, and a per-pair count inside the rank pass moved spill code into the pair
            // scheduler's loop (+20 % time at four tiles per CU).
            if (tile_guard) {
                unsigned int mine = 0;
                for (int e = tid; e < total; e += nthreads) {
                    const uint32_t rec = t.own_pack[e];
                    // (entries hold `other`; the self of entry e is found through the offsets)
                    int lo = 0, hi = n - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if ((int)t.own_off(cur)[mid] <= e) lo = mid; else hi = mid - 1;
                    }
                    if (t.wr[lo].x + t.wr[rec & EGG_IDX].x < eps) ++mine;
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
                if (lane == 0 && mine) atomicAdd(&A.status->visits[min(pass_seq, EGG_MAX_PASSES - 1)], 0ull - (unsigned long long)mine);
            }

            PROF(6)  // transpose
            // -------------------------------------- dataflow execution of the pair projections
            unsigned int spins = 0;
            int solved = t.pinv ? execute_dataflow<GLOBAL_STATE, true>(t, cur, n, tid, nthreads, A.overlap_factor,
                                                                       A.collision_compliance, eps, total, A.spin_sleep, spins)
                                : execute_dataflow<GLOBAL_STATE, false>(t, cur, n, tid, nthreads, A.overlap_factor,
                                                                        A.collision_compliance, eps, total, A.spin_sleep, spins);
            spins_total += spins;
            __syncthreads();
            PROF(7)  // DAG
            {   // every particle must have finished its whole sequence
                bool short_ = false;
                for (int i = tid; i < n; i += nthreads) {
                    uint32_t want = (t.own_off(cur)[i + 1] - t.own_off(cur)[i]) + (t.inc_off[i + 1] - t.inc_off[i]);
                    short_ |= t.done[i] != want;
                }
                (void)solved;
                if (__syncthreads_or(short_)) {  // cannot happen with consistent lists
                    const int timed_out = __syncthreads_or(spins == 0xFFFFFFFFu);
                    if (tid == 0) atomicExch(&A.status->fail_stall, timed_out ? 2 : 1);
                    return;
                }
            }

            // ------------------------------ clear policy between passes, L:1905-1912
            // (cleared: the next pass is fresh; otherwise the hash lists and collided survive into the next
            // sub-step (Q3) and, with one pass per sub-step, pile up)
            if (tid == 0)
                t.sc[10] = (c + 1 < A.n_collision_steps) ? 0 : (int)(((unsigned int)t.sc[10] << 1) | (this_cut ? 2u : 0u));
        }

        // ------------------------------------------------ post-solve, L:1690-1693
        for (int i = tid; i < n; i += nthreads) {
            const double2 ps = t.pos[i], pv = regs ? rprev : t.prev[i];
            const double2 v = make_double2((ps.x - pv.x) / sub_delta, (ps.y - pv.y) / sub_delta);
            if (regs)
                rvel = v;
            else
                t.vel[i] = v;
        }
        __syncthreads();
    }

    PROF(8)  // post-solve
    // ------------------------------------------------------------- write back
    // particle-wise (thread i <-> particle i while n <= threads, so the velocity is still in its
    // registers); each atom's occupied cell box is reduced with LDS atomics
    int slack = 0x7FFFFFFF;
    for (int i = tid; i < n; i += nthreads) {
        const int k = t.aslot[i];
        const int g = t.aglob[k] + (i - t.aoff[k]);
        const int32_t *cl = &t.aclaim[4 * k];
        const double2 ps = t.pos[i], v = regs ? rvel : t.vel[i];
        A.x_out[g] = ps.x;
        A.y_out[g] = ps.y;
        A.vx_out[g] = v.x;
        A.vy_out[g] = v.y;
        double fcx = floor(ps.x / A.cell_size), fcy = floor(ps.y / A.cell_size);
        int cx = (fcx >= -2.0e9 && fcx <= 2.0e9) ? (int)fcx : 0x7FFFFFF0;
        int cy = (fcy >= -2.0e9 && fcy <= 2.0e9) ? (int)fcy : 0x7FFFFFF0;
        atomicMin(&t.aaabb[4 * k + 0], cx);
        atomicMin(&t.aaabb[4 * k + 1], cy);
        atomicMax(&t.aaabb[4 * k + 2], cx);
        atomicMax(&t.aaabb[4 * k + 3], cy);
        // how far the particle moved in the LAST sub-step (= end velocity * sub_delta), per direction,
        // in 1/16 px: pre-solve continues from exactly that (x += dt * damping * v, L:1411-1418), so the
        // host can predict the next step's travel and size the claims with it
        const double2 pvl = regs ? rprev : t.prev[i];
        const double ddx = ps.x - pvl.x, ddy = ps.y - pvl.y;
        const int qx = (int)fmin(fmax(ddx * 16.0, -2.0e9), 2.0e9), qy = (int)fmin(fmax(ddy * 16.0, -2.0e9), 2.0e9);
        atomicMax(&t.adisp[4 * k + (qx >= 0 ? 0 : 1)], abs(qx));
        atomicMax(&t.adisp[4 * k + (qy >= 0 ? 2 : 3)], abs(qy));
        slack = min(slack, min(min(cx - cl[0], cl[2] - cx), min(cy - cl[1], cl[3] - cy)));
    }
    __syncthreads();
    for (int q = tid; q < 4 * na; q += nthreads) {
        const int atom = A.tile_atoms[a_begin + (q >> 2)];
        A.atom_aabb_out[4 * atom + (q & 3)] = t.aaabb[q];
        A.atom_disp_out[4 * atom + (q & 3)] = t.adisp[q];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) slack = min(slack, __shfl_xor(slack, d, 64));
    bad = __any(bad);
    if (lane == 0) {
        atomicMin(&A.status->min_slack, slack);
        if (bad) atomicExch(&A.status->fail_claim, 1);
        atomicMax(&A.status->max_list, (unsigned long long)max_list);
        if (tid == 0) atomicAdd(&A.status->rounds, spins_total);
    }
    PROF(9)  // write back
#ifdef EGG_PROFILE
    const unsigned long long rounds_total = spins_total;
#endif
    PROF_FLUSH
}

}  // namespace

// tiles of up to 256 particles (the common case): 4 waves, up to 512 registers per lane, no spills
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) egg_step_kernel(EggStepArgs A) {
    egg_step_body<false, false>(A);
}
// the same with registers capped at 96 (5 waves per SIMD): when thousands of tiles queue for the
// chip, six resident tiles per CU beat the spill-free build's four (measured: 2.75 -> 2.15 ms per
// step at 4096 batches), while a tile that has its CU to itself is ~3 % slower
extern "C" __global__ void __launch_bounds__(256, 5) egg_step_kernel_occ(EggStepArgs A) {
    egg_step_body<false, false>(A);
}
// up to 512 threads: tiles that have a CU to themselves run with three lanes per particle, which build the
// visit lists column-wise; the pair scheduler still uses one lane per particle.  Registers are capped at
// 168 (three waves per SIMD): the two waves per SIMD of such a workgroup then leave room for a wave of the
// OTHER particle type's launch, which runs concurrently on its own stream -- uncapped (186 registers)
// the yolk tiles waited for the white tiles to finish (+50 us per step in config 2).
extern "C" __global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3)))
egg_step_kernel_wide(EggStepArgs A) {
    egg_step_body<false, false, true>(A);
}
extern "C" __global__ void __launch_bounds__(1024) egg_step_kernel_gl(EggStepArgs A) { egg_step_body<true, false>(A); }
extern "C" __global__ void __launch_bounds__(1024) egg_step_kernel_gs(EggStepArgs A) { egg_step_body<true, true>(A); }
// Several launch classes -- of both particle types -- in ONE launch: the grid is the concatenation of the
// classes' tiles, every class brings its own arguments (EggStepArgs::threads tells its tiles how many of
// the workgroup's waves to keep).  Two launches on two streams overlap only if they land in different
// hardware queues, and on a full chip the second one's tiles finish ~0.12 ms after the first's whatever the
// order or the stream priorities; as blocks of one grid they are dispatched in order, small tiles first.
// The grid is: tiles of the secondary classes (slots 1..3 of the parameter, small tiles), then the tiles of
// the primary class (slot 0: the white tiles, nearly all of the work).  The primary class reads its arguments
// like the single-class kernels do; the secondary ones read theirs where they lie in the kernel-argument
// segment (scalar loads at a uniform offset) -- indexing the by-value parameter made the compiler stage all
// four structs through scratch for every lane.
typedef const __attribute__((address_space(4))) EggStepArgs EggStepArgsK;
template <bool WIDE>
__device__ __forceinline__ void egg_step_multi(const EggStepArgs4 &P) {
    const int n_secondary = P.a[1].n_tiles + P.a[2].n_tiles + P.a[3].n_tiles;
    int tile = (int)blockIdx.x;
    if (tile >= n_secondary) {
        egg_step_body<false, false, WIDE>(P.a[0], tile - n_secondary);
        return;
    }
    EggStepArgsK *K = (EggStepArgsK *)__builtin_amdgcn_kernarg_segment_ptr();  // EggStepArgs4 is the only parameter
    int k = 1;
    while (k < 3 && tile >= K[k].n_tiles) {
        tile -= K[k].n_tiles;
        ++k;
    }
    egg_step_body<false, false, WIDE, false, EggStepArgsK>(K[k], tile);
}
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) egg_step_kernel_multi(EggStepArgs4 P) {
    egg_step_multi<false>(P);
}
extern "C" __global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3)))
egg_step_kernel_multi_wide(EggStepArgs4 P) {
    egg_step_multi<true>(P);
}
extern "C" __global__ void __launch_bounds__(256, 5) egg_step_kernel_multi_occ(EggStepArgs4 P) { egg_step_multi<false>(P); }

// More than two hash generations alive (one collision pass per sub-step, three or more sub-steps): the same
// three storage variants with the general list builder compiled in.  A rare configuration; no wide / occupancy
// tuned instances.
extern "C" __global__ void __launch_bounds__(256) egg_step_kernel_mg(EggStepArgs A) { egg_step_body<false, false, false, true>(A); }
extern "C" __global__ void __launch_bounds__(1024) egg_step_kernel_gl_mg(EggStepArgs A) { egg_step_body<true, false, false, true>(A); }
extern "C" __global__ void __launch_bounds__(1024) egg_step_kernel_gs_mg(EggStepArgs A) { egg_step_body<true, true, false, true>(A); }

#ifdef EGG_PROFILE
// developer microbenchmark (diagnostic build only): cycles per dependent projection of one wave,
// operands in registers (mode 0), or through LDS like the dataflow loop (mode 1)
extern "C" __global__ void egg_microbench_kernel(int mode, int iters, int active_lanes, unsigned long long *out) {
    __shared__ double2 lpos[128];
    __shared__ double2 lwr[128];
    __shared__ int lbatch[2];
    __shared__ uint16_t lslot[128];
    Tile t;
    t.abatch = lbatch;
    t.aslot = lslot;
    const int lane = threadIdx.x & 63;  // several waves (mode >= 2 only) run the same probe side by side
    lpos[lane] = make_double2(10.0 + lane * 0.37, 5.0 + lane * 0.11);
    lpos[lane + 64] = make_double2(17.0 + lane * 0.35, 9.0 + lane * 0.13);
    lwr[lane] = make_double2(0.7 + 0.001 * lane, 4.0);
    lwr[lane + 64] = make_double2(0.8, 4.0);
    lslot[lane] = 0;
    lslot[lane + 64] = 0;
    if (lane < 2) lbatch[lane] = lane;
    __syncthreads();
    double2 pa = lpos[lane], pb = lpos[lane + 64];
    const double2 wra = lwr[lane], wrb = lwr[lane + 64];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode >= 2) {
        // raw issue/latency probes: 2/3/4 = 1/2/4 independent fma chains, 5 = rcp chain, 6 = rsq chain,
        // 7 = compare + exec-mask branch per fma, 8 = LDS store -> load round trip
        double x0 = pa.x, x1 = pa.y, x2 = pb.x, x3 = pb.y;
        const double m = 1.0000001, c = 1e-9;
        if (lane < active_lanes) {
            for (int i = 0; i < iters; ++i) {
                if (mode == 2) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) x0 = __builtin_fma(x0, m, c);
                } else if (mode == 3) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        x0 = __builtin_fma(x0, m, c);
                        x1 = __builtin_fma(x1, m, c);
                    }
                } else if (mode == 4) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        x0 = __builtin_fma(x0, m, c);
                        x1 = __builtin_fma(x1, m, c);
                        x2 = __builtin_fma(x2, m, c);
                        x3 = __builtin_fma(x3, m, c);
                    }
                } else if (mode == 5) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) x0 = __builtin_amdgcn_rcp(x0);
                } else if (mode == 6) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) x0 = __builtin_amdgcn_rsq(x0);
                } else if (mode == 7) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        if (x0 > 1e300) x0 = x0 * 0.5;  // never taken; compare + saveexec + branch
                        __asm__ volatile("" : "+v"(x0));
                        x0 = __builtin_fma(x0, m, c);
                    }
                } else if (mode == 14) {  // calibration: 32 x s_nop 15 = 512 core cycles
#pragma unroll
                    for (int u = 0; u < 32; ++u) __asm__ volatile("s_nop 15");
                } else if (mode == 9) {
                    float f = (float)x0;
#pragma unroll
                    for (int u = 0; u < 32; ++u) f = __builtin_fmaf(f, 1.0000001f, 1e-9f);
                    x0 = f;
                } else if (mode == 10) {
                    int q = (int)x0 + i;
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        q = q * 3 + 1;
                        __asm__ volatile("" : "+v"(q));
                    }
                    x0 = q;
                } else if (mode == 11) {
                    int acc = 0;
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        acc += (x0 > (double)u) ? 1 : 0;  // v_cmp_f64 + v_addc / cndmask, no branch
                        __asm__ volatile("" : "+v"(acc));
                    }
                    x0 += acc;
                } else if (mode == 12) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        x0 = x0 + c;
                        __asm__ volatile("" : "+v"(x0));
                    }
                } else if (mode == 13) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        x0 = x0 * m;
                        __asm__ volatile("" : "+v"(x0));
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        lpos[lane].x = x0;
                        __asm__ volatile("" ::: "memory");
                        x0 = lpos[lane].x;
                        __asm__ volatile("" : "+v"(x0));
                    }
                }
            }
        }
        pa.x = x0 + x1;
        pb.y = x2 + x3;
    } else if (lane < active_lanes) {
        for (int i = 0; i < iters; ++i) {
            if (mode == 1) {
                pa = lpos[lane];
                pb = lpos[lane + 64];
            }
            project_pair<false>([&]() { return t.abatch[t.aslot[lane]] == t.abatch[t.aslot[lane + 64]]; },
                                (iters & 0x40000000) != 0, pa, pb, wra, wrb, wra, 2.0, 36.0, 1e-8);
            if (mode == 1) {
                lpos[lane] = pa;
                lpos[lane + 64] = pb;
            }
            // keep the pair in range so that every iteration takes the full path
            pb.x = pb.x - 0.9 * (pb.x - pa.x) * 0.01;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = (unsigned long long)__double_as_longlong(pa.x + pb.y);
    }
}
extern "C" void egg_microbench(int mode, int iters, int active_lanes, unsigned long long *cycles) {
    unsigned long long *d = nullptr;
    (void)hipMalloc((void **)&d, 16);
    // mode >= 100: (mode % 100) run by mode / 100 waves of one workgroup
    const int waves = mode >= 100 ? mode / 100 : 1;
    hipLaunchKernelGGL(egg_microbench_kernel, dim3(1), dim3(64 * waves), 0, 0, mode % 100, iters, active_lanes, d);
    unsigned long long h[2] = {0, 0};
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    *cycles = h[0];
}
extern "C" void egg_prof_read(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(egg_prof), sizeof(egg_prof));
}
extern "C" void egg_prof_reset() {
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(egg_prof), z, sizeof(z));
}
#endif

// ---------------------------------------------------------------------------
// self test of the scaling-free division and square root: mismatches against `/` and sqrt() on
// random operands drawn across their windows (counter-based xorshift per thread)
extern "C" __global__ void egg_selftest_arith_kernel(unsigned long long seed, int per_thread,
                                                      unsigned long long *mismatches) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1);
    unsigned long long bad = 0;
    for (int k = 0; k < per_thread; ++k) {
        unsigned long long v[2];
        for (int q = 0; q < 2; ++q) {
            z ^= z << 13;
            z ^= z >> 7;
            z ^= z << 17;
            unsigned long long mant = z & 0xFFFFFFFFFFFFFull;
            z ^= z << 13;
            z ^= z >> 7;
            z ^= z << 17;
            // exponents: mostly the solver's range (2^-40 .. 2^20), sometimes the whole window
            unsigned long long e = (k & 7) ? 1023ull - 40ull + (z >> 11) % 61ull : 723ull + (z >> 11) % 601ull;
            unsigned long long sign = (z >> 5) & 1ull;
            v[q] = (sign << 63) | (e << 52) | mant;
        }
        double n = __longlong_as_double((long long)v[0]), d = __longlong_as_double((long long)v[1]);
        double want = n / d;
        double got = egg_div_with_rcp(n, d, egg_rcp_refined(d));
        if (__double_as_longlong(want) != __double_as_longlong(got)) ++bad;
        // square root: |n| spans [2^-300, 2^300]; its square (up to rounding) spans the sqrt window
        double x = (k & 1) ? fabs(n) : n * n;
        if (x >= 0x1p-600 && x <= 0x1p600) {
            double ws = sqrt(x), gs = egg_sqrt_core(x), gr, rr;
            if (__double_as_longlong(ws) != __double_as_longlong(gs)) ++bad;
            // the projection's normalisation: numerator / sqrt(x) through the reciprocal the root core yields
            egg_sqrt_rcp_core(x, gr, rr);
            if (__double_as_longlong(ws) != __double_as_longlong(gr)) ++bad;
            double wq = d / ws, gq = egg_div_with_rcp(d, gr, rr);
            if (__double_as_longlong(wq) != __double_as_longlong(gq)) ++bad;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

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

// ---------------------------------------------------------------------------
// The per-type reductions the reference keeps in its environment for :draw() (L:1669-1718, L:1795-1815).

// order-preserving map double -> unsigned (so that atomicMin / atomicMax work on doubles)
__device__ inline unsigned long long egg_ordered_key(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// keys[0..5] = min(x - r), min(y - r), max(x + r), max(y + r), max r, max |v| as ordered keys, initialised by
// the host to +inf, +inf, -inf, -inf, 0, 0 (the reference's start values, L:1670-1675).  min / max are exact
// whatever the order; NaN never wins a comparison in the reference and is skipped here.
extern "C" __global__ void egg_env_bounds_kernel(const double *x, const double *y, const double *vx, const double *vy,
                                                   const double *radius, int n, unsigned long long *keys) {
    double lo_x = __longlong_as_double(0x7FF0000000000000ll), lo_y = lo_x, hi_x = -lo_x, hi_y = -lo_x, mr = 0.0, mv = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double px = x[i], py = y[i], r = radius[i], ux = vx[i], uy = vy[i];
        const double a = px - r, b = py - r, c = px + r, d = py + r;
        const double m = sqrt(ux * ux + uy * uy);  // math.magnitude, math.lua:96-98
        if (a < lo_x) lo_x = a;
        if (b < lo_y) lo_y = b;
        if (c > hi_x) hi_x = c;
        if (d > hi_y) hi_y = d;
        if (r > mr) mr = r;
        if (m > mv) mv = m;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        lo_x = fmin(lo_x, __shfl_xor(lo_x, s, 64));
        lo_y = fmin(lo_y, __shfl_xor(lo_y, s, 64));
        hi_x = fmax(hi_x, __shfl_xor(hi_x, s, 64));
        hi_y = fmax(hi_y, __shfl_xor(hi_y, s, 64));
        mr = fmax(mr, __shfl_xor(mr, s, 64));
        mv = fmax(mv, __shfl_xor(mv, s, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&keys[0], egg_ordered_key(lo_x));
        atomicMin(&keys[1], egg_ordered_key(lo_y));
        atomicMax(&keys[2], egg_ordered_key(hi_x));
        atomicMax(&keys[3], egg_ordered_key(hi_y));
        atomicMax(&keys[4], egg_ordered_key(mr));
        atomicMax(&keys[5], egg_ordered_key(mv));
    }
}

// Block b adds up array b (x, y, last_x, last_y) in particle order, like the reference's loops (L:1700-1701,
// L:1803-1804): floating-point addition does not associate, so the sum is serial.  One wave: every lane holds one
// value of a 64-value row, the rows are added in order through constant-lane readlanes (two readlanes and one
// dependent v_add_f64 per value; a counted loop with a lane index in an SGPR cost six dependent instructions,
// ~90 cycles, per value).  Three blocks of four rows rotate without register copies, so the loads of the block
// after next are in flight while a block is added.  out[b] = the sum.
namespace {
constexpr int kSumRows = 4;  // rows of 64 values per block
__device__ __forceinline__ void sums_load(const double *a, int n, int base, int lane, double (&r)[kSumRows]) {
#pragma unroll
    for (int d = 0; d < kSumRows; ++d) {
        // (unconditional, index clamped: rows past the end are never added, and a load inside a branch would make
        // every wait a wait for ALL loads in flight)
        r[d] = a[min(base + d * 64 + lane, n - 1)];
    }
}
__device__ __forceinline__ double sums_add(double s, int n, int base, const double (&r)[kSumRows]) {
#pragma unroll
    for (int d = 0; d < kSumRows; ++d) {
        const int m = n - (base + d * 64);  // values of this row
        const int lo = (int)(__double_as_longlong(r[d]) & 0xFFFFFFFFll), hi = (int)(__double_as_longlong(r[d]) >> 32);
        if (m >= 64) {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const unsigned int l = (unsigned int)__builtin_amdgcn_readlane(lo, k), u = (unsigned int)__builtin_amdgcn_readlane(hi, k);
                s = s + __longlong_as_double((long long)(((unsigned long long)u << 32) | l));
            }
        } else {
            for (int k = 0; k < m; ++k) {
                const unsigned int l = (unsigned int)__builtin_amdgcn_readlane(lo, k), u = (unsigned int)__builtin_amdgcn_readlane(hi, k);
                s = s + __longlong_as_double((long long)(((unsigned long long)u << 32) | l));
            }
        }
    }
    return s;
}
}  // namespace

extern "C" __global__ void __launch_bounds__(64) egg_env_sums_kernel(const double *a0, const double *a1, const double *a2,
                                                                      const double *a3, int n, double *out) {
    const double *a = blockIdx.x == 0 ? a0 : blockIdx.x == 1 ? a1 : blockIdx.x == 2 ? a2 : a3;
    constexpr int kBlock = 64 * kSumRows;
    const int lane = threadIdx.x;
    double s = 0.0;
    double A[kSumRows], B[kSumRows], C[kSumRows];
    sums_load(a, n, 0, lane, A);
    sums_load(a, n, kBlock, lane, B);
    for (int base = 0; base < n; base += 3 * kBlock) {
        sums_load(a, n, base + 2 * kBlock, lane, C);
        s = sums_add(s, n, base, A);
        sums_load(a, n, base + 3 * kBlock, lane, A);
        s = sums_add(s, n, base + kBlock, B);
        sums_load(a, n, base + 4 * kBlock, lane, B);
        s = sums_add(s, n, base + 2 * kBlock, C);
    }
    if (lane == 0) out[blockIdx.x] = s;
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
