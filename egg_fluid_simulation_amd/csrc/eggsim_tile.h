// eggsim_tile.h -- device-side building blocks shared by the step kernels (eggsim_step.hip: one fused
// launch per step, for few tiles per CU) and the packed pipeline (eggsim_packed.hip: one launch per phase,
// pair projections of many islands packed into full waves): the LDS image of a tile, the cell grid lookup,
// the visit-list rules (simulation_handler.lua:1568-1590, "L:") and the XPBD pair projection (L:1514-1545).
// Not part of the public ABI.  Everything here must be compiled with -ffp-contract=off, no fast-math.
#pragma once
#include <hip/hip_runtime.h>
#include "eggsim_device.h"

#define EGG_EMPTY_KEY 0xFFFFFFFFu
#define EGG_IDX 0x7FFFu

namespace {

// LDS image of one tile.  Double-buffered arrays ("this pass" / "previous un-cleared pass") are
// addressed by arithmetic, never through pointer tables, so that a runtime buffer index does not
// force the struct into scratch memory.
struct Tile {
    double2 *pos;   // [n] (x, y)
    double2 *wr;    // [n] (inverse mass, radius)
    double2 *prev;  // [n] position at the start of the sub-step  } only for tiles with more particles than
    double2 *vel;   // [n]                                          } threads; otherwise in registers
    double *atx, *aty, *afd;
    uint32_t *ckey_b;     // [2][nmax]     packed tile-relative cell of each particle
    uint32_t *cell_b;     // [2][ccap]     per cell (start << 16 | count); dense grid or hash slots
    uint32_t *hkeys_b;    // [2][ccap]     hash mode only: cell key of each slot
    uint32_t *own_off_b;  // [2][nmax + 1] CSR offsets of the visit lists
    uint32_t *inc_off;    // [nmax + 1]    CSR offsets of the incoming lists (transposition)
    uint32_t *fill;       // [nmax]        scratch counters
    uint32_t *done;       // [nmax]        pairs finished so far per particle (the dataflow counters)
    uint32_t *own_pack;   // [lcap]        visit lists: other | rank of the pair in other's sequence << 16
                          //               (rank 0 until the rank pass has run)
    uint32_t *inc_tmp;    // [lcap]        scratch: incoming (self | visit-list position << 16); with the lists in
                          //               global memory [lcap] 64-bit entries (self | position << 32): no 16-bit bound
    double2 *pinv;        // [lcap] or null: per visit entry (refined reciprocal of the pair's divisor, its minimum
                          //               distance): the position-independent part of the projection, computed
                          //               by all lanes when the lists are ranked instead of by the serial scheduler
    int32_t *aclaim, *aoff, *abatch, *aglob, *aaabb, *adisp;
    int32_t *sc;  // scalars: 2 particle count; 3 origin x; 4 origin y; 5 misc; 6 gw; 7 gh; 10 cut history; 11 mass-guard pairs of the pass
    uint16_t *hitems_b;   // [2][nmax]     particles sorted by cell, ascending index inside a cell
    uint16_t *pslot, *aslot, *nlo;
    uint16_t *own_ent_b;  // [2][lcap]     exact-budget mode only: plain copies of this and the previous pass's lists
    int s_n, s_c, s_o, s_l;
    int n, na, ccap, lcap, use_grid, gw, ncell;
    __device__ uint32_t *ckey(int b) const { return ckey_b + b * s_n; }
    __device__ uint32_t *cell(int b) const { return cell_b + b * s_c; }
    __device__ uint32_t *hkeys(int b) const { return hkeys_b + b * s_c; }
    __device__ uint32_t *own_off(int b) const { return own_off_b + b * s_o; }
    __device__ uint16_t *hitems(int b) const { return hitems_b + b * s_n; }
    __device__ uint16_t *own_ent(int b) const { return own_ent_b + b * s_l; }
};

__device__ inline unsigned char *carve(unsigned char *&p, size_t bytes) {
    unsigned char *q = p;
    p += (bytes + 15) & ~(size_t)15;
    return q;
}

// inclusive prefix sum over the 64 lanes with data-parallel-primitive moves (no LDS traffic): log steps
// inside each row of 16 lanes, then lane 15 of a row is broadcast into the next row (rows 1 and 3), then
// lane 31 into the upper half.  Lanes that a move does not reach receive 0.
__device__ inline int wave_incl_scan(int v, int /*lane*/) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// exclusive prefix sum of cnt[0..len) into off[0..len], off[len] = total, by the whole workgroup: every thread takes a run of consecutive entries, one
// wave scan, one barrier for the per-wave totals.  PACK writes the cell records (start << 16 | count)
// and no total.  cnt and off may be the same array.  The caller puts barriers before (cnt complete) and
// after (off visible); wtot holds one word per wave.
template <bool PACK>
__device__ inline void block_exclusive_scan(const uint32_t *cnt, uint32_t *off, int len, int tid, int nthreads,
                                            uint32_t *wtot) {
    const int per = (len + nthreads - 1) / nthreads;
    const int b0 = tid * per, b1 = min(b0 + per, len);
    uint32_t sum = 0;
    for (int q = b0; q < b1; ++q) sum += cnt[q];
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t incl = (uint32_t)wave_incl_scan((int)sum, lane);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    // totals of the waves before this one: one load per lane (at most 16 waves), summed along the row
    const int l16 = lane & 15;
    int before = (l16 < wave) ? (int)wtot[l16] : 0;
    before += __builtin_amdgcn_update_dpp(0, before, 0x111, 0xf, 0xf, true);
    before += __builtin_amdgcn_update_dpp(0, before, 0x112, 0xf, 0xf, true);
    before += __builtin_amdgcn_update_dpp(0, before, 0x114, 0xf, 0xf, true);
    before += __builtin_amdgcn_update_dpp(0, before, 0x118, 0xf, 0xf, true);
    uint32_t run = incl - sum + (uint32_t)__builtin_amdgcn_readlane(before, 15);
    for (int q = b0; q < b1; ++q) {
        const uint32_t v = cnt[q];
        off[q] = PACK ? ((run << 16) | v) : run;
        run += v;
    }
    if (!PACK && tid == nthreads - 1) off[len] = run;
}

__device__ inline int cell_slot(uint32_t ka, uint32_t kb) {
    // slot of cell ka in the 3x3 loop around cell kb (x offset outer, y inner; L:1568-1569), or -1
    int dcx = (int)(ka >> 16) - (int)(kb >> 16);
    int dcy = (int)(ka & 0xFFFFu) - (int)(kb & 0xFFFFu);
    if (dcx < -1 || dcx > 1 || dcy < -1 || dcy > 1) return -1;
    return (dcx + 1) * 3 + (dcy + 1);
}

__device__ inline uint32_t hash_cell(uint32_t key, int cap) { return (key * 2654435761u) >> 7 & (uint32_t)(cap - 1); }

// (start << 16 | count) of a cell's item list; 0 when the cell is empty
__device__ inline uint32_t cell_meta(const Tile &t, int buf, uint32_t key) {
    if (t.use_grid) return t.cell(buf)[(int)(key & 0xFFFFu) * t.gw + (int)(key >> 16)];
    const uint32_t *keys = t.hkeys(buf);
    uint32_t h = hash_cell(key, t.ccap);
    for (int probe = 0; probe < t.ccap; ++probe) {
        uint32_t k = keys[h];
        if (k == key) return t.cell(buf)[h];
        if (k == EGG_EMPTY_KEY) return 0;
        h = (h + 1) & (uint32_t)(t.ccap - 1);
    }
    return 0;
}

// ---- f64 division and square root without the hardware expansions' scaling steps -----------
// hipcc expands `n / d` to v_div_scale x2, v_rcp_f64, two FMA Newton steps, q = n * r,
// rem = fma(-d, q, n), v_div_fmas (= fma(rem, r, q) with scale fix-up), v_div_fixup (special
// values), and sqrt(x) to an input ldexp, v_rsq_f64, a Goldschmidt iteration, an output ldexp and
// a class test for 0 / inf.  A lone wave issues one FP64 instruction per ~8 cycles, and the pair
// projection below sits on the step's critical path, so every instruction counts.  The functions
// here are the same arithmetic minus the scaling / special-value steps, which are the identity
// while all operands and results are normal numbers far from the exponent limits.  Inside that
// window the results are bit-identical to `/` and sqrt() (egg_selftest_arith compares them on
// random operands on the device); outside it the projection uses the operators.
__device__ __forceinline__ double egg_rcp_refined(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
__device__ __forceinline__ double egg_div_with_rcp(double n, double d, double r) {
    double q = n * r;
    double rem = __builtin_fma(-d, q, n);
    return __builtin_fma(rem, r, q);
}
__device__ __forceinline__ double egg_sqrt_core(double x) {  // valid for x in [2^-600, 2^600]
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return g;
}
// the same square root, plus a refined reciprocal of the root: the Goldschmidt iteration carries
// h ~ 1 / (2 sqrt(x)) to ~2^-51, one Newton step on 2h gives the quality of egg_rcp_refined(root) without
// the v_rcp_f64 and one of its two Newton steps (the quotient below only needs a faithful reciprocal)
__device__ __forceinline__ void egg_sqrt_rcp_core(double x, double &root, double &rroot) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    root = g;
    const double r0 = h + h;
    const double e = __builtin_fma(-g, r0, 1.0);
    rroot = __builtin_fma(r0, e, r0);
}
#define EGG_ARITH_LO 0x1p-300
#define EGG_ARITH_HI 0x1p300

// XPBD distance projection between two particles, L:1514-1545 with the collision caller
// L:1632-1654 (and the numerically dead cohesion block L:1603-1630), written operator by operator
// like the reference.  pa/pb are the current positions, wra/wrb the (inverse mass, radius) records.
// This is the general path; the pair scheduler calls it only for the rare pairs project_pair
// below cannot prove to be inside the window of its hand-expanded arithmetic.
// same_batch() is only evaluated for exactly coincident particles (it may be a slow lookup).
template <class SameBatch>
__device__ __forceinline__ void project_pair_reference(SameBatch same_batch, double2 &pa, double2 &pb, double2 wra,
                                                    double2 wrb, double overlap, double compliance, double eps) {
    const double wa = wra.x, wb = wrb.x;
    const double wsum = wa + wb;
    if (wsum < eps) return;  // L:1601
    const double dx = pb.x - pa.x, dy = pb.y - pa.y;
    const double d2 = dx * dx + dy * dy;
    if (d2 <= 0.0) {
        if (same_batch()) {
            // cohesion fires only for coincident same-batch particles (interaction distance 0,
            // L:1608-1616); its corrections are +0 for self and -0 for other, so the only
            // observable effect is x_self + 0.0 (turns -0.0 into +0.0)
            pa.x = pa.x + 0.0;
            pa.y = pa.y + 0.0;
        }
    }
    const double min_distance = overlap * (wra.y + wrb.y);
    if (d2 <= min_distance * min_distance) {
        const double divisor = wsum + compliance;
        const double current = sqrt(d2);
        const double violation = current - min_distance;
        double nx, ny;
        if (current < eps) {  // math.normalize, math.lua:53-60
            nx = 0.0;
            ny = 0.0;
        } else {
            nx = dx / current;
            ny = dy / current;
        }
        double cax, cay, cbx, cby;
        if (divisor < eps) {
            cax = cay = cbx = cby = 0.0;
        } else {
            double correction = -violation / divisor;
            const double max_correction = fabs(violation);
            if (correction < -max_correction) correction = -max_correction;
            if (correction > max_correction) correction = max_correction;
            cax = -nx * correction * wa;
            cay = -ny * correction * wa;
            cbx = nx * correction * wb;
            cby = ny * correction * wb;
        }
        pa.x = pa.x + cax;
        pa.y = pa.y + cay;
        pb.x = pb.x + cbx;
        pb.y = pb.y + cby;
    }
}

// is the pair's position-independent data outside what the fast path below may assume?  Evaluated
// once per pair when the visit lists are ranked (all lanes busy there) and kept as a flag bit in
// the list entry, so the serial pair scheduler does not pay for these comparisons.  NaN fails every
// comparison and therefore sets the flag.
__device__ __forceinline__ bool pair_needs_reference(double2 wra, double2 wrb, double overlap, double compliance,
                                                     double eps) {
    const double wsum = wra.x + wrb.x;
    const double divisor = wsum + compliance;
    const double min_distance = overlap * (wra.y + wrb.y);
    const double md2 = min_distance * min_distance;
    const bool fine = ((int)(wsum >= eps) & (int)(divisor >= eps) & (int)(divisor <= EGG_ARITH_HI) &
                       (int)(md2 <= 0x1p600)) != 0;
    return !fine;
}

// The same projection for the pair scheduler: one branch for "within range", one for "inside the
// arithmetic window", the rest straight-line.  `slow` is pair_needs_reference() of the pair.
// Window argument: !slow gives wsum >= eps, eps <= divisor <= 2^300 and md2 <= 2^600, hence
// d2 <= md2 <= 2^600; current >= eps (eps >= 2^-300 is enforced by the host) gives d2 >= 2^-601 up
// to rounding, far inside the square root's window [2^-767, ...), and covers the normalisation's
// denominator; numerators must be non-zero normal numbers (a zero numerator would lose its sign in the
// hand expansion).  Anything else -- including NaN, which fails every comparison -- takes the
// reference path from the unchanged inputs.
template <bool CACHED, class SameBatch>
__device__ __forceinline__ void project_pair(SameBatch same_batch, bool slow, double2 &pa, double2 &pb,
                                             double2 wra, double2 wrb, double2 cached, double overlap, double compliance,
                                             double eps) {
    const double dx = pb.x - pa.x, dy = pb.y - pa.y;
    const double d2 = dx * dx + dy * dy;
    const double min_distance = CACHED ? cached.y : overlap * (wra.y + wrb.y);
    const double md2 = min_distance * min_distance;
    if (((int)(d2 <= md2) | (int)slow) != 0) {
        double current, r_current;
        egg_sqrt_rcp_core(d2, current, r_current);
        const double violation = current - min_distance;
        const bool fast = ((int)!slow & (int)(current >= eps) & (int)(fabs(dx) >= EGG_ARITH_LO) &
                           (int)(fabs(dy) >= EGG_ARITH_LO) & (int)(fabs(violation) >= EGG_ARITH_LO)) != 0;
        if (__builtin_expect(fast, 1)) {
            const double wa = wra.x, wb = wrb.x;
            const double divisor = (wa + wb) + compliance;
            const double r_divisor = CACHED ? cached.x : egg_rcp_refined(divisor);
            const double nx = egg_div_with_rcp(dx, current, r_current);
            const double ny = egg_div_with_rcp(dy, current, r_current);
            double correction = egg_div_with_rcp(-violation, divisor, r_divisor);
            // clamp(correction, -|violation|, |violation|): no NaN or signed-zero subtleties here
            // (|violation| > 0), so max/min are the reference's two comparisons
            __asm__("v_max_f64 %0, %1, -|%2|" : "=v"(correction) : "v"(correction), "v"(violation));
            __asm__("v_min_f64 %0, %1, |%2|" : "=v"(correction) : "v"(correction), "v"(violation));
            // -nx * correction == -(nx * correction) bit for bit (rounding is sign-symmetric)
            const double tx = nx * correction, ty = ny * correction;
            pa.x = pa.x + -tx * wa;
            pa.y = pa.y + -ty * wa;
            pb.x = pb.x + tx * wb;
            pb.y = pb.y + ty * wb;
        } else {
            project_pair_reference(same_batch, pa, pb, wra, wrb, overlap, compliance, eps);
        }
    }
}

// project_pair<true> without branches on its dependent chain, for the packed executor: one wave walks an island's
// levels and every level waits for the one before it, so a compare + exec-mask branch in the middle of the chain (two of
// them in project_pair) is dead time for the whole island.  All lanes run the hand-expanded arithmetic on whatever they
// hold (no traps; garbage stays in the lane), the results are taken by select where the reference would have taken
// them, and only the rare pairs outside the arithmetic window (or flagged slow) branch, afterwards, into the reference
// path from the unchanged inputs.  Same operations in the same order as project_pair: same bits.
// Returns whether pa / pb changed (the caller stores them or not by choosing the ADDRESS: the select of the address
// needs only `fast`, known halfway down the chain, so neither a data select nor a store branch sits behind the last add).
// after_first_use(): called once the positions have been consumed for the first time (the executor issues the scalar
// load of a later chunk's descriptor there: in front of that point it would be waited for together with the positions).
// wa / wb: the inverse masses; divisor = (wa + wb) + compliance, cached = (its refined reciprocal, the minimum distance),
// md2 = minimum distance squared: all computed by the caller off the chain.  fetch_wr(wra, wrb): the whole (inverse
// mass, radius) records, asked for only by the rare pair that takes the reference path.
template <class SameBatch, class Hook, class FetchWr>
__device__ __forceinline__ bool project_pair_predicated(SameBatch same_batch, Hook after_first_use, FetchWr fetch_wr, bool active,
                                                        bool slow, double2 &pa, double2 &pb, double wa, double wb, double divisor,
                                                        double2 cached, double md2, double overlap, double compliance, double eps) {
    const double dx = pb.x - pa.x, dy = pb.y - pa.y;
    after_first_use();
    const double d2 = dx * dx + dy * dy;
    const double min_distance = cached.y;
    const bool in_range = active && (((int)(d2 <= md2) | (int)slow) != 0);
    double current, r_current;
    egg_sqrt_rcp_core(d2, current, r_current);
    const double violation = current - min_distance;
    const bool fast = ((int)!slow & (int)(current >= eps) & (int)(fabs(dx) >= EGG_ARITH_LO) & (int)(fabs(dy) >= EGG_ARITH_LO) &
                       (int)(fabs(violation) >= EGG_ARITH_LO)) != 0;
    const double nx = egg_div_with_rcp(dx, current, r_current);
    const double ny = egg_div_with_rcp(dy, current, r_current);
    double correction = egg_div_with_rcp(-violation, divisor, cached.x);
    __asm__("v_max_f64 %0, %1, -|%2|" : "=v"(correction) : "v"(correction), "v"(violation));
    __asm__("v_min_f64 %0, %1, |%2|" : "=v"(correction) : "v"(correction), "v"(violation));
    const double tx = nx * correction, ty = ny * correction;
    const double ax = pa.x + -tx * wa, ay = pa.y + -ty * wa;
    const double bx = pb.x + tx * wb, by = pb.y + ty * wb;
    const double2 pa0 = pa, pb0 = pb;
    pa = make_double2(ax, ay);
    pb = make_double2(bx, by);
    if (__builtin_expect(in_range && !fast, 0)) {  // (the common case falls through one untaken branch)
        double2 qa = pa0, qb = pb0, wra, wrb;
        fetch_wr(wra, wrb);
        project_pair_reference(same_batch, qa, qb, wra, wrb, overlap, compliance, eps);
        pa = qa;
        pb = qb;
    }
    return in_range;  // out of range: pa / pb hold what the arithmetic made of the inputs, not to be stored
}

// Hash generations.  The reference clears the cell lists and `collided` only BETWEEN the collision passes
// of a sub-step (L:1905-1912), so the first pass of a later sub-step still sees the previous pass (Q3);
// with a single collision pass per sub-step nothing is ever cleared inside a step and every sub-step
// adds a generation.  Buffers form a ring of G; age 0 is this pass, age d the pass d rebuilds ago.
struct PassCtx {
    int cur;         // buffer of this pass
    int prev;        // buffer of the previous pass (age 1)
    int live;        // older generations still alive (0 = fresh pass)
    int G;           // ring size
    int stale;       // live > 0
    int prev_uncut;  // the previous pass visited every adjacent pair (no budget cut)
    unsigned int cut_mask;  // bit d: the pass of age d was cut by the budget
    __device__ int buf(int age) const { return (cur + G - age % G) % G; }
};

// ------------------------------------------------------------------ visit lists
//
// own(i): the partners particle i visits as `self`, in the reference's attempt order.

// fresh pass (hash and collided were cleared): adjacency is symmetric, the smaller index visits,
// so own(i) = { j > i in the 3x3 cells }, cells in loop order, ascending j inside a cell.
// FILL also counts, per partner, how many selves visit it (t.done doubles as that counter until
// the pair scheduler starts): the transposition below needs it and the atomic needs no return.
// One column (x offset p - 1) of the 3x3 loop: the visit list of i is the concatenation of its three
// columns, so three lanes can build it side by side when the workgroup has lanes to spare (wide kernel);
// otherwise one lane walks the columns in turn (enum_fresh below).
// MODE 0: count; 1: fill (write other | self << 16 to dst, count other's incoming pairs); 2: count and keep up
// to `cap` partners in the 16-bit staging slots at dst, so that the fill needs no second enumeration
template <int MODE>
__device__ inline int enum_fresh_column(const Tile &t, int cur, int i, int p, uint32_t *dst, int cap = 0) {
    const uint32_t ki = t.ckey(cur)[i];
    const uint16_t *items = t.hitems(cur);
    uint32_t m[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) m[r] = cell_meta(t, cur, (uint32_t)((int)ki + (p - 1) * 65536 + (r - 1)));
    int count = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int st = (int)(m[r] >> 16), cn = (int)(m[r] & 0xFFFFu);
        for (int e = 0; e < cn; ++e) {
            const int j = items[st + e];
            if (j > i) {
                if (MODE == 1) {
                    dst[count] = (uint32_t)j | ((uint32_t)i << 16);
                    atomicAdd(&t.done[j], 1u);
                } else if (MODE == 2) {
                    if (count < cap) ((uint16_t *)dst)[count] = (uint16_t)j;
                }
                ++count;
            }
        }
    }
    return count;
}

// the whole neighbourhood, column after column (a fully unrolled nine-cell version with every load in
// flight at once was ~5 % faster for a lone tile but cost 25-45 registers, i.e. a wave per SIMD)
template <bool FILL>
__device__ inline int enum_fresh(const Tile &t, int cur, int i, uint32_t *dst) {
    int count = 0;
    for (int p = 0; p < 3; ++p) count += enum_fresh_column<FILL ? 1 : 0>(t, cur, i, p, FILL ? dst + count : dst);
    return count;
}

// is the unordered pair {i, j} in `collided` from the previous pass?  (ko* = previous cells)
__device__ inline bool in_prev(const Tile &t, const PassCtx &c, int i, int j, uint32_t koi, uint32_t koj) {
    if (cell_slot(koj, koi) < 0) return false;  // never met in the previous (fresh) pass
    if (c.prev_uncut) return true;
    int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t *off = t.own_off(c.prev);
    const uint16_t *ent = t.own_ent(c.prev);
    for (uint32_t e = off[lo]; e < off[lo + 1]; ++e)
        if (ent[e] == (uint16_t)hi) return true;
    return false;
}

// stale pass: does self i visit j when it meets it at 3x3 slot s through j's old (isnew = 0) or
// new cell?  kn* / ko* = this pass's / the previous pass's cell of i and j.
__device__ inline bool accept_stale(const Tile &t, const PassCtx &c, int i, int j, int s, int isnew, uint32_t kni,
                                    uint32_t koi, uint32_t knj, uint32_t koj) {
    if (j == i) return false;
    // j can sit in i's attempt order twice (old cell and new cell): first occurrence wins
    if (!isnew) {
        int s2 = cell_slot(knj, kni);
        if (s2 >= 0 && s2 < s) return false;
    } else {
        int s2 = cell_slot(koj, kni);
        if (s2 >= 0 && s2 <= s) return false;
    }
    if (in_prev(t, c, i, j, koi, koj)) return false;
    if (j < i) {  // j's loop ran first: did it meet i?
        if (cell_slot(kni, knj) >= 0 || cell_slot(koi, knj) >= 0) return false;
    }
    return true;
}

template <int MODE>  // as in enum_fresh_column
__device__ inline int enum_stale(const Tile &t, const PassCtx &c, int i, uint32_t *dst, int s0 = 0, int s1 = 9, int cap = 0) {
    const uint32_t *kn = t.ckey(c.cur), *ko = t.ckey(c.prev);
    const uint32_t kni = kn[i], koi = ko[i];
    const bool settled = c.prev_uncut && kni == koi;
    int count = 0;
    for (int s = s0; s < s1; ++s) {
        const uint32_t nk = (uint32_t)((int)kni + (s / 3 - 1) * 65536 + (s % 3 - 1));
        const uint32_t mo = cell_meta(t, c.prev, nk), mn = cell_meta(t, c.cur, nk);
#pragma unroll
        for (int isnew = 0; isnew < 2; ++isnew) {
            const uint32_t m = isnew ? mn : mo;
            const uint16_t *items = t.hitems(isnew ? c.cur : c.prev);
            const int st = (int)(m >> 16), cn = (int)(m & 0xFFFFu);
            for (int e0 = 0; e0 < cn; e0 += 4) {
                int j[4];
                uint32_t knj[4], koj[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) j[q] = (int)items[min(st + e0 + q, t.n - 1)];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    knj[q] = kn[j[q]];
                    koj[q] = ko[j[q]];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    // (shortcut for the common case: neither particle changed its cell since the previous pass, so the
                    // pair was adjacent then as well and -- that pass not being cut -- is in `collided`)
                    if (e0 + q < cn && !(settled && knj[q] == koj[q]) &&
                        accept_stale(t, c, i, j[q], s, isnew, kni, koi, knj[q], koj[q])) {
                        if (MODE == 1) {
                            dst[count] = (uint32_t)j[q] | ((uint32_t)i << 16);
                            atomicAdd(&t.done[j[q]], 1u);
                        } else if (MODE == 2) {
                            if (count < cap) ((uint16_t *)dst)[count] = (uint16_t)j[q];
                        }
                        ++count;
                    }
            }
        }
    }
    return count;
}

// ---- more than two generations alive (one collision pass per sub-step, three or more sub-steps) ----
// The same rules, stated over ages: a cell's list holds the oldest generation's entries first.

// is {i, j} in `collided`?  It was attempted at the pass of age p iff one of them met an entry (of age >= p)
// of the other in the 3x3 cells around the cell it was in at that pass; a pass the budget cut only counts
// for what it actually visited.
__device__ inline bool in_collided_multi(const Tile &t, const PassCtx &c, int i, int j) {
    for (int p = 1; p <= c.live; ++p) {
        const int bp = c.buf(p);
        const uint32_t kpi = t.ckey(bp)[i], kpj = t.ckey(bp)[j];
        bool adj = false;
        for (int b = p; b <= c.live && !adj; ++b) {
            const int bb = c.buf(b);
            adj = cell_slot(t.ckey(bb)[j], kpi) >= 0 || cell_slot(t.ckey(bb)[i], kpj) >= 0;
        }
        if (!adj) continue;
        if (!((c.cut_mask >> p) & 1u)) return true;
        const uint32_t *off = t.own_off(bp);
        const uint16_t *ent = t.own_ent(bp);
        for (uint32_t e = off[i]; e < off[i + 1]; ++e)
            if (ent[e] == (uint16_t)j) return true;
        for (uint32_t e = off[j]; e < off[j + 1]; ++e)
            if (ent[e] == (uint16_t)i) return true;
    }
    return false;
}

// does self i visit j when it meets j's entry of age d at 3x3 slot s?
__device__ inline bool accept_multi(const Tile &t, const PassCtx &c, int i, int j, int s, int d, uint32_t kni) {
    if (j == i) return false;
    // j sits in i's attempt order once per generation that put it into the neighbourhood: first occurrence wins
    for (int d2 = c.live; d2 >= 0; --d2) {
        const int s2 = cell_slot(t.ckey(c.buf(d2))[j], kni);
        if (s2 >= 0 && (s2 < s || (s2 == s && d2 > d))) return false;
    }
    if (in_collided_multi(t, c, i, j)) return false;
    if (j < i) {  // j's loop ran first in this pass: did it meet any entry of i?
        const uint32_t knj = t.ckey(c.cur)[j];
        for (int a = 0; a <= c.live; ++a)
            if (cell_slot(t.ckey(c.buf(a))[i], knj) >= 0) return false;
    }
    return true;
}

template <bool FILL>
__device__ inline int enum_stale_multi(const Tile &t, const PassCtx &c, int i, uint32_t *dst, int s0 = 0, int s1 = 9) {
    const uint32_t kni = t.ckey(c.cur)[i];
    int count = 0;
    for (int s = s0; s < s1; ++s) {
        const uint32_t nk = (uint32_t)((int)kni + (s / 3 - 1) * 65536 + (s % 3 - 1));
        for (int d = c.live; d >= 0; --d) {
            const int bd = c.buf(d);
            const uint32_t m = cell_meta(t, bd, nk);
            const uint16_t *items = t.hitems(bd);
            const int st = (int)(m >> 16), cn = (int)(m & 0xFFFFu);
            for (int e = 0; e < cn; ++e) {
                const int j = (int)items[st + e];
                if (accept_multi(t, c, i, j, s, d, kni)) {
                    if (FILL) {
                        dst[count] = (uint32_t)j | ((uint32_t)i << 16);
                        atomicAdd(&t.done[j], 1u);
                    }
                    ++count;
                }
            }
        }
    }
    return count;
}

}  // namespace
