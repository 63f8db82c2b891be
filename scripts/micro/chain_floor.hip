// chain_floor.hip -- what one level of the packed executor costs when the wave does NOTHING but the dependent chain:
// positions of both particles from LDS, the branch-free projection (eggsim_tile.h, the executor's own arithmetic),
// positions back to LDS.  Variants: (0) pair constants and inverse masses already in registers, (1) the same plus
// three 16-byte LDS reads of a prepared record per level (what a helper wave could leave in an LDS ring), (2) as the
// executor does today: two 16-byte global gathers + the reciprocal of the divisor per level (one level ahead).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ../../egg_fluid_simulation_amd/csrc -o build/chain_floor chain_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "eggsim_tile.h"

template <int MODE>
__global__ void __launch_bounds__(64) chain(int levels, const double2 *gwr, unsigned long long *out, double2 *sink) {
    __shared__ double2 lpos[1408];
    __shared__ double2 ring[64 * 3 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1408; i += 64) lpos[i] = make_double2(100.0 + 7.0 * (i % 37) + 0.013 * i, 50.0 + 9.0 * (i / 37) + 0.007 * i);
    for (int i = lane; i < 64 * 3 * 4; i += 64) ring[i] = make_double2(0.0263, 16.0);
    __syncthreads();
    const double overlap = 2.0, compliance = 36.0, eps = 1e-8;
    int ga = MODE == 3 ? lane : lane * 2, gb = MODE == 3 ? 64 + lane : lane * 2 + 1;
    double2 wa = gwr[ga], wb = gwr[gb];
    double2 pc = make_double2(egg_rcp_refined((wa.x + wb.x) + compliance), overlap * (wa.y + wb.y));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < levels; ++l) {
        if (MODE == 1) {  // a prepared record per level from an LDS ring: (wa), (wb), (rdiv, mind)
            const double2 *r = ring + ((l & 3) * 64 + lane) * 3;
            wa = make_double2(wa.x * 0 + r[0].x * 38.0, 4.0);
            wb = make_double2(r[1].x * 38.0, 4.0);
            pc = r[2];
        } else if (MODE == 2) {
            wa = gwr[ga];
            wb = gwr[gb];
            pc = make_double2(egg_rcp_refined((wa.x + wb.x) + compliance), overlap * (wa.y + wb.y));
        }
        double2 pa = lpos[ga], pb = lpos[gb];
        const bool store = project_pair_predicated([&]() { return false; }, []() {}, [&](double2 &x, double2 &y) { x = wa; y = wb; }, true, false, pa, pb, wa.x, wb.x, (wa.x + wb.x) + compliance, pc, pc.y * pc.y, overlap, compliance, eps);
        lpos[store ? ga : 1344 + lane] = pa;
        lpos[store ? gb : 1344 + lane] = pb;
        // next level: other particles (a dependent chain through LDS: b of this level is a of the next lane's)
        if (MODE == 3) {  // consecutive 16-byte slots per lane group: no LDS bank conflicts at all
            ga = (ga + 128) % 1280;
            gb = (gb + 128) % 1280;
        } else {  // scattered slots, like the particles of a sorted pair list
            ga = (ga + 129) % 1280;
            gb = (gb + 131) % 1280;
            if (ga == gb) gb = (gb + 1) % 1280;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = lpos[lane];
}

// The same chain with NO address arithmetic on it (eight precomputed slot pairs per lane, the loop unrolled by eight) and
// only `ACTIVE` lanes carrying a pair, the others doing what the executor's idle lanes do (read slot 0, write their own
// spare slot).  LAYOUT 0: active lanes packed at the low end, random slots; 1: active lanes spread evenly over the wave,
// random slots; 2: packed, slots chosen so that no two lanes of an instruction share a bank group.
template <int ACTIVE, int LAYOUT>
__global__ void __launch_bounds__(64) chain2(int levels, const double2 *gwr, unsigned long long *out, double2 *sink) {
    __shared__ double2 lpos[1408];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1408; i += 64) lpos[i] = make_double2(100.0 + 7.0 * (i % 37) + 0.013 * i, 50.0 + 9.0 * (i / 37) + 0.007 * i);
    __syncthreads();
    const double overlap = 2.0, compliance = 36.0, eps = 1e-8;
    const int stride = LAYOUT == 1 ? 64 / ACTIVE : 1;
    const bool active = (lane % stride) == 0 && lane / stride < ACTIVE;
    const int k = lane / stride;  // which pair of the level
    int sa[8], sb[8];
    uint32_t h = 0x9E3779B9u * (uint32_t)(lane + 1) + blockIdx.x;
    for (int u = 0; u < 8; ++u) {
        if (LAYOUT == 2) {
            sa[u] = ((k * 2 + u * 128) % 1280);
            sb[u] = ((k * 2 + 1 + u * 128) % 1280);
        } else {
            h = h * 1664525u + 1013904223u;
            sa[u] = (int)((h >> 8) % 1280u);
            h = h * 1664525u + 1013904223u;
            sb[u] = (int)((h >> 8) % 1280u);
            if (sb[u] == sa[u]) sb[u] = (sb[u] + 1) % 1280;
        }
        if (!active) sa[u] = sb[u] = 0;
    }
    const double2 wa = gwr[lane * 2], wb = gwr[lane * 2 + 1];
    const double2 pc = make_double2(egg_rcp_refined((wa.x + wb.x) + compliance), overlap * (wa.y + wb.y));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < levels; l += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            double2 pa = lpos[sa[u]], pb = lpos[sb[u]];
            const bool store = project_pair_predicated([&]() { return false; }, []() {}, [&](double2 &x, double2 &y) { x = wa; y = wb; }, active, false, pa, pb, wa.x, wb.x, (wa.x + wb.x) + compliance, pc, pc.y * pc.y, overlap, compliance, eps);
            lpos[store ? sa[u] : 1344 + lane] = pa;
            lpos[store ? sb[u] : 1344 + lane] = pb;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = lpos[lane];
}

// The executor as a CONSUMER: everything but the dependent chain comes ready-made from a ring in LDS (what a helper wave
// on another SIMD would leave there): per lane and level the two LDS addresses + flags (8 B) and three 16-byte records
// (inverse masses; reciprocal of the divisor, minimum distance; divisor, minimum distance squared).  The records of
// level l + 1 are read in the shadow of level l's position reads.  CHECK: also compare a progress counter each level
// (scalar compare on a cached copy, as the consumer of a live ring would).
template <int CHECK>
__global__ void __launch_bounds__(64) chain3(int levels, const double2 *gwr, unsigned long long *out, double2 *sink) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    __shared__ v2d lpos[1408];
    __shared__ v2d ringA[8 * 64], ringC[8 * 64], ringD[8 * 64];
    __shared__ v2u ringQ[8 * 64];
    __shared__ uint32_t produced;
    const int lane = threadIdx.x;
    for (int i = lane; i < 1408; i += 64) {
        v2d v;
        v.x = 100.0 + 7.0 * (i % 37) + 0.013 * i;
        v.y = 50.0 + 9.0 * (i / 37) + 0.007 * i;
        lpos[i] = v;
    }
    const double overlap = 2.0, compliance = 36.0, eps = 1e-8;
    uint32_t h = 0x9E3779B9u * (uint32_t)(lane + 1) + blockIdx.x;
    typedef __attribute__((address_space(3))) v2d lds_v2d;
    lds_v2d *const lp = (lds_v2d *)lpos;
    for (int u = 0; u < 8; ++u) {
        h = h * 1664525u + 1013904223u;
        uint32_t a = (h >> 8) % 1280u;
        h = h * 1664525u + 1013904223u;
        uint32_t b = (h >> 8) % 1280u;
        if (b == a) b = (b + 1) % 1280;
        const double2 wa = gwr[a], wb = gwr[b];
        v2d A, C, D;
        A.x = wa.x;
        A.y = wb.x;
        const double div = (wa.x + wb.x) + compliance;
        C.x = egg_rcp_refined(div);
        C.y = overlap * (wa.y + wb.y);
        D.x = div;
        D.y = C.y * C.y;
        v2u Q;
        Q.x = (uint32_t)(uintptr_t)(lp + a) | 1u;  // bit 0: the lane has a pair
        Q.y = (uint32_t)(uintptr_t)(lp + b);
        ringA[u * 64 + lane] = A;
        ringC[u * 64 + lane] = C;
        ringD[u * 64 + lane] = D;
        ringQ[u * 64 + lane] = Q;
    }
    if (lane == 0) produced = 0x7FFFFFFF;
    __syncthreads();
    lds_v2d *const spare = lp + 1344 + lane;
    v2d A = ringA[lane], C = ringC[lane], D = ringD[lane];
    v2u Q = ringQ[lane];
    uint32_t seen = 0;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define PIN(x) __asm__ volatile("" : "+v"(x))
    for (int l = 0; l < levels; l += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            lds_v2d *const qa = (lds_v2d *)(uintptr_t)(Q.x & ~15u), *const qb = (lds_v2d *)(uintptr_t)Q.y;
            const bool has = (Q.x & 1u) != 0, slow = (Q.x & 2u) != 0;
            const v2d va = *qa, vb = *qb;
            __builtin_amdgcn_sched_barrier(0);
            if (CHECK) {
                if (__builtin_expect((int)seen <= l + u + 1, 0)) {
                    do seen = __builtin_amdgcn_readfirstlane(*(volatile uint32_t *)&produced);
                    while ((int)seen <= l + u + 1);
                }
            }
            const v2d A1 = ringA[((u + 1) & 7) * 64 + lane], C1 = ringC[((u + 1) & 7) * 64 + lane], D1 = ringD[((u + 1) & 7) * 64 + lane];
            const v2u Q1 = ringQ[((u + 1) & 7) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
            double2 pa = make_double2(va.x, va.y), pb = make_double2(vb.x, vb.y);
            const bool store = project_pair_predicated([&]() { return false; }, []() {}, [&](double2 &x, double2 &y) { x = make_double2(A.x, 4.0); y = make_double2(A.y, 4.0); }, has, slow, pa, pb, A.x, A.y, D.x,
                                                       make_double2(C.x, C.y), D.y, overlap, compliance, eps);
            v2d oa, ob;
            oa.x = pa.x;
            oa.y = pa.y;
            ob.x = pb.x;
            ob.y = pb.y;
            *(store ? qa : spare) = oa;
            *(store ? qb : spare) = ob;
            __builtin_amdgcn_sched_barrier(0);
            A = A1;
            C = C1;
            D = D1;
            Q = Q1;
        }
    }
#undef PIN
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    if (lane == 0 && blockIdx.x == 0) printf("   (chain3: %llu cycle-counter ticks in %llu ticks of the 100 MHz clock: %.0f MHz)\n", t1 - t0, __builtin_amdgcn_s_memrealtime() - r0, 100.0 * (double)(t1 - t0) / (double)(__builtin_amdgcn_s_memrealtime() - r0));
    const v2d r = lpos[lane];
    sink[blockIdx.x * 64 + lane] = make_double2(r.x, r.y);
}

template <class K>
double run_kernel(K kernel, int blocks, int levels, const double2 *gwr) {
    unsigned long long *d;
    double2 *sink;
    hipMalloc(&d, blocks * 8);
    hipMalloc(&sink, blocks * 64 * 16);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, 0, levels, gwr, d, sink);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    hipFree(sink);
    std::sort(h.begin(), h.end());
    return (double)h[blocks / 2] / levels;
}

template <int MODE>
double run(int blocks, int levels, const double2 *gwr) {
    unsigned long long *d;
    double2 *sink;
    hipMalloc(&d, blocks * 8);
    hipMalloc(&sink, blocks * 64 * 16);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(chain<MODE>, dim3(blocks), dim3(64), 0, 0, levels, gwr, d, sink);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    hipFree(sink);
    std::sort(h.begin(), h.end());
    return (double)h[blocks / 2] / levels;
}

int main() {
    std::vector<double2> wr(1280);
    for (int i = 0; i < 1280; ++i) wr[i] = make_double2(1.0 / (1.0 + 0.8 * ((i * 37) % 100) / 100.0), 4.0);
    double2 *gwr;
    hipMalloc(&gwr, 1280 * 16);
    hipMemcpy(gwr, wr.data(), 1280 * 16, hipMemcpyHostToDevice);
    printf("cycles per level (median over workgroups), one wave per workgroup\n");
    printf("%-60s %10s %10s\n", "variant", "1 wave", "512 waves");
    printf("%-60s %10.1f %10.1f\n", "chain only (constants in registers)", run<0>(1, 2000, gwr), run<0>(512, 2000, gwr));
    printf("%-60s %10.1f %10.1f\n", "chain + 3 x 16 B prepared record from an LDS ring", run<1>(1, 2000, gwr), run<1>(512, 2000, gwr));
    printf("%-60s %10.1f %10.1f\n", "chain + 2 global gathers + reciprocal (unpipelined)", run<2>(1, 2000, gwr), run<2>(512, 2000, gwr));
    printf("%-60s %10.1f %10.1f\n", "chain only, conflict-free LDS slots", run<3>(1, 2000, gwr), run<3>(512, 2000, gwr));
    printf("\nno address arithmetic on the chain (eight precomputed slot pairs per lane):\n");
#define ROW(label, A, L) printf("%-60s %10.1f %10.1f\n", label, run_kernel(chain2<A, L>, 1, 2000, gwr), run_kernel(chain2<A, L>, 512, 2000, gwr))
    ROW("64 pairs, random slots", 64, 0);
    ROW("64 pairs, conflict-free slots", 64, 2);
    ROW("16 pairs in lanes 0-15, random slots", 16, 0);
    ROW("16 pairs in every 4th lane, random slots", 16, 1);
    ROW("16 pairs in lanes 0-15, conflict-free slots", 16, 2);
    ROW("8 pairs in lanes 0-7, random slots", 8, 0);
    ROW("8 pairs in every 8th lane, random slots", 8, 1);
    ROW("32 pairs in lanes 0-31, random slots", 32, 0);
    ROW("32 pairs in every 2nd lane, random slots", 32, 1);
    printf("\nthe executor as the consumer of a ring of ready-made records in LDS:\n");
    printf("%-60s %10.1f %10.1f\n", "4 LDS reads per level, no progress check", run_kernel(chain3<0>, 1, 2000, gwr), run_kernel(chain3<0>, 512, 2000, gwr));
    printf("%-60s %10.1f %10.1f\n", "4 LDS reads per level + progress check", run_kernel(chain3<1>, 1, 2000, gwr), run_kernel(chain3<1>, 512, 2000, gwr));
    return 0;
}
