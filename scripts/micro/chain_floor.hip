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
        const bool store = project_pair_predicated([&]() { return false; }, true, false, pa, pb, wa, wb, pc, overlap, compliance, eps);
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
    return 0;
}
