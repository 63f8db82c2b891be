// lds_atomic_order.hip -- in which order does ONE ds_add_rtn_u32 wave-instruction serve lanes that hit the same LDS
// address?  If it is ascending lane order, the returned values of a single atomic instruction are an exact
// "number of earlier lanes with my key" (a multi-prefix), which the packed pipeline's level pass could use to rank
// the entries of a pair stream without a serial walk.  Not an architectural guarantee: this only measures.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const unsigned *keys, int trials, int nkeys, unsigned long long *bad, unsigned *first_bad) {
    __shared__ unsigned cnt[1024];
    const int lane = threadIdx.x;
    unsigned long long mism = 0;
    for (int t = 0; t < trials; ++t) {
        for (int k = lane; k < nkeys; k += 64) cnt[k] = 0;
        __syncthreads();
        const unsigned key = keys[(size_t)(blockIdx.x * trials + t) * 64 + lane];
        const unsigned got = atomicAdd(&cnt[key], 1u);
        // exact answer: earlier lanes with the same key
        unsigned want = 0;
        for (int j = 0; j < 64; ++j) {
            const unsigned kj = __shfl(key, j, 64);
            want += (j < lane && kj == key) ? 1u : 0u;
        }
        if (got != want) {
            ++mism;
            if (atomicCAS(&first_bad[0], 0u, 1u) == 0u) {
                first_bad[1] = lane;
                first_bad[2] = key;
                first_bad[3] = got;
                first_bad[4] = want;
            }
        }
        __syncthreads();
    }
    for (int d = 32; d >= 1; d >>= 1) mism += __shfl_xor(mism, d, 64);
    if (lane == 0) atomicAdd(bad, mism);
}

int main() {
    const int blocks = 1024, trials = 256;
    for (int nkeys : {1, 2, 3, 5, 8, 16, 32, 48, 64, 200, 1000}) {
        std::vector<unsigned> h((size_t)blocks * trials * 64);
        unsigned s = 12345u + nkeys;
        for (auto &v : h) {
            s = s * 1664525u + 1013904223u;
            v = (s >> 8) % nkeys;
        }
        unsigned *dk;
        unsigned long long *dbad;
        unsigned *dfirst;
        hipMalloc(&dk, h.size() * 4);
        hipMalloc(&dbad, 8);
        hipMalloc(&dfirst, 32);
        hipMemcpy(dk, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemset(dbad, 0, 8);
        hipMemset(dfirst, 0, 32);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, dk, trials, nkeys, dbad, dfirst);
        unsigned long long bad = 0;
        unsigned fb[8];
        hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
        hipMemcpy(fb, dfirst, 32, hipMemcpyDeviceToHost);
        printf("keys drawn from %4d values: %llu of %llu lane results differ from ascending-lane order", nkeys, bad,
               (unsigned long long)blocks * trials * 64);
        if (bad) printf("  (first: lane %u key %u got %u want %u)", fb[1], fb[2], fb[3], fb[4]);
        printf("\n");
        hipFree(dk);
        hipFree(dbad);
        hipFree(dfirst);
    }
    return 0;
}
