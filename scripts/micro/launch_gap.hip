// Developer microbenchmark: what does one more dependent kernel in a stream cost on this GPU?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void spin(int *p, long long ticks) {
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
int main() {
    int *d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {1, 512, 4096}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a, s);
            for (int k = 0; k < 2000; ++k) hipLaunchKernelGGL(tiny, dim3(grid), dim3(64), 0, s, d);
            hipEventRecord(b, s); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("tiny kernel, grid %d: %.2f us per dependent launch\n", grid, 1e3 * ms / 2000);
        }
    }
    {   // the same chain as a graph (captured once, launched repeatedly)
        hipGraph_t graph; hipGraphExec_t exec;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(tiny, dim3(512), dim3(64), 0, s, d);
        hipStreamEndCapture(s, &graph);
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a, s);
            for (int k = 0; k < 10; ++k) hipGraphLaunch(exec, s);
            hipEventRecord(b, s); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("graph of 200 dependent tiny kernels, grid 512: %.2f us per kernel\n", 1e3 * ms / 2000);
        }
    }
    for (long long ticks : {2000LL, 10000LL}) {   // 100 MHz counter: 20 us / 100 us of work
        hipEventRecord(a, s);
        for (int k = 0; k < 500; ++k) hipLaunchKernelGGL(spin, dim3(512), dim3(64), 0, s, d, ticks);
        hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("kernel spinning %lld ticks of the 100 MHz counter (%.0f us), grid 512: %.2f us per launch\n", ticks, ticks / 100.0, 1e3 * ms / 500);
    }
    return 0;
}
