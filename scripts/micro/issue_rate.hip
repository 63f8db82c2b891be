// issue_rate.hip -- what ONE wave can issue on gfx950, measured with exact instruction streams (inline asm blocks,
// nothing for the compiler to reorder).  The packed executor runs one wave per island with ~110 instructions per
// dependency level: whether a level costs (dependent FP64 latency x chain length) or (instructions x issue cost)
// decides what can shorten it.  Build: hipcc --offload-arch=gfx950 -O2 -o build/issue_rate issue_rate.hip
//
//   ./issue_rate            every mode at 1 wave on the chip, 1 wave per SIMD of one CU, 2 / 4 waves per SIMD of one CU,
//                           and 1 wave per SIMD on every CU
// Output: shader cycles (s_memtime) per instruction, median over the waves of the launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int MODE>
__global__ void probe(int iters, unsigned long long *out, double seed) {
    __shared__ double lds[64 * 16 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double x0 = seed + lane, x1 = seed * 2 + lane, x2 = seed * 3 + lane, x3 = seed * 5 + lane;
    double x4 = seed * 7 + lane, x5 = seed * 11, x6 = seed * 13, x7 = seed * 17;
    const double m = 1.0000001, c = 1e-9;
    float f0 = (float)x0, f1 = (float)x1, f2 = (float)x2, f3 = (float)x3;
    const float fm = 1.0000001f, fc = 1e-9f;
    unsigned u0 = lane, u1 = lane * 3, u2 = lane * 5, u3 = lane * 7;
    double *my = lds + (wave * 64 + lane) * 4;
    my[0] = x0;
    my[1] = x1;
    my[2] = x2;
    my[3] = x3;
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 p0 = {x0, x1}, p1 = {x2, x3};
    unsigned addr = (unsigned)(size_t)my;  // LDS byte address (low 32 bits of the generic pointer are the LDS offset)
    addr = (unsigned)((wave * 64 + lane) * 32);
    ((unsigned *)lds)[(wave * 64 + lane) * 8] = addr;  // pointer chase: every slot points to itself
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // one dependent f64 fma chain
            asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(x0) : "v"(m), "v"(c));
        } else if (MODE == 1) {  // 4 independent chains, interleaved
            asm volatile(REP16("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m), "v"(c));
        } else if (MODE == 2) {  // 8 independent chains
            asm volatile(REP16("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n")
                         REP16("v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(c));
        } else if (MODE == 3) {
            asm volatile(REP64("v_add_f64 %0, %0, %1\n") : "+v"(x0) : "v"(c));
        } else if (MODE == 4) {
            asm volatile(REP64("v_mul_f64 %0, %0, %1\n") : "+v"(x0) : "v"(m));
        } else if (MODE == 5) {
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(f0) : "v"(fm), "v"(fc));
        } else if (MODE == 6) {
            asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fm), "v"(fc));
        } else if (MODE == 7) {
            asm volatile(REP64("v_rsq_f64 %0, %0\n") : "+v"(x0));
        } else if (MODE == 8) {
            asm volatile(REP64("v_rcp_f64 %0, %0\n") : "+v"(x0));
        } else if (MODE == 9) {
            asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(u0) : "v"(u1));
        } else if (MODE == 10) {
            asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(lane));
        } else if (MODE == 11) {
            asm volatile(REP64("s_nop 0\n"));
        } else if (MODE == 12) {  // LDS store -> load of the same word, dependent (64 round trips)
            asm volatile(REP64("ds_write_b64 %1, %0\n ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(x0) : "v"(addr) : "memory");
        } else if (MODE == 13) {  // pointer chase: dependent ds_read_b32
            asm volatile(REP64("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(addr) : : "memory");
        } else if (MODE == 14) {  // the executor's level shape: two 16-byte reads, one fma on them, two 16-byte writes
            asm volatile(REP16("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)\n"
                               "ds_write_b128 %2, %0\n ds_write_b128 %2, %1 offset:16\n")
                         : "+v"(p0), "+v"(p1) : "v"(addr), "v"(m), "v"(c) : "memory");
        } else if (MODE == 15) {  // one dependent f64 chain with an independent integer add after every link
            asm volatile(REP64("v_fma_f64 %0, %0, %2, %3\n v_add_u32 %1, %1, %4\n") : "+v"(x0), "+v"(u0) : "v"(m), "v"(c), "v"(lane));
        } else if (MODE == 16) {  // one dependent f64 chain with THREE independent integer adds after every link
            asm volatile(REP64("v_fma_f64 %0, %0, %4, %5\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %6\n")
                         : "+v"(x0), "+v"(u0), "+v"(u1), "+v"(u2) : "v"(m), "v"(c), "v"(lane));
        } else if (MODE == 17) {  // compare + select on the chain
            asm volatile(REP64("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n") : "+v"(x0), "+v"(x1), "+v"(u0) : "v"(u1) : "vcc");
        } else if (MODE == 18) {  // two dependent chains f64: does a second chain hide in the first one's latency?
            asm volatile(REP64("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n") : "+v"(x0), "+v"(x1) : "v"(m), "v"(c));
        } else if (MODE == 19) {  // scalar ALU
            asm volatile(REP64("s_add_u32 s20, s20, 1\n") : : : "s20", "scc");
        } else if (MODE == 21) {  // LDS atomic add with return, distinct addresses, one at a time (latency)
            asm volatile(REP16("ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)\n") : "=&v"(u1) : "v"(addr), "v"(u2) : "memory");
        } else if (MODE == 22) {  // ... sixteen back to back (throughput)
            asm volatile(REP16("ds_add_rtn_u32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(u1) : "v"(addr), "v"(u2) : "memory");
        } else if (MODE == 23) {  // LDS atomic add without return
            asm volatile(REP16("ds_add_u32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(u2) : "memory");
        } else if (MODE == 24) {  // with return, all 64 lanes on ONE address
            asm volatile(REP16("ds_add_rtn_u32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(u1) : "v"(addr & 0u), "v"(u2) : "memory");
        } else if (MODE == 25) {  // with return, 8 lanes active
            if (lane < 8) asm volatile(REP16("ds_add_rtn_u32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(u1) : "v"(addr), "v"(u2) : "memory");
        } else if (MODE == 26) {  // ds_bpermute_b32
            asm volatile(REP16("ds_bpermute_b32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(u1) : "v"(addr & 0xFCu), "v"(u2) : "memory");
        } else if (MODE == 27) {  // plain ds_write_b32 for comparison
            asm volatile(REP16("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(u2) : "memory");
        } else if (MODE == 28) {  // plain ds_read_b32 for comparison
            asm volatile(REP16("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(u1) : "v"(addr) : "memory");
        } else if (MODE == 20) {  // LDS read issue rate, independent (no wait until the end of the block)
            asm volatile(REP16("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n") "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(p0), "=v"(p1) : "v"(addr) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    if (p0.x + p1.y + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f0 + f1 + f2 + f3 == 12345.678 && u0 + u1 + u2 + u3 + addr == 77) out[0] = 0;
}

struct Mode {
    int id;
    const char *name;
    int instrs;  // per block
};

template <int MODE>
double run(int blocks, int threads, int iters, int instrs) {
    const int waves = blocks * threads / 64;
    unsigned long long *d = nullptr;
    (void)hipMalloc(&d, waves * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, iters, d, 1.5);
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    std::sort(h.begin(), h.end());
    return (double)h[waves / 2] / ((double)iters * instrs);
}

#define ROW(M, NAME, N)                                                                                          \
    printf("%-58s %8.2f %8.2f %8.2f %8.2f %8.2f\n", NAME, run<M>(1, 64, 2000, N), run<M>(1, 256, 2000, N),          \
           run<M>(1, 512, 2000, N), run<M>(1, 1024, 2000, N), run<M>(256, 256, 2000, N));

int main() {
    printf("cycles per instruction (median over waves)\n%-58s %8s %8s %8s %8s %8s\n", "stream", "1 wave", "1/SIMD", "2/SIMD", "4/SIMD",
           "1/SIMD,256CU");
    ROW(0, "v_fma_f64, one dependent chain", 64)
    ROW(18, "v_fma_f64, 2 chains", 128)
    ROW(1, "v_fma_f64, 4 chains", 64)
    ROW(2, "v_fma_f64, 8 chains", 128)
    ROW(3, "v_add_f64 dependent", 64)
    ROW(4, "v_mul_f64 dependent", 64)
    ROW(5, "v_fma_f32 dependent", 64)
    ROW(6, "v_fma_f32, 4 chains", 64)
    ROW(7, "v_rsq_f64 dependent", 64)
    ROW(8, "v_rcp_f64 dependent", 64)
    ROW(9, "v_add_u32 dependent", 64)
    ROW(10, "v_add_u32, 4 chains", 64)
    ROW(11, "s_nop 0", 64)
    ROW(19, "s_add_u32 dependent", 64)
    ROW(15, "v_fma_f64 dependent + 1 independent v_add_u32 (per pair)", 64)
    ROW(16, "v_fma_f64 dependent + 3 independent v_add_u32 (per group)", 64)
    ROW(17, "v_cmp_lt_f64 + v_cndmask (per pair)", 64)
    ROW(12, "ds_write_b64 -> ds_read_b64 round trip", 64)
    ROW(13, "ds_read_b32 pointer chase", 64)
    ROW(14, "2 ds_read_b128, wait, 2 ds_write_b128 of them (per level)", 16)
    ROW(20, "ds_read_b128 independent (per read)", 32)
    ROW(28, "ds_read_b32 x16 back to back (per read)", 16)
    ROW(27, "ds_write_b32 x16 back to back (per write)", 16)
    ROW(21, "ds_add_rtn_u32, distinct addresses, waited (latency)", 16)
    ROW(22, "ds_add_rtn_u32 x16 back to back (per atomic)", 16)
    ROW(23, "ds_add_u32 (no return) x16 back to back", 16)
    ROW(24, "ds_add_rtn_u32 x16, 64 lanes on one address", 16)
    ROW(25, "ds_add_rtn_u32 x16, 8 lanes active", 16)
    ROW(26, "ds_bpermute_b32 x16 back to back", 16)
    return 0;
}
