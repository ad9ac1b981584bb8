// tools/microbench/salu_latency.hip -- what a wave's serial scalar code costs on gfx950 (development probe, not part of the product):
// cycles (s_memtime) per instruction of (a) a dependent s_add chain, (b) a readlane -> scalar -> readlane chain as the parsers'
// chain walks run it, (c) a loop of one s_sub + one TAKEN s_cbranch, (d) dependent VALU adds, (e) dependent ds_read chain,
// each alone on a CU and with `waves` wavefronts per CU running the same code.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/salu_latency.hip -o /tmp/salu && /tmp/salu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__global__ void k_salu(unsigned long long* out, int iters)
{
    unsigned s = (unsigned)iters;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) asm volatile(REP64("s_add_u32 %0, %0, 1\n\t") : "+s"(s) :: "scc");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = s; }
}
__global__ void k_readlane(unsigned long long* out, int iters)
{
    unsigned v = (threadIdx.x + 1) & 63, s = 0, t;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_readlane_b32 %1, %2, %0\n\ts_and_b32 %0, %1, 63\n\ts_nop 0\n\ts_nop 0\n\t") : "+s"(s), "=&s"(t) : "v"(v) : "scc");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = s; }
}
__global__ void k_branch(unsigned long long* out, int iters)
{
    unsigned s = (unsigned)iters * 64;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("1:\n\ts_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1b\n\t" : "+s"(s) :: "scc");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = s; }
}
__global__ void k_valu(unsigned long long* out, int iters)
{
    unsigned v = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) asm volatile(REP64("v_add_u32 %0, %0, 1\n\t") : "+v"(v));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (v == 0x12345) out[0] = 1;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = v; }
}
__global__ void k_lds(unsigned long long* out, int iters)
{
    __shared__ unsigned tab[1024];
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)tab;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = base + ((i * 4 + 68) & 4092);
    __syncthreads();
    unsigned a = base + (threadIdx.x & 63) * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) asm volatile(REP16("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(a) :: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = a + tab[0]; }
}
__global__ void k_mixed(unsigned long long* out, int iters)   // the walk's shape: readlane, cmp, not-taken branch, bitset
{
    unsigned v = 1, cur = 0, t; unsigned long long m = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_readlane_b32 %2, %3, %1\n\ts_cmp_lt_u32 %2, 64\n\ts_cbranch_scc0 9f\n\ts_bitset1_b64 %0, %1\n\ts_nop 0\n\t") "9:\n\t" : "+s"(m), "+s"(cur), "=&s"(t) : "v"(v) : "scc");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = m; }
}

template <class K> void run(const char* name, K k, int per_iter, int cus)
{
    unsigned long long* d; hipMalloc(&d, 16 * 65536);
    const int iters = 200;
    for (int waves : {1, 4, 8, 16, 24, 32}) {
        const int wg = waves <= 16 ? 64 : 64; const int blocks = cus * waves;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(wg), 0, 0, d, iters); hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(blocks), dim3(wg), 0, 0, d, iters); hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * blocks); hipMemcpy(h.data(), d, 16 * blocks, hipMemcpyDeviceToHost);
        double s = 0; for (int b = 0; b < blocks; ++b) s += (double)h[2 * b];
        printf("%-10s %2d waves/CU: %7.2f ticks per instruction (%d instructions per trip)\n", name, waves, s / blocks / ((double)iters * per_iter), per_iter); fflush(stdout);
    }
    hipFree(d);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clock %d kHz; s_memtime ticks\n", p.gcnArchName, p.multiProcessorCount, p.clockRate); fflush(stdout);
    {   // tick rate: s_memtime against the wall clock
        unsigned long long* d; hipMalloc(&d, 64);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL(k_salu, dim3(1), dim3(64), 0, 0, d, 20000); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("s_memtime: %.1f ticks per microsecond\n", (double)h[0] / (ms * 1e3)); fflush(stdout);
    }
    const int cus = p.multiProcessorCount;
    run("salu", k_salu, 64, cus); run("readlane", k_readlane, 64, cus); run("branch", k_branch, 192, cus);
    run("valu", k_valu, 64, cus); run("lds", k_lds, 32, cus); run("walk4", k_mixed, 80, cus);
    return 0;
}
