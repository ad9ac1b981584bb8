// tools/stream_overlap_probe.cpp -- do small kernels on different non-blocking streams run side by side on this platform, and
// does a pageable hipMemcpyAsync + sync on another stream get in their way?  (development tool)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(unsigned long long ticks, int* sink) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } if (ticks == 1) *sink = 1; }
int main()
{
    const int NS = 16;
    hipStream_t st[NS], cp; for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&cp, hipStreamNonBlocking);
    int* sink; hipMalloc(&sink, 4);
    const unsigned long long ticks = 500000;            // 100 MHz wall clock: 5 ms
    hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st[0], ticks, sink); hipDeviceSynchronize();
    for (int n : {1, 4, 8, 16}) {
        const double t0 = now();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st[i], ticks, sink);
        hipDeviceSynchronize();
        printf("%2d kernels of 5 ms on %2d streams: %.1f ms\n", n, n, now() - t0);
    }
    const size_t bytes = (size_t)32 << 20; char* h = (char*)malloc(bytes * 8); memset(h, 1, bytes * 8); void* d; hipMalloc(&d, bytes * 8);
    {
        const double t0 = now();
        for (int i = 0; i < 8; ++i) {
            hipMemcpyAsync((char*)d + i * bytes, h + i * bytes, bytes, hipMemcpyHostToDevice, cp); hipStreamSynchronize(cp);
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st[i], ticks, sink);
        }
        const double t1 = now();
        hipDeviceSynchronize();
        printf("8 x (32 MiB pageable copy + 5 ms kernel on its own stream): copies+launches %.1f ms, then %.1f ms more\n", t1 - t0, now() - t1);
    }
    return 0;
}
