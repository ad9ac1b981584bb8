// tools/host_copy_probe.cpp -- what the host link gives a drop-in that is handed PAGEABLE buffers (development tool, not product):
// pageable hipMemcpy both ways, pinned hipMemcpy, hipHostRegister of the caller's buffer, and host memcpy into pinned staging
// with 1 .. 16 threads.  build: hipcc -O2 -o /tmp/host_copy_probe tools/host_copy_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = (size_t)1 << 30;
    char* pageable = (char*)malloc(n); memset(pageable, 1, n);
    char* pageable2 = (char*)malloc(n); memset(pageable2, 2, n);
    char* pinned = nullptr; hipHostMalloc((void**)&pinned, n, hipHostMallocDefault); memset(pinned, 3, n);
    void* dev = nullptr; hipMalloc(&dev, n);
    hipMemcpy(dev, pinned, n, hipMemcpyHostToDevice);
    auto t = [&](const char* what, auto f) { f(); double t0 = now(); f(); f(); double dt = (now() - t0) / 2; printf("%-44s %7.1f ms  %6.1f GB/s\n", what, dt * 1e3, n / dt / 1e9); };
    t("hipMemcpy H2D pageable", [&] { hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice); });
    t("hipMemcpy D2H pageable", [&] { hipMemcpy(pageable2, dev, n, hipMemcpyDeviceToHost); });
    t("hipMemcpy H2D pinned", [&] { hipMemcpy(dev, pinned, n, hipMemcpyHostToDevice); });
    t("hipMemcpy D2H pinned", [&] { hipMemcpy(pinned, dev, n, hipMemcpyDeviceToHost); });
    {
        double t0 = now(); hipError_t e = hipHostRegister(pageable, n, hipHostRegisterDefault); double t1 = now();
        printf("%-44s %7.1f ms  (%s)\n", "hipHostRegister 1 GiB", (t1 - t0) * 1e3, hipGetErrorString(e));
        if (e == hipSuccess) {
            t("hipMemcpy H2D registered", [&] { hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice); });
            t0 = now(); hipHostUnregister(pageable); printf("%-44s %7.1f ms\n", "hipHostUnregister", (now() - t0) * 1e3);
        }
    }
    for (int th : {1, 2, 4, 8, 16}) {
        char what[64]; snprintf(what, sizeof(what), "host memcpy pageable -> pinned, %d threads", th);
        t(what, [&] {
            std::vector<std::thread> v;
            for (int i = 0; i < th; ++i) v.emplace_back([&, i] { const size_t a = n / th * i, b = i + 1 == th ? n : n / th * (i + 1); memcpy(pinned + a, pageable + a, b - a); });
            for (auto& x : v) x.join();
        });
    }
    // slabs through two pinned buffers: memcpy (4 threads) of slab k+1 while slab k is on the wire
    {
        const size_t slab = (size_t)64 << 20; const int th = 4;
        hipStream_t s; hipStreamCreate(&s);
        char* st[2]; hipHostMalloc((void**)&st[0], slab, 0); hipHostMalloc((void**)&st[1], slab, 0);
        hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
        auto run = [&] {
            for (size_t k = 0; k * slab < n; ++k) {
                hipEventSynchronize(ev[k & 1]);
                std::vector<std::thread> v;
                for (int i = 0; i < th; ++i) v.emplace_back([&, i] { const size_t a = slab / th * i; memcpy(st[k & 1] + a, pageable + k * slab + a, slab / th); });
                for (auto& x : v) x.join();
                hipMemcpyAsync((char*)dev + k * slab, st[k & 1], slab, hipMemcpyHostToDevice, s);
                hipEventRecord(ev[k & 1], s);
            }
            hipStreamSynchronize(s);
        };
        t("H2D pageable via 2 pinned slabs of 64 MiB, 4 thr", run);
    }
    return 0;
}
