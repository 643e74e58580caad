// How fast does host->device go from (a) fresh malloc'd pages, (b) the same pages again,
// (c) hipHostRegister'd pages, (d) hipHostMalloc'd pages?  134 MB like the headline K/V.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 134217728;
    void *d; hipMalloc(&d, bytes);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 2; ++rep) {
        char *h = (char *)malloc(bytes); memset(h, 1, bytes);
        double t0 = now(); hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t1 = now();
        hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t2 = now();
        double r0 = now(); hipHostRegister(h, bytes, hipHostRegisterDefault); double r1 = now();
        hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double r2 = now();
        hipHostUnregister(h); double r3 = now();
        printf("malloc: first copy %.2f ms (%.1f GB/s), second %.2f ms (%.1f GB/s); register %.2f ms, copy %.2f ms (%.1f GB/s), unregister %.2f ms\n",
               t1 - t0, bytes / (t1 - t0) / 1e6, t2 - t1, bytes / (t2 - t1) / 1e6, r1 - r0, r2 - r1, bytes / (r2 - r1) / 1e6, r3 - r2);
        free(h);
        void *p; double a0 = now(); hipHostMalloc(&p, bytes, hipHostMallocDefault); double a1 = now(); memset(p, 2, bytes); double a2 = now();
        hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double a3 = now();
        hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double a4 = now();
        hipHostFree(p);
        printf("hipHostMalloc %.2f ms, memset %.2f ms, H2D %.2f ms (%.1f GB/s), D2H %.2f ms (%.1f GB/s)\n",
               a1 - a0, a2 - a1, a3 - a2, bytes / (a3 - a2) / 1e6, a4 - a3, bytes / (a4 - a3) / 1e6);
    }
    return 0;
}
