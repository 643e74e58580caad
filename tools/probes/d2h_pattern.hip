// Follow-up of d2h_probe: the probe's device->host copies were ALL the copy engine's (MEMORY_COPY_DEVICE_TO_HOST, 55 GB/s, true
// overlap beside a chip-filling kernel); the boundary's are __amd_rocclr_copyBuffer shader launches.  What differs?  One pattern per
// process (count kernels vs memory copies with rocprofv3 --kernel-trace --memory-copy-trace --stats):
//   d2h_pattern <pattern> [bytes]
//   0 = copy on a stream of its own, nothing else                       3 = as 1, copy stream has high priority
//   1 = kernel on stream A, event, copy stream waits for the event      4 = as 1, the event was created with hipEventDisableTiming
//   2 = kernel on the copy's own stream                                 5 = as 4 + high priority + 8 copies back to back (the pipeline's shape)
//   6 = as 5, but the destination is an offset inside a big hipHostMalloc block and the source an offset inside a big hipMalloc block
//   7 = as 5, destination from hipHostMalloc in ANOTHER thread          8 = as 5 with a second device->host stream active at the same time
//   9 = as 4, but the producer is LONG (~2 ms): the copy is enqueued while the kernel it waits for is still running (the pipeline
//       enqueues everything up front)                                  10 = as 9, the host waits for the event and THEN enqueues the copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void produce_slow(float *p, size_t n, int iters) {
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)i + (a == 1.5f ? 1.f : 0.f);
}
__global__ void produce(float *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)i;
}
int main(int argc, char **argv) {
    const int pat = argc > 1 ? atoi(argv[1]) : 0;
    const size_t bytes = argc > 2 ? (size_t)atol(argv[2]) : (size_t)4 << 20;
    const int reps = pat >= 5 ? 8 : 4;
    char *d = nullptr, *h = nullptr;
    const size_t big = bytes * 10;
    CK(hipMalloc((void **)&d, big));
    if (pat == 7) {
        std::thread t([&] { (void)hipSetDevice(0); (void)hipHostMalloc((void **)&h, big, hipHostMallocPortable); });
        t.join();
        if (!h) return 2;
    } else {
        CK(hipHostMalloc((void **)&h, big, hipHostMallocPortable));
    }
    memset(h, 0, big);
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t a, c, c2;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    if (pat == 3 || pat >= 5) CK(hipStreamCreateWithPriority(&c, hipStreamNonBlocking, hi));
    else CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&c2, hipStreamNonBlocking, hi));
    hipEvent_t ev[8];
    for (auto &e : ev) {
        if (pat >= 4) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        else CK(hipEventCreate(&e));
    }
    CK(hipDeviceSynchronize());
    for (int round = 0; round < 2; ++round) {
        double t0 = now();
        for (int r = 0; r < reps; ++r) {
            char *src = d + (pat == 6 ? (size_t)(r + 1) * bytes + 4096 : 0);
            char *dst = h + (pat == 6 ? (size_t)(r + 1) * bytes + 8192 : (size_t)(pat >= 5 ? r : 0) * bytes);
            if (pat == 9 || pat == 10) {
                hipLaunchKernelGGL(produce_slow, dim3(256), dim3(256), 0, a, (float *)src, bytes / 4, 400000);
                CK(hipEventRecord(ev[r], a));
                if (pat == 9) CK(hipStreamWaitEvent(c, ev[r], 0));
                else CK(hipEventSynchronize(ev[r]));
            } else if (pat == 1 || pat >= 3) {
                hipLaunchKernelGGL(produce, dim3(256), dim3(256), 0, a, (float *)src, bytes / 4);
                CK(hipEventRecord(ev[r], a));
                CK(hipStreamWaitEvent(c, ev[r], 0));
            } else if (pat == 2) {
                hipLaunchKernelGGL(produce, dim3(256), dim3(256), 0, c, (float *)src, bytes / 4);
            }
            CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c));
            if (pat == 8) CK(hipMemcpyAsync(h + 9 * bytes, d + 9 * bytes, bytes, hipMemcpyDeviceToHost, c2));
            if (pat < 5) CK(hipStreamSynchronize(c));
        }
        CK(hipDeviceSynchronize());
        double t1 = now();
        if (round) printf("pattern %d: %d copies of %zu bytes in %.3f ms (%.1f GB/s incl. the producers)\n", pat, reps, bytes, t1 - t0,
                          reps * bytes / (t1 - t0) / 1e6);
    }
    return 0;
}
