// stream_flag_probe.hip -- can a RUNNING kernel be told that a host->device copy has landed, and does it then read the
// copied bytes?  (Round 5: the persistent chunk-streaming launch of the host pipeline rests on this.)
//   hipcc --offload-arch=gfx950 -O2 -o stream_flag_probe stream_flag_probe.hip && ./stream_flag_probe
// A chip-filling kernel (512 workgroups x 64 KiB LDS) spins -- bounded by the wall clock, never forever -- on a flag
// word; a second stream copies 64 MiB of new data over a buffer the previous kernel left in the caches, then
// raises the flag; the kernel then reads the buffer and counts stale values.  Varied: where the flag lives (ordinary
// device memory / fine-grained device memory / page-locked host memory), who raises it (a 4-byte copy on the copy
// stream = the copy engine, hipStreamWriteValue32, the host CPU), the scope of the polling load, the fence behind it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill(float *d, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = v;
}
__global__ void reader(const float *d, size_t n, float *sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += d[i];
    if (acc == 123.456f) *sink = acc;
}

struct Out { unsigned long long wait_ticks_max, mismatches, timeouts, seen; };

// load_scope: 0 agent, 1 system.  fence: 0 none, 1 agent acquire, 2 system acquire.  lds_dma: read through global_load_lds
__global__ __launch_bounds__(256) void waiter(unsigned *flag, unsigned want, const float *data, size_t n, float expect,
                                              int load_scope, int fence, unsigned long long limit_ticks, Out *out) {
    extern __shared__ float smem[];
    __shared__ int ok;
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        int seen = 0;
        for (;;) {
            const unsigned v = load_scope ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                          : __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == want) { seen = 1; break; }
            if (wall_clock64() - t0 > limit_ticks) break;
            __builtin_amdgcn_s_sleep(20);
        }
        ok = seen;
        const unsigned long long dt = wall_clock64() - t0;
        atomicMax(&out->wait_ticks_max, dt);
        if (seen) atomicAdd(&out->seen, 1ull); else atomicAdd(&out->timeouts, 1ull);
    }
    __syncthreads();
    if (!ok) return;
    if (fence == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (fence == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        bad += data[i] != expect;
    smem[threadIdx.x] = (float)bad;
    if (bad) atomicAdd(&out->mismatches, bad);
}

int main() {
    CK(hipSetDevice(0));
    const size_t n = 16u << 20;                       // 64 MiB of floats
    float *data, *sink, *hnew;
    CK(hipMalloc(&data, n * 4));
    CK(hipMalloc(&sink, 4));
    CK(hipHostMalloc(&hnew, n * 4, hipHostMallocPortable));
    unsigned *f_coarse, *f_fine = nullptr, *f_host, *h_word;
    CK(hipMalloc(&f_coarse, 64));
    if (hipExtMallocWithFlags((void **)&f_fine, 64, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); f_fine = nullptr; }
    CK(hipHostMalloc(&f_host, 64, hipHostMallocPortable | hipHostMallocCoherent));
    CK(hipHostMalloc(&h_word, 64, hipHostMallocPortable));
    Out *out;
    CK(hipHostMalloc(&out, sizeof(Out), hipHostMallocPortable));
    hipStream_t s_run, s_cp;
    CK(hipStreamCreateWithFlags(&s_run, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_cp, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void *)waiter, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    printf("fine-grained device allocation: %s\n", f_fine ? "ok" : "UNAVAILABLE");
    const unsigned long long limit = 30000000ull;     // 300 ms at 100 MHz
    unsigned gen = 100;
    const char *fk_name[] = {"device coarse", "device fine-grained", "host pinned coherent"};
    const char *raise_name[] = {"4-byte H2D copy on the copy stream", "hipStreamWriteValue32 on the copy stream", "host CPU store after hipStreamSynchronize(copy)"};
    for (int fk = 0; fk < 3; ++fk) {
        unsigned *flag = fk == 0 ? f_coarse : fk == 1 ? f_fine : f_host;
        if (!flag) continue;
        for (int raise = 0; raise < 3; ++raise) {
            if (raise == 2 && fk != 2) continue;      // the CPU can only store to host memory
            if (raise == 0 && fk == 2) continue;      // (host -> host "copy": not a copy-engine job)
            for (int scope = 0; scope < 2; ++scope)
                for (int fence = 0; fence < 3; ++fence) {
                    ++gen;
                    const float oldv = (float)gen, newv = (float)gen + 0.5f;
                    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, s_run, data, n, oldv);
                    hipLaunchKernelGGL(reader, dim3(2048), dim3(256), 0, s_run, data, n, sink);   // old values into the caches
                    CK(hipStreamSynchronize(s_run));
                    for (size_t i = 0; i < n; ++i) hnew[i] = newv;
                    memset(out, 0, sizeof(Out));
                    *h_word = gen;
                    if (fk == 2) *flag = 0; else CK(hipMemset(flag, 0, 4));
                    CK(hipDeviceSynchronize());
                    hipLaunchKernelGGL(waiter, dim3(512), dim3(256), 64 * 1024, s_run, flag, gen, data, n, newv, scope, fence, limit, out);
                    std::this_thread::sleep_for(std::chrono::milliseconds(3));     // the kernel is running and polling
                    const auto t0 = std::chrono::steady_clock::now();
                    CK(hipMemcpyAsync(data, hnew, n * 4, hipMemcpyHostToDevice, s_cp));
                    hipError_t er = hipSuccess;
                    if (raise == 0) er = hipMemcpyAsync(flag, h_word, 4, hipMemcpyHostToDevice, s_cp);
                    else if (raise == 1) er = hipStreamWriteValue32(s_cp, flag, gen, 0);
                    else { CK(hipStreamSynchronize(s_cp)); __atomic_store_n(flag, gen, __ATOMIC_RELEASE); }
                    const double enq_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    if (er != hipSuccess) { printf("flag %-22s raise %-48s -> API error %s\n", fk_name[fk], raise_name[raise], hipGetErrorString(er)); (void)hipGetLastError(); CK(hipDeviceSynchronize()); continue; }
                    CK(hipStreamSynchronize(s_cp));
                    const double cp_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    CK(hipStreamSynchronize(s_run));
                    printf("flag %-22s raise %-48s load %-6s fence %-6s : seen %3llu timeouts %3llu stale values %10llu  longest wait %7.2f ms (copy done after %.2f ms, enqueue took %.2f)\n",
                           fk_name[fk], raise_name[raise], scope ? "system" : "agent", fence == 0 ? "none" : fence == 1 ? "agent" : "system",
                           out->seen, out->timeouts, out->mismatches, out->wait_ticks_max / 1e5, cp_ms, enq_ms);
                    fflush(stdout);
                }
        }
    }
    return 0;
}
