// stream_flag_probe2.hip -- the first probe's question again, under the REAL conditions of the streamed launch (round 5, after
// it timed out in the engine): the polling kernel now owns EVERY register of every SIMD (256 VGPRs x 2 waves, 64 KiB LDS
// x 2 workgroups per CU: no other wave can become resident), several streams exist, and events are recorded between
// the copies the way the host pipeline does.  Which way of raising a ready word still works?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Out { unsigned long long wait_ticks_max, mismatches, timeouts, seen; };

// 2 workgroups of 4 waves per CU, each wave with (nearly) 256 VGPRs live across the wait: nothing else fits on a SIMD
__global__ __launch_bounds__(256, 2) void hog_waiter(unsigned *flag, unsigned want, const float *data, size_t n, float expect,
                                                      unsigned long long limit_ticks, Out *out, float *sink) {
    extern __shared__ float smem[];
    float r[250];
#pragma unroll
    for (int i = 0; i < 250; ++i) r[i] = (float)(threadIdx.x * 3 + i) * 0.5f + (float)blockIdx.x;
#pragma unroll
    for (int i = 0; i < 250; ++i) asm volatile("" : "+v"(r[i]));
    __shared__ int ok;
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        int seen = 0;
        for (;;) {
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == want) { seen = 1; break; }
            if (wall_clock64() - t0 > limit_ticks) break;
            __builtin_amdgcn_s_sleep(20);
        }
        ok = seen;
        atomicMax(&out->wait_ticks_max, wall_clock64() - t0);
        if (seen) atomicAdd(&out->seen, 1ull); else atomicAdd(&out->timeouts, 1ull);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 250; ++i) asm volatile("" : "+v"(r[i]));
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 250; ++i) acc += r[i];
    if (acc == 12345.678f) *sink = acc;
    smem[threadIdx.x] = acc;
    if (!ok) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad += data[i] != expect;
    if (bad) atomicAdd(&out->mismatches, bad);
}
__global__ void fill(float *d, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = v;
}

int main() {
    CK(hipSetDevice(0));
    const size_t n = 4u << 20;                        // 16 MiB of floats
    float *data, *sink, *hnew;
    CK(hipMalloc(&data, n * 4));
    CK(hipMalloc(&sink, 4));
    CK(hipHostMalloc(&hnew, n * 4, hipHostMallocPortable));
    unsigned *f_fine, *f_coarse, *f_host, *h_word, *h_words;
    if (hipExtMallocWithFlags((void **)&f_fine, 1 << 16, hipDeviceMallocFinegrained) != hipSuccess) { printf("no fine-grained memory\n"); return 1; }
    CK(hipMalloc(&f_coarse, 1 << 16));
    CK(hipHostMalloc(&f_host, 64, hipHostMallocPortable | hipHostMallocCoherent));
    CK(hipHostMalloc(&h_word, 64, hipHostMallocPortable));
    CK(hipHostMalloc(&h_words, 1 << 16, hipHostMallocPortable));
    Out *out;
    CK(hipHostMalloc(&out, sizeof(Out), hipHostMallocPortable));
    hipStream_t s_run, s_cp, s_in, s_out, s_comm;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&s_cp, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_in, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&s_run, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_out, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&s_comm, hipStreamNonBlocking, hi));
    hipEvent_t ev[4], tev[2];
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : tev) CK(hipEventCreate(&e));
    CK(hipFuncSetAttribute((const void *)hog_waiter, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void *)hog_waiter));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, hog_waiter, 256, 64 * 1024));
    printf("hog_waiter: %d registers per lane, %zu B static LDS, occupancy %d workgroups per CU\n", fa.numRegs, fa.sharedSizeBytes, occ);
    const unsigned long long limit = 30000000ull;     // 300 ms at 100 MHz
    unsigned gen = 500;
    const char *names[] = {"4-byte H2D copy, fine-grained device word", "4-byte H2D copy, coarse device word",
                           "the same behind hipEventRecord + a stream-wait on another stream (as the host pipeline)",
                           "hipStreamWriteValue32, fine-grained word", "64 KiB H2D copy of generation words, fine-grained area",
                           "host CPU store to a page-locked host word after hipEventSynchronize(copy)",
                           "NO data copy at all: only the 4-byte H2D copy (fine-grained word)",
                           "4-byte H2D copy issued BEFORE the kernel's launch (control: must be seen at once)",
                           "256 B H2D copy of generation words", "1 KiB H2D copy of generation words", "4 KiB H2D copy of generation words",
                           "16 KiB H2D copy of generation words", "32 KiB H2D copy of generation words"};
    const size_t sweep[] = {256, 1024, 4096, 16384, 32768};
    for (int mode = 0; mode < 13; ++mode) {
        ++gen;
        const float oldv = (float)gen, newv = (float)gen + 0.5f;
        unsigned *flag = (mode == 1) ? f_coarse : (mode == 5) ? f_host : f_fine;
        hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, s_run, data, n, oldv);
        CK(hipStreamSynchronize(s_run));
        for (size_t i = 0; i < n; ++i) hnew[i] = newv;
        for (size_t i = 0; i < (1 << 14); ++i) h_words[i] = gen;
        memset(out, 0, sizeof(Out));
        *h_word = gen;
        if (mode == 5) *flag = 0; else CK(hipMemset(flag, 0, 1 << 16));
        CK(hipDeviceSynchronize());
        if (mode == 7) { CK(hipMemcpyAsync(data, hnew, n * 4, hipMemcpyHostToDevice, s_cp)); CK(hipMemcpyAsync(flag, h_word, 4, hipMemcpyHostToDevice, s_cp)); CK(hipStreamSynchronize(s_cp)); }
        CK(hipEventRecord(tev[0], s_run));
        hipLaunchKernelGGL(hog_waiter, dim3(512), dim3(256), 64 * 1024, s_run, flag, gen, data, n, newv, limit, out, sink);
        CK(hipEventRecord(tev[1], s_run));
        std::this_thread::sleep_for(std::chrono::milliseconds(3));
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t er = hipSuccess;
        if (mode != 6 && mode != 7) CK(hipMemcpyAsync(data, hnew, n * 4, hipMemcpyHostToDevice, s_cp));
        if (mode == 2) { CK(hipEventRecord(ev[0], s_cp)); CK(hipStreamWaitEvent(s_in, ev[0], 0)); CK(hipEventRecord(ev[1], s_in)); }
        if (mode == 0 || mode == 1 || mode == 2 || mode == 6) er = hipMemcpyAsync(flag, h_word, 4, hipMemcpyHostToDevice, s_cp);
        else if (mode == 3) er = hipStreamWriteValue32(s_cp, flag, gen, 0);
        else if (mode == 4) er = hipMemcpyAsync(flag, h_words, 1 << 16, hipMemcpyHostToDevice, s_cp);
        else if (mode >= 8) er = hipMemcpyAsync(flag, h_words, sweep[mode - 8], hipMemcpyHostToDevice, s_cp);
        else if (mode == 5) { CK(hipEventRecord(ev[2], s_cp)); CK(hipEventSynchronize(ev[2])); __atomic_store_n(flag, gen, __ATOMIC_RELEASE); }
        const double enq_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (er != hipSuccess) { printf("%-95s -> API error %s\n", names[mode], hipGetErrorString(er)); (void)hipGetLastError(); }
        CK(hipStreamSynchronize(s_cp));
        const double cp_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CK(hipStreamSynchronize(s_run));
        const double run_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%-95s : seen %3llu timeouts %3llu stale %8llu longest wait %7.2f ms | enqueue %.2f ms, copy stream drained after %.2f ms, kernel done after %.2f ms\n",
               names[mode], out->seen, out->timeouts, out->mismatches, out->wait_ticks_max / 1e5, enq_ms, cp_ms, run_ms);
        fflush(stdout);
    }
    return 0;
}
