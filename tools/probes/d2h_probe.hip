// Which engine carries a device->host copy into page-locked memory on this runtime, and what does it cost beside a
// chip-filling kernel?  (The boundary's result rows leave as `__amd_rocclr_copyBuffer` SHADER launches in the kernel
// traces: 4.2 MB in 0.5 ms next to a fused kernel -- they get compute units only as that kernel's workgroups retire.)
//   d2h_probe [bytes]        one line per variant: how the copy was issued, alone / beside the hog, ms and GB/s
// Variants: hipMemcpyAsync D2H (hipHostMalloc default / portable / non-coherent / hipHostRegister'ed malloc), on a stream that
// never launched a kernel and on one that just did; a ZERO-COPY kernel (G workgroups storing straight into the mapped host
// pointer) alone and beside the hog; the hog alone.  Run under rocprofv3 --kernel-trace --memory-copy-trace to see which of the
// copies are kernels, and with AMD_LOG_LEVEL=4 for the runtime's own "HSA Copy copy_engine=..." lines.
// Env knobs of the runtime worth a separate process each: GPU_FORCE_BLIT_COPY_SIZE, HSA_ENABLE_SDMA, GPU_BLIT_ENGINE_TYPE.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// fills every SIMD's registers like the fp32 fused kernel: 2 workgroups of 256 threads with 256 VGPRs each per CU, 70 KiB of LDS
// each; spins `iters` rounds of dependent FMAs
__global__ __launch_bounds__(256, 2) void hog(float *out, int iters) {
    extern __shared__ float lds[];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");          // 256 VGPRs a wave: two workgroups own every register of a CU's SIMDs
    float a = threadIdx.x, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) a = a * b + 0.5f;
    }
    lds[threadIdx.x] = a;
    __syncthreads();
    if (a == 1234.5f) out[blockIdx.x] = lds[(threadIdx.x + 1) & 255];
}

__global__ void tiny(float *p) { if (threadIdx.x == 0) p[0] += 1.f; }

// zero-copy egress: each workgroup streams its share of src into the mapped host pointer with 16-byte stores
__global__ __launch_bounds__(256) void store_host(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        float4 v = src[i];
        __builtin_nontemporal_store(v.x, &dst[i].x);
        __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z);
        __builtin_nontemporal_store(v.w, &dst[i].w);
    }
}
__global__ __launch_bounds__(256) void store_host_plain(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main(int argc, char **argv) {
    const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) : (size_t)16 << 20;
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    float *d, *hogout;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&hogout, 1 << 20));
    CK(hipMemset(d, 1, bytes));
    CK(hipFuncSetAttribute((const void *)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    hipStream_t s_copy, s_copy2, s_hog, s_hi;
    CK(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_copy2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_hog, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi));
    void *h_def, *h_port, *h_nc, *h_reg;
    CK(hipHostMalloc(&h_def, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h_port, bytes, hipHostMallocPortable));
    CK(hipHostMalloc(&h_nc, bytes, hipHostMallocNonCoherent));
    h_reg = aligned_alloc(4096, bytes);
    memset(h_reg, 0, bytes);
    CK(hipHostRegister(h_reg, bytes, hipHostRegisterDefault));
    memset(h_def, 0, bytes); memset(h_port, 0, bytes); memset(h_nc, 0, bytes);
    // how long does the hog run?  (sized to ~3 ms)
    int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(hog, dim3(2 * cus), dim3(256), 70 * 1024, s_hog, hogout, iters);
        double t0 = now(); CK(hipStreamSynchronize(s_hog)); double t1 = now();
        hipLaunchKernelGGL(hog, dim3(2 * cus), dim3(256), 70 * 1024, s_hog, hogout, iters);
        t0 = now(); CK(hipStreamSynchronize(s_hog)); t1 = now();
        if (rep == 2) printf("hog alone: %d workgroups, %.3f ms\n", 2 * cus, t1 - t0);
        else iters = (int)(iters * 3.0 / (t1 - t0 > 0.01 ? t1 - t0 : 0.01));
    }
    struct { const char *name; void *p; } dsts[] = {{"hipHostMalloc default", h_def}, {"hipHostMalloc portable", h_port},
                                                    {"hipHostMalloc non-coherent", h_nc}, {"hipHostRegister'ed", h_reg}};
    auto timed_copy = [&](const char *what, void *dst, hipStream_t st, bool beside_hog, bool kernel_first) -> int {
        if (beside_hog) hipLaunchKernelGGL(hog, dim3(2 * cus), dim3(256), 70 * 1024, s_hog, hogout, iters);
        if (kernel_first) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d);
        double t0 = now();
        CK(hipMemcpyAsync(dst, d, bytes, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t1 = now();
        CK(hipStreamSynchronize(s_hog));
        double t2 = now();
        printf("%-28s %-26s %s: copy done after %.3f ms (%.1f GB/s)%s\n", what, kernel_first ? "behind a kernel on its stream" : "copy-only stream",
               beside_hog ? "beside the hog" : "alone         ", t1 - t0, bytes / (t1 - t0) / 1e6,
               beside_hog ? (t2 - t0 > t1 - t0 + 0.2 ? "  [hog still running: true overlap]" : "  [ended with the hog]") : "");
        return 0;
    };
    for (auto &ds : dsts) {
        for (int rep = 0; rep < 2; ++rep) timed_copy(ds.name, ds.p, s_copy, false, false);
        timed_copy(ds.name, ds.p, s_copy, true, false);
        timed_copy(ds.name, ds.p, s_copy2, false, true);
        timed_copy(ds.name, ds.p, s_copy2, true, true);
    }
    timed_copy("portable, high-priority stream", h_port, s_hi, true, false);
    // pieces: 4 x bytes/4 back to back beside the hog (the boundary's row pieces)
    {
        hipLaunchKernelGGL(hog, dim3(2 * cus), dim3(256), 70 * 1024, s_hog, hogout, iters);
        double t0 = now();
        for (int i = 0; i < 4; ++i)
            CK(hipMemcpyAsync((char *)h_port + i * (bytes / 4), (char *)d + i * (bytes / 4), bytes / 4, hipMemcpyDeviceToHost, s_copy));
        CK(hipStreamSynchronize(s_copy));
        double t1 = now();
        CK(hipStreamSynchronize(s_hog));
        printf("4 pieces of %zu bytes beside the hog: %.3f ms (%.1f GB/s)\n", bytes / 4, t1 - t0, bytes / (t1 - t0) / 1e6);
    }
    // zero-copy store kernels
    float4 *hp = nullptr;
    CK(hipHostGetDevicePointer((void **)&hp, h_port, 0));
    for (int G : {4, 8, 16, 32, 64, 256, 1024}) {
        for (int nt = 0; nt < 2; ++nt) {
            for (int beside = 0; beside < 2; ++beside) {
                if (beside) hipLaunchKernelGGL(hog, dim3(2 * cus), dim3(256), 70 * 1024, s_hog, hogout, iters);
                double t0 = now();
                if (nt) hipLaunchKernelGGL(store_host, dim3(G), dim3(256), 0, s_hi, (const float4 *)d, hp, bytes / 16);
                else hipLaunchKernelGGL(store_host_plain, dim3(G), dim3(256), 0, s_hi, (const float4 *)d, hp, bytes / 16);
                CK(hipStreamSynchronize(s_hi));
                double t1 = now();
                CK(hipStreamSynchronize(s_hog));
                double t2 = now();
                printf("zero-copy store kernel, %4d workgroups, %s stores, %s: %.3f ms (%.1f GB/s)%s\n", G, nt ? "non-temporal" : "plain       ",
                       beside ? "beside the hog" : "alone         ", t1 - t0, bytes / (t1 - t0) / 1e6,
                       beside ? (t2 - t0 > t1 - t0 + 0.2 ? "  [true overlap]" : "  [ended with the hog]") : "");
            }
        }
    }
    printf("check: host[0] = %08x host[last] = %08x\n", ((unsigned *)h_port)[0], ((unsigned *)h_port)[bytes / 4 - 1]);
    return 0;
}
