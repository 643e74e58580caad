// Can a 2-D (pitched) host->device copy from page-locked memory land while a chip-filling kernel is resident -- i.e. is it the
// copy engine's, like the linear copies the streamed launch is fed with, or a shader's (which would wait for the launch to end)?
// A streamed bf16 launch would need the Vt image's column ranges: 512 rows x (keys x 2 bytes) with the image's pitch.
//   h2d_2d_probe            lines: width x height, alone and beside the hog, ms, GB/s, whether it finished before the hog
// Run with LD_PRELOAD=<torch>/lib/libamdhip64.so:<torch>/lib/libhsa-runtime64.so for the runtime the Python processes load.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ __launch_bounds__(256, 2) void hog(float *out, int iters) {
    extern __shared__ float lds[];
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    float a = threadIdx.x, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) a = a * b + 0.5f;
    }
    lds[threadIdx.x] = a;
    __syncthreads();
    if (a == 1234.5f) out[blockIdx.x] = lds[(threadIdx.x + 1) & 255];
}
int main() {
    int v = 0;
    CK(hipRuntimeGetVersion(&v));
    printf("hipRuntimeGetVersion %d\n", v);
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t dpitch = 65536 * 2 + 256, rows = 512;     // the Vt image of config 5: 512 rows of 65536 (+ pad) bf16
    char *d, *h;
    float *hogout;
    CK(hipMalloc((void **)&d, dpitch * rows));
    CK(hipMalloc((void **)&hogout, 1 << 20));
    CK(hipHostMalloc((void **)&h, dpitch * rows, hipHostMallocPortable));
    memset(h, 1, dpitch * rows);
    CK(hipFuncSetAttribute((const void *)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024));
    hipStream_t s_cp, s_hog;
    CK(hipStreamCreateWithFlags(&s_cp, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_hog, hipStreamNonBlocking));
    int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(hog, dim3(2 * cus), dim3(256), 70 * 1024, s_hog, hogout, iters);
        CK(hipStreamSynchronize(s_hog));
        double t0 = now();
        hipLaunchKernelGGL(hog, dim3(2 * cus), dim3(256), 70 * 1024, s_hog, hogout, iters);
        CK(hipStreamSynchronize(s_hog));
        double t1 = now();
        if (rep == 2) printf("hog alone %.3f ms\n", t1 - t0);
        else iters = (int)(iters * 4.0 / (t1 - t0 > 0.01 ? t1 - t0 : 0.01));
    }
    for (size_t width : {(size_t)4096, (size_t)8192, (size_t)16384, (size_t)32768, (size_t)131072}) {
        for (int packed_src = 0; packed_src < 2; ++packed_src) {
            const size_t spitch = packed_src ? width : dpitch;          // host image laid out per group, or like the device image
            for (int beside = 0; beside < 2; ++beside) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (beside) hipLaunchKernelGGL(hog, dim3(2 * cus), dim3(256), 70 * 1024, s_hog, hogout, iters);
                    double t0 = now();
                    CK(hipMemcpy2DAsync(d + 1024, dpitch, h, spitch, width, rows, hipMemcpyHostToDevice, s_cp));
                    CK(hipStreamSynchronize(s_cp));
                    double t1 = now();
                    CK(hipStreamSynchronize(s_hog));
                    double t2 = now();
                    if (rep) printf("2-D H2D %6zu B x %zu rows (%5.1f MB), source %s, %s: %.3f ms (%.1f GB/s)%s\n", width, rows, width * rows / 1e6,
                                    packed_src ? "packed       " : "device's pitch", beside ? "beside the hog" : "alone         ", t1 - t0,
                                    width * rows / (t1 - t0) / 1e6, beside ? (t2 - t1 > 0.3 ? "  [landed while the hog ran: copy engine]" : "  [ENDED WITH THE HOG: a shader]") : "");
                }
            }
        }
    }
    // the linear reference: the same bytes as one row range
    for (int beside = 0; beside < 2; ++beside) {
        if (beside) hipLaunchKernelGGL(hog, dim3(2 * cus), dim3(256), 70 * 1024, s_hog, hogout, iters);
        double t0 = now();
        CK(hipMemcpyAsync(d, h, 8192 * rows, hipMemcpyHostToDevice, s_cp));
        CK(hipStreamSynchronize(s_cp));
        double t1 = now();
        CK(hipStreamSynchronize(s_hog));
        double t2 = now();
        printf("linear H2D %.1f MB %s: %.3f ms (%.1f GB/s)%s\n", 8192 * rows / 1e6, beside ? "beside the hog" : "alone", t1 - t0, 8192 * rows / (t1 - t0) / 1e6,
               beside ? (t2 - t1 > 0.3 ? "  [copy engine]" : "  [a shader]") : "");
    }
    return 0;
}
