// What one SIMD of an MI355X issues per cycle when several waves share it: vector ALU only, scalar ALU only, and both interleaved
// 1:1 in every wave's stream.  (is a scalar instruction free beside vector work?)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define V8(i)                                                                                                        \
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"        \
    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define S8                                                                                                            \
    "s_add_u32 %10, %10, 1\n s_add_u32 %11, %11, 3\n s_add_u32 %12, %12, 5\n s_add_u32 %13, %13, 7\n"                   \
    "s_add_u32 %10, %10, 1\n s_add_u32 %11, %11, 3\n s_add_u32 %12, %12, 5\n s_add_u32 %13, %13, 7\n"
#define VS8                                                                                                           \
    "v_fma_f32 %0, %0, %8, %9\n s_add_u32 %10, %10, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 %11, %11, 3\n"              \
    "v_fma_f32 %2, %2, %8, %9\n s_add_u32 %12, %12, 5\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 %13, %13, 7\n"              \
    "v_fma_f32 %4, %4, %8, %9\n s_add_u32 %10, %10, 1\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 %11, %11, 3\n"              \
    "v_fma_f32 %6, %6, %8, %9\n s_add_u32 %12, %12, 5\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 %13, %13, 7\n"
// a taken branch every 8 vector instructions
#define VB8(L)                                                                                                         \
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"        \
    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"        \
    "s_branch 1f\n s_nop 0\n s_nop 0\n 1:\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile(V8(0) V8(1) V8(2) V8(3) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(a), "v"(b), "s"(s0), "s"(s1), "s"(s2), "s"(s3));
        else if (MODE == 1)
            asm volatile(S8 S8 S8 S8 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(a), "+v"(b), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        else if (MODE == 2)
            asm volatile(VS8 VS8 VS8 VS8 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(a), "+v"(b), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        else
            asm volatile(VB8(0) VB8(1) VB8(2) VB8(3) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(a), "v"(b), "s"(s0), "s"(s1), "s"(s2), "s"(s3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)(s0 + s1 + s2 + s3);
}

template <int MODE> double run(float *out, int wgs, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, 16, 0.999f, 0.001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    int clk_khz = 0;
    CHECK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    float *out;
    CHECK(hipMalloc(&out, (size_t)cus * 16 * 256 * 4));
    const int iters = 4000;
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("%d CUs, nominal clock %.0f MHz; %d iterations of the loop body per wave; time per loop body PER SIMD in ns and in cycles at the nominal clock\n", cus, clk_khz / 1e3, iters);
    const char *names[4] = {"32 v_fma_f32", "32 s_add_u32", "32 v_fma_f32 + 32 s_add_u32 interleaved", "32 v_fma_f32 + 4 taken s_branch"};
    for (int wps = 1; wps <= 8; wps *= 2) {       // waves per SIMD = workgroups per CU (a 256-thread workgroup puts one wave on each SIMD)
        double t[4];
        t[0] = run<0>(out, cus * wps, iters);
        t[1] = run<1>(out, cus * wps, iters);
        t[2] = run<2>(out, cus * wps, iters);
        t[3] = run<3>(out, cus * wps, iters);
        for (int m = 0; m < 4; ++m) {
            const double ns_body = t[m] * 1e6 / iters / wps;      // per loop body of ONE wave, amortised over the waves of the SIMD
            printf("%d waves/SIMD  %-42s  %8.3f ms   %7.2f ns per body per SIMD = %6.1f cycles\n", wps, names[m], t[m], ns_body, ns_body * clk_khz * 1e-6);
        }
    }
    return 0;
}
