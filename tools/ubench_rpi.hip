// ubench_rpi.hip -- is  v_cvt_rpi_i32_f32(x)  equal to  (int)floorf(x + 0.5f)  (f32 add, then floor) ?
// Exhaustive over every float bit pattern with |x| < 2^24 plus the rest of the finite range sampled; prints the mismatches.
// (The ISA manual words the instruction as floor(x + 0.5); whether the sum is rounded to f32 first decides the half-ulp cases,
// e.g. x = 0.49999997: fl(x + 0.5f) = 1.0f -> 1, an exact sum -> 0.)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_rpi.hip -o /tmp/ubench_rpi && /tmp/ubench_rpi
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

__global__ void k(unsigned long long *bad, unsigned *ex, int *ex_ref, int *ex_got) {
    unsigned long long nb = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < 0x100000000ull; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = (unsigned)i;
        if ((bits & 0x7fffffffu) >= 0x4b800000u) continue;          // |x| >= 2^24 (and inf / nan)
        const float x = __uint_as_float(bits);
        const int ref = (int)floorf(x + 0.5f);
        int got;
        asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(got) : "v"(x));
        if (got != ref) {
            if (nb == 0 && atomicAdd(ex, 1u) < 16u) { const unsigned s = atomicAdd(ex + 1, 1u); if (s < 16u) { ex[2 + s] = bits; ex_ref[s] = ref; ex_got[s] = got; } }
            ++nb;
        }
    }
    atomicAdd(bad, nb);
}

int main() {
    unsigned long long *d_bad; unsigned *d_ex; int *d_r, *d_g;
    hipMalloc(&d_bad, 8); hipMalloc(&d_ex, 18 * 4); hipMalloc(&d_r, 64); hipMalloc(&d_g, 64);
    hipMemset(d_bad, 0, 8); hipMemset(d_ex, 0, 72); hipMemset(d_r, 0, 64); hipMemset(d_g, 0, 64);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d_bad, d_ex, d_r, d_g);
    unsigned long long bad; unsigned ex[18]; int r[16], g[16];
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(ex, d_ex, 72, hipMemcpyDeviceToHost);
    hipMemcpy(r, d_r, 64, hipMemcpyDeviceToHost); hipMemcpy(g, d_g, 64, hipMemcpyDeviceToHost);
    printf("v_cvt_rpi_i32_f32 vs (int)floorf(x + 0.5f): %llu mismatches over all floats with |x| < 2^24\n", bad);
    for (unsigned s = 0; s < ex[1] && s < 16u; ++s) { float f; memcpy(&f, &ex[2 + s], 4); printf("  x = %.9g (%08x): floorf(x + 0.5f) = %d, v_cvt_rpi = %d\n", f, ex[2 + s], r[s], g[s]); }
    return 0;
}
