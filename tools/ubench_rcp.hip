// ubench_rcp.hip -- is  r = v_rcp_f32(z); e = fma(-z, r, 1); r' = fma(e, r, r)  equal to the IEEE quotient 1.0f / z ?
// Exhaustive over every positive float bit pattern (2^31 values); prints the number of mismatches and the sub-ranges they
// fall in.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_rcp.hip -o /tmp/ubench_rcp && /tmp/ubench_rcp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

__global__ void k(unsigned long long *bad, unsigned *first, unsigned *last, unsigned long long *bad_by_exp) {
    unsigned long long nb = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < 0x7f800000ull; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = (unsigned)i;
        const float z = __uint_as_float(bits);
        const float ref = 1.0f / z;
        float r = __builtin_amdgcn_rcpf(z);
        const float e = fmaf(-z, r, 1.0f);
        r = fmaf(e, r, r);
        if (__float_as_uint(r) != __float_as_uint(ref)) {
            ++nb;
            atomicMin(first, bits);
            atomicMax(last, bits);
            atomicAdd(bad_by_exp + (bits >> 23), 1ull);
        }
    }
    atomicAdd(bad, nb);
}

int main() {
    unsigned long long *d_bad, *d_exp; unsigned *d_first, *d_last;
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 4); hipMalloc(&d_last, 4); hipMalloc(&d_exp, 256 * 8);
    hipMemset(d_bad, 0, 8); hipMemset(d_first, 0xff, 4); hipMemset(d_last, 0, 4); hipMemset(d_exp, 0, 256 * 8);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d_bad, d_first, d_last, d_exp);
    unsigned long long bad, ex[256]; unsigned first, last;
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&last, d_last, 4, hipMemcpyDeviceToHost); hipMemcpy(ex, d_exp, 256 * 8, hipMemcpyDeviceToHost);
    float ff, lf; memcpy(&ff, &first, 4); memcpy(&lf, &last, 4);
    printf("mismatches: %llu of 2139095040 positive finite floats; first %08x (%g) last %08x (%g)\n", bad, first, ff, last, lf);
    for (int e = 0; e < 256; ++e) if (ex[e]) printf("  exponent field %3d (2^%d): %llu\n", e, e - 127, ex[e]);
    return 0;
}
