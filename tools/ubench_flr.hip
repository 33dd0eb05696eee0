// ubench_flr.hip -- is  v_cvt_flr_i32_f32(y)  equal to  v_cvt_i32_f32(v_floor_f32(y))  (what (int)floorf(y) compiles to) for EVERY float
// bit pattern (saturation, infinities and NaN included)?  And is the integer window test the pair kernel uses,
// (unsigned)floor(x + 0.5f) <= W - 1, the same decision as  x >= -0.5f && x < W - 0.5f  for every float x and every W in 1 .. 65536?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_flr.hip -o /tmp/ubench_flr && /tmp/ubench_flr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

__global__ void k(unsigned long long *bad, unsigned *ex) {
    unsigned long long nb = 0, nw = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < 0x100000000ull; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = (unsigned)i;
        const float x = __uint_as_float(bits);
        float fl;
        int ref, got;
        asm volatile("v_floor_f32 %0, %1" : "=v"(fl) : "v"(x));
        asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(ref) : "v"(fl));
        asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(got) : "v"(x));
        if (got != ref) {
            if (atomicAdd(ex, 1u) < 8u) ex[2 + atomicAdd(ex + 1, 1u) % 8u] = bits;
            ++nb;
        }
        // the window test: u = flr(x + 0.5f) as an unsigned number against W - 1, for a spread of W (all of them for |x| near W is
        // what matters: W = round(|x|) - 2 .. + 2, plus powers of two and the image widths in use)
        const float y = x + 0.5f;
        int u;
        asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(u) : "v"(y));
        const int c = (fabsf(x) < 70000.0f) ? (int)x : 0;
        const int ws[12] = {c - 2, c - 1, c, c + 1, c + 2, c + 3, 1, 640, 1080, 1920, 3840, 65536};
        for (int j = 0; j < 12; ++j) {
            const int W = ws[j];
            if (W < 1 || W > 65536) continue;
            const bool a = x >= -0.5f && x < (float)W - 0.5f;
            const bool b = (unsigned)u <= (unsigned)(W - 1);
            if (a != b) {
                if (atomicAdd(ex + 10, 1u) < 8u) { const unsigned s = atomicAdd(ex + 11, 1u) % 8u; ex[12 + 2 * s] = bits; ex[13 + 2 * s] = (unsigned)W; }
                ++nw;
            }
        }
    }
    atomicAdd(bad, nb);
    atomicAdd(bad + 1, nw);
}

int main() {
    unsigned long long *d_bad; unsigned *d_ex;
    hipMalloc(&d_bad, 16); hipMalloc(&d_ex, 32 * 4);
    hipMemset(d_bad, 0, 16); hipMemset(d_ex, 0, 128);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d_bad, d_ex);
    unsigned long long bad[2]; unsigned ex[32];
    hipMemcpy(bad, d_bad, 16, hipMemcpyDeviceToHost); hipMemcpy(ex, d_ex, 128, hipMemcpyDeviceToHost);
    printf("v_cvt_flr_i32_f32 vs v_cvt_i32_f32(v_floor_f32): %llu mismatches over all 2^32 bit patterns\n", bad[0]);
    for (unsigned s = 0; s < ex[1] && s < 8u; ++s) { float f; memcpy(&f, &ex[2 + s], 4); printf("  x = %.9g (%08x)\n", f, ex[2 + s]); }
    printf("(unsigned)flr(x + 0.5f) <= W - 1  vs  x >= -0.5f && x < W - 0.5f: %llu mismatches over all 2^32 x, 12 widths each\n", bad[1]);
    for (unsigned s = 0; s < ex[11] && s < 8u; ++s) { float f; memcpy(&f, &ex[12 + 2 * s], 4); printf("  x = %.9g (%08x), W = %u\n", f, ex[12 + 2 * s], ex[13 + 2 * s]); }
    return 0;
}
