// Micro-benchmark for DESIGN.md section 7: can the L2's integer atomic units do the read-modify-write of the voxel records,
// so that a CU's vector-memory pipeline (TCP) carries no pending record reads while it serves the depth gathers?
//   mode 0  RMW by load / add / store, 16 B per lane, 4 KB bricks          (the FREE-brick path of tsdf_integrate_kernel)
//   mode 1  no-return global_atomic_add (32-bit) x 2 per record             (sum += q, weight += 1)
//   mode 2  no-return global_atomic_add_x2 (64-bit) x 1 per record          (weight << 32 | biased sum)
//   mode 3  gathers only (8 x 64-lane dword gathers per brick from an 8 MB image)
//   mode 4  gathers + mode 0 per brick    mode 5  gathers + mode 1    mode 6  gathers + mode 2
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench_atomics.hip -o /tmp/uba && /tmp/uba
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
constexpr int W = 1080, H = 1920;

__device__ __forceinline__ float gathers(const float *__restrict__ img, unsigned brick, int lane) {
    const unsigned h = brick * 2654435761u;
    const int u0 = (int)(h % (unsigned)(W - 96)), v0 = (int)((h >> 12) % (unsigned)(H - 96));
    const int la = lane & 7, lb = lane >> 3;
    float d[8], acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = img[(v0 + (int)(8.6f * (float)k + 0.61f * (float)lb)) * W + u0 + (int)(8.6f * (float)la + 0.37f * (float)lb)];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float x = d[k];
#pragma unroll
        for (int r = 0; r < 20; ++r) x = fmaf(x, 1.0001f, 0.5f) * 0.999f;
        acc += x;
    }
    return acc;
}

__global__ __launch_bounds__(256) void ub(const float *img, int2 *grid, const unsigned *bricks, unsigned nb, int mode, float *sink) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc = 0.f;
    for (unsigned i = blockIdx.x * 4 + wid; i < nb; i += gridDim.x * 4) {
        const unsigned b = bricks[i];
        int q = 32767;
        if (mode >= 3) { const float g = gathers(img, b, lane); acc += g; q = 32767 - ((int)g & 1); }
        if (mode == 3) continue;
        const int m = mode >= 4 ? mode - 4 : mode;
        if (m == 0) {
            int4 *recs = reinterpret_cast<int4 *>(grid + ((size_t)b << 9));
            int4 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = recs[k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 4; ++k) { r[k].x += q; r[k].y += 1; r[k].z += q; r[k].w += 1; recs[k * 64 + lane] = r[k]; }
        } else if (m == 1) {
            int *w = reinterpret_cast<int *>(grid + ((size_t)b << 9));
#pragma unroll
            for (int k = 0; k < 16; ++k)                       // 16 instructions x 256 B: dword (k * 64 + lane) of the brick's 1024
                __hip_atomic_fetch_add(w + k * 64 + lane, (lane & 1) ? 1 : q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned long long *w = reinterpret_cast<unsigned long long *>(grid + ((size_t)b << 9));
#pragma unroll
            for (int k = 0; k < 8; ++k)                        // 8 instructions x 512 B
                __hip_atomic_fetch_add(w + k * 64 + lane, (1ull << 32) + (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    float *img; int2 *grid; unsigned *bricks; float *sink;
    const size_t nvox = 512ull * 512 * 512;
    CK(hipMalloc(&img, (size_t)W * H * 4)); CK(hipMalloc(&grid, nvox * 8)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(img, 0, (size_t)W * H * 4)); CK(hipMemset(grid, 0, nvox * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char *names[7] = {"RMW load/add/store 16 B", "atomic add u32 x2", "atomic add u64 x1", "gathers only", "gathers + RMW", "gathers + atomic u32", "gathers + atomic u64"};
    for (unsigned nb : {10000u, 16000u, 26000u}) {
        std::vector<unsigned> hb(nb);
        unsigned s = 12345;
        for (unsigned i = 0; i < nb; ++i) { s = s * 1664525u + 1013904223u; hb[i] = (s >> 8) % 262144u; }
        CK(hipMalloc(&bricks, nb * 4));
        CK(hipMemcpy(bricks, hb.data(), nb * 4, hipMemcpyHostToDevice));
        for (int blocks : {1024, 1536}) {
            for (int mode = 0; mode < 7; ++mode) {
                for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(ub, dim3(blocks), dim3(256), 0, 0, img, grid, bricks, nb, mode, sink);
                CK(hipDeviceSynchronize());
                const int reps = 20;
                CK(hipEventRecord(a, 0));
                for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(ub, dim3(blocks), dim3(256), 0, 0, img, grid, bricks, nb, mode, sink);
                CK(hipEventRecord(b, 0));
                CK(hipEventSynchronize(b));
                float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
                const double us = 1e3 * ms / reps, mb = nb * 8192.0 / 1e6;
                printf("bricks %5u blocks %4d  %-26s %7.2f us  (%.1f MB read+write -> %.2f TB/s)\n", nb, blocks, names[mode], us, mb, mode == 3 ? 0.0 : mb / us / 1e0 * 1e-6 * 1e6 / 1e6);
            }
        }
        CK(hipFree(bricks));
    }
    return 0;
}
