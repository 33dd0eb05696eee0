// Micro-benchmark behind DESIGN.md 7.3: what do the two halves of a MIXED brick cost a CU, alone and together?
//   G: 8 x 64-lane dword gathers from a brick-like footprint of an 8 MB image (L2/L1 resident), + the projection-sized ALU
//   R: 8 x (8-B-per-lane load, add, store) on a 4 KB record block of a 1 GiB grid (streams from HBM)
// modes: G only, R only, G then R in every wave (the kernel's structure), G in even waves / R in odd waves (specialised).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench_mixed.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

constexpr int W = 1080, H = 1920;

__device__ __forceinline__ float gather_half(const float *__restrict__ img, unsigned brick, int lane, float spacing, bool swap = false) {
    // footprint origin from a hash of the brick id; lanes = 8 x 8 voxels, 8 layers along the image-vertical axis
    const unsigned h = brick * 2654435761u;
    const int u0 = (int)(h % (unsigned)(W - 96)), v0 = (int)((h >> 12) % (unsigned)(H - 96));
    // swap: the 8 lanes of a group walk along the view direction (sub-pixel parallax) and the groups across the image,
    // so the 4 lanes the texture unit takes per step share a cache line
    const int la = swap ? lane >> 3 : lane & 7, lb = swap ? lane & 7 : lane >> 3;
    float acc = 0.f;
    float d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int u = u0 + (int)(spacing * (float)la + 0.37f * (float)lb);
        const int v = v0 + (int)(spacing * (float)k + 0.61f * (float)lb);
        d[k] = img[v * W + u];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {                  // stand-in for the ~45 VALU instructions per voxel of the real path
        float x = d[k];
#pragma unroll
        for (int r = 0; r < 20; ++r) x = fmaf(x, 1.0001f, 0.5f) * 0.999f;
        acc += x;
    }
    return acc;
}

__device__ __forceinline__ void record_half(int2 *__restrict__ grid, unsigned brick, int lane, int q) {
    int2 *recs = grid + ((size_t)brick << 9);
    int2 r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = recs[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) { r[k].x += q; r[k].y += 1; recs[k * 64 + lane] = r[k]; }
}

// gather half reading a 16-bit (millimetre) depth image: half the bytes under the same footprint
__device__ __forceinline__ float gather_half_u16(const unsigned short *__restrict__ img, unsigned brick, int lane, float spacing) {
    const unsigned h = brick * 2654435761u;
    const int u0 = (int)(h % (unsigned)(W - 96)), v0 = (int)((h >> 12) % (unsigned)(H - 96));
    const int la = lane & 7, lb = lane >> 3;
    float acc = 0.f;
    float d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int u = u0 + (int)(spacing * (float)la + 0.37f * (float)lb);
        const int v = v0 + (int)(spacing * (float)k + 0.61f * (float)lb);
        d[k] = (float)img[v * W + u] * 0.001f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float x = d[k];
#pragma unroll
        for (int r = 0; r < 20; ++r) x = fmaf(x, 1.0001f, 0.5f) * 0.999f;
        acc += x;
    }
    return acc;
}

// 16-B-per-lane variant of the record half: 4 loads + 4 stores per brick
__device__ __forceinline__ void record_half16(int2 *__restrict__ grid, unsigned brick, int lane, int q) {
    int4 *recs = reinterpret_cast<int4 *>(grid + ((size_t)brick << 9));
    int4 r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = recs[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) { r[k].x += q; r[k].y += 1; r[k].z += q; r[k].w += 1; recs[k * 64 + lane] = r[k]; }
}

// predicated variants: pred 1 = a pseudo-random half of the lanes of every instruction, pred 2 = half of the 64-B rows
// (8 consecutive lanes) of every instruction, pred 3 = every other instruction entirely
__device__ __forceinline__ void record_half_pred(int2 *__restrict__ grid, unsigned brick, int lane, int q, int pred) {
    int2 *recs = grid + ((size_t)brick << 9);
    int2 r[8];
    bool on[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned h = (brick * 8u + (unsigned)k) * 2654435761u;
        on[k] = pred == 1 ? (((h >> (lane & 31)) ^ (unsigned)lane) & 1u) != 0 : pred == 2 ? (((h >> (lane >> 3)) & 1u) != 0) : (k & 1) == 0;
        if (on[k]) r[k] = recs[k * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (on[k]) { r[k].x += q; r[k].y += 1; recs[k * 64 + lane] = r[k]; }
}

// mode 0: G, 1: R, 2: G then R per wave, 3: even waves G / odd waves R (each wave does twice as many bricks of its kind)
// mode 4: R with 16 B per lane, 5/6/7: R with half the lanes / half the rows / half the instructions predicated off
__global__ __launch_bounds__(256) void ub(const float *img, int2 *grid, const unsigned *bricks, unsigned nbricks, int mode, float spacing,
                                          float *sink) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc = 0.f;
    if (mode <= 2) {
        for (unsigned i = blockIdx.x * 4 + wid; i < nbricks; i += gridDim.x * 4) {
            const unsigned b = bricks[i];
            float g = 0.f;
            if (mode != 1) g = gather_half(img, b, lane, spacing);
            if (mode != 0) record_half(grid, b, lane, mode == 2 ? (int)g & 7 : 3);
            acc += g;
        }
    } else if (mode == 9) {
        for (unsigned i = blockIdx.x * 4 + wid; i < nbricks; i += gridDim.x * 4)
            acc += gather_half_u16(reinterpret_cast<const unsigned short *>(img), bricks[i], lane, spacing);
    } else if (mode == 8) {
        for (unsigned i = blockIdx.x * 4 + wid; i < nbricks; i += gridDim.x * 4) acc += gather_half(img, bricks[i], lane, spacing, true);
    } else if (mode >= 4) {
        for (unsigned i = blockIdx.x * 4 + wid; i < nbricks; i += gridDim.x * 4) {
            const unsigned b = bricks[i];
            if (mode == 4) record_half16(grid, b, lane, 3);
            else record_half_pred(grid, b, lane, 3, mode - 4);
        }
    } else {
        const bool is_g = (wid & 1) == 0;
        const unsigned half = wid >> 1;
        for (unsigned i = blockIdx.x * 2 + half; i < nbricks; i += gridDim.x * 2) {
            const unsigned b = bricks[i];
            if (is_g) acc += gather_half(img, b, lane, spacing);
            else record_half(grid, b, lane, 3);
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const unsigned nb = 16000;                          // MIXED bricks of the headline frame
    float *img; int2 *grid; unsigned *bricks; float *sink;
    const size_t nvox = 512ull * 512 * 512;
    CK(hipMalloc(&img, (size_t)W * H * 4)); CK(hipMalloc(&grid, nvox * 8)); CK(hipMalloc(&bricks, nb * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(img, 0, (size_t)W * H * 4)); CK(hipMemset(grid, 0, nvox * 8));
    std::vector<unsigned> hb(nb);
    unsigned s = 12345;
    for (unsigned i = 0; i < nb; ++i) { s = s * 1664525u + 1013904223u; hb[i] = (s >> 8) % 262144u; }
    CK(hipMemcpy(bricks, hb.data(), nb * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char *names[10] = {"G only", "R only", "G then R per wave", "G waves | R waves", "R 16 B per lane", "R half the lanes", "R half the rows", "R half the instr.", "G, depth-axis lanes adjacent", "G, 16-bit depth image"};
    for (int blocks : {1024}) {
        for (float spacing : {4.3f, 8.6f}) {
            for (int mode = 0; mode < 10; ++mode) {
                for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(ub, dim3(blocks), dim3(256), 0, 0, img, grid, bricks, nb, mode, spacing, sink);
                CK(hipDeviceSynchronize());
                const int reps = 20;
                CK(hipEventRecord(a, 0));
                for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(ub, dim3(blocks), dim3(256), 0, 0, img, grid, bricks, nb, mode, spacing, sink);
                CK(hipEventRecord(b, 0));
                CK(hipEventSynchronize(b));
                float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
                printf("blocks %4d  pixel spacing %.1f  %-20s %7.2f us\n", blocks, spacing, names[mode], 1e3f * ms / reps);
            }
        }
    }
    return 0;
}
