// ubench_gather.hip -- how many scattered depth-pixel reads per second does the chip deliver?  The roof the TSDF update's gathers
// run under: a 4-B read per lane whose 64-B sector is (mostly) not in L1/L2, from a 265 MB pool of 32 images of 1080 x 1920.
//   random:   every lane a pseudo-random pixel (64 sectors per wave instruction), 8 independent loads in flight per lane
//   footprint: the pattern of a 4x4x4 sub-brick seen at 5 px per voxel: 16 rows per wave instruction, 4 lanes 5 px apart per row
//              (an 80-B run: 1-2 sectors), four such rows sharing an image row block; rows far apart between instructions
// Prints sector requests per second assuming every distinct 64-B sector of an instruction is one request.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o /tmp/ubench_gather && /tmp/ubench_gather
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void g_random(const float *__restrict__ p, unsigned npix, int iters, float *sink) {
    float acc = 0.f;
    unsigned long long s = (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
    for (int k = 0; k < iters; ++k) {
        unsigned ix[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s = s * 6364136223846793005ull + 1442695040888963407ull; ix[j] = (unsigned)((s >> 24) % npix); }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[ix[j]];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    if (acc == 1.2345f) *sink = 1;
}

// lane -> (x = lane & 3, y = lane >> 2 & 3, z = lane >> 4): pixel (u0 + 5 x + z, v0 + 5 y + 2 z) of image f
__global__ __launch_bounds__(256) void g_footprint(const float *__restrict__ p, int W, int H, int nimg, int iters, float *sink) {
    float acc = 0.f;
    const int lane = threadIdx.x & 63;
    const int x = lane & 3, y = (lane >> 2) & 3, z = lane >> 4;
    unsigned long long s = (unsigned long long)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 0x9E3779B97F4A7C15ull + 777ull;
    for (int k = 0; k < iters; ++k) {
        unsigned ix[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;       // wave-uniform: where this sub-brick's footprint lies
            const unsigned r = (unsigned)(s >> 24);
            const int f = (int)(r % (unsigned)nimg), u0 = (int)((r >> 5) % (unsigned)(W - 32)), v0 = (int)((r >> 17) % (unsigned)(H - 32));
            ix[j] = (unsigned)f * (unsigned)(W * H) + (unsigned)((v0 + 5 * y + 2 * z) * W + u0 + 5 * x + z);
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[ix[j]];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    if (acc == 1.2345f) *sink = 1;
}

int main() {
    const int W = 1920, H = 1080, NI = 32;
    const size_t npix = (size_t)W * H * NI;
    float *buf = nullptr, *sink = nullptr;
    if (hipMalloc(&buf, npix * 4) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, npix * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int blocks : {1024, 2048, 4096}) {
        const int iters = 64;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(g_random, dim3(blocks), dim3(256), 0, 0, buf, (unsigned)npix, iters, sink);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
            const double n = (double)blocks * 256 * iters * 8;
            if (rep) printf("random     blocks %5d: %8.3f ms  %7.2f G lane-reads/s = sector requests/s  (%7.1f GB/s at 64 B)\n", blocks, ms, n / ms * 1e-6, n * 64 / ms * 1e-6);
        }
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(g_footprint, dim3(blocks), dim3(256), 0, 0, buf, W, H, NI, iters, sink);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
            const double instr = (double)blocks * 4 * iters * 8;       // wave instructions; ~16 rows x 1.3 sectors each
            if (rep) printf("footprint  blocks %5d: %8.3f ms  %7.2f G wave-gathers/s  (x ~21 sectors = %7.2f G sector requests/s, %7.1f GB/s at 64 B)\n",
                            blocks, ms, instr / ms * 1e-6, instr * 21 / ms * 1e-6, instr * 21 * 64 / ms * 1e-6);
        }
    }
    return 0;
}
