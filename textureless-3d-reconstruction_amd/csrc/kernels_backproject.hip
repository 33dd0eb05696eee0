// kernels_backproject.hip -- rows a3/a4/a5: fused projection factors + mask + pose transform + BGR->RGB,
// with order-preserving compaction (the reference's boolean fancy-indexing keeps row-major pixel order,
// depth_to_reconstruction.py:364-366, 381).  Also the u16-mm depth conversion of row a2 (D2R:85-90).
//
// Arithmetic: fp64 intermediates, exactly the reference's sequence (xf=(u-cx)/fx in fp64, x=xf*z,
// P_w = R^T P_c - R^T t, cast to f32).  MI355X fp64 VALU is full rate enough that this kernel stays
// bound by its 15 B/point of output; bytes per frame: read H*W*(4+3)/s^2, write N*15.
#include "tl3d_internal.h"
#include "bp_device.h"

namespace tl3d {

__global__ __launch_bounds__(256) void u16_to_f32_kernel(const uint16_t *__restrict__ in, float *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = mm_to_m(in[i]);              // == .astype(float32) / 1000.0 (D2R:90)
}

__global__ __launch_bounds__(256) void bp_count_kernel(Cam cam, BpArgs a, const float *__restrict__ depth,
                                                       unsigned *__restrict__ block_counts) {
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long ns = (long long)a.Ws * a.Hs;
    bool ok = false;
    if (s < ns) {
        const int vs = (int)(s / a.Ws), us = (int)(s - (long long)vs * a.Ws);
        ok = bp_valid_only(cam, a, depth, us * a.sub, vs * a.sub);
    }
    const int c = __syncthreads_count(ok);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (unsigned)c;
}

// single-block exclusive scan of n block counts -> 64-bit offsets (+ total)
__global__ __launch_bounds__(1024) void scan_kernel(const unsigned *__restrict__ counts, unsigned long long *__restrict__ offsets,
                                                    int n, unsigned long long *__restrict__ total) {
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = min(n, lo + per);
    unsigned long long s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned long long v = (t >= off) ? part[t - off] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = part[t] - s;          // exclusive prefix of this thread's span
    for (int i = lo; i < hi; ++i) {
        offsets[i] = run;
        run += counts[i];
    }
    if (t == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void bp_write_kernel(Cam cam, BpArgs a, PoseD p, const float *__restrict__ depth,
                                                       const uint8_t *__restrict__ bgr,
                                                       const unsigned long long *__restrict__ offsets,
                                                       float *__restrict__ xyz, uint8_t *__restrict__ rgb,
                                                       unsigned long long cap) {
    __shared__ unsigned wave_tot[4];
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long ns = (long long)a.Ws * a.Hs;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bool ok = false;
    float pt[3];
    int u = 0, v = 0;
    if (s < ns) {
        const int vs = (int)(s / a.Ws), us = (int)(s - (long long)vs * a.Ws);
        u = us * a.sub;
        v = vs * a.sub;
        ok = bp_pixel(cam, a, p, depth, u, v, pt);
    }
    const unsigned long long m = __ballot(ok);
    const unsigned below = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wid] = __popcll(m);
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wid; ++w) base += wave_tot[w];
    if (ok) {
        const unsigned long long o = offsets[blockIdx.x] + base + below;
        if (o < cap) {
            xyz[3 * o + 0] = pt[0];
            xyz[3 * o + 1] = pt[1];
            xyz[3 * o + 2] = pt[2];
            uint8_t r = 0, g = 0, b = 0;
            if (bgr) {
                const uint8_t *px = bgr + 3 * ((size_t)v * cam.W + u);
                b = px[0]; g = px[1]; r = px[2];
            }
            rgb[3 * o + 0] = r;
            rgb[3 * o + 1] = g;
            rgb[3 * o + 2] = b;
        }
    }
}

int launch_u16_to_f32(hipStream_t s, const uint16_t *in, float *out, size_t n) {
    if (n == 0) return TL3D_OK;
    hipLaunchKernelGGL(u16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_bp_count(hipStream_t s, const Cam &cam, const BpArgs &a, const float *depth, unsigned *block_counts, int nblocks) {
    hipLaunchKernelGGL(bp_count_kernel, dim3(nblocks), dim3(256), 0, s, cam, a, depth, block_counts);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_scan(hipStream_t s, const unsigned *counts, unsigned long long *offsets, int n, unsigned long long *total) {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, counts, offsets, n, total);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_bp_write(hipStream_t s, const Cam &cam, const BpArgs &a, const PoseD &p, const float *depth, const uint8_t *bgr,
                    const unsigned long long *offsets, int nblocks, float *xyz, uint8_t *rgb, unsigned long long cap) {
    hipLaunchKernelGGL(bp_write_kernel, dim3(nblocks), dim3(256), 0, s, cam, a, p, depth, bgr, offsets, xyz, rgb, cap);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

}  // namespace tl3d
