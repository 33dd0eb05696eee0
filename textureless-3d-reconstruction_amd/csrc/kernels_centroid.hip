// kernels_centroid.hip -- row a7 (fusion half): voxel-centroid accumulators with the semantics of Open3D
// voxel_down_sample as the reference calls it (depth_to_reconstruction.py:404-410; DER:635-640):
//   index = floor((p - origin)/voxel) in fp64 from the f32 point, per-voxel mean of points and colours.
// The np.vstack of every cloud (D2R:401-402) is never materialised: each frame's points go straight from the
// depth map into integer accumulators (exact, order-free => bit-reproducible, and the multi-GPU merge is a sum).
//
// record (32 B, one 32-B sector): u64[4] = { sx | sy<<32, sz | n<<32, sr | sg<<32, sb }, s* in units of voxel/4096.
// One thread per sampled pixel.  Points that share a voxel are combined on chip before anything reaches the grid: first
// runs of adjacent lanes in the wave (segmented shuffle scan), then -- per-frame kernel -- across the workgroup's 32 x 8
// tile of samples in an LDS table keyed by the record index (LDS-staged voxel-block accumulation); only one set of four
// 64-bit atomics per distinct voxel and tile goes to HBM.
#include "bp_device.h"
#include "tl3d_internal.h"

namespace tl3d {

struct CenAdd {
    unsigned long long rec;      // record index, ~0ull if none
    unsigned long long a, b, c, d;
};

__device__ __forceinline__ CenAdd centroid_key(const Grid &g, const float p[3], unsigned r8, unsigned g8, unsigned b8) {
    CenAdd o;
    o.rec = ~0ull;
    const double org[3] = {g.oxd, g.oyd, g.ozd};
    const int dims[3] = {g.nx, g.ny, g.nz};
    int idx[3];
    unsigned long long q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double rc = ((double)p[a] - org[a]) / g.vsd;
        const double fl = floor(rc);
        if (!(fl >= 0.0 && fl < (double)dims[a])) return o;
        idx[a] = (int)fl;
        int qq = (int)((rc - fl) * 4096.0);
        if (qq > 4095) qq = 4095;
        q[a] = (unsigned long long)qq;
    }
    o.rec = (unsigned long long)vox_index(idx[0], idx[1], idx[2], g.nbx, g.nby);
    o.a = q[0] | (q[1] << 32);
    o.b = q[2] | (1ull << 32);
    o.c = (unsigned long long)r8 | ((unsigned long long)g8 << 32);
    o.d = (unsigned long long)b8;
    return o;
}

__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int d) {
    const unsigned lo = __shfl_up((unsigned)(v & 0xffffffffull), d), hi = __shfl_up((unsigned)(v >> 32), d);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_down_u64(unsigned long long v, int d) {
    const unsigned lo = __shfl_down((unsigned)(v & 0xffffffffull), d), hi = __shfl_down((unsigned)(v >> 32), d);
    return ((unsigned long long)hi << 32) | lo;
}

// Neighbouring pixels of an image row usually land in the same voxel (a 5 mm voxel spans ~8 pixels at 1 m), so a wave
// first sums each run of adjacent lanes with equal record index (segmented inclusive scan: 6 shuffle steps) and only the
// last lane of a run issues the four 64-bit atomics.  Integer sums: the grid is the same bit for bit, with ~4-8x fewer
// atomics.  Every lane of the wave must call this (the shuffles need a full EXEC mask).
__device__ __forceinline__ bool centroid_reduce_runs(CenAdd &k) {
    const int lane = threadIdx.x & 63;
    const unsigned long long prev_rec = shfl_up_u64(k.rec, 1), next_rec = shfl_down_u64(k.rec, 1);   // all lanes, no short-circuit
    const bool head = (lane == 0) | (prev_rec != k.rec);
    int start = head ? lane : 0;                                  // first lane of my run = max of head positions so far
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(start, d);
        if (lane >= d) start = max(start, o);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long oa = shfl_up_u64(k.a, d), ob = shfl_up_u64(k.b, d), oc = shfl_up_u64(k.c, d), od = shfl_up_u64(k.d, d);
        if (lane - d >= start) { k.a += oa; k.b += ob; k.c += oc; k.d += od; }
    }
    const bool tail = (lane == 63) | (next_rec != k.rec);
    return tail && k.rec != ~0ull;                                // this lane now carries its run's sums
}

__device__ __forceinline__ void centroid_commit_runs(CenAdd k, unsigned long long *__restrict__ grid) {
    if (centroid_reduce_runs(k)) {
        unsigned long long *rec = grid + 4 * k.rec;
        atomicAdd(rec + 0, k.a);
        atomicAdd(rec + 1, k.b);
        atomicAdd(rec + 2, k.c);
        atomicAdd(rec + 3, k.d);
    }
}

// LDS table of the per-frame kernel: open addressing, 1024 slots for the at most 1024 samples of a 32 x 32 tile (never more
// distinct keys than slots, so every probe sequence ends); key = record index + 1 (0 = empty), four 64-bit sums per slot.
// Dense sampling only (stride 1 and 2: 74 / 18 pixels per 5 mm voxel at 1 m); from stride 3 on neighbouring samples
// rarely share a voxel and the points go straight to the grid (centroid_direct_kernel).
constexpr int CEN_TW = 32, CEN_TH = 32, CEN_SLOTS = 1024;

__device__ __forceinline__ CenAdd centroid_sample(const Cam &cam, const Grid &g, const BpArgs &a, const PoseD &p, const float *__restrict__ depth,
                                                  const uint8_t *__restrict__ bgr, int us, int vs, bool &valid) {
    CenAdd k;
    k.rec = ~0ull;
    k.a = k.b = k.c = k.d = 0;
    valid = false;
    if (us < a.Ws && vs < a.Hs) {
        const int u = us * a.sub, v = vs * a.sub;
        float pt[3];
        if (bp_pixel(cam, a, p, depth, u, v, pt)) {
            valid = true;
            unsigned r8 = 0, g8 = 0, b8 = 0;
            if (bgr) {
                const uint8_t *px = bgr + 3 * ((size_t)v * cam.W + u);
                b8 = px[0]; g8 = px[1]; r8 = px[2];
            }
            k = centroid_key(g, pt, r8, g8, b8);
        }
    }
    return k;
}

// statistics: one pair of adds per workgroup into one of 256 counter lines (a single hot word would serialise the whole
// launch: same-address atomics retire at ~90 per microsecond)
__device__ __forceinline__ void centroid_stats(int nvalid_thread, int nkept_thread, unsigned long long *__restrict__ counters) {
    __shared__ int s_n[2];
    if (threadIdx.x == 0) { s_n[0] = 0; s_n[1] = 0; }
    __syncthreads();
    int v = nvalid_thread, k = nkept_thread;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { v += __shfl_down(v, d); k += __shfl_down(k, d); }
    if ((threadIdx.x & 63) == 0 && v) { atomicAdd(&s_n[0], v); atomicAdd(&s_n[1], k); }
    __syncthreads();
    if (threadIdx.x == 0 && s_n[0]) {
        unsigned long long *line = counters + (size_t)(blockIdx.x & 255) * 8;
        atomicAdd(line + 0, (unsigned long long)s_n[1]);
        atomicAdd(line + 1, (unsigned long long)(s_n[0] - s_n[1]));
    }
}

__global__ __launch_bounds__(256) void centroid_frame_kernel(Cam cam, Grid g, BpArgs a, PoseD p, const float *__restrict__ depth,
                                                             const uint8_t *__restrict__ bgr, unsigned long long *__restrict__ grid,
                                                             unsigned long long *__restrict__ counters, int tiles_x) {
    __shared__ unsigned long long s_key[CEN_SLOTS];
    __shared__ unsigned long long s_val[CEN_SLOTS][4];
    for (int h = threadIdx.x; h < CEN_SLOTS; h += 256) {
        s_key[h] = 0ull;
        s_val[h][0] = 0ull; s_val[h][1] = 0ull; s_val[h][2] = 0ull; s_val[h][3] = 0ull;
    }
    __syncthreads();
    // the workgroup's samples: a 32 x 32 tile of the (sub-sampled) image, 8 rows at a time, so that voxels are shared
    // across rows too (a 5 mm voxel spans ~8 x 8 pixels at 1 m)
    const int tx = (int)(blockIdx.x % (unsigned)tiles_x), ty = (int)(blockIdx.x / (unsigned)tiles_x);
    const int us = tx * CEN_TW + (threadIdx.x & (CEN_TW - 1));
    int nvalid = 0, nkept = 0;
#pragma unroll
    for (int r = 0; r < CEN_TH / 8; ++r) {
        const int vs = ty * CEN_TH + r * 8 + (threadIdx.x >> 5);
        bool valid;
        CenAdd k = centroid_sample(cam, g, a, p, depth, bgr, us, vs, valid);
        nvalid += valid ? 1 : 0;
        nkept += (k.rec != ~0ull) ? 1 : 0;
        if (centroid_reduce_runs(k)) {                            // one lane per run of equal voxels: into the LDS table
            const unsigned long long key = k.rec + 1ull;
            unsigned h = (unsigned)((k.rec * 0x9E3779B97F4A7C15ull) >> 54) & (CEN_SLOTS - 1);
            for (;;) {
                const unsigned long long old = atomicCAS(&s_key[h], 0ull, key);
                if (old == 0ull || old == key) break;
                h = (h + 1u) & (CEN_SLOTS - 1);
            }
            atomicAdd(&s_val[h][0], k.a);
            atomicAdd(&s_val[h][1], k.b);
            atomicAdd(&s_val[h][2], k.c);
            atomicAdd(&s_val[h][3], k.d);
        }
    }
    centroid_stats(nvalid, nkept, counters);                      // (its barriers also complete the table)
    for (int h = threadIdx.x; h < CEN_SLOTS; h += 256) {          // one set of grid atomics per distinct voxel of the tile
        const unsigned long long key = s_key[h];
        if (key != 0ull) {
            unsigned long long *rec = grid + 4 * (key - 1ull);
            atomicAdd(rec + 0, s_val[h][0]);
            atomicAdd(rec + 1, s_val[h][1]);
            atomicAdd(rec + 2, s_val[h][2]);
            atomicAdd(rec + 3, s_val[h][3]);
        }
    }
}

// sparse sampling (stride >= 3): one thread per sample in row-major order, runs combined in the wave, straight to the grid
__global__ __launch_bounds__(256) void centroid_direct_kernel(Cam cam, Grid g, BpArgs a, PoseD p, const float *__restrict__ depth,
                                                              const uint8_t *__restrict__ bgr, unsigned long long *__restrict__ grid,
                                                              unsigned long long *__restrict__ counters) {
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long ns = (long long)a.Ws * a.Hs;
    const int vs = (int)(s / a.Ws), us = (int)(s - (long long)vs * a.Ws);
    bool valid;
    CenAdd k = centroid_sample(cam, g, a, p, depth, bgr, s < ns ? us : a.Ws, vs, valid);
    centroid_stats(valid ? 1 : 0, (k.rec != ~0ull) ? 1 : 0, counters);
    centroid_commit_runs(k, grid);
}

__global__ __launch_bounds__(256) void centroid_points_kernel(Grid g, const float *__restrict__ xyz, const uint8_t *__restrict__ rgb,
                                                              long long n, unsigned long long *__restrict__ grid,
                                                              unsigned long long *__restrict__ counters) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    CenAdd k;
    k.rec = ~0ull;
    k.a = k.b = k.c = k.d = 0;
    bool valid = false;
    if (i < n) {
        valid = true;
        const float pt[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        k = centroid_key(g, pt, rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    }
    centroid_stats(valid ? 1 : 0, (k.rec != ~0ull) ? 1 : 0, counters);
    centroid_commit_runs(k, grid);
}

// per-block min/max of a point list -> slab[block][6]
__global__ __launch_bounds__(256) void bounds_kernel(const float *__restrict__ xyz, long long n, float *__restrict__ slab) {
    __shared__ float sm[4][6];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int d = 32; d > 0; d >>= 1) {
            mn[a] = fminf(mn[a], __shfl_down(mn[a], d));
            mx[a] = fmaxf(mx[a], __shfl_down(mx[a], d));
        }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0)
        for (int a = 0; a < 3; ++a) { sm[wid][a] = mn[a]; sm[wid][3 + a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sm[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, sm[w][threadIdx.x]) : fmaxf(v, sm[w][threadIdx.x]);
        slab[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void add_i32_kernel(int4 *__restrict__ dst, const int4 *__restrict__ src, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        int4 a = dst[i];
        const int4 b = src[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        dst[i] = a;
    }
}

__global__ __launch_bounds__(256) void add_u64_kernel(ulonglong2 *__restrict__ dst, const ulonglong2 *__restrict__ src, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        ulonglong2 a = dst[i];
        const ulonglong2 b = src[i];
        a.x += b.x; a.y += b.y;
        dst[i] = a;
    }
}

int launch_centroid_frame(hipStream_t s, const Cam &cam, const Grid &g, const BpArgs &a, const PoseD &p, const float *depth,
                          const uint8_t *bgr, unsigned long long *grid, unsigned long long *counters) {
    if (a.sub >= 3) {
        const long long ns = (long long)a.Ws * a.Hs;
        hipLaunchKernelGGL(centroid_direct_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, s, cam, g, a, p, depth, bgr, grid, counters);
    } else {
        const int tiles_x = (a.Ws + CEN_TW - 1) / CEN_TW, tiles_y = (a.Hs + CEN_TH - 1) / CEN_TH;
        hipLaunchKernelGGL(centroid_frame_kernel, dim3((unsigned)tiles_x * (unsigned)tiles_y), dim3(256), 0, s, cam, g, a, p, depth, bgr, grid,
                           counters, tiles_x);
    }
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_centroid_points(hipStream_t s, const Grid &g, const float *xyz, const uint8_t *rgb, long long n,
                           unsigned long long *grid, unsigned long long *counters) {
    if (n <= 0) return TL3D_OK;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(centroid_points_kernel, dim3(nb), dim3(256), 0, s, g, xyz, rgb, n, grid, counters);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_bounds(hipStream_t s, const float *xyz, long long n, float *slab, int nblocks) {
    hipLaunchKernelGGL(bounds_kernel, dim3(nblocks), dim3(256), 0, s, xyz, n, slab);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_add_i32(hipStream_t s, int *dst, const int *src, size_t n) {
    const size_t n4 = n / 4;     // n is a multiple of 1024 (bricks)
    const unsigned nb = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_i32_kernel, dim3(nb ? nb : 1), dim3(256), 0, s, (int4 *)dst, (const int4 *)src, n4);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_add_u64(hipStream_t s, unsigned long long *dst, const unsigned long long *src, size_t n) {
    const size_t n2 = n / 2;
    const unsigned nb = (unsigned)((n2 + 255) / 256 < 4096 ? (n2 + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_u64_kernel, dim3(nb ? nb : 1), dim3(256), 0, s, (ulonglong2 *)dst, (const ulonglong2 *)src, n2);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

}  // namespace tl3d
