// kernels_extract.hip -- N4 / row a7 read-back half: fused grid -> point list, in ascending record order.
//   mode TL3D_EXTRACT_CENTROID: one point per occupied voxel = mean of the accumulated points and colours, i.e. what
//       Open3D voxel_down_sample returns to the reference (depth_to_reconstruction.py:410, 417-418), optionally gated
//       by the TSDF (weight / |mean tsdf|) as an outlier filter.
//   mode TL3D_EXTRACT_TSDF: zero crossings of the mean TSDF along the +x,+y,+z voxel edges (no reference code).
// Two passes (count -> single-block scan -> write) so the output order is deterministic.
// Same fp64 expressions as oracle/tl3d_oracle.c: orc_extract.
#include "tl3d_internal.h"

namespace tl3d {

struct ExtArgs {
    int mode, min_count, min_weight;
    double max_abs;
};

__device__ __forceinline__ void rec_coords(size_t idx, int nbx, int nby, int &i, int &j, int &k) {
    const size_t b = idx >> 9;
    const int l = (int)(idx & 511);
    const int bx = (int)(b % (size_t)nbx), by = (int)((b / (size_t)nbx) % (size_t)nby);
    const int bz = (int)(b / ((size_t)nbx * (size_t)nby));
    in_brick_coords(l, i, j, k);
    i |= bx << 3;
    j |= by << 3;
    k |= bz << 3;
}

__device__ __forceinline__ void mean_colour(const unsigned long long *__restrict__ rec, unsigned long long n, uint8_t c[3]) {
    c[0] = (uint8_t)((rec[2] & 0xffffffffull) / n);
    c[1] = (uint8_t)((rec[2] >> 32) / n);
    c[2] = (uint8_t)((rec[3] & 0xffffffffull) / n);
}

// Returns the number of points record idx emits; when WRITE, stores them at out index o, o+1, ...
template <bool WRITE>
__device__ __forceinline__ int extract_record(const Grid &g, const ExtArgs &a, const int2 *__restrict__ tsdf,
                                              const unsigned long long *__restrict__ cen, size_t idx, float *__restrict__ xyz,
                                              uint8_t *__restrict__ rgb, unsigned long long o, unsigned long long cap) {
    int ijk[3];
    rec_coords(idx, g.nbx, g.nby, ijk[0], ijk[1], ijk[2]);
    const double org[3] = {g.oxd, g.oyd, g.ozd};
    const double off[3] = {g.offx, g.offy, g.offz};
    if (a.mode == TL3D_EXTRACT_CENTROID) {
        const unsigned long long *rec = cen_record(g, cen, idx);
        if (!rec) return 0;
        const unsigned long long r1 = rec[1];
        const unsigned long long n = r1 >> 32;
        if (n < (unsigned long long)a.min_count) return 0;
        if (tsdf && a.min_weight > 0) {
            const int2 tw = tsdf_record(g, tsdf, idx);
            if (tw.y < a.min_weight) return 0;
            const double mean = (double)tw.x / ((double)tw.y * 32767.0);
            if (!(fabs(mean) <= a.max_abs)) return 0;
        }
        if (WRITE && o < cap) {
            const unsigned long long r0 = rec[0];
            const unsigned long long s[3] = {r0 & 0xffffffffull, r0 >> 32, r1 & 0xffffffffull};
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double f = ((double)s[ax] + 0.5 * (double)n) / ((double)n * 4096.0);
                xyz[3 * o + ax] = (float)(org[ax] + (off[ax] + (double)ijk[ax] + f) * g.vsd);
            }
            uint8_t c[3];
            mean_colour(rec, n, c);
            rgb[3 * o + 0] = c[0]; rgb[3 * o + 1] = c[1]; rgb[3 * o + 2] = c[2];
        }
        return 1;
    }
    // TSDF zero crossings
    const int mw = a.min_weight < 1 ? 1 : a.min_weight;
    const int2 ta_ = tsdf_record(g, tsdf, idx);
    if (ta_.y < mw) return 0;
    const double ta = (double)ta_.x / ((double)ta_.y * 32767.0);
    if (!(fabs(ta) < 0.98)) return 0;
    const int dims[3] = {g.nx, g.ny, g.nz};
    int emitted = 0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        int nb[3] = {ijk[0], ijk[1], ijk[2]};
        nb[e] += 1;
        if (nb[e] >= dims[e]) continue;
        const size_t jdx = vox_index(nb[0], nb[1], nb[2], g.nbx, g.nby);
        const int2 tb_ = tsdf_record(g, tsdf, jdx);
        if (tb_.y < mw) continue;
        const double tb = (double)tb_.x / ((double)tb_.y * 32767.0);
        if (!(fabs(tb) < 0.98)) continue;
        if (!(ta * tb < 0.0)) continue;
        if (WRITE && o + emitted < cap) {
            const unsigned long long oo = o + emitted;
            const double r0 = fabs(ta), r1 = fabs(tb);
            const double frac = r0 / (r0 + r1);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const double cc = org[ax] + ((double)ijk[ax] + 0.5) * g.vsd;
                xyz[3 * oo + ax] = (float)(ax == e ? cc + frac * g.vsd : cc);
            }
            uint8_t c[3] = {128, 128, 128};
            if (cen) {
                const size_t first = (r0 <= r1) ? idx : jdx, second = (r0 <= r1) ? jdx : idx;
                const unsigned long long *ra = cen_record(g, cen, first);
                const unsigned long long na = ra ? ra[1] >> 32 : 0ull;
                if (na > 0) {
                    mean_colour(ra, na, c);
                } else {
                    const unsigned long long *rb = cen_record(g, cen, second);
                    const unsigned long long nb2 = rb ? rb[1] >> 32 : 0ull;
                    if (nb2 > 0) mean_colour(rb, nb2, c);
                }
            }
            rgb[3 * oo + 0] = c[0]; rgb[3 * oo + 1] = c[1]; rgb[3 * oo + 2] = c[2];
        }
        ++emitted;
    }
    return emitted;
}

__global__ __launch_bounds__(256) void extract_count_kernel(Grid g, ExtArgs a, const int2 *__restrict__ tsdf,
                                                            const unsigned long long *__restrict__ cen, size_t nvox,
                                                            unsigned *__restrict__ block_counts) {
    __shared__ unsigned sm[4];
    unsigned cnt = 0;
    const size_t base = (size_t)blockIdx.x * EXTRACT_CHUNK;
#pragma unroll 1
    for (int it = 0; it < EXTRACT_CHUNK / 256; ++it) {
        const size_t idx = base + (size_t)it * 256 + threadIdx.x;
        if (idx < nvox) cnt += (unsigned)extract_record<false>(g, a, tsdf, cen, idx, nullptr, nullptr, 0, 0);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(256) void extract_write_kernel(Grid g, ExtArgs a, const int2 *__restrict__ tsdf,
                                                            const unsigned long long *__restrict__ cen, size_t nvox,
                                                            const unsigned long long *__restrict__ offsets,
                                                            float *__restrict__ xyz, uint8_t *__restrict__ rgb,
                                                            unsigned long long cap) {
    __shared__ unsigned sm[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned long long run = offsets[blockIdx.x];
    const size_t base = (size_t)blockIdx.x * EXTRACT_CHUNK;
#pragma unroll 1
    for (int it = 0; it < EXTRACT_CHUNK / 256; ++it) {
        const size_t idx = base + (size_t)it * 256 + threadIdx.x;
        const unsigned c = (idx < nvox) ? (unsigned)extract_record<false>(g, a, tsdf, cen, idx, nullptr, nullptr, 0, 0) : 0u;
        unsigned inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned tv = __shfl_up(inc, d);
            if (lane >= d) inc += tv;
        }
        if (lane == 63) sm[wid] = inc;
        __syncthreads();
        unsigned wbase = 0;
        for (int w = 0; w < wid; ++w) wbase += sm[w];
        const unsigned total = sm[0] + sm[1] + sm[2] + sm[3];
        __syncthreads();
        if (c) extract_record<true>(g, a, tsdf, cen, idx, xyz, rgb, run + wbase + (inc - c), cap);
        run += total;
    }
}

int launch_extract_count(hipStream_t s, const Grid &g, int mode, int min_count, int min_weight, double max_abs,
                         const int2 *tsdf, const unsigned long long *cen, unsigned *block_counts, int nblocks) {
    ExtArgs a{mode, min_count < 1 ? 1 : min_count, min_weight, max_abs};
    const size_t nvox = (size_t)g.nx * g.ny * g.nz;
    hipLaunchKernelGGL(extract_count_kernel, dim3(nblocks), dim3(256), 0, s, g, a, tsdf, cen, nvox, block_counts);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_extract_write(hipStream_t s, const Grid &g, int mode, int min_count, int min_weight, double max_abs,
                         const int2 *tsdf, const unsigned long long *cen, const unsigned long long *offsets, int nblocks,
                         float *xyz, uint8_t *rgb, unsigned long long cap) {
    ExtArgs a{mode, min_count < 1 ? 1 : min_count, min_weight, max_abs};
    const size_t nvox = (size_t)g.nx * g.ny * g.nz;
    hipLaunchKernelGGL(extract_write_kernel, dim3(nblocks), dim3(256), 0, s, g, a, tsdf, cen, nvox, offsets, xyz, rgb, cap);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

}  // namespace tl3d
