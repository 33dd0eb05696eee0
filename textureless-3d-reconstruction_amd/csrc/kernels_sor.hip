// kernels_sor.hip -- row f1: statistical outlier removal with the semantics of Open3D
// remove_statistical_outlier as the reference calls it (depth_to_reconstruction.py:412-415, nb_neighbors=20,
// std_ratio=2.0): per point the mean distance to its k nearest neighbours (the point itself included, as the
// KNN search returns it), mu / sigma (Bessel-corrected) of those means, keep  0 < mean < mu + ratio*sigma.
// Restated in oracle/ref_numpy.py: statistical_outlier_open3d (Open3D itself: parity unpinned).
//
// Exact k-NN without a tree: points are counting-sorted into a uniform cell grid (cell ~ 2 voxels); each point
// scans the cells around it in growing cubes and stops once its k-th best distance is inside the scanned radius.
// Distances in fp64 (Open3D works on double points); the k-best list lives in registers (static indexing).
#include <math.h>

#include <vector>

#include "tl3d_internal.h"

namespace tl3d {

struct CellGrid {
    double ox, oy, oz, cell, inv_cell;
    int nx, ny, nz;
};

__device__ __forceinline__ int cell_of(const CellGrid &cg, const float *__restrict__ p, int &cx, int &cy, int &cz) {
    cx = min(cg.nx - 1, max(0, (int)floor(((double)p[0] - cg.ox) * cg.inv_cell)));
    cy = min(cg.ny - 1, max(0, (int)floor(((double)p[1] - cg.oy) * cg.inv_cell)));
    cz = min(cg.nz - 1, max(0, (int)floor(((double)p[2] - cg.oz) * cg.inv_cell)));
    return (cz * cg.ny + cy) * cg.nx + cx;
}

__global__ __launch_bounds__(256) void sor_count_kernel(CellGrid cg, const float *__restrict__ xyz, long long n,
                                                        unsigned *__restrict__ cell_count) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cx, cy, cz;
    atomicAdd(cell_count + cell_of(cg, xyz + 3 * i, cx, cy, cz), 1u);
}

// block sums of 1024-element chunks
__global__ __launch_bounds__(256) void sor_chunk_sum_kernel(const unsigned *__restrict__ v, long long n, unsigned *__restrict__ sums) {
    __shared__ unsigned sm[4];
    const long long base = (long long)blockIdx.x * 1024;
    unsigned s = 0;
    for (int k = 0; k < 4; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < n) s += v[i];
    }
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// exclusive scan inside each 1024-element chunk, offset by the scanned chunk sums
__global__ __launch_bounds__(256) void sor_chunk_scan_kernel(const unsigned *__restrict__ v, long long n,
                                                             const unsigned long long *__restrict__ chunk_off,
                                                             unsigned *__restrict__ start) {
    __shared__ unsigned sm[4];
    const long long base = (long long)blockIdx.x * 1024;
    const long long i0 = base + (long long)threadIdx.x * 4;
    unsigned a[4], s = 0;
    for (int k = 0; k < 4; ++k) { a[k] = (i0 + k < n) ? v[i0 + k] : 0u; s += a[k]; }
    unsigned inc = s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) sm[wid] = inc;
    __syncthreads();
    unsigned run = (unsigned)chunk_off[blockIdx.x] + inc - s;
    for (int w = 0; w < wid; ++w) run += sm[w];
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < n) start[i0 + k] = run;
        run += a[k];
    }
}

__global__ __launch_bounds__(256) void sor_fill_kernel(CellGrid cg, const float *__restrict__ xyz, long long n,
                                                       const unsigned *__restrict__ cell_start, unsigned *__restrict__ cell_fill,
                                                       float *__restrict__ sorted_xyz) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cx, cy, cz;
    const int c = cell_of(cg, xyz + 3 * i, cx, cy, cz);
    const unsigned pos = cell_start[c] + atomicAdd(cell_fill + c, 1u);
    sorted_xyz[3 * (size_t)pos + 0] = xyz[3 * i + 0];
    sorted_xyz[3 * (size_t)pos + 1] = xyz[3 * i + 1];
    sorted_xyz[3 * (size_t)pos + 2] = xyz[3 * i + 2];
}

template <int KMAX>
__global__ __launch_bounds__(256) void sor_knn_kernel(CellGrid cg, const float *__restrict__ xyz, long long n, int k,
                                                      const unsigned *__restrict__ cell_start,
                                                      const unsigned *__restrict__ cell_count,
                                                      const float *__restrict__ sorted_xyz, double *__restrict__ mean_d) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    int cx, cy, cz;
    cell_of(cg, xyz + 3 * i, cx, cy, cz);
    double best[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) best[j] = 1e300;
    const int rmax = max(cg.nx, max(cg.ny, cg.nz));
    for (int r = 0; r <= rmax; ++r) {
        // shell of Chebyshev radius r around (cx,cy,cz)
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cz + dz;
            if (z < 0 || z >= cg.nz) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int y = cy + dy;
                if (y < 0 || y >= cg.ny) continue;
                const bool face = (abs(dz) == r) || (abs(dy) == r);
                const int step = face ? 1 : max(1, 2 * r);            // interior rows: only the two end cells
                for (int dx = -r; dx <= r; dx += step) {
                    const int x = cx + dx;
                    if (x < 0 || x >= cg.nx) continue;
                    const int c = (z * cg.ny + y) * cg.nx + x;
                    const unsigned s = cell_start[c], e = s + cell_count[c];
                    for (unsigned q = s; q < e; ++q) {
                        const double ddx = (double)sorted_xyz[3 * (size_t)q] - px, ddy = (double)sorted_xyz[3 * (size_t)q + 1] - py;
                        const double ddz = (double)sorted_xyz[3 * (size_t)q + 2] - pz;
                        double v = ddx * ddx + ddy * ddy + ddz * ddz;
                        if (v < best[KMAX - 1]) {
#pragma unroll
                            for (int j = 0; j < KMAX; ++j) {           // ordered insert, static indices -> registers
                                const double b = best[j];
                                const bool sw = v < b;
                                best[j] = sw ? v : b;
                                v = sw ? b : v;
                            }
                        }
                    }
                }
            }
        }
        // everything outside the scanned cube is at least r*cell away
        const double reach = (double)r * cg.cell;
        double kth = 1e300;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (j == k - 1) kth = best[j];
        if (kth <= reach * reach) break;
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
        if (j < k) s += sqrt(best[j]);
    mean_d[i] = s / (double)k;
}

// pass 0: count and sum of the positive means; pass 1: sum of squared deviations; pass 2: write the mask
__global__ __launch_bounds__(256) void sor_stat_kernel(const double *__restrict__ mean_d, long long n, int pass, double mu,
                                                       double thr, double *__restrict__ slab, uint8_t *__restrict__ keep) {
    __shared__ double sm[4][2];
    double a = 0.0, b = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double m = mean_d[i];
        if (pass == 0) {
            if (m > 0.0) { a += 1.0; b += m; }
        } else if (pass == 1) {
            if (m > 0.0) { const double d = m - mu; a += d * d; }
        } else {
            const bool kp = (m > 0.0) && (m < thr);
            keep[i] = kp ? 1 : 0;
            if (kp) a += 1.0;
        }
    }
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_down(a, d); b += __shfl_down(b, d); }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = a; sm[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        slab[2 * blockIdx.x] = ((sm[0][0] + sm[1][0]) + sm[2][0]) + sm[3][0];
        slab[2 * blockIdx.x + 1] = ((sm[0][1] + sm[1][1]) + sm[2][1]) + sm[3][1];
    }
}

#define SOR_HIP(x)                                                                                  \
    do {                                                                                            \
        hipError_t e__ = (x);                                                                       \
        if (e__ != hipSuccess) { rc = set_err(TL3D_E_HIP, "%s failed: %s", #x, hipGetErrorString(e__)); goto done; } \
    } while (0)

int sor_run(tl3d_ctx *ctx, const float *xyz, long long n, int k, double std_ratio, double cell, uint8_t *keep, long long *kept) {
    hipStream_t s = ctx->stream;
    int rc = TL3D_OK;
    unsigned *cell_count = nullptr, *cell_start = nullptr, *cell_fill = nullptr, *chunk_sums = nullptr;
    unsigned long long *chunk_off = nullptr;
    float *sorted = nullptr;
    double *mean_d = nullptr, *slab = nullptr;
    if (k > n) k = (int)n;
    // bounds
    double mn[3], mx[3];
    rc = tl3d_points_bounds(ctx, xyz, n, mn, mx);
    if (rc) return rc;
    CellGrid cg;
    for (;;) {
        cg.cell = cell;
        cg.inv_cell = 1.0 / cell;
        cg.ox = mn[0]; cg.oy = mn[1]; cg.oz = mn[2];
        const double ex = floor((mx[0] - mn[0]) / cell) + 1, ey = floor((mx[1] - mn[1]) / cell) + 1, ez = floor((mx[2] - mn[2]) / cell) + 1;
        if (ex * ey * ez <= 134217728.0 && ex < 2e9 && ey < 2e9 && ez < 2e9) {
            cg.nx = (int)ex; cg.ny = (int)ey; cg.nz = (int)ez;
            break;
        }
        cell *= 2.0;
    }
    {
        const long long ncell = (long long)cg.nx * cg.ny * cg.nz;
        const int nchunks = (int)((ncell + 1023) / 1024);
        const unsigned nb = (unsigned)((n + 255) / 256);
        const int nred = 1024;
        std::vector<double> h(2 * nred);
        double m = 0, sum = 0, mu = 0, sq = 0, sigma = 0, thr = 0, cnt = 0;
        SOR_HIP(hipMalloc(&cell_count, ncell * sizeof(unsigned)));
        SOR_HIP(hipMalloc(&cell_start, ncell * sizeof(unsigned)));
        SOR_HIP(hipMalloc(&cell_fill, ncell * sizeof(unsigned)));
        SOR_HIP(hipMalloc(&chunk_sums, (size_t)nchunks * sizeof(unsigned)));
        SOR_HIP(hipMalloc(&chunk_off, ((size_t)nchunks + 1) * sizeof(unsigned long long)));
        SOR_HIP(hipMalloc(&sorted, (size_t)n * 3 * sizeof(float)));
        SOR_HIP(hipMalloc(&mean_d, (size_t)n * sizeof(double)));
        SOR_HIP(hipMalloc(&slab, 2 * nred * sizeof(double)));
        SOR_HIP(hipMemsetAsync(cell_count, 0, ncell * sizeof(unsigned), s));
        SOR_HIP(hipMemsetAsync(cell_fill, 0, ncell * sizeof(unsigned), s));
        hipLaunchKernelGGL(sor_count_kernel, dim3(nb), dim3(256), 0, s, cg, xyz, n, cell_count);
        hipLaunchKernelGGL(sor_chunk_sum_kernel, dim3(nchunks), dim3(256), 0, s, cell_count, ncell, chunk_sums);
        rc = launch_scan(s, chunk_sums, chunk_off, nchunks, chunk_off + nchunks);
        if (rc) goto done;
        hipLaunchKernelGGL(sor_chunk_scan_kernel, dim3(nchunks), dim3(256), 0, s, cell_count, ncell, chunk_off, cell_start);
        hipLaunchKernelGGL(sor_fill_kernel, dim3(nb), dim3(256), 0, s, cg, xyz, n, cell_start, cell_fill, sorted);
        if (k <= 32)
            hipLaunchKernelGGL(sor_knn_kernel<32>, dim3(nb), dim3(256), 0, s, cg, xyz, n, k, cell_start, cell_count, sorted, mean_d);
        else
            hipLaunchKernelGGL(sor_knn_kernel<64>, dim3(nb), dim3(256), 0, s, cg, xyz, n, k, cell_start, cell_count, sorted, mean_d);
        SOR_HIP(hipGetLastError());
        // mu, sigma over the positive means (fixed-order sums of the block partials)
        hipLaunchKernelGGL(sor_stat_kernel, dim3(nred), dim3(256), 0, s, mean_d, n, 0, 0.0, 0.0, slab, keep);
        SOR_HIP(hipMemcpyAsync(h.data(), slab, 2 * nred * sizeof(double), hipMemcpyDeviceToHost, s));
        SOR_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < nred; ++i) { m += h[2 * i]; sum += h[2 * i + 1]; }
        if (m > 1.0) {
            mu = sum / m;
            hipLaunchKernelGGL(sor_stat_kernel, dim3(nred), dim3(256), 0, s, mean_d, n, 1, mu, 0.0, slab, keep);
            SOR_HIP(hipMemcpyAsync(h.data(), slab, 2 * nred * sizeof(double), hipMemcpyDeviceToHost, s));
            SOR_HIP(hipStreamSynchronize(s));
            for (int i = 0; i < nred; ++i) sq += h[2 * i];
            sigma = sqrt(sq / (m - 1.0));
            thr = mu + std_ratio * sigma;
        } else {
            thr = INFINITY;                         // <= 1 valid point: keep every point with a positive mean
        }
        hipLaunchKernelGGL(sor_stat_kernel, dim3(nred), dim3(256), 0, s, mean_d, n, 2, mu, thr, slab, keep);
        SOR_HIP(hipMemcpyAsync(h.data(), slab, 2 * nred * sizeof(double), hipMemcpyDeviceToHost, s));
        SOR_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < nred; ++i) cnt += h[2 * i];
        *kept = (long long)cnt;
    }
done:
    (void)hipStreamSynchronize(s);
    if (cell_count) (void)hipFree(cell_count);
    if (cell_start) (void)hipFree(cell_start);
    if (cell_fill) (void)hipFree(cell_fill);
    if (chunk_sums) (void)hipFree(chunk_sums);
    if (chunk_off) (void)hipFree(chunk_off);
    if (sorted) (void)hipFree(sorted);
    if (mean_d) (void)hipFree(mean_d);
    if (slab) (void)hipFree(slab);
    return rc;
}

}  // namespace tl3d
