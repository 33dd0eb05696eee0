// kernels_grid.hip -- whole-grid utilities that are not on the per-frame path: the largest voxel weight of the TSDF channel
// (int32 headroom check before a merge or a long scan), the sparse form of the multi-GPU merge (which bricks hold anything;
// their records packed into / unpacked from one contiguous block) and the stream-concurrency probe used by the measurement code.
#include <chrono>
#include <vector>

#include "tl3d_internal.h"

namespace tl3d {

// largest record weight: 16-B loads (two records per lane), wave shuffle, one atomicMax per workgroup
__global__ __launch_bounds__(256) void max_weight_kernel(const int4 *__restrict__ grid2, size_t n2, int *__restrict__ out) {
    __shared__ int sm[4];
    int m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const int4 r = grid2[i];
        m = max(m, max(r.y, r.w));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_down(m, d));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(sm[0], sm[1]), max(sm[2], sm[3])));
}

int launch_max_weight(hipStream_t s, const int2 *grid, size_t nvox, int *d_out) {
    TL3D_HIP(hipMemsetAsync(d_out, 0, sizeof(int), s));
    const size_t n2 = nvox / 2;                           // nvox is a multiple of 512
    const unsigned nb = (unsigned)((n2 + 255) / 256 < 2048 ? (n2 + 255) / 256 : 2048);
    hipLaunchKernelGGL(max_weight_kernel, dim3(nb ? nb : 1), dim3(256), 0, s, reinterpret_cast<const int4 *>(grid), n2, d_out);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// ---- sparse merge: which bricks hold anything, and their records as one contiguous block --------------------------------
// map[b] |= 1 when brick b has a voxel with a TSDF weight or a centroid count.  One wave per brick; TSDF: 16 B per lane x 4,
// centroid: the record's second word (sz | n << 32) of 8 records per lane.
__global__ __launch_bounds__(256) void touched_bricks_kernel(const int4 *__restrict__ tsdf2, const unsigned long long *__restrict__ cen,
                                                             unsigned nbricks, unsigned char *__restrict__ map) {
    const int lane = threadIdx.x & 63;
    for (unsigned b = blockIdx.x * 4u + (threadIdx.x >> 6); b < nbricks; b += gridDim.x * 4u) {
        bool any = false;
        if (tsdf2) {
            const int4 *r = tsdf2 + ((size_t)b << 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int4 v = r[k * 64 + lane];
                any = any || v.y != 0 || v.w != 0;
            }
        }
        if (cen) {
            const unsigned long long *r = cen + ((size_t)b << 11);
#pragma unroll
            for (int k = 0; k < 8; ++k) any = any || (r[(size_t)(k * 64 + lane) * 4 + 1] >> 32) != 0ull;
        }
        if (__ballot(any) != 0ull && lane == 0) map[b] = 1;
    }
}

// rows of `words16` 16-byte words each: dst[i] = src[idx[i]] (pack) or dst[idx[i]] = src[i] (unpack)
template <bool PACK>
__global__ __launch_bounds__(256) void brick_rows_kernel(int4 *__restrict__ grid, const unsigned *__restrict__ idx, long long n, unsigned words16,
                                                         int4 *__restrict__ packed) {
    for (long long i = blockIdx.x; i < n; i += gridDim.x) {
        int4 *g = grid + (size_t)idx[i] * words16;
        int4 *q = packed + (size_t)i * words16;
        for (unsigned w = threadIdx.x; w < words16; w += 256) {
            if (PACK) q[w] = g[w];
            else g[w] = q[w];
        }
    }
}

int launch_touched_bricks(hipStream_t s, const int2 *tsdf, const unsigned long long *cen, unsigned nbricks, unsigned char *map) {
    const unsigned nb = (nbricks + 3u) / 4u < 4096u ? (nbricks + 3u) / 4u : 4096u;
    hipLaunchKernelGGL(touched_bricks_kernel, dim3(nb ? nb : 1), dim3(256), 0, s, reinterpret_cast<const int4 *>(tsdf), cen, nbricks, map);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_brick_rows(hipStream_t s, bool pack, void *grid, const unsigned *idx, long long n, unsigned bytes_per_brick, void *packed) {
    if (n <= 0) return TL3D_OK;
    const unsigned nb = (unsigned)(n < 8192 ? n : 8192);
    if (pack)
        hipLaunchKernelGGL(brick_rows_kernel<true>, dim3(nb), dim3(256), 0, s, static_cast<int4 *>(grid), idx, n, bytes_per_brick / 16u, static_cast<int4 *>(packed));
    else
        hipLaunchKernelGGL(brick_rows_kernel<false>, dim3(nb), dim3(256), 0, s, static_cast<int4 *>(grid), idx, n, bytes_per_brick / 16u, static_cast<int4 *>(packed));
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// One wave that keeps its queue busy for `ticks` of the 100 MHz constant clock (bounded by an iteration cap, so it always
// ends).  N of them on N streams finish in ~spin when every stream has a hardware queue of its own, in ceil(N / Q) x spin
// when the runtime multiplexes the streams onto Q queues (GPU_MAX_HW_QUEUES, read when the HIP runtime initialises).
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long ticks, unsigned *__restrict__ sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned it = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks && it < 40000000u) {
        __builtin_amdgcn_s_sleep(8);
        ++it;
    }
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = it;
}

int probe_hw_queues(int n_streams, double spin_ms, double *elapsed_ms) {
    std::vector<hipStream_t> st((size_t)n_streams, nullptr);
    unsigned *sink = nullptr;
    int rc = TL3D_OK;
    if (hipMalloc(&sink, sizeof(unsigned)) != hipSuccess) return set_err(TL3D_E_NOMEM, "alloc failed");
    for (int i = 0; i < n_streams && rc == TL3D_OK; ++i)
        if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) rc = set_err(TL3D_E_HIP, "stream create failed");
    if (rc == TL3D_OK) {
        const unsigned long long ticks = (unsigned long long)(spin_ms * 1e5);      // 100 MHz
        for (int i = 0; i < n_streams; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[i], 100ull, sink);   // warm: code load, queue creation
        for (int i = 0; i < n_streams; ++i) (void)hipStreamSynchronize(st[i]);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n_streams; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[i], ticks, sink);
        for (int i = 0; i < n_streams; ++i)
            if (hipStreamSynchronize(st[i]) != hipSuccess) rc = set_err(TL3D_E_HIP, "probe sync failed");
        *elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (hipGetLastError() != hipSuccess) rc = set_err(TL3D_E_HIP, "probe launch failed");
    }
    for (int i = 0; i < n_streams; ++i)
        if (st[i]) (void)hipStreamDestroy(st[i]);
    (void)hipFree(sink);
    return rc;
}

}  // namespace tl3d
