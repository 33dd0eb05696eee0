// kernels_grid.hip -- whole-grid utilities that are not on the per-frame path: the largest voxel weight of the TSDF channel
// (int32 headroom check before a merge or a long scan), the sparse form of the multi-GPU merge (which bricks hold anything;
// their records packed into / unpacked from one contiguous block) and the stream-concurrency probe used by the measurement code.
#include <chrono>
#include <vector>

#include "tl3d_internal.h"

namespace tl3d {

// Largest weight a reader of the channel would see: per brick, the largest record weight (if the brick has records) PLUS the
// brick's pending free-space count -- exact whether or not the counts have been folded into the records.  One wave per brick,
// 16-B loads (two records per lane), wave shuffle, one atomicMax per workgroup.
__global__ __launch_bounds__(256) void max_weight_kernel(Grid g, const int4 *__restrict__ pool2, const unsigned *__restrict__ free_cnt, unsigned nbricks,
                                                         int *__restrict__ out) {
    __shared__ int sm[4];
    const int lane = threadIdx.x & 63;
    int m = 0;
    for (unsigned b = blockIdx.x * 4u + (threadIdx.x >> 6); b < nbricks; b += gridDim.x * 4u) {
        const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane((int)brick_slot(g.tsdf_tab, b));
        int w = 0;
        if (slot < SLOT_FULL) {
            const int4 *r = pool2 + ((size_t)slot << 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int4 v = r[k * 64 + lane];
                w = max(w, max(v.y, v.w));
            }
        }
        if (free_cnt) w += (int)__builtin_amdgcn_readfirstlane((int)free_cnt[b]);
        m = max(m, w);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_down(m, d));
    if (lane == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(sm[0], sm[1]), max(sm[2], sm[3])));
}

int launch_max_weight(hipStream_t s, const Grid &g, const int2 *pool, int *d_out) {
    TL3D_HIP(hipMemsetAsync(d_out, 0, sizeof(int), s));
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    const unsigned nb = (nbricks + 3u) / 4u < 4096u ? (nbricks + 3u) / 4u : 4096u;
    hipLaunchKernelGGL(max_weight_kernel, dim3(nb ? nb : 1), dim3(256), 0, s, g, reinterpret_cast<const int4 *>(pool), g.free_cnt, nbricks, d_out);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// the same over a plain dense record array (the other grid of a merge)
__global__ __launch_bounds__(256) void max_weight_dense_kernel(const int4 *__restrict__ grid2, size_t n2, int *__restrict__ out) {
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

int launch_max_weight_dense(hipStream_t s, const int2 *grid, size_t nvox, int *d_out) {
    TL3D_HIP(hipMemsetAsync(d_out, 0, sizeof(int), s));
    const size_t n2 = nvox / 2;                           // nvox is a multiple of 512
    const unsigned nb = (unsigned)((n2 + 255) / 256 < 2048 ? (n2 + 255) / 256 : 2048);
    hipLaunchKernelGGL(max_weight_dense_kernel, dim3(nb ? nb : 1), dim3(256), 0, s, reinterpret_cast<const int4 *>(grid), n2, d_out);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// ---- sparse merge: which bricks hold anything, and their records as one contiguous block --------------------------------
// map[b] |= 1 when brick b has a voxel with a TSDF weight or a centroid count.  One wave per brick; TSDF: 16 B per lane x 4,
// centroid: the record's second word (sz | n << 32) of 8 records per lane.  (Pending free-space counts are not records: they
// travel as the small per-brick counter array, tl3d_grid_device_ptr(TL3D_CH_FREE).)
// SUB: the map has EIGHT bytes per brick, one per 4x4x4 sub-brick (64 contiguous records: tl3d_internal.h in_brick_index).
template <bool SUB>
__global__ __launch_bounds__(256) void touched_bricks_kernel(Grid g, const int4 *__restrict__ tsdf2, const unsigned long long *__restrict__ cen,
                                                             unsigned nbricks, unsigned char *__restrict__ map) {
    const int lane = threadIdx.x & 63;
    for (unsigned b = blockIdx.x * 4u + (threadIdx.x >> 6); b < nbricks; b += gridDim.x * 4u) {
        unsigned subs = 0u;                                           // (wave-uniform) sub-bricks of b that hold anything
        if (tsdf2) {
            const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane((int)brick_slot(g.tsdf_tab, b));
            if (slot < SLOT_FULL) {
                const int4 *r = tsdf2 + ((size_t)slot << 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) {                          // 16 B = records 2 (64 k + lane), + 1: sub-brick 2 k (lanes 0-31), 2 k + 1 (32-63)
                    const int4 v = r[k * 64 + lane];
                    const unsigned long long m = __ballot(v.y != 0 || v.w != 0);
                    subs |= ((m & 0xffffffffull) ? 1u : 0u) << (2 * k) | ((m >> 32) ? 1u : 0u) << (2 * k + 1);
                }
            }
        }
        if (cen) {
            const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane((int)brick_slot(g.cen_tab, b));
            if (slot < SLOT_FULL) {
                const unsigned long long *r = cen + ((size_t)slot << 11);
#pragma unroll
                for (int k = 0; k < 8; ++k) subs |= (__ballot((r[(size_t)(k * 64 + lane) * 4 + 1] >> 32) != 0ull) ? 1u : 0u) << k;
            }
        }
        if (SUB) {
            if (lane < 8 && ((subs >> lane) & 1u)) map[(size_t)b * 8 + lane] = 1;
        } else if (subs && lane == 0) {
            map[b] = 1;
        }
    }
}

// Rows of `words16` 16-byte words each, row i <-> virtual brick idx[i] (idx == nullptr: row i <-> brick i, a dense image of the
// channel).  MODE 0: row = the brick's records (zeros when it has none; TSDF rows get the pending free-space count folded in when
// add_free is set: the image a reader of a dense grid would see).  MODE 1: the brick's records = row.  MODE 2: records += row
// (32-bit lanes for the TSDF channel, 64-bit for the centroid channel).  MODE 1 / 2 give a brick without records a slot when its
// row holds anything.
// SUB: a row is ONE 4x4x4 SUB-BRICK (64 records), idx[i] = brick * 8 + sub-brick: what the multi-GPU merge sends, so that a brick
// whose surface crosses two of its sub-bricks does not travel whole (the centroid channel: 2 KB instead of 16 KB).
template <int MODE, bool IS_TSDF, bool SUB = false>
__global__ __launch_bounds__(256) void brick_rows_kernel(Grid g, int4 *__restrict__ pool, const unsigned *__restrict__ idx, long long n,
                                                         int4 *__restrict__ rows, int add_free) {
    constexpr unsigned words16_brick = IS_TSDF ? 256u : 1024u;
    constexpr unsigned words16 = SUB ? words16_brick / 8u : words16_brick;
    __shared__ unsigned s_slot;
    __shared__ int s_any;
    unsigned *table = IS_TSDF ? g.tsdf_tab : g.cen_tab;
    for (long long i = blockIdx.x; i < n; i += gridDim.x) {
        const unsigned id = idx ? idx[i] : (unsigned)i;
        const unsigned brick = SUB ? id >> 3 : id;
        const size_t in_brick = SUB ? (size_t)(id & 7u) * words16 : 0;
        int4 *q = rows + (size_t)i * words16;
        if (MODE == 0) {
            const unsigned slot = brick_slot(table, brick);
            const unsigned c = (IS_TSDF && add_free && g.free_cnt) ? g.free_cnt[brick] : 0u;
            const int dq = (int)(c * 32767u), dw = (int)c;
            for (unsigned w = threadIdx.x; w < words16; w += 256) {
                int4 v = make_int4(0, 0, 0, 0);
                if (slot < SLOT_FULL) v = pool[((size_t)slot * words16_brick) + in_brick + w];
                if (IS_TSDF) { v.x += dq; v.y += dw; v.z += dq; v.w += dw; }
                q[w] = v;
            }
        } else {
            if (threadIdx.x == 0) s_any = 0;
            __syncthreads();
            int any = 0;
            for (unsigned w = threadIdx.x; w < words16; w += 256) {
                const int4 v = q[w];
                any |= (v.x | v.y | v.z | v.w) != 0;
            }
            if (any) s_any = 1;
            __syncthreads();
            if (threadIdx.x == 0) s_slot = s_any ? brick_slot_ensure(table, g.cursors + (IS_TSDF ? 0 : 2), IS_TSDF ? g.tsdf_cap : g.cen_cap, brick) : brick_slot(table, brick);
            __syncthreads();
            const unsigned slot = s_slot;
            if (slot < SLOT_FULL) {
                int4 *r = pool + (size_t)slot * words16_brick + in_brick;
                for (unsigned w = threadIdx.x; w < words16; w += 256) {
                    if (MODE == 1) {
                        r[w] = q[w];
                    } else if (IS_TSDF) {
                        int4 a = r[w];
                        const int4 b = q[w];
                        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                        r[w] = a;
                    } else {
                        ulonglong2 a = reinterpret_cast<ulonglong2 *>(r)[w];
                        const ulonglong2 b = reinterpret_cast<const ulonglong2 *>(q)[w];
                        a.x += b.x; a.y += b.y;
                        reinterpret_cast<ulonglong2 *>(r)[w] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

int launch_touched_bricks(hipStream_t s, const Grid &g, const int2 *tsdf, const unsigned long long *cen, unsigned nbricks, unsigned char *map, bool sub) {
    const unsigned nb = (nbricks + 3u) / 4u < 4096u ? (nbricks + 3u) / 4u : 4096u;
    if (sub) hipLaunchKernelGGL(touched_bricks_kernel<true>, dim3(nb ? nb : 1), dim3(256), 0, s, g, reinterpret_cast<const int4 *>(tsdf), cen, nbricks, map);
    else hipLaunchKernelGGL(touched_bricks_kernel<false>, dim3(nb ? nb : 1), dim3(256), 0, s, g, reinterpret_cast<const int4 *>(tsdf), cen, nbricks, map);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// mode 0 rows <- bricks, 1 bricks <- rows, 2 bricks += rows; idx may be null (row i = brick i)
int launch_brick_rows(hipStream_t s, const Grid &g, int mode, bool is_tsdf, void *pool, const unsigned *idx, long long n, void *rows, bool add_free, bool sub) {
    if (n <= 0) return TL3D_OK;
    const unsigned nb = (unsigned)(n < 16384 ? n : 16384);
#define TL3D_ROWS(M_, T_, S_) hipLaunchKernelGGL((brick_rows_kernel<M_, T_, S_>), dim3(nb), dim3(256), 0, s, g, static_cast<int4 *>(pool), idx, n, static_cast<int4 *>(rows), add_free ? 1 : 0)
    if (sub) {
        if (is_tsdf) { if (mode == 0) TL3D_ROWS(0, true, true); else if (mode == 1) TL3D_ROWS(1, true, true); else TL3D_ROWS(2, true, true); }
        else { if (mode == 0) TL3D_ROWS(0, false, true); else if (mode == 1) TL3D_ROWS(1, false, true); else TL3D_ROWS(2, false, true); }
    } else {
        if (is_tsdf) { if (mode == 0) TL3D_ROWS(0, true, false); else if (mode == 1) TL3D_ROWS(1, true, false); else TL3D_ROWS(2, true, false); }
        else { if (mode == 0) TL3D_ROWS(0, false, false); else if (mode == 1) TL3D_ROWS(1, false, false); else TL3D_ROWS(2, false, false); }
    }
#undef TL3D_ROWS
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// table[b] = b (a dense grid: every brick has its records)
__global__ __launch_bounds__(256) void iota_kernel(unsigned *__restrict__ t, unsigned n) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) t[i] = i;
}
int launch_iota(hipStream_t s, unsigned *t, unsigned n) {
    hipLaunchKernelGGL(iota_kernel, dim3(n / 256 + 1 < 2048 ? n / 256 + 1 : 2048), dim3(256), 0, s, t, n);
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
