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
    const double off[3] = {g.offx, g.offy, g.offz};
    const int dims[3] = {g.nx, g.ny, g.nz};
    int idx[3];
    unsigned long long q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double rc = ((double)p[a] - org[a]) / g.vsd;
        const double fl = floor(rc);
        const double loc = fl - off[a];                       // (integers below 2^53: exact) this grid's voxel, a block of the lattice
        if (!(loc >= 0.0 && loc < (double)dims[a])) return o;
        idx[a] = (int)loc;
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

// Neighbouring pixels of an image row usually land in the same voxel (a 5 mm voxel spans ~8 pixels at 1 m), so a wave
// first sums each run of adjacent lanes with equal record index and only the last lane of a run goes on (to the LDS table
// or to the grid).  Integer sums: the grid is the same bit for bit, with ~4-8x fewer atomics.
// The scan works inside rows of 16 lanes with DPP row shifts (plain vector-ALU moves): a wave-wide scan through
// ds_bpermute shuffles cost ~2 us per call -- 54 dependent LDS round trips -- which was 16 of the 45 us of a stride-1 frame;
// a run that crosses a row boundary is simply two runs.  Payload packed into five dwords for the scan (64 lanes x 4095 < 2^18,
// counts < 2^7, colour sums < 2^14).  Every lane of the wave must call this (DPP reads need the neighbours' registers live).
template <int N> __device__ __forceinline__ unsigned dpp_row_shr(unsigned v) {      // lane i of a 16-lane row <- lane i - N (0 beyond the row)
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xF, 0xF, true);
}
template <int N> __device__ __forceinline__ unsigned dpp_row_shl(unsigned v) {      // lane i <- lane i + N (0 beyond the row)
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xF, 0xF, true);
}
__device__ __forceinline__ bool centroid_reduce_runs(CenAdd &k) {
    const int li = threadIdx.x & 15;                                // lane within its row
    const unsigned rlo = (unsigned)(k.rec & 0xffffffffull), rhi = (unsigned)(k.rec >> 32);
    // (every DPP read sits in uniform control flow: a read from a lane that a branch has switched off returns 0)
    const unsigned plo = dpp_row_shr<1>(rlo), phi = dpp_row_shr<1>(rhi), nlo = dpp_row_shl<1>(rlo), nhi = dpp_row_shl<1>(rhi);
    const bool same_prev = (li != 0) & (plo == rlo) & (phi == rhi);
    const bool same_next = (li != 15) & (nlo == rlo) & (nhi == rhi);
    unsigned start = same_prev ? 0u : (unsigned)li;                 // first lane of my run = max of the head positions so far
    {
        unsigned o;
        o = dpp_row_shr<1>(start); start = max(start, o);
        o = dpp_row_shr<2>(start); start = max(start, o);
        o = dpp_row_shr<4>(start); start = max(start, o);
        o = dpp_row_shr<8>(start); start = max(start, o);
    }
    unsigned w0 = (unsigned)(k.a & 0xffffffffull), w1 = (unsigned)(k.a >> 32), w2 = (unsigned)(k.b & 0xffffffffull);
    unsigned w3 = (unsigned)(k.b >> 32) | ((unsigned)(k.c & 0xffffffffull) << 7);           // n | r << 7
    unsigned w4 = (unsigned)(k.c >> 32) | ((unsigned)k.d << 14);                             // g | b << 14
#define CEN_SCAN_STEP(N_)                                                                             \
    {                                                                                                 \
        const unsigned o0 = dpp_row_shr<N_>(w0), o1 = dpp_row_shr<N_>(w1), o2 = dpp_row_shr<N_>(w2);   \
        const unsigned o3 = dpp_row_shr<N_>(w3), o4 = dpp_row_shr<N_>(w4);                             \
        if ((unsigned)li >= start + N_) { w0 += o0; w1 += o1; w2 += o2; w3 += o3; w4 += o4; }          \
    }
    CEN_SCAN_STEP(1)
    CEN_SCAN_STEP(2)
    CEN_SCAN_STEP(4)
    CEN_SCAN_STEP(8)
#undef CEN_SCAN_STEP
    k.a = (unsigned long long)w0 | ((unsigned long long)w1 << 32);
    k.b = (unsigned long long)w2 | ((unsigned long long)(w3 & 0x7fu) << 32);
    k.c = (unsigned long long)(w3 >> 7) | ((unsigned long long)(w4 & 0x3fffu) << 32);
    k.d = (unsigned long long)(w4 >> 14);
    return !same_next && k.rec != ~0ull;                            // this lane now carries its run's sums
}

// Tails of a wave straight to the grid.  The L2's atomic units are bound by REQUESTS (an 8-B add costs a 64-B request) and
// lanes that add to consecutive words in one instruction share one: the tails are listed in LDS and lanes 4j .. 4j+3 add
// the four words of tail j's record.  stage: this wave's [64][5] words.
__device__ __forceinline__ void centroid_commit_runs(const Grid &g, CenAdd k, unsigned long long *__restrict__ grid, unsigned long long (*stage)[5],
                                                     unsigned long long *__restrict__ counters = nullptr) {
    const bool tail = centroid_reduce_runs(k);
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(tail);
    if (tail) {
        unsigned long long *e = stage[__popcll(m & ((1ull << lane) - 1ull))];
        e[0] = k.rec; e[1] = k.a; e[2] = k.b; e[3] = k.c; e[4] = k.d;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n4 = 4 * (int)__popcll(m);
    if (lane == 0 && m && counters) atomicAdd(counters + (size_t)(blockIdx.x & 255) * 8 + 2, (unsigned long long)__popcll(m));     // records updated (counted)
    for (int base = 0; base < n4; base += 64) {                     // (wave-uniform trip count: wave_slots needs every lane)
        const int i = base + lane;
        const bool want = i < n4;
        const unsigned long long rec = want ? stage[i >> 2][0] : 0ull;
        const unsigned slot = wave_slots(g.cen_tab, g.cursors + 2, g.cen_cap, (unsigned)(rec >> 9), want);      // first touch of a sparse grid's brick: a slot
        if (want && slot < SLOT_FULL) atomicAdd(grid + 4 * (((unsigned long long)slot << 9) | (rec & 511ull)) + (i & 3), stage[i >> 2][1 + (i & 3)]);
    }
}

// LDS table of the per-frame kernel: open addressing, one slot per sample of a tile (never more
// distinct keys than slots, so every probe sequence ends); key = record index + 1 (0 = empty), four 64-bit sums per slot.
// Dense sampling only (stride 1 and 2: 74 / 18 pixels per 5 mm voxel at 1 m); from stride 3 on neighbouring samples
// rarely share a voxel and the points go straight to the grid (centroid_direct_kernel).
constexpr int CEN_TW = 32, CEN_TH = 16;     // 32 frames per launch: 32 x 32 tiles (44 KB of LDS, three workgroups per CU) 9.5 us per frame,
                                            // 32 x 16 (22 KB, seven) 7.3, 32 x 8 8.0 -- smaller tiles send more partial sums to the grid

// One frame of a BATCH of accumulations (tl3d_accumulate_centroid collects up to TL3D_CEN_MAXBATCH frames of one stride per launch;
// blockIdx.y picks the frame).  One frame per launch -- 500 workgroups of a 1080p frame at the reference's stride 2 on 256 CUs, 45 KB
// of LDS each -- left the chip two thirds empty and every launch paid its ramp: 26.7 us per frame beside the TSDF batches (round
// 3).  The descriptors travel as the arguments of a tiny kernel (captured at launch, 16 per launch) and are read through the
// constant address space: a wave-uniform index, scalar loads.
typedef const CenFrame __attribute__((address_space(4))) *CenPtr;
struct CenChunk { CenFrame f[16]; };
static_assert(sizeof(CenChunk) <= 4096, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(256) void cen_desc_upload_kernel(CenChunk c, CenFrame *__restrict__ dst, int n) {
    const unsigned *src = reinterpret_cast<const unsigned *>(&c);
    unsigned *out = reinterpret_cast<unsigned *>(dst);
    const int words = n * (int)(sizeof(CenFrame) / 4);
    for (int i = threadIdx.x; i < words; i += 256) out[i] = src[i];
}
__device__ __forceinline__ void cen_frame_load(CenPtr F, BpArgs &a, PoseD &p, const float *&depth, const uint8_t *&bgr) {
    a.sub = F->a.sub; a.Ws = F->a.Ws; a.Hs = F->a.Hs; a.flags = F->a.flags;
    a.scale = F->a.scale; a.min_d = F->a.min_d; a.max_d = F->a.max_d; a.zero = F->a.zero;
#pragma unroll
    for (int i = 0; i < 9; ++i) p.r[i] = F->p.r[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) p.ct[i] = F->p.ct[i];
    depth = F->depth;
    bgr = F->bgr;
}

// xf / yf: the context's tables of (u - cx) / fx and (v - cy) / fy (the same IEEE quotients bp_pixel computes, without the
// two fp64 divisions); colour: the pixel's three bytes in one unaligned 4-byte load (frame buffers have 16 B of slack)
__device__ __forceinline__ CenAdd centroid_sample(const Cam &cam, const Grid &g, const BpArgs &a, const PoseD &p, const float *__restrict__ depth,
                                                  const uint8_t *__restrict__ bgr, const double *__restrict__ xf, const double *__restrict__ yf,
                                                  int us, int vs, bool &valid) {
    CenAdd k;
    k.rec = ~0ull;
    k.a = k.b = k.c = k.d = 0;
    valid = false;
    if (us < a.Ws && vs < a.Hs) {
        const int u = us * a.sub, v = vs * a.sub;
        const size_t pix = (size_t)v * cam.W + u;
        const float d32 = depth[pix];
        unsigned w = 0u;
        if (bgr) __builtin_memcpy(&w, bgr + 3 * pix, 4);
        float pt[3];
        if (bp_point_f(a, p, d32, xf[u], yf[v], pt)) {
            valid = true;
            k = centroid_key(g, pt, (w >> 16) & 0xffu, (w >> 8) & 0xffu, w & 0xffu);       // bytes b, g, r
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

template <int VAR>
__global__ __launch_bounds__(256) void centroid_frame_kernel(Cam cam, Grid g, const CenFrame *__restrict__ frames, const double *__restrict__ xf,
                                                             const double *__restrict__ yf, unsigned long long *__restrict__ grid,
                                                             unsigned long long *__restrict__ counters, int tiles_x) {
    BpArgs a;
    PoseD p;
    const float *depth;
    const uint8_t *bgr;
    cen_frame_load((CenPtr)frames + blockIdx.y, a, p, depth, bgr);
    constexpr int CEN_SLOTS = CEN_TW * CEN_TH;
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
        CenAdd k = centroid_sample(cam, g, a, p, depth, VAR == 4 ? nullptr : bgr, xf, yf, us, vs, valid);
        nvalid += valid ? 1 : 0;
        nkept += (k.rec != ~0ull) ? 1 : 0;
        if (VAR == 3) { if (k.rec == 12345ull) nkept += (int)k.a; continue; }      // timing ablation: samples only
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
    // One set of grid atomics per distinct voxel of the tile.  The L2's atomic units are bound by REQUESTS, not bytes (an 8-B
    // add costs a 64-B request: 1.2 M of them per 1080p frame at stride 2 were 23 of the kernel's 36 us), and lanes that add to
    // consecutive words in one instruction share a request: so the used slots are listed first and lanes 4k .. 4k+3 add the
    // four words of one record (32 B, one request instead of four).
    __shared__ unsigned s_used[CEN_SLOTS];
    __shared__ unsigned s_nused;
    if (threadIdx.x == 0) s_nused = 0u;
    __syncthreads();
    for (int h = threadIdx.x; h < CEN_SLOTS; h += 256) {
        const bool used = s_key[h] != 0ull;
        const unsigned long long m = __ballot(used);
        unsigned base = 0u;
        if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(&s_nused, (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (used) s_used[base + (unsigned)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (unsigned)h;
    }
    __syncthreads();
    if (VAR == 2) return;                                         // timing ablation: no grid atomics
    if (threadIdx.x == 0 && s_nused) atomicAdd(counters + (size_t)(blockIdx.x & 255) * 8 + 2, (unsigned long long)s_nused);      // records updated (counted)
    const unsigned n4 = 4u * s_nused;
    for (unsigned base = 0; base < n4; base += 256) {              // (wave-uniform trip count: wave_slots needs every lane)
        const unsigned i = base + threadIdx.x;
        const bool want = i < n4;
        const unsigned h = want ? s_used[i >> 2] : 0u, c = i & 3u;
        const unsigned long long rec = want ? s_key[h] - 1ull : 0ull;
        const unsigned slot = wave_slots(g.cen_tab, g.cursors + 2, g.cen_cap, (unsigned)(rec >> 9), want);      // first touch of a sparse grid's brick: a slot
        if (want && slot < SLOT_FULL) atomicAdd(grid + 4 * (((unsigned long long)slot << 9) | (rec & 511ull)) + c, s_val[h][c]);
    }
}

// sparse sampling (stride >= 3): one thread per sample in row-major order, runs combined in the wave, straight to the grid
__global__ __launch_bounds__(256) void centroid_direct_kernel(Cam cam, Grid g, const CenFrame *__restrict__ frames, const double *__restrict__ xf,
                                                              const double *__restrict__ yf, unsigned long long *__restrict__ grid,
                                                              unsigned long long *__restrict__ counters) {
    BpArgs a;
    PoseD p;
    const float *depth;
    const uint8_t *bgr;
    cen_frame_load((CenPtr)frames + blockIdx.y, a, p, depth, bgr);
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long ns = (long long)a.Ws * a.Hs;
    const int vs = (int)(s / a.Ws), us = (int)(s - (long long)vs * a.Ws);
    bool valid;
    CenAdd k = centroid_sample(cam, g, a, p, depth, bgr, xf, yf, s < ns ? us : a.Ws, vs, valid);
    centroid_stats(valid ? 1 : 0, (k.rec != ~0ull) ? 1 : 0, counters);
    __shared__ unsigned long long s_stage[4][64][5];
    centroid_commit_runs(g, k, grid, s_stage[threadIdx.x >> 6], counters);
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
    __shared__ unsigned long long s_stage[4][64][5];
    centroid_commit_runs(g, k, grid, s_stage[threadIdx.x >> 6], counters);
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

// n frames of ONE stride (a.sub, a.Ws, a.Hs equal) in one launch; `dev` holds the descriptors on the device (>= n entries)
int launch_centroid_batch(hipStream_t s, const Cam &cam, const Grid &g, int n, const CenFrame *host, CenFrame *dev, const double *xf, const double *yf,
                          unsigned long long *grid, unsigned long long *counters) {
    if (n < 1 || n > TL3D_CEN_MAXBATCH) return set_err(TL3D_E_INVALID, "bad centroid batch size %d", n);
    for (int i0 = 0; i0 < n; i0 += 16) {
        CenChunk ch;
        const int m = n - i0 < 16 ? n - i0 : 16;
        memcpy(ch.f, host + i0, (size_t)m * sizeof(CenFrame));
        if (m < 16) memset(ch.f + m, 0, (size_t)(16 - m) * sizeof(CenFrame));
        hipLaunchKernelGGL(cen_desc_upload_kernel, dim3(1), dim3(256), 0, s, ch, dev + i0, m);
        TL3D_HIP(hipGetLastError());
    }
    const BpArgs &a = host[0].a;
    if (a.sub >= 3) {
        const long long ns = (long long)a.Ws * a.Hs;
        hipLaunchKernelGGL(centroid_direct_kernel, dim3((unsigned)((ns + 255) / 256), n), dim3(256), 0, s, cam, g, dev, xf, yf, grid, counters);
    } else {
        // experiments flavour only -- TL3D_CEN_VARIANT: timing ablations (2: no grid atomics, 3: samples only, 4: no colour loads); DESIGN.md 7.5
#ifdef TL3D_EXPERIMENTS
        static const int var = getenv("TL3D_CEN_VARIANT") ? atoi(getenv("TL3D_CEN_VARIANT")) : 0;
#endif
        const int tiles_x = (a.Ws + CEN_TW - 1) / CEN_TW, tiles_y = (a.Hs + CEN_TH - 1) / CEN_TH;
        const dim3 gr((unsigned)tiles_x * (unsigned)tiles_y, n);
#define CEN_LAUNCH(V_) hipLaunchKernelGGL((centroid_frame_kernel<V_>), gr, dim3(256), 0, s, cam, g, dev, xf, yf, grid, counters, tiles_x)
#ifdef TL3D_EXPERIMENTS
        if (var == 2) CEN_LAUNCH(2);
        else if (var == 3) CEN_LAUNCH(3);
        else if (var == 4) CEN_LAUNCH(4);
        else
#endif
            CEN_LAUNCH(0);
#undef CEN_LAUNCH
    }
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// tl3d_count_bricks: the bricks a frame's samples fall into take a slot in the brick table handed in (a scratch table: no records
// exist); the table's cursor ends up as the number of distinct bricks
__global__ __launch_bounds__(256) void centroid_mark_kernel(Cam cam, Grid g, BpArgs a, PoseD p, const float *__restrict__ depth,
                                                            const double *__restrict__ xf, const double *__restrict__ yf) {
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long ns = (long long)a.Ws * a.Hs;
    const int vs = (int)(s / a.Ws), us = (int)(s - (long long)vs * a.Ws);
    bool valid;
    const CenAdd k = centroid_sample(cam, g, a, p, depth, nullptr, xf, yf, s < ns ? us : a.Ws, vs, valid);
    const bool want = k.rec != ~0ull;
    // neighbouring samples share bricks: one lane per run of equal bricks asks
    const unsigned brick = want ? (unsigned)(k.rec >> 9) : 0xffffffffu;
    const unsigned prev = __shfl_up(brick, 1);
    const bool ask = want && ((threadIdx.x & 63) == 0 || prev != brick);
    (void)wave_slots(g.cen_tab, g.cursors + 2, g.cen_cap, brick, ask);
}
int launch_centroid_mark(hipStream_t s, const Cam &cam, const Grid &g, const BpArgs &a, const PoseD &p, const float *depth, const double *xf, const double *yf) {
    const long long ns = (long long)a.Ws * a.Hs;
    hipLaunchKernelGGL(centroid_mark_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, s, cam, g, a, p, depth, xf, yf);
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
