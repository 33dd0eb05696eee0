// kernels_backproject.hip -- rows a3/a4/a5: fused projection factors + mask + pose transform + BGR->RGB,
// with order-preserving compaction (the reference's boolean fancy-indexing keeps row-major pixel order,
// depth_to_reconstruction.py:364-366, 381).  Also the u16-mm depth conversion of row a2 (D2R:85-90).
//
// Arithmetic: fp64 intermediates, exactly the reference's sequence (xf=(u-cx)/fx in fp64, x=xf*z,
// P_w = R^T P_c - R^T t, cast to f32).  MI355X fp64 VALU is full rate enough that this kernel stays
// bound by its 15 B/point of output; bytes per frame: read H*W*(4+3)/s^2, write N*15.
//
// ONE launch (bp_fused_kernel): every workgroup takes a ticket (its position in sample order), counts the survivors of
// its 2048 samples, publishes the count as an 8-byte {status, value} granule and finds its output offset by a decoupled
// look-back over its predecessors' granules (64 at a time, one per lane).  Survivors are then computed and staged
// through LDS in output order, so xyz leaves as 16-byte stores of a contiguous range
// and rgb as 16-byte stores of packed bytes (the round-1 kernel wrote 12-byte-strided floats and single bytes per lane,
// after a count kernel, a single-workgroup scan and a host round trip for the size: 77 us per 1080x1920 frame).
#include <stdlib.h>

#include "tl3d_internal.h"
#include "bp_device.h"

namespace tl3d {

__global__ __launch_bounds__(256) void u16_to_f32_kernel(const uint16_t *__restrict__ in, float *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = mm_to_m(in[i]);              // == .astype(float32) / 1000.0 (D2R:90)
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

// ---- one-launch back-projection -------------------------------------------------------------------------------
#ifndef TL3D_BP_SLEEP
#define TL3D_BP_SLEEP 1                        // x 64 cycles between two polls of a granule
#endif
// Samples per workgroup (TILE, a template parameter: 8, 4 or 2 per thread).  A tile walks a chain of ~8 dependent memory round trips
// (ticket, loads, publish, look-back, arrival, stores) whatever its size, so what a launch costs is that chain times the tiles a
// CU has to take one after another, and a level of few samples wants SMALL tiles: the reference's default stride 2 at 1080p is
// 518 k samples = 253 tiles of 2048, one workgroup on every CU and nothing beside it to hide the chain (see bp_tile_samples).
constexpr int BP_TILE_MAX = 2048;
constexpr int BP_WIN = 1024;                   // predecessors one look-back round inspects (4 per thread)
constexpr unsigned long long BP_AGG = 1ull << 62, BP_PFX = 2ull << 62, BP_VAL = (1ull << 62) - 1ull;
constexpr unsigned BP_SPIN_LIMIT = 1u << 17;   // polls of one granule (~0.1 s) before giving up (sets the error word)
// state words: [0] ticket, [1] error word, [8 + 8 g] finished tiles of shard g (g = tile % 8, one 64-byte line each),
// [72] finished shards, [BP_HDR + b] granule of tile b
constexpr int BP_HDR = 80;
constexpr unsigned BP_F_STATIC_ORDER = 0x40000000u;   // internal BpArgs.flags bit: tile = blockIdx.x (see launch_bp_fused)

// Granule of tile b = {status << 62 | value}: AGG | own count as soon as the tile has counted, PFX | inclusive prefix
// once it knows its offset.  Written by ONE 8-byte agent-scope store, polled by 8-byte agent-scope loads: the data is the
// flag, no fence needed (MI355X guide, Guideline 16, form R2).  All words are zero between launches: the buffer is zeroed
// when allocated and the tile that FINISHES last re-arms it (by then every tile has stopped polling), so a launch needs
// no memset in front of it and has no per-launch arguments (it can be captured and replayed).
template <bool WRITE, int BP_TILE>
__global__ __launch_bounds__(256) void bp_fused_kernel(Cam cam, BpArgs a, PoseD p, const float *__restrict__ depth,
                                                       const uint8_t *__restrict__ bgr, const double *__restrict__ xf,
                                                       const double *__restrict__ yf, unsigned long long *state,
                                                       float *__restrict__ xyz, uint8_t *__restrict__ rgb, unsigned long long cap,
                                                       unsigned long long *__restrict__ total_out) {
    constexpr int BP_PER = BP_TILE / 256;
    // one LDS block, carved by hand: [xyz staging | rgb staging | per-(iteration, wave) counts | reduction scratch | scalars]
    constexpr int XYZ_B = WRITE ? (3 * BP_TILE + 4) * 4 : 16, RGB_B = WRITE ? 3 * BP_TILE + 16 : 16;
    constexpr int O_WAVE = XYZ_B + RGB_B, O_RED = O_WAVE + BP_PER * 16, O_SC = O_RED + 64;
    __shared__ __attribute__((aligned(16))) unsigned char smem[O_SC + 32];
    float *const s_xyz = reinterpret_cast<float *>(smem);
    uint8_t *const s_rgb = smem + XYZ_B;
    unsigned *const s_wave = reinterpret_cast<unsigned *>(smem + O_WAVE);          // [j * 4 + w]: survivors, then exclusive prefix
    unsigned long long *const s_red = reinterpret_cast<unsigned long long *>(smem + O_RED);   // [0..3] per-wave partials, [4] result
    unsigned long long &s_excl = *reinterpret_cast<unsigned long long *>(smem + O_SC);
    unsigned &s_tile = *reinterpret_cast<unsigned *>(smem + O_SC + 8);
    unsigned &s_cnt = *reinterpret_cast<unsigned *>(smem + O_SC + 12);
    unsigned &s_flag = *reinterpret_cast<unsigned *>(smem + O_SC + 16);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    unsigned long long *const gran = state + BP_HDR;
    // A tile waits for the COUNTS of all its predecessors, which every started tile publishes without waiting for anybody.
    // So the only requirement is that a tile's predecessors have started.  Dynamic order: tiles are numbered by a ticket
    // taken at workgroup start -- unconditional, but 1000 returning atomics on one word retire at ~88 per microsecond
    // (12 us of a 1080p frame).  Static order (tile = blockIdx.x) is used when every tile of the launch fits on the chip at
    // once (then nobody can wait for a workgroup that has no slot) -- see launch_bp_fused for the fallback.
    if (tid == 0) {
        s_tile = (a.flags & BP_F_STATIC_ORDER) ? blockIdx.x : (unsigned)__hip_atomic_fetch_add(state, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_flag = 0u;
    }
    __syncthreads();
    const unsigned tile = s_tile;
    const long long ns = (long long)a.Ws * a.Hs;
    const long long base = (long long)tile * BP_TILE;
    // ---- phase 1: this thread's samples (sample = base + j * 256 + tid): depth AND colour loads issued together (the
    // colours of rejected pixels are loaded too: one round trip instead of two on the tile's critical path), validity,
    // survivors per wave
    unsigned long long ball[BP_PER];
    float dval[BP_PER];
    unsigned col[BP_PER];
    const long long s0 = base + tid;
    const int v0 = (int)(s0 / a.Ws), u0 = (int)(s0 - (long long)v0 * a.Ws);
    {
        int v = v0, u = u0;
#pragma unroll
        for (int j = 0; j < BP_PER; ++j) {
            const long long sj = base + (long long)j * 256 + tid;
            const bool in = sj < ns;
            const size_t pix = in ? (size_t)(v * a.sub) * cam.W + (size_t)(u * a.sub) : 0;
            dval[j] = depth[pix];
            col[j] = 0u;
            if (WRITE && bgr) {
                unsigned w;                                // the pixel's b, g, r bytes in one unaligned load (frame buffers have 16 B of slack)
                __builtin_memcpy(&w, bgr + 3 * pix, 4);
                col[j] = ((w >> 16) & 0xffu) | (w & 0xff00u) | ((w & 0xffu) << 16);                      // r | g << 8 | b << 16
            }
            u += 256;
            while (u >= a.Ws) { u -= a.Ws; ++v; }
        }
#pragma unroll
        for (int j = 0; j < BP_PER; ++j) {
            const long long sj = base + (long long)j * 256 + tid;
            ball[j] = __ballot(sj < ns && bp_valid_value(a, dval[j]));
            if (lane == 0) s_wave[j * 4 + wid] = (unsigned)__popcll(ball[j]);
        }
    }
    __syncthreads();
    if (wid == 0) {                            // exclusive scan of the 32 per-(iteration, wave) counts, in sample order
        const unsigned c = lane < BP_PER * 4 ? s_wave[lane] : 0u;
        unsigned inc = c;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const unsigned t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane < BP_PER * 4) s_wave[lane] = inc - c;
        if (lane == BP_PER * 4 - 1) {
            s_cnt = inc;
            s_excl = 0ull;
            __hip_atomic_store(gran + tile, (tile == 0 ? BP_PFX : BP_AGG) | (unsigned long long)inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();                           // s_cnt and the scanned s_wave are visible to every wave
    const unsigned cnt_tile = s_cnt;
    // ---- look-back, by the whole workgroup: thread t inspects the predecessors at distance t, t + 256, t + 512, t + 768
    // (distance 0 = the tile just in front); counts are summed up to and including the nearest published prefix.  With
    // every tile of a frame resident at once nobody holds a prefix early, so a round has to be wide: 1024 predecessors
    // per round cost one load latency, where a 64-wide window would walk 16 rounds for the last tile of a 1080p frame.
    if (tile != 0) {
        unsigned long long excl = 0ull;
        long long j = (long long)tile - 1;
        for (;;) {
            unsigned long long g[4];
            unsigned near_pfx = 0xffffffffu;                                     // smallest distance with a prefix, this thread
            bool failed = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long mine = j - (long long)(k * 256 + tid);
                g[k] = BP_PFX;                                                   // in front of tile 0: prefix 0
                if (mine >= 0) {
                    unsigned spins = 0;
                    for (;;) {
                        // an agent-scope load is served by this XCD's L2, which may go on returning the copy it fetched before
                        // the predecessor (on another XCD) published -- seen in the batched ICP kernel, for seconds, on an otherwise
                        // quiet chip.  After a few misses the poll becomes a read-modify-write (add of zero),
                        // which executes at the memory side and always returns the current word.
                        g[k] = spins < 16u ? __hip_atomic_load(gran + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : __hip_atomic_fetch_add(gran + mine, a.zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((g[k] >> 62) != 0ull) break;
                        // every spin ends: a bound, and one tile's time-out ends every other tile's wait at once
                        if (++spins > BP_SPIN_LIMIT) { failed = true; break; }
                        if ((spins & 255u) == 0u && __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { failed = true; break; }
                        __builtin_amdgcn_s_sleep(TL3D_BP_SLEEP);
                    }
                }
                if ((g[k] >> 62) == 2ull && near_pfx == 0xffffffffu) near_pfx = (unsigned)(k * 256 + tid);
            }
            if (failed) {
                __hip_atomic_store(state + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_flag = 1u;
            }
            // nearest prefix over the workgroup
            unsigned m = near_pfx;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, d));
            if (lane == 0) s_red[wid] = m;
            __syncthreads();
            const unsigned stop = (unsigned)min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
            __syncthreads();
            unsigned long long v = 0ull;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((unsigned)(k * 256 + tid) <= stop) v += g[k] & BP_VAL;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
            if (lane == 0) s_red[wid] = v;
            __syncthreads();
            excl += (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
            const bool give_up = s_flag != 0u;
            __syncthreads();
            if (stop != 0xffffffffu || give_up) break;                           // uniform over the workgroup
            j -= BP_WIN;
        }
        if (tid == 0) {
            // the inclusive prefix goes out at once (also after a time-out, so that every successor ends too)
            __hip_atomic_store(gran + tile, BP_PFX | ((excl + cnt_tile) & BP_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
        }
        __syncthreads();
    }
    const unsigned long long excl = s_excl;
    if (tile == gridDim.x - 1u && tid == 0) *total_out = excl + cnt_tile;       // the last tile in sample order knows the total
    // ---- done polling.  The tile that gets here last re-arms the scratch for the next launch (nobody polls any more; the
    // others may still be staging and writing, which touches none of it).  "Last" through a two-level count: 8 shard
    // counters on lines of their own (tile % 8), then one counter of finished shards -- a single word would serialise
    // 1000 returning atomics (6.5 us of a 1080p frame).  Placed here, the count's round trip runs under phase 2.
    const unsigned shard = tile & 7u, in_shard = (gridDim.x - shard + 7u) >> 3;
    unsigned long long arrived = 0ull;         // tid 0: issued here, looked at after phase 2
    if (tid == 0) arrived = __hip_atomic_fetch_add(state + 8 + 8 * shard, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // points this tile may write: [excl, min(excl + cnt, cap))
    const unsigned long long room = cap > excl ? cap - excl : 0ull;
    const unsigned nw = (unsigned)(room < (unsigned long long)cnt_tile ? room : (unsigned long long)cnt_tile);
    if (WRITE && nw != 0) {
        // ---- phase 2: compute the survivors and stage them in output order.  LDS element i <-> global element
        // (g0 - phase + i): a 16-byte chunk of an LDS array is 16-byte aligned in global memory too, whatever the offset
        // and the caller's base alignment
        const unsigned long long g0 = 3ull * excl;                               // first dword of xyz / first byte of rgb
        const unsigned xphase = (unsigned)((g0 + ((reinterpret_cast<uintptr_t>(xyz) >> 2) & 3u)) & 3ull);
        const unsigned bphase = (unsigned)((g0 + (reinterpret_cast<uintptr_t>(rgb) & 15u)) & 15ull);
        const unsigned long long below = (1ull << lane) - 1ull;
        int v = v0, u = u0;
#pragma unroll
        for (int j = 0; j < BP_PER; ++j) {
            if ((ball[j] >> lane) & 1ull) {
                const unsigned r = s_wave[j * 4 + wid] + (unsigned)__popcll(ball[j] & below);
                float pt[3] = {0.0f, 0.0f, 0.0f};
                // xf[u] = (u - cx) / fx and yf[v] = (v - cy) / fy come from tables (the reference caches the same two maps,
                // D2R:287-295): the same IEEE quotients, without two fp64 divisions per pixel
                (void)bp_point_f(a, p, dval[j], xf[u * a.sub], yf[v * a.sub], pt);
                s_xyz[xphase + 3 * r + 0] = pt[0];
                s_xyz[xphase + 3 * r + 1] = pt[1];
                s_xyz[xphase + 3 * r + 2] = pt[2];
                s_rgb[bphase + 3 * r + 0] = (uint8_t)(col[j] & 0xffu);
                s_rgb[bphase + 3 * r + 1] = (uint8_t)((col[j] >> 8) & 0xffu);
                s_rgb[bphase + 3 * r + 2] = (uint8_t)((col[j] >> 16) & 0xffu);
            }
            u += 256;
            while (u >= a.Ws) { u -= a.Ws; ++v; }
        }
        __syncthreads();
        {   // xyz: dwords [xphase, xphase + 3 nw) of s_xyz; gx + i is the home of s_xyz[i]
            float *__restrict__ gx = xyz + g0 - xphase;
            const unsigned lo = xphase, hi = xphase + 3u * nw;
            const unsigned body_lo = (lo + 3u) & ~3u, body_hi = hi & ~3u;
            if (body_lo < body_hi) {
                for (unsigned i = lo + tid; i < body_lo; i += 256) gx[i] = s_xyz[i];
                for (unsigned c = body_lo / 4u + tid; c < body_hi / 4u; c += 256)
                    *reinterpret_cast<float4 *>(gx + 4u * c) = *reinterpret_cast<const float4 *>(s_xyz + 4u * c);
                for (unsigned i = body_hi + tid; i < hi; i += 256) gx[i] = s_xyz[i];
            } else {
                for (unsigned i = lo + tid; i < hi; i += 256) gx[i] = s_xyz[i];
            }
        }
        {   // rgb: bytes [bphase, bphase + 3 nw) of s_rgb; gb + i is the home of s_rgb[i]
            uint8_t *__restrict__ gb = rgb + g0 - bphase;
            const unsigned lo = bphase, hi = bphase + 3u * nw;
            const unsigned body_lo = (lo + 15u) & ~15u, body_hi = hi & ~15u;
            if (body_lo < body_hi) {
                for (unsigned i = lo + tid; i < body_lo; i += 256) gb[i] = s_rgb[i];
                for (unsigned c = body_lo / 16u + tid; c < body_hi / 16u; c += 256)
                    *reinterpret_cast<uint4 *>(gb + 16u * c) = *reinterpret_cast<const uint4 *>(s_rgb + 16u * c);
                for (unsigned i = body_hi + tid; i < hi; i += 256) gb[i] = s_rgb[i];
            } else {
                for (unsigned i = lo + tid; i < hi; i += 256) gb[i] = s_rgb[i];
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned last = 0u;
        if (arrived == (unsigned long long)in_shard - 1ull) {
            const unsigned shards = gridDim.x < 8u ? gridDim.x : 8u;
            if (__hip_atomic_fetch_add(state + 72, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)shards - 1ull) last = 2u;
        }
        s_flag = last;
    }
    __syncthreads();
    if (s_flag == 2u) {
        for (unsigned i = tid; i < gridDim.x; i += 256) __hip_atomic_store(gran + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 8) __hip_atomic_store(state + 8 + 8 * tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(state, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(state + 72, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// min / max of a frame's back-projected points without materialising them (the scene-bounding pass of the pipeline used
// to read every frame's point list back to the host just to take its extent): per-workgroup partials, reduced by the host.
__global__ __launch_bounds__(256) void bp_bounds_kernel(Cam cam, BpArgs a, PoseD p, const float *__restrict__ depth,
                                                        const double *__restrict__ xf, const double *__restrict__ yf,
                                                        float *__restrict__ slab) {
    __shared__ float sm[4][6];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const long long ns = (long long)a.Ws * a.Hs;
    for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < ns; s += (long long)gridDim.x * 256) {
        const int vs = (int)(s / a.Ws), us = (int)(s - (long long)vs * a.Ws);
        const int u = us * a.sub, v = vs * a.sub;
        float pt[3];
        if (bp_point_f(a, p, depth[(size_t)v * cam.W + u], xf[u], yf[v], pt)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], pt[k]); mx[k] = fmaxf(mx[k], pt[k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int d = 32; d > 0; d >>= 1) {
            mn[k] = fminf(mn[k], __shfl_down(mn[k], d));
            mx[k] = fmaxf(mx[k], __shfl_down(mx[k], d));
        }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 3; ++k) { sm[wid][k] = mn[k]; sm[wid][3 + k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sm[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, sm[w][threadIdx.x]) : fmaxf(v, sm[w][threadIdx.x]);
        slab[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

int launch_bp_bounds(hipStream_t s, const Cam &cam, const BpArgs &a, const PoseD &p, const float *depth, const double *xf, const double *yf,
                     float *slab, int nblocks) {
    hipLaunchKernelGGL(bp_bounds_kernel, dim3(nblocks), dim3(256), 0, s, cam, a, p, depth, xf, yf, slab);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// tile size of a launch: a function of the number of samples only (the output does not depend on it: order-preserving either way)
int bp_tile_samples(const BpArgs &a) {
#ifdef TL3D_EXPERIMENTS
    if (const char *e = getenv("TL3D_BP_TILE")) {
        const int t = atoi(e);
        if (t == 512 || t == 1024 || t == 2048) return t;
    }
#endif
    const long long ns = (long long)a.Ws * a.Hs;
    return ns > (3ll << 19) ? 2048 : (ns > (3ll << 17) ? 1024 : 512);        // > 1.5 M samples: 2048; > 393 k: 1024; else 512
}

int bp_fused_tiles(const BpArgs &a) {
    const long long ns = (long long)a.Ws * a.Hs;
    const long long tile = bp_tile_samples(a);
    const long long nt = (ns + tile - 1) / tile;
    return (int)(nt < 1 ? 1 : nt);
}

int bp_state_words(const BpArgs &a) {               // sized for the smallest tile: a context's scratch serves every launch shape
    const long long ns = (long long)a.Ws * a.Hs;
    return BP_HDR + (int)((ns + 511) / 512) + 1;
}

// state: device buffer of at least bp_state_words(a) 64-bit words, ALL ZERO (zeroed once by the owner; the kernel leaves it
// zero again, except the error word [1]); total_out: device word that receives the count
int launch_bp_fused(hipStream_t s, const Cam &cam, const BpArgs &a, const PoseD &p, const float *depth, const uint8_t *bgr,
                    const double *xf, const double *yf, unsigned long long *state, float *xyz, uint8_t *rgb, unsigned long long cap,
                    unsigned long long *total_out, bool force_dynamic) {
    const int nt = bp_fused_tiles(a);
    // Static tile order while every tile of the launch can be resident at once (>= 4 workgroups of this kernel fit a CU by
    // registers and LDS; 256 CUs): no tile can then be kept off the chip by tiles that wait for it.  If other work holds
    // slots AND workgroups were dispatched out of order, a wait could starve: it is bounded, ends in the error word, and
    // the blocking entry point then repeats the call in dynamic order (force_dynamic).  (Experiments flavour: TL3D_BP_ORDER=dynamic|static pins it.)
#ifdef TL3D_EXPERIMENTS
    static const int pin = getenv("TL3D_BP_ORDER") ? (getenv("TL3D_BP_ORDER")[0] == 'd' ? 1 : 2) : 0;
#else
    constexpr int pin = 0;
#endif
    BpArgs ad = a;
    const int tile = bp_tile_samples(a);
    const int ti = tile == 2048 ? 0 : (tile == 1024 ? 1 : 2);
    using Kern = void (*)(Cam, BpArgs, PoseD, const float *, const uint8_t *, const double *, const double *, unsigned long long *, float *, uint8_t *,
                          unsigned long long, unsigned long long *);
    static const Kern kw[3] = {bp_fused_kernel<true, 2048>, bp_fused_kernel<true, 1024>, bp_fused_kernel<true, 512>};
    static const Kern kc[3] = {bp_fused_kernel<false, 2048>, bp_fused_kernel<false, 1024>, bp_fused_kernel<false, 512>};
    // how many workgroups of this kernel the device holds at once, from the runtime (registers, LDS, CU count), not a constant
    static int resident[3] = {-1, -1, -1};
    if (resident[ti] < 0) {
        int per_cu = 0, cus = 0, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kw[ti], 256, 0) == hipSuccess && per_cu > 0 && cus > 0)
            resident[ti] = per_cu * cus;
        else {
            (void)hipGetLastError();
            resident[ti] = 0;                              // unknown: always dynamic order
        }
    }
    const bool stat = pin == 2 || (pin == 0 && !force_dynamic && nt <= resident[ti]);
    if (stat) ad.flags |= BP_F_STATIC_ORDER;
    if (xyz && rgb)
        hipLaunchKernelGGL(kw[ti], dim3(nt), dim3(256), 0, s, cam, ad, p, depth, bgr, xf, yf, state, xyz, rgb, cap, total_out);
    else
        hipLaunchKernelGGL(kc[ti], dim3(nt), dim3(256), 0, s, cam, ad, p, depth, bgr, xf, yf, state, (float *)nullptr, (uint8_t *)nullptr, 0ull, total_out);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_u16_to_f32(hipStream_t s, const uint16_t *in, float *out, size_t n) {
    if (n == 0) return TL3D_OK;
    hipLaunchKernelGGL(u16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_scan(hipStream_t s, const unsigned *counts, unsigned long long *offsets, int n, unsigned long long *total) {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, counts, offsets, n, total);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

}  // namespace tl3d
