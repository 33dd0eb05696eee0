// kernels_tsdf.hip -- row a11: TSDF integration of one depth frame into the brick-major grid.
// No reference code exists for this row; the convention is the oracle's
// (oracle/tl3d_oracle.c: orc_tsdf_integrate) and both sides evaluate the same f32 sequence, so the integer
// grid {sum of rint(tsdf*32767), weight} is bit-identical.
//
// Launches per frame, all on the ctx stream:
//   1. depth_tiles_kernel     32x32-pixel tiles -> (min, max, all-valid) of the valid scaled depth   4 B/pixel read
//   2. tile_pyramid_kernel    1 workgroup of 4 waves: 2x2 reductions of the tiles up to a single tile
//   3. brick_cull_kernel      one lane per 8^3 brick (one wave per 4x4x4-brick cell) classifies it from <= 16
//                             pyramid lookups:
//        SKIP   outside the view, no valid depth under it, or more than trunc behind every surface it can see
//        FREE   wholly inside the image, every pixel under it valid, and at least trunc in front of every
//               surface: every voxel gets exactly tsdf = 1 (q = 32767), so no depth lookup is needed
//        MIXED  everything else (near a surface, on the image border, straddling the camera plane)
//      and appends it to a compact list (MIXED from the front, FREE from the back).
//   4. tsdf_integrate_kernel  one 64-lane wave per listed brick (4 KB of records, contiguous; a lane owns
//        4 x 16 B = 8 voxels, a wave-instruction moves 1 KB).  FREE bricks are a pure streaming
//        read-modify-write; MIXED bricks run the per-voxel rule (project, gather depth, truncate) and load /
//        store only the 16-B pairs that change.
// Every classification is conservative with respect to the per-voxel rule, so the result is the oracle's bit
// for bit whatever the view.  The list balances the work (a static brick->block map leaves most of the chip idle:
// the frustum covers a fraction of the grid).
// Algorithmic bytes per launch = 8 B x (records read + written), counted by the kernel in counting mode
// ("counted, never estimated"), + 4 B x H x W for the depth frame.
#include <stdlib.h>

#include "tl3d_internal.h"
#include "bp_device.h"

namespace tl3d {

struct TsdfConst {
    float mind, maxd, sc, wlim, hlim;
    int xcd_group;
    int free_counted;            // FREE bricks were counted in the per-brick free-space counters by the classification: not listed
};

constexpr int TILE = 32;
constexpr int TILE_SHIFT = 5;
constexpr int MAX_LEVELS = 12;
constexpr unsigned FREE_FLAG = 0x80000000u;
constexpr unsigned XCD_GROUPS = 8;

struct Pyramid {
    int nlev;
    int ntx[MAX_LEVELS], nty[MAX_LEVELS], off[MAX_LEVELS];     // per level: tiles in x / y, offset into the float4 array
};

// per-frame arguments of the prep kernels: one launch prepares one frame or two (blockIdx.y picks the frame), since the
// prep chains of later frames crawl beside the update kernels and their launches, not their work, are what costs
struct PrepFrame {
    const void *depth;
    float4 *tiles;
    unsigned *list, *list_counts;
    unsigned char *cls;
    TsdfConst c;
    PoseF pose;
};
struct PrepFrames { PrepFrame f[2]; };

// tile = (dmin, dmax, allvalid ? 1 : 0, unused)

// Depth source of the TSDF kernels: the f32 frame, or the 16-bit millimetre image it was converted from (uploads of kind
// TL3D_DEPTH_U16_MM keep it): same value through the same conversion as u16_to_f32_kernel (mm_to_m), half the bytes per pixel and
// so half the cache lines under a brick's footprint (tools/ubench_mixed.hip: the gather half 17.9 -> 10.0 us).
__device__ __forceinline__ float ld_depth(const float *__restrict__ p, size_t i) { return p[i]; }
__device__ __forceinline__ float ld_depth(const uint16_t *__restrict__ p, size_t i) { return mm_to_m(p[i]); }
__device__ __forceinline__ void ld_depth4(const float *__restrict__ p, size_t i, float dd[4]) {
    const float4 t4 = *reinterpret_cast<const float4 *>(p + i);
    dd[0] = t4.x; dd[1] = t4.y; dd[2] = t4.z; dd[3] = t4.w;
}
__device__ __forceinline__ void ld_depth4(const uint16_t *__restrict__ p, size_t i, float dd[4]) {
    const ushort4 t4 = *reinterpret_cast<const ushort4 *>(p + i);
    dd[0] = mm_to_m(t4.x); dd[1] = mm_to_m(t4.y); dd[2] = mm_to_m(t4.z); dd[3] = mm_to_m(t4.w);
}

// ---- 1. depth tiles --------------------------------------------------------------------------------------
template <typename DT>
__global__ __launch_bounds__(256) void depth_tiles_kernel(Cam cam, PrepFrames P, int ntx, int nty) {
    const PrepFrame &F = P.f[blockIdx.y];
    const TsdfConst c = F.c;
    const DT *__restrict__ depth = static_cast<const DT *>(F.depth);
    float4 *__restrict__ tiles = F.tiles;
    unsigned *__restrict__ list_counts = F.list_counts;
    __shared__ float smin[4], smax[4];
    __shared__ int sbad[4];
    if (blockIdx.x == 0 && threadIdx.x < 2) list_counts[threadIdx.x] = 0u;      // reset the brick-list cursors
    // one 32x32-pixel tile per workgroup iteration: 8 lanes x 16 B cover a tile row, 32 rows -> 256 threads
    const int q4 = threadIdx.x & 7, row = threadIdx.x >> 3;
    const bool vec = (cam.W & 3) == 0;                              // rows are 16-B aligned
    for (int tile = blockIdx.x; tile < ntx * nty; tile += gridDim.x) {
        const int tx = tile % ntx, ty = tile / ntx;
        float mn = INFINITY, mx = -INFINITY;
        int bad = 0;
        const int u0 = tx * TILE + q4 * 4, v = ty * TILE + row;
        if (v < cam.H && u0 < cam.W) {
            float dd[4];
            int nv = min(4, cam.W - u0);
            if (vec) {
                ld_depth4(depth, (size_t)v * cam.W + u0, dd);
            } else {
                for (int k = 0; k < 4; ++k) dd[k] = (k < nv) ? ld_depth(depth, (size_t)v * cam.W + u0 + k) : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nv) {
                    const float d = dd[k] * c.sc;
                    if (d > c.mind && d < c.maxd) {
                        mn = fminf(mn, d);
                        mx = fmaxf(mx, d);
                    } else {
                        bad = 1;
                    }
                }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            mn = fminf(mn, __shfl_down(mn, d));
            mx = fmaxf(mx, __shfl_down(mx, d));
            bad |= __shfl_down(bad, d);
        }
        if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; sbad[threadIdx.x >> 6] = bad; }
        __syncthreads();
        if (threadIdx.x == 0)
            tiles[tile] = make_float4(fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3])),
                                      fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])),
                                      (sbad[0] | sbad[1] | sbad[2] | sbad[3]) ? 0.0f : 1.0f, 0.0f);
        __syncthreads();
    }
}

// ---- 2. pyramid ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_pyramid_kernel(Pyramid py, PrepFrames P) {
    float4 *__restrict__ tiles = P.f[blockIdx.x].tiles;
    for (int L = 1; L < py.nlev; ++L) {
        const int n = py.ntx[L] * py.nty[L];
        const float4 *__restrict__ src = tiles + py.off[L - 1];
        float4 *__restrict__ dst = tiles + py.off[L];
        for (int i = threadIdx.x; i < n; i += 256) {
            const int x = i % py.ntx[L], y = i / py.ntx[L];
            float4 r = make_float4(INFINITY, -INFINITY, 1.0f, 0.0f);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int sx = 2 * x + dx, sy = 2 * y + dy;
                    if (sx < py.ntx[L - 1] && sy < py.nty[L - 1]) {
                        const float4 s = src[sy * py.ntx[L - 1] + sx];
                        r.x = fminf(r.x, s.x);
                        r.y = fmaxf(r.y, s.y);
                        r.z = fminf(r.z, s.z);
                    }
                }
            dst[i] = r;
        }
        __syncthreads();        // level L complete (and visible to the block) before level L+1 reads it
    }
}

// ---- 3. brick classification ------------------------------------------------------------------------------
// One wave per cell of 4x4x4 bricks, one lane per brick.  A cell whose bounding sphere misses the view exits after
// a handful of instructions (most of the grid); the four waves of a workgroup pool their survivors so the two list
// cursors see one atomic per workgroup, not per wave (same-address returning atomics retire at ~90 per microsecond).
// Margins: 1.5 px on projected bounds, 1 % of a voxel on depths, 0.1 % on the truncation distance.
__device__ __forceinline__ bool sphere_in_view(const Frustum &fr, float x, float y, float z, float rad) {
    return (z + rad > 0.0f) && (fr.lx * x + fr.lz * z >= -rad) && (fr.rx * x + fr.rz * z >= -rad) &&
           (fr.ty * y + fr.tz * z >= -rad) && (fr.by * y + fr.bz * z >= -rad);
}

__global__ __launch_bounds__(256) void brick_cull_kernel(Cam cam, Grid g, PrepFrames P, Frustum fr, Pyramid py,
                                                         unsigned *__restrict__ free_cnt) {
    const PrepFrame &F = P.f[blockIdx.y];
    const PoseF pose = F.pose;
    const float4 *__restrict__ tiles = F.tiles;
    unsigned *__restrict__ list = F.list;
    unsigned *__restrict__ list_counts = F.list_counts;
    unsigned char *__restrict__ cls_map = F.cls;
    __shared__ unsigned s_cnt[4][2];
    __shared__ unsigned s_base[2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int ncx = (g.nbx + 3) >> 2, ncy = (g.nby + 3) >> 2, ncz = (g.nbz + 3) >> 2;
    const int cell = blockIdx.x * 4 + wid;
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    int cls = 0;                                                  // 0 skip, 1 mixed, 2 free
    int brick = 0;
    bool in_grid = false;
    if (cell < ncx * ncy * ncz) {
        const int ccx = cell % ncx, ccy = (cell / ncx) % ncy, ccz = cell / (ncx * ncy);
        {
            const int bx0 = ccx * 4 + (lane & 3), by0 = ccy * 4 + ((lane >> 2) & 3), bz0 = ccz * 4 + (lane >> 4);
            in_grid = bx0 < g.nbx && by0 < g.nby && bz0 < g.nbz;
            brick = (bz0 * g.nby + by0) * g.nbx + bx0;
        }
        const float crad = 27.712812f * g.vs * 1.01f;               // half diagonal of a 32^3-voxel cell, +1 %
        const float qx = fmaf((float)(ccx * 32 + 16), g.vs, g.ox), qy = fmaf((float)(ccy * 32 + 16), g.vs, g.oy);
        const float qz = fmaf((float)(ccz * 32 + 16), g.vs, g.oz);
        const float ex = pose.r[0] * qx + pose.r[1] * qy + pose.r[2] * qz + pose.t[0];
        const float ey = pose.r[3] * qx + pose.r[4] * qy + pose.r[5] * qz + pose.t[1];
        const float ez = pose.r[6] * qx + pose.r[7] * qy + pose.r[8] * qz + pose.t[2];
        const int bx = ccx * 4 + (lane & 3), by = ccy * 4 + ((lane >> 2) & 3), bz = ccz * 4 + (lane >> 4);
        if (sphere_in_view(fr, ex, ey, ez, crad) && in_grid) {
            const float rad = 6.9282032f * g.vs * 1.01f;             // half diagonal of a brick, +1 %
            const float wx = fmaf((float)(bx * 8 + 4), g.vs, g.ox);
            const float wy = fmaf((float)(by * 8 + 4), g.vs, g.oy);
            const float wz = fmaf((float)(bz * 8 + 4), g.vs, g.oz);
            const float cxm = pose.r[0] * wx + pose.r[1] * wy + pose.r[2] * wz + pose.t[0];
            const float cym = pose.r[3] * wx + pose.r[4] * wy + pose.r[5] * wz + pose.t[1];
            const float czm = pose.r[6] * wx + pose.r[7] * wy + pose.r[8] * wz + pose.t[2];
            if (sphere_in_view(fr, cxm, cym, czm, rad)) {
                cls = 1;
                if (czm - rad > 1e-3f) {
                    // all 8 corners are in front of the camera: the voxel centres project inside the corners' pixel box
                    const float h = 4.0f * g.vs;
                    float umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float sx = (k & 1) ? h : -h, sy = (k & 2) ? h : -h, sz = (k & 4) ? h : -h;
                        const float x = cxm + pose.r[0] * sx + pose.r[1] * sy + pose.r[2] * sz;
                        const float y = cym + pose.r[3] * sx + pose.r[4] * sy + pose.r[5] * sz;
                        const float z = czm + pose.r[6] * sx + pose.r[7] * sy + pose.r[8] * sz;
                        const float iz = __builtin_amdgcn_rcpf(z);     // 1 ulp; the 1.5 px margin absorbs it
                        const float u = cam.fx * x * iz + cam.cx, v = cam.fy * y * iz + cam.cy;
                        umin = fminf(umin, u); umax = fmaxf(umax, u);
                        vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
                        zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
                    }
                    umin -= 1.5f; umax += 1.5f; vmin -= 1.5f; vmax += 1.5f;
                    if (umax < 0.0f || vmax < 0.0f || umin > (float)(cam.W - 1) || vmin > (float)(cam.H - 1)) {
                        cls = 0;
                    } else {
                        const bool inside = umin >= 0.0f && vmin >= 0.0f && umax <= (float)(cam.W - 1) && vmax <= (float)(cam.H - 1);
                        const int pu0 = max(0, (int)floorf(umin)), pu1 = min(cam.W - 1, (int)ceilf(umax));
                        const int pv0 = max(0, (int)floorf(vmin)), pv1 = min(cam.H - 1, (int)ceilf(vmax));
                        // finest level at which the pixel box spans at most 4 tiles per axis: 16 independent lookups
                        int L = 0;
                        while (L < py.nlev - 1 && (((pu1 >> (TILE_SHIFT + L)) - (pu0 >> (TILE_SHIFT + L))) > 3 ||
                                                   ((pv1 >> (TILE_SHIFT + L)) - (pv0 >> (TILE_SHIFT + L))) > 3))
                            ++L;
                        const int tu0 = pu0 >> (TILE_SHIFT + L), tu1 = pu1 >> (TILE_SHIFT + L);
                        const int tv0 = pv0 >> (TILE_SHIFT + L), tv1 = pv1 >> (TILE_SHIFT + L);
                        const float4 *__restrict__ lv = tiles + py.off[L];
                        const int nt = py.ntx[L];
                        float4 a = make_float4(INFINITY, -INFINITY, 1.0f, 0.0f);
#pragma unroll
                        for (int dv = 0; dv < 4; ++dv)
#pragma unroll
                            for (int du = 0; du < 4; ++du) {
                                const float4 b = lv[min(tv0 + dv, tv1) * nt + min(tu0 + du, tu1)];   // clamped: repeats are harmless
                                a.x = fminf(a.x, b.x);
                                a.y = fmaxf(a.y, b.y);
                                a.z = fminf(a.z, b.z);
                            }
                        const float m = 0.01f * g.vs;
                        if (!(a.y > -INFINITY) || (zmin - m > a.y + g.trunc)) {
                            cls = 0;        // no valid depth under the brick, or the brick is > trunc behind all it can see
                        } else if (inside && a.z > 0.5f && (a.x - (zmax + m) >= g.trunc * 1.001f)) {
                            cls = 2;        // every voxel: in image, valid depth, sdf >= trunc  =>  tsdf == 1 exactly
                        }
                    }
                }
            }
        }
    }
    // the class of EVERY brick of the grid, for the kernel that updates two frames per visit (it asks whether a brick on one
    // frame's list is on the other's too)
    if (in_grid) cls_map[brick] = (unsigned char)cls;
    const unsigned long long mm = __ballot(cls == 1), mf = __ballot(cls == 2);
    if (lane == 0) { s_cnt[wid][0] = (unsigned)__popcll(mm); s_cnt[wid][1] = (unsigned)__popcll(mf); }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned tot = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        s_base[threadIdx.x] = tot ? atomicAdd(list_counts + threadIdx.x, tot) : 0u;
    }
    __syncthreads();
    unsigned bm = s_base[0], bf = s_base[1];
    for (int w = 0; w < wid; ++w) { bm += s_cnt[w][0]; bf += s_cnt[w][1]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (cls == 1) list[bm + __popcll(mm & below)] = (unsigned)brick;
    if (cls == 2) {
        // Free space: every voxel of the brick gets exactly (+32767, +1).  With per-brick counters that is ONE integer add
        // here instead of a 4 KB read + 4 KB write by the update kernel; the counters are folded into the records before
        // anything reads them (fold_free_kernel).  Without counters the brick goes on the list (from the back).
        if (free_cnt) atomicAdd(free_cnt + brick, 1u);
        else list[nbricks - 1u - (bf + __popcll(mf & below))] = (unsigned)brick | FREE_FLAG;
    }
}

// records += count x (32767, 1) for every brick with a pending free-space count; the count returns to zero.  One wave per
// brick, 16 B per lane.  Runs on the main stream before anything reads the TSDF channel (download, merge, extraction,
// weight check), i.e. once per scan, not per frame.
__global__ __launch_bounds__(256) void fold_free_kernel(int2 *__restrict__ grid, unsigned *__restrict__ free_cnt, unsigned nbricks) {
    const int lane = threadIdx.x & 63;
    for (unsigned b = blockIdx.x * 4u + (threadIdx.x >> 6); b < nbricks; b += gridDim.x * 4u) {
        const unsigned c = __builtin_amdgcn_readfirstlane(free_cnt[b]);
        if (c == 0u) continue;
        int4 *__restrict__ recs = reinterpret_cast<int4 *>(grid + ((size_t)b << 9));
        const int dq = (int)(c * 32767u), dw = (int)c;
        int4 r0 = recs[lane], r1 = recs[64 + lane], r2 = recs[128 + lane], r3 = recs[192 + lane];
        r0.x += dq; r0.y += dw; r0.z += dq; r0.w += dw;
        r1.x += dq; r1.y += dw; r1.z += dq; r1.w += dw;
        r2.x += dq; r2.y += dw; r2.z += dq; r2.w += dw;
        r3.x += dq; r3.y += dw; r3.z += dq; r3.w += dw;
        recs[lane] = r0; recs[64 + lane] = r1; recs[128 + lane] = r2; recs[192 + lane] = r3;
        if (lane == 0) free_cnt[b] = 0u;
    }
}

// ---- 4. integration --------------------------------------------------------------------------------------
// Per-voxel rule, split in two so that a wave can issue all its depth gathers before it needs any of them.
// project(): camera-space voxel centre -> clamped pixel address + "could update" flag (no memory access).
// finish():  depth value -> quantised tsdf + final flag.  Together they are exactly orc_tsdf_integrate's sequence.
__device__ __forceinline__ bool tsdf_project(const Cam &cam, const TsdfConst &c, float xc, float yc, float zc, int &pix) {
    bool ok = zc > 0.0f;
    const float inv = 1.0f / zc;
    const float uf = fmaf(cam.fx * xc, inv, cam.cx);
    const float vf = fmaf(cam.fy * yc, inv, cam.cy);
    ok = ok && (uf >= -0.5f && uf < c.wlim && vf >= -0.5f && vf < c.hlim);
    int u = (int)floorf(uf + 0.5f), v = (int)floorf(vf + 0.5f);
    u = min(max(u, 0), cam.W - 1);          // always a legal address, so the gather needs no branch
    v = min(max(v, 0), cam.H - 1);
    pix = v * cam.W + u;
    return ok;
}

__device__ __forceinline__ bool tsdf_finish(const Grid &g, const TsdfConst &c, bool ok, float draw, float zc, int &q) {
    const float d = draw * c.sc;
    ok = ok && (d > c.mind && d < c.maxd);
    const float sdf = d - zc;
    ok = ok && (sdf >= -g.trunc);
    const float tsdf = fminf(1.0f, sdf * g.inv_trunc);
    q = (int)rintf(tsdf * 32767.0f);
    return ok;
}

// DBG: timing experiments only (TL3D_DEBUG_ONLY=1 -> MIXED bricks only, 2 -> FREE bricks only); results incomplete.
// MAP: lane -> voxel mapping of the MIXED path.  A gather costs the texture path roughly one step per distinct image
// row it touches, so the 8 voxels a lane owns run along the grid axis that is most vertical in the image (chosen
// per launch from R) and the 64 lanes of one gather instruction share (almost) one image row band:
//   MAP 1: lanes = (x, y), lane loop over z   records k*64 + lane          (512 B contiguous per instruction)
//   MAP 2: lanes = (x, z), lane loop over y   records z*64 + k*8 + x       (8 segments of 64 B)
//   MAP 0: lanes = (x pair, y, z pair), 16-B pairs -- the generic layout, used when x is the vertical axis
// VAR: how a MIXED brick's records reach the registers (MAP 1 / 2 only; the grid is the same bit for bit):
//   0  predicated loads after the per-voxel decision: only records that change are read (fewest bytes, but the read-modify-
//      write waits for projection + gather + decision, then for HBM)
//   1  all 512 records loaded up front, before the projection: the HBM latency runs under the projection and the gathers;
//      stores stay predicated.  Reads 4 KB per MIXED brick whatever changes (counted as read).
//   2  as 1, and the NEXT listed brick's records are requested right behind this brick's gathers (one brick of look-ahead
//      per wave): the record stream never waits for the gather phase.
// FREEB: the list may hold free-space bricks (TL3D_FREE_COUNTERS=0, the round-1 formulation); false = they were counted by
// the classification, the kernel only sees MIXED bricks (a separate instantiation, so that profiles tell the two apart).
template <bool COUNT, int DBG, int MAP, typename DT, int VAR, bool FREEB>
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(Cam cam, Grid g, PoseF pose, TsdfConst c,
                                                             const DT *__restrict__ depth,
                                                             const unsigned *__restrict__ list,
                                                             const unsigned *__restrict__ list_counts,
                                                             int2 *__restrict__ grid, unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    // the trip count below comes from device memory: clamp it to what the list can hold, so that no ordering mistake
    // upstream can ever turn into an unbounded loop or an out-of-range list read
    unsigned nmixed = min(list_counts[0], nbricks), nfree = min(list_counts[1], nbricks - nmixed);
    const unsigned nfree_classified = nfree;
    if (!FREEB || c.free_counted) nfree = 0;                // counted by the classification, not listed
    if (DBG == 1 || DBG == 3 || DBG == 4) nfree = 0;
    if (DBG == 2) nmixed = 0;        // (list offsets below stay valid: FREE entries are addressed from the back)
    const unsigned nlist = nmixed + nfree;
    unsigned nread = 0, nwritten = 0;
    // Blocks with the same blockIdx % 8 share an XCD and its L2 (a placement habit of the dispatcher: a speed choice, never
    // a correctness one).  Each such group consumes one contiguous eighth of the list -- the classification emits bricks
    // in grid order, so an eighth is a slab of the volume and its depth lookups stay in one region of the image.
    const bool grouped = XCD_GROUPS > 1 && (gridDim.x % XCD_GROUPS) == 0 && c.xcd_group != 0;
    const unsigned ngrp = grouped ? XCD_GROUPS : 1u;
    const unsigned grp = grouped ? blockIdx.x % XCD_GROUPS : 0u, bi = grouped ? blockIdx.x / XCD_GROUPS : blockIdx.x;
    // list positions: [0, npair) alternate MIXED / FREE, [npair, nlist) hold what is left of the larger class; every
    // group takes its eighth of both regions, so all groups see the same MIXED : FREE ratio
    const unsigned npair = 2u * min(nmixed, nfree);
    const unsigned per_a = ((npair + ngrp - 1u) / ngrp + 1u) & ~1u;           // even: MIXED/FREE pairs stay together
    const unsigned per_b = (nlist - npair + ngrp - 1u) / ngrp;
    const unsigned a_beg = min(npair, grp * per_a), a_len = min(npair, a_beg + per_a) - a_beg;
    const unsigned b_beg = min(nlist, npair + grp * per_b), b_len = min(nlist, b_beg + per_b) - b_beg;
    const unsigned lstep = (gridDim.x / ngrp) * 4u;
    const unsigned ntask = a_len + b_len;
    // list entry of task t (wave-uniform): MIXED entries sit at the front of the list, FREE entries at the back (filled
    // downwards).  They are consumed interleaved (even slots MIXED, odd slots FREE while both last): MIXED bricks are bound
    // by the texture-address path, FREE bricks by HBM, so mixing them on every CU overlaps the two instead of running them
    // back to back.
    auto entry_of = [&](unsigned t) -> unsigned {
        const unsigned li = t < a_len ? a_beg + t : b_beg + (t - a_len);
        unsigned src;
        if (li < npair) src = (li & 1u) ? nbricks - 1u - (li >> 1) : (li >> 1);
        else if (nmixed > nfree) src = li - nfree;                                   // remaining MIXED entries
        else src = nbricks - 1u - (li - nmixed);                                     // remaining FREE entries
        return __builtin_amdgcn_readfirstlane(list[src]);
    };
    constexpr bool PREF = (VAR >= 1) && (MAP != 0);
    const int la = lane & 7, lb = lane >> 3;
    auto rec_index = [&](int k) -> int { return MAP == 1 ? (k * 64 + lane) : (lb * 64 + k * 8 + la); };
    int2 pre[8];                                            // VAR 2: records of the brick this wave handles next
    bool pre_valid = false;
    unsigned e_next = 0;
    bool have_next = bi * 4u + wid < ntask;
    if (have_next) e_next = entry_of(bi * 4u + wid);
    for (unsigned t0 = bi * 4u; t0 < ntask; t0 += lstep) {
        if (!have_next) break;
        const unsigned e = e_next;
        if (VAR == 2) {                                     // the entry after this one (wave-uniform)
            have_next = t0 + lstep + wid < ntask;
            e_next = have_next ? entry_of(t0 + lstep + wid) : 0u;
        } else {
            have_next = t0 + lstep + wid < ntask;
            if (have_next) e_next = entry_of(t0 + lstep + wid);
        }
        const int brick = (int)(e & ~FREE_FLAG);
        int4 *__restrict__ recs = reinterpret_cast<int4 *>(grid + ((size_t)brick << 9));
        if (FREEB && (e & FREE_FLAG)) {
            int4 r0 = recs[lane], r1 = recs[64 + lane], r2 = recs[128 + lane], r3 = recs[192 + lane];
            if (VAR == 2 && PREF && have_next && !(e_next & FREE_FLAG) && !pre_valid) {   // keep the look-ahead primed across a FREE brick
                const int2 *__restrict__ nx2 = reinterpret_cast<const int2 *>(grid + ((size_t)(e_next & ~FREE_FLAG) << 9));
#pragma unroll
                for (int k = 0; k < 8; ++k) pre[k] = nx2[rec_index(k)];
                pre_valid = true;
            }
            r0.x += 32767; r0.y += 1; r0.z += 32767; r0.w += 1;
            r1.x += 32767; r1.y += 1; r1.z += 32767; r1.w += 1;
            r2.x += 32767; r2.y += 1; r2.z += 32767; r2.w += 1;
            r3.x += 32767; r3.y += 1; r3.z += 32767; r3.w += 1;
            recs[lane] = r0; recs[64 + lane] = r1; recs[128 + lane] = r2; recs[192 + lane] = r3;
            if (COUNT) { nread += 8; nwritten += 8; }
            continue;
        }
        const int bx = brick % g.nbx;
        const int by = (brick / g.nbx) % g.nby;
        const int bz = brick / (g.nbx * g.nby);
        if constexpr (MAP == 0) {
            // phase 1: project the lane's 8 voxels, issue the 8 depth gathers back to back
            float zc[8], dv[8];
            bool ok[8];
    #pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pr = it * 64 + lane;
                const int i = bx * 8 + ((pr & 3) << 1);
                const int j = by * 8 + ((pr >> 2) & 7);
                const int k = bz * 8 + (pr >> 5);
                const float py = fmaf((float)j + 0.5f, g.vs, g.oy);
                const float pz = fmaf((float)k + 0.5f, g.vs, g.oz);
                const float ax = fmaf(pose.r[1], py, fmaf(pose.r[2], pz, pose.t[0]));
                const float ay = fmaf(pose.r[4], py, fmaf(pose.r[5], pz, pose.t[1]));
                const float az = fmaf(pose.r[7], py, fmaf(pose.r[8], pz, pose.t[2]));
    #pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float px = fmaf((float)(i + hh) + 0.5f, g.vs, g.ox);
                    const float xc = fmaf(pose.r[0], px, ax), yc = fmaf(pose.r[3], px, ay);
                    zc[2 * it + hh] = fmaf(pose.r[6], px, az);
                    int pix;
                    ok[2 * it + hh] = tsdf_project(cam, c, xc, yc, zc[2 * it + hh], pix);
                    dv[2 * it + hh] = ld_depth(depth, (size_t)pix);
                }
            }
            // phase 2: decide; phase 3: load the 16-B pairs that change; phase 4: add and store them
            int q[8];
    #pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) ok[e8] = tsdf_finish(g, c, ok[e8], dv[e8], zc[e8], q[e8]);
            int4 rec[4];
    #pragma unroll
            for (int it = 0; it < 4; ++it)
                if (ok[2 * it] | ok[2 * it + 1]) rec[it] = recs[it * 64 + lane];
    #pragma unroll
            for (int it = 0; it < 4; ++it)
                if (ok[2 * it] | ok[2 * it + 1]) {
                    if (ok[2 * it]) { rec[it].x += q[2 * it]; rec[it].y += 1; }
                    if (ok[2 * it + 1]) { rec[it].z += q[2 * it + 1]; rec[it].w += 1; }
                    recs[it * 64 + lane] = rec[it];
                    if (COUNT) { nread += 2; nwritten += 2; }
                }
    
        } else {
            // lane-owned column of 8 voxels along the image-vertical grid axis; records are touched 8 B at a time
            const int i = bx * 8 + la;                                     // x is a lane axis in both maps
            const float px = fmaf((float)i + 0.5f, g.vs, g.ox);
            float zc[8], dv[8];
            bool ok[8];
            int2 *__restrict__ recs2 = reinterpret_cast<int2 *>(recs);
            int2 rec[8];
            if (PREF) {
                if (VAR == 2 && pre_valid) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) rec[k] = pre[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) rec[k] = recs2[rec_index(k)];
                }
                pre_valid = false;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = by * 8 + (MAP == 1 ? lb : k);
                const int kk = bz * 8 + (MAP == 1 ? k : lb);
                const float py = fmaf((float)j + 0.5f, g.vs, g.oy);
                const float pz = fmaf((float)kk + 0.5f, g.vs, g.oz);
                const float xc = fmaf(pose.r[0], px, fmaf(pose.r[1], py, fmaf(pose.r[2], pz, pose.t[0])));
                const float yc = fmaf(pose.r[3], px, fmaf(pose.r[4], py, fmaf(pose.r[5], pz, pose.t[1])));
                zc[k] = fmaf(pose.r[6], px, fmaf(pose.r[7], py, fmaf(pose.r[8], pz, pose.t[2])));
                int pix;
                ok[k] = tsdf_project(cam, c, xc, yc, zc[k], pix);
                dv[k] = (DBG == 3) ? 1.0f + 1e-6f * (float)(pix & 1023) : ld_depth(depth, (size_t)pix);
            }
            if (VAR == 2 && PREF && have_next && !(e_next & FREE_FLAG)) {     // right behind the gathers: the next brick's records
                const int2 *__restrict__ nx2 = reinterpret_cast<const int2 *>(grid + ((size_t)(e_next & ~FREE_FLAG) << 9));
#pragma unroll
                for (int k = 0; k < 8; ++k) pre[k] = nx2[rec_index(k)];
                pre_valid = true;
            }
            int q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) ok[k] = tsdf_finish(g, c, ok[k], dv[k], zc[k], q[k]);
            if (!PREF) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (ok[k]) rec[k] = (DBG == 4) ? make_int2(q[k], k) : recs2[rec_index(k)];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (ok[k]) {
                    rec[k].x += q[k];
                    rec[k].y += 1;
                    if (DBG == 4) { if (rec[k].x == 0x7fffffff) recs2[0] = rec[k]; }      // keep the values alive, never store
                    else recs2[rec_index(k)] = rec[k];
                    if (COUNT) { nread += PREF ? 0 : 1; nwritten += 1; }
                }
            if (COUNT && PREF) nread += 8;
        }
    }
    if (COUNT) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            nread += __shfl_down(nread, d);
            nwritten += __shfl_down(nwritten, d);
        }
        if (lane == 0) {
            atomicAdd(counters + 2, (unsigned long long)nread);
            atomicAdd(counters + 3, (unsigned long long)nwritten);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicAdd(counters + 4, (unsigned long long)(nmixed + nfree_classified));
            atomicAdd(counters + 5, (unsigned long long)nfree_classified);
            if (c.free_counted) atomicAdd(counters + 6, (unsigned long long)nfree_classified);
        }
    }
}

// Two frames per visit of a brick.  The update is bound by the CU's vector L1: a brick's depth gathers and its record lines
// queue there one after the other.  The gathers are per frame; the records need not be: a brick that two
// consecutive frames both see near a surface is read and written ONCE for both (integer sums: the same grid bit for bit).
// Tasks = frame A's MIXED list, then frame B's; a brick of A's list that is MIXED in B too (B's class map) takes both
// frames' gathers before its one read-modify-write; a brick of B's list that was on A's is skipped.  MAP 1 / 2 as above.
template <bool COUNT, int MAP, typename DT>
__global__ __launch_bounds__(256) void tsdf_pair_kernel(Cam cam, Grid g, PoseF poseA, PoseF poseB, TsdfConst cA, TsdfConst cB,
                                                        const DT *__restrict__ depthA, const DT *__restrict__ depthB,
                                                        const unsigned *__restrict__ listA, const unsigned *__restrict__ countsA,
                                                        const unsigned char *__restrict__ clsA,
                                                        const unsigned *__restrict__ listB, const unsigned *__restrict__ countsB,
                                                        const unsigned char *__restrict__ clsB,
                                                        int2 *__restrict__ grid, unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    const unsigned nA = min(countsA[0], nbricks), nB = min(countsB[0], nbricks);        // clamped: see tsdf_integrate_kernel
    const unsigned ntask = nA + nB;
    unsigned nread = 0, nwritten = 0;
    // blocks that share an XCD (blockIdx % 8) take one contiguous eighth of the tasks: a slab of the volume
    const bool grouped = XCD_GROUPS > 1 && (gridDim.x % XCD_GROUPS) == 0 && cA.xcd_group != 0;
    const unsigned ngrp = grouped ? XCD_GROUPS : 1u;
    const unsigned grp = grouped ? blockIdx.x % XCD_GROUPS : 0u, bi = grouped ? blockIdx.x / XCD_GROUPS : blockIdx.x;
    // every group takes its eighth of BOTH lists (B's entries are mostly skipped: a group that got only those would idle)
    const unsigned perA = (nA + ngrp - 1u) / ngrp, perB = (nB + ngrp - 1u) / ngrp;
    const unsigned a_beg = min(nA, grp * perA), a_len = min(nA, a_beg + perA) - a_beg;
    const unsigned b_beg = min(nB, grp * perB), b_len = min(nB, b_beg + perB) - b_beg;
    const unsigned lstep = (gridDim.x / ngrp) * 4u;
    const int la = lane & 7, lb = lane >> 3;
    (void)ntask;
    auto process = [&](unsigned brick, bool useA, bool useB) {        // useA / useB wave-uniform
        const int bx = (int)(brick % (unsigned)g.nbx), by = (int)((brick / (unsigned)g.nbx) % (unsigned)g.nby), bz = (int)(brick / (unsigned)(g.nbx * g.nby));
        int2 *__restrict__ recs2 = grid + ((size_t)brick << 9);
        const int i = bx * 8 + la;
        const float px = fmaf((float)i + 0.5f, g.vs, g.ox);
        int qs[8], ws[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { qs[k] = 0; ws[k] = 0; }
        // both frames' gathers are issued before either frame's values are needed (one wait for 16 instead of two for 8)
        float zcA[8], dvA[8], zcB[8], dvB[8];
        bool okA[8], okB[8];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (!(f ? useB : useA)) continue;
            const PoseF &pose = f ? poseB : poseA;
            const TsdfConst &c = f ? cB : cA;
            const DT *__restrict__ depth = f ? depthB : depthA;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = by * 8 + (MAP == 1 ? lb : k);
                const int kk = bz * 8 + (MAP == 1 ? k : lb);
                const float py = fmaf((float)j + 0.5f, g.vs, g.oy);
                const float pz = fmaf((float)kk + 0.5f, g.vs, g.oz);
                const float xc = fmaf(pose.r[0], px, fmaf(pose.r[1], py, fmaf(pose.r[2], pz, pose.t[0])));
                const float yc = fmaf(pose.r[3], px, fmaf(pose.r[4], py, fmaf(pose.r[5], pz, pose.t[1])));
                const float z = fmaf(pose.r[6], px, fmaf(pose.r[7], py, fmaf(pose.r[8], pz, pose.t[2])));
                int pix;
                const bool ok = tsdf_project(cam, c, xc, yc, z, pix);
                const float d = ld_depth(depth, (size_t)pix);
                if (f) { zcB[k] = z; okB[k] = ok; dvB[k] = d; } else { zcA[k] = z; okA[k] = ok; dvA[k] = d; }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int q;
            if (useA && tsdf_finish(g, cA, okA[k], dvA[k], zcA[k], q)) { qs[k] += q; ws[k] += 1; }
            if (useB && tsdf_finish(g, cB, okB[k], dvB[k], zcB[k], q)) { qs[k] += q; ws[k] += 1; }
        }
        int2 rec[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (ws[k]) rec[k] = recs2[MAP == 1 ? (k * 64 + lane) : (lb * 64 + k * 8 + la)];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (ws[k]) {
                rec[k].x += qs[k];
                rec[k].y += ws[k];
                recs2[MAP == 1 ? (k * 64 + lane) : (lb * 64 + k * 8 + la)] = rec[k];
                if (COUNT) { nread += 1; nwritten += 1; }
            }
    };
    // one list entry per wave and trip, A's list then B's; the entry of the NEXT trip (and its class in the other frame) is
    // fetched before this trip's brick is processed.  Most of B's entries are on A's list too and are skipped.
    {
        const unsigned ntot = a_len + b_len;
        auto fetch = [&](unsigned t, unsigned &brick, int &other) {
            const bool fromA = t < a_len;
            brick = (unsigned)__builtin_amdgcn_readfirstlane((int)(fromA ? listA[a_beg + t] : listB[b_beg + (t - a_len)]));
            other = __builtin_amdgcn_readfirstlane((int)(fromA ? clsB[brick] : clsA[brick]));
        };
        unsigned t = bi * 4u + wid;
        unsigned brick_n = 0;
        int other_n = 0;
        if (t < ntot) fetch(t, brick_n, other_n);
        for (; t < ntot; t += lstep) {
            const unsigned brick = brick_n;
            const int other = other_n;
            if (t + lstep < ntot) fetch(t + lstep, brick_n, other_n);
            const bool fromA = t < a_len;
            if (!fromA && other == 1) continue;                      // handled from A's list
            process(brick, fromA, fromA ? other == 1 : true);
        }
    }
    if (COUNT) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            nread += __shfl_down(nread, d);
            nwritten += __shfl_down(nwritten, d);
        }
        if (lane == 0) {
            atomicAdd(counters + 2, (unsigned long long)nread);
            atomicAdd(counters + 3, (unsigned long long)nwritten);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const unsigned fA = min(countsA[1], nbricks - nA), fB = min(countsB[1], nbricks - nB);
            atomicAdd(counters + 4, (unsigned long long)(nA + fA) + (nB + fB));
            atomicAdd(counters + 5, (unsigned long long)fA + fB);
            atomicAdd(counters + 6, (unsigned long long)fA + fB);
        }
    }
}

static Pyramid make_pyramid(const Cam &cam) {
    Pyramid p;
    memset(&p, 0, sizeof(p));
    int nx = (cam.W + TILE - 1) / TILE, ny = (cam.H + TILE - 1) / TILE, off = 0, L = 0;
    for (;;) {
        p.ntx[L] = nx; p.nty[L] = ny; p.off[L] = off;
        off += nx * ny;
        ++L;
        if ((nx == 1 && ny == 1) || L == MAX_LEVELS) break;
        nx = (nx + 1) / 2;
        ny = (ny + 1) / 2;
    }
    p.nlev = L;
    return p;
}

static size_t pyramid_tiles(const Pyramid &p) { return (size_t)p.off[p.nlev - 1] + (size_t)p.ntx[p.nlev - 1] * p.nty[p.nlev - 1]; }

size_t tsdf_scratch_bytes(const Cam &cam, const Grid &g) {
    const Pyramid p = make_pyramid(cam);
    const size_t nbricks = (size_t)g.nbx * g.nby * g.nbz;
    return 256 + pyramid_tiles(p) * sizeof(float4) + (nbricks + 64) * sizeof(unsigned) + ((nbricks + 255) & ~(size_t)255);
}

struct TsdfScratch {
    unsigned *list_counts;       // [0] mixed, [1] free
    float4 *tiles;
    unsigned *list;
    unsigned char *cls;          // [nbricks] class of every brick (0 skip, 1 mixed, 2 free)
    Pyramid py;
};

static TsdfScratch carve(const Cam &cam, const Grid &g, void *scratch) {
    TsdfScratch t;
    t.py = make_pyramid(cam);
    // scratch layout: [list_counts (256 B)] [tile pyramid] [brick list] [brick classes]
    t.list_counts = reinterpret_cast<unsigned *>(scratch);
    t.tiles = reinterpret_cast<float4 *>(reinterpret_cast<char *>(scratch) + 256);
    t.list = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(scratch) + 256 + pyramid_tiles(t.py) * sizeof(float4));
    t.cls = reinterpret_cast<unsigned char *>(t.list + ((size_t)g.nbx * g.nby * g.nbz + 64));
    return t;
}

static TsdfConst make_const(const Cam &cam, float scale, float mind, float maxd, bool free_counted = false) {
    TsdfConst c;
    c.free_counted = free_counted ? 1 : 0;
    c.mind = mind;
    c.maxd = maxd;
    c.sc = scale;
    c.wlim = (float)cam.W - 0.5f;
    c.hlim = (float)cam.H - 0.5f;
    static const int xg = getenv("TL3D_XCD_GROUP") ? atoi(getenv("TL3D_XCD_GROUP")) : 1;
    c.xcd_group = xg;
    return c;
}

// depth tiles + pyramid + brick classification -> compact brick list in scratch, for one frame or for two in the same
// three launches (n = 2: both depth images of one kind)
int launch_tsdf_prepare(hipStream_t s, const Cam &cam, const Grid &g, int n, const PoseF *p, const Frustum &fr, const void *const *depth,
                        bool depth_u16, const float *scale, float mind, float maxd, void *const *scratch, unsigned *free_cnt) {
    PrepFrames P;
    memset(&P, 0, sizeof(P));
    Pyramid py = make_pyramid(cam);
    for (int i = 0; i < n; ++i) {
        const TsdfScratch t = carve(cam, g, scratch[i]);
        P.f[i].depth = depth[i];
        P.f[i].tiles = t.tiles;
        P.f[i].list = t.list;
        P.f[i].list_counts = t.list_counts;
        P.f[i].cls = t.cls;
        P.f[i].c = make_const(cam, scale[i], mind, maxd);
        P.f[i].pose = p[i];
    }
    const int ntiles = py.ntx[0] * py.nty[0];
    if (depth_u16)
        hipLaunchKernelGGL(depth_tiles_kernel<uint16_t>, dim3(ntiles < 1024 ? ntiles : 1024, n), dim3(256), 0, s, cam, P, py.ntx[0], py.nty[0]);
    else
        hipLaunchKernelGGL(depth_tiles_kernel<float>, dim3(ntiles < 1024 ? ntiles : 1024, n), dim3(256), 0, s, cam, P, py.ntx[0], py.nty[0]);
    TL3D_HIP(hipGetLastError());
    if (py.nlev > 1) {
        hipLaunchKernelGGL(tile_pyramid_kernel, dim3(n), dim3(256), 0, s, py, P);
        TL3D_HIP(hipGetLastError());
    }
    const int ncells = ((g.nbx + 3) / 4) * ((g.nby + 3) / 4) * ((g.nbz + 3) / 4);
    hipLaunchKernelGGL(brick_cull_kernel, dim3((ncells + 3) / 4, n), dim3(256), 0, s, cam, g, P, fr, py, free_cnt);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// the dominant kernel: read-modify-write of the listed bricks
int launch_fold_free(hipStream_t s, const Grid &g, int2 *grid, unsigned *free_cnt) {
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    const unsigned nb = (nbricks + 3u) / 4u < 2048u ? (nbricks + 3u) / 4u : 2048u;
    hipLaunchKernelGGL(fold_free_kernel, dim3(nb ? nb : 1), dim3(256), 0, s, grid, free_cnt, nbricks);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// which lane map the update of a frame with pose p uses: the grid axis most vertical in the image (largest |R[1][a]|):
// y -> MAP 2, z -> MAP 1, x -> generic MAP 0
int tsdf_lane_map(const PoseF &p) {
    static const int force_map = getenv("TL3D_TSDF_MAP") ? atoi(getenv("TL3D_TSDF_MAP")) : -1;
    const float ax = fabsf(p.r[3]), ay = fabsf(p.r[4]), az = fabsf(p.r[5]);
    int map = (ay >= ax && ay >= az) ? 2 : (az >= ax ? 1 : 0);
    if (force_map >= 0 && force_map <= 2) map = force_map;
    return map;
}

// two prepared frames in one launch (callers check tsdf_lane_map(pA) == tsdf_lane_map(pB) != 0, both depth images of one kind,
// free-space bricks counted)
int launch_tsdf_update_pair(hipStream_t s, const Cam &cam, const Grid &g, const PoseF &pA, const PoseF &pB, const void *depthA, const void *depthB,
                            bool depth_u16, float scaleA, float scaleB, float mind, float maxd, int2 *grid, void *scratchA, void *scratchB,
                            unsigned long long *counters, bool count) {
    const TsdfConst cA = make_const(cam, scaleA, mind, maxd, true), cB = make_const(cam, scaleB, mind, maxd, true);
    const TsdfScratch tA = carve(cam, g, scratchA), tB = carve(cam, g, scratchB);
    const int nbricks = g.nbx * g.nby * g.nbz;
    // 6 workgroups per CU, as the one-frame kernel.  (While every frame had a prep chain of its own those chains, which crawl
    // beside this kernel, decided the frame rate and 4 per CU was the better trade: 29.2k against 26.6k frames/s; with one chain
    // per two frames it is 28.5k at 4 per CU, 30.7k at 6.)
    static const int max_blk = getenv("TL3D_UPDATE_BLOCKS") ? atoi(getenv("TL3D_UPDATE_BLOCKS")) : 1536;
    int nblk = (nbricks + 3) / 4;
    if (nblk > max_blk) nblk = max_blk;
    const int map = tsdf_lane_map(pA);
#define TL3D_LAUNCH_PAIR(C_, M_)                                                                                                        \
    do {                                                                                                                              \
        if (depth_u16)                                                                                                                \
            hipLaunchKernelGGL((tsdf_pair_kernel<C_, M_, uint16_t>), dim3(nblk), dim3(256), 0, s, cam, g, pA, pB, cA, cB,              \
                               static_cast<const uint16_t *>(depthA), static_cast<const uint16_t *>(depthB), tA.list, tA.list_counts,  \
                               tA.cls, tB.list, tB.list_counts, tB.cls, grid, counters);                                               \
        else                                                                                                                          \
            hipLaunchKernelGGL((tsdf_pair_kernel<C_, M_, float>), dim3(nblk), dim3(256), 0, s, cam, g, pA, pB, cA, cB,                 \
                               static_cast<const float *>(depthA), static_cast<const float *>(depthB), tA.list, tA.list_counts,        \
                               tA.cls, tB.list, tB.list_counts, tB.cls, grid, counters);                                               \
    } while (0)
    if (map == 2) { if (count) TL3D_LAUNCH_PAIR(true, 2); else TL3D_LAUNCH_PAIR(false, 2); }
    else { if (count) TL3D_LAUNCH_PAIR(true, 1); else TL3D_LAUNCH_PAIR(false, 1); }
#undef TL3D_LAUNCH_PAIR
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_tsdf_update(hipStream_t s, const Cam &cam, const Grid &g, const PoseF &p, const void *depth, bool depth_u16, float scale, float mind,
                       float maxd, int2 *grid, void *scratch, unsigned long long *counters, bool count, bool free_counted) {
    const TsdfConst c = make_const(cam, scale, mind, maxd, free_counted);
    const TsdfScratch t = carve(cam, g, scratch);
    const int nbricks = g.nbx * g.nby * g.nbz;
    // 6 workgroups per CU.  The update saturates from 4 per CU upwards; with the update launches batched
    // back to back, 6 measured best (frame period 59.5 us vs 62.3 at 4 and 60.2 at 7): the prep kernels of later frames,
    // which run beside it at stream priority, then get just the slots they need
    static const int max_blk = getenv("TL3D_UPDATE_BLOCKS") ? atoi(getenv("TL3D_UPDATE_BLOCKS")) : 1536;
    int nblk = (nbricks + 3) / 4;
    if (nblk > max_blk) nblk = max_blk;
    static const int dbg = getenv("TL3D_DEBUG_ONLY") ? atoi(getenv("TL3D_DEBUG_ONLY")) : 0;
    const int map = tsdf_lane_map(p);
#define TL3D_LAUNCH_UPD(C_, D_, M_, V_, F_)                                                                                            \
    do {                                                                                                                         \
        if (depth_u16)                                                                                                           \
            hipLaunchKernelGGL((tsdf_integrate_kernel<C_, D_, M_, uint16_t, V_, F_>), dim3(nblk), dim3(256), 0, s, cam, g, p, c, \
                               static_cast<const uint16_t *>(depth), t.list, t.list_counts, grid, counters);                      \
        else                                                                                                                     \
            hipLaunchKernelGGL((tsdf_integrate_kernel<C_, D_, M_, float, V_, F_>), dim3(nblk), dim3(256), 0, s, cam, g, p, c,    \
                               static_cast<const float *>(depth), t.list, t.list_counts, grid, counters);                         \
    } while (0)
#define TL3D_LAUNCH_MAPF(C_, D_, V_, F_)                       \
    do {                                                       \
        if (map == 2) TL3D_LAUNCH_UPD(C_, D_, 2, V_, F_);      \
        else if (map == 1) TL3D_LAUNCH_UPD(C_, D_, 1, V_, F_); \
        else TL3D_LAUNCH_UPD(C_, D_, 0, V_, F_);               \
    } while (0)
#define TL3D_LAUNCH_MAP(C_, D_, V_)                             \
    do {                                                        \
        if (free_counted) TL3D_LAUNCH_MAPF(C_, D_, V_, false);  \
        else TL3D_LAUNCH_MAPF(C_, D_, V_, true);                \
    } while (0)
#define TL3D_LAUNCH_VAR(C_)                         \
    do {                                            \
        if (var == 2) TL3D_LAUNCH_MAP(C_, 0, 2);    \
        else if (var == 1) TL3D_LAUNCH_MAP(C_, 0, 1); \
        else TL3D_LAUNCH_MAP(C_, 0, 0);             \
    } while (0)
    static const int var = getenv("TL3D_TSDF_VARIANT") ? atoi(getenv("TL3D_TSDF_VARIANT")) : 0;
    if (count) TL3D_LAUNCH_VAR(true);
    else if (dbg == 1) TL3D_LAUNCH_MAP(false, 1, 0);
    else if (dbg == 2) TL3D_LAUNCH_MAP(false, 2, 0);
    else if (dbg == 3) TL3D_LAUNCH_MAP(false, 3, 0);
    else if (dbg == 4) TL3D_LAUNCH_MAP(false, 4, 0);
    else TL3D_LAUNCH_VAR(false);
#undef TL3D_LAUNCH_VAR
#undef TL3D_LAUNCH_MAPF
#undef TL3D_LAUNCH_MAP
#undef TL3D_LAUNCH_UPD
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

}  // namespace tl3d
