// kernels_tsdf.hip -- row a11: TSDF integration of depth frames into the brick-major grid.
// No reference code exists for this row; the convention is the oracle's
// (oracle/tl3d_oracle.c: orc_tsdf_integrate) and both sides evaluate the same f32 sequence, so the integer
// grid {sum of rint(tsdf*32767), weight} is bit-identical.
//
// Launches per frame (prep chain on a side stream, update on the main stream):
//   1. depth_tiles_kernel     (min, max, all-valid) of the valid scaled depth over 8x8-, 16x16- and 32x32-pixel tiles
//                             (pyramid levels 0-2), 4 B/pixel read
//   2. tile_pyramid_kernel    1 workgroup: 2x2 reductions of level 2 up to a single tile
//   3. brick_cull_kernel      one lane per 8^3 brick (one wave per 4x4x4-brick cell) classifies it from <= 16 pyramid
//                             lookups:
//        SKIP   outside the view, no valid depth under it, or more than trunc behind every surface it can see
//        FREE   wholly inside the image, every pixel under it valid, and at least trunc in front of every
//               surface: every voxel gets exactly tsdf = 1 (q = 32767): ONE add to the brick's free-space counter
//        MIXED  everything else (near a surface, on the image border, straddling the camera plane): compact list
//   4. subbrick_classify_kernel  one wave per LISTED brick: its eight 4x4x4 SUB-BRICKS by the same three rules (8 lanes per
//        sub-brick: one extreme voxel centre each, <= 8 pyramid lookups) -> a (mixed, free) bit mask per brick.  On the
//        headline sequence 3.4 of the 8 sub-bricks of a MIXED brick stay MIXED.  For the second frame of a prepared pair also
//        the list of its bricks that are NOT on the first frame's list ("solo" list).
//   5. tsdf_update_kernel     one 64-lane wave per listed brick and visit, for ONE frame or for TWO consecutive frames
//        (a brick both frames see near a surface is read and written once for both): the voxels of MIXED sub-bricks are
//        projected and gather a depth value; FREE sub-bricks stream (+32767, +1) into their 512 B of records; SKIP
//        sub-bricks are not touched.  A sub-brick's 64 records are contiguous (the in-brick record order,
//        tl3d_internal.h: in_brick_index), so every record access of a wave is one 512-B run.
// Every classification is conservative with respect to the per-voxel rule, so the result is the oracle's bit
// for bit whatever the view.  The list balances the work (a static brick->block map leaves most of the chip idle:
// the frustum covers a fraction of the grid).
// Algorithmic bytes per launch = 8 B x (records read + written), counted by the kernel in counting mode
// ("counted, never estimated"), + 8 B per counted free-space brick + the depth frame(s).
#include <stdlib.h>

#include "tl3d_internal.h"
#include "bp_device.h"

namespace tl3d {

struct TsdfConst {
    float mind, maxd, sc, wlim, hlim;
};

constexpr int TILE0 = 8;                 // level-0 tiles are 8x8 pixels
constexpr int TILE0_SHIFT = 3;
constexpr int REGION = 32;               // one workgroup pass of the tiles kernel: 32x32 pixels = levels 0, 1, 2 of that region
constexpr int MAX_LEVELS = 14;           // 8 px << 13 = 65 536 px
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
    unsigned *list, *list_counts;        // list_counts: [0] listed (MIXED) bricks, [1] free-space bricks (counted), [2] solo list
    unsigned char *cls;
    unsigned short *sub;                 // [nbricks] sub-brick masks of the LISTED bricks: bits 0-7 mixed, bits 8-15 free
    unsigned *solo;                      // second frame of a prepared pair: its listed bricks that the first frame does not list
    TsdfConst c;
    PoseF pose;
};
struct PrepFrames { PrepFrame f[2]; };

// one frame of an update launch
struct UpdFrame {
    PoseF pose;
    TsdfConst c;
    const void *depth;
    const unsigned *list, *counts;       // compact list of the frame's MIXED bricks, counts[0] = its length (as frame B of a pair:
                                         // the solo list, counts[2])
    const unsigned char *cls;            // class of every brick of the grid in this frame (0 skip, 1 mixed, 2 free)
    const unsigned short *sub;           // sub-brick masks of its listed bricks
};

// tile = (dmin, dmax, allvalid ? 1 : 0, unused)

// Depth source of the TSDF kernels: the f32 frame, or the 16-bit millimetre image it was converted from (uploads of kind
// TL3D_DEPTH_U16_MM keep it): same value through the same conversion as u16_to_f32_kernel (mm_to_m), half the bytes per pixel and
// so half the cache lines under a brick's footprint.
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
// One 32x32-pixel region per workgroup pass: 8 lanes x 16 B cover a region row, 32 rows -> 256 threads.  Wave w holds rows
// 8w .. 8w+7 = one row of four level-0 tiles (lanes with equal q4 >> 1); the region's level-1 and level-2 tiles are combined
// through LDS.
template <typename DT>
__global__ __launch_bounds__(256) void depth_tiles_kernel(Cam cam, PrepFrames P, Pyramid py, int nrx, int nry) {
    const PrepFrame &F = P.f[blockIdx.y];
    const TsdfConst c = F.c;
    const DT *__restrict__ depth = static_cast<const DT *>(F.depth);
    float4 *__restrict__ tiles = F.tiles;
    unsigned *__restrict__ list_counts = F.list_counts;
    __shared__ float s_mn[4][4], s_mx[4][4];
    __shared__ int s_bad[4][4];
    if (blockIdx.x == 0 && threadIdx.x < 3) list_counts[threadIdx.x] = 0u;      // reset the brick-list cursors
    const int q4 = threadIdx.x & 7, row = threadIdx.x >> 3, wid = threadIdx.x >> 6;
    const bool vec = (cam.W & 3) == 0;                              // rows are 16-B aligned
    for (int region = blockIdx.x; region < nrx * nry; region += gridDim.x) {
        const int rx = region % nrx, ry = region / nrx;
        float mn = INFINITY, mx = -INFINITY;
        int bad = 0;
        const int u0 = rx * REGION + q4 * 4, v = ry * REGION + row;
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
        // level 0: lanes (row & 7, q4) with equal q4 >> 1: butterfly over lane bits 0, 3, 4, 5
#pragma unroll
        for (int d = 1; d <= 32; d = (d == 1 ? 8 : d << 1)) {
            mn = fminf(mn, __shfl_xor(mn, d));
            mx = fmaxf(mx, __shfl_xor(mx, d));
            bad |= __shfl_xor(bad, d);
        }
        const int lane = threadIdx.x & 63;
        if ((lane & 0x39) == 0) {                                   // lanes 0, 2, 4, 6: one per level-0 tile of this wave
            const int ax = lane >> 1;
            const int tx = rx * 4 + ax, ty = ry * 4 + wid;
            if (tx < py.ntx[0] && ty < py.nty[0]) tiles[py.off[0] + ty * py.ntx[0] + tx] = make_float4(mn, mx, bad ? 0.0f : 1.0f, 0.0f);
            s_mn[wid][ax] = mn; s_mx[wid][ax] = mx; s_bad[wid][ax] = bad;
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            // threads 0..3: the region's level-1 tiles; thread 4: its level-2 tile
            const int l1x = threadIdx.x & 1, l1y = (threadIdx.x >> 1) & 1;
            const int x0 = threadIdx.x < 4 ? 2 * l1x : 0, x1 = threadIdx.x < 4 ? x0 + 2 : 4;
            const int y0 = threadIdx.x < 4 ? 2 * l1y : 0, y1 = threadIdx.x < 4 ? y0 + 2 : 4;
            float a = INFINITY, b = -INFINITY;
            int bd = 0;
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) { a = fminf(a, s_mn[y][x]); b = fmaxf(b, s_mx[y][x]); bd |= s_bad[y][x]; }
            if (threadIdx.x < 4) {
                const int tx = rx * 2 + l1x, ty = ry * 2 + l1y;
                if (py.nlev > 1 && tx < py.ntx[1] && ty < py.nty[1]) tiles[py.off[1] + ty * py.ntx[1] + tx] = make_float4(a, b, bd ? 0.0f : 1.0f, 0.0f);
            } else if (py.nlev > 2) {
                tiles[py.off[2] + ry * py.ntx[2] + rx] = make_float4(a, b, bd ? 0.0f : 1.0f, 0.0f);
            }
        }
        __syncthreads();
    }
}

// ---- 2. pyramid: levels 3 .. from level 2 ------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_pyramid_kernel(Pyramid py, PrepFrames P) {
    float4 *__restrict__ tiles = P.f[blockIdx.x].tiles;
    for (int L = 3; L < py.nlev; ++L) {
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
// a handful of instructions (most of the grid); the four waves of a workgroup pool their survivors so the list
// cursor sees one atomic per workgroup, not per wave (same-address returning atomics retire at ~90 per microsecond).
// Margins: 1.5 px on projected bounds, 1 % of a voxel on depths, 0.1 % on the truncation distance.
__device__ __forceinline__ bool sphere_in_view(const Frustum &fr, float x, float y, float z, float rad) {
    return (z + rad > 0.0f) && (fr.lx * x + fr.lz * z >= -rad) && (fr.rx * x + fr.rz * z >= -rad) &&
           (fr.ty * y + fr.tz * z >= -rad) && (fr.by * y + fr.bz * z >= -rad);
}

// The three rules on one box of voxels, given what the pyramid says about the pixels under it: a = (min, max, all-valid) of
// the valid scaled depth over a superset of the box's pixel footprint, [zmin, zmax] the camera depths of its voxel centres,
// inside = the footprint (widened by 1.5 px) lies wholly in the image.  0 skip, 1 mixed, 2 free.
__device__ __forceinline__ int classify_box(const Grid &g, float4 a, float zmin, float zmax, bool inside) {
    const float m = 0.01f * g.vs;
    if (!(a.y > -INFINITY) || (zmin - m > a.y + g.trunc)) return 0;      // no valid depth under the box, or > trunc behind all it can see
    if (inside && a.z > 0.5f && (a.x - (zmax + m) >= g.trunc * 1.001f)) return 2;   // every voxel: in image, valid depth, sdf >= trunc => tsdf == 1 exactly
    return 1;
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
                        while (L < py.nlev - 1 && (((pu1 >> (TILE0_SHIFT + L)) - (pu0 >> (TILE0_SHIFT + L))) > 3 ||
                                                   ((pv1 >> (TILE0_SHIFT + L)) - (pv0 >> (TILE0_SHIFT + L))) > 3))
                            ++L;
                        const int tu0 = pu0 >> (TILE0_SHIFT + L), tu1 = pu1 >> (TILE0_SHIFT + L);
                        const int tv0 = pv0 >> (TILE0_SHIFT + L), tv1 = pv1 >> (TILE0_SHIFT + L);
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
                        cls = classify_box(g, a, zmin, zmax, inside);
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
        s_base[threadIdx.x] = tot ? atomicAdd(list_counts + threadIdx.x, tot) : 0u;      // [1]: free-space bricks, counted only
    }
    __syncthreads();
    unsigned bm = s_base[0];
    for (int w = 0; w < wid; ++w) bm += s_cnt[w][0];
    const unsigned long long below = (1ull << lane) - 1ull;
    if (cls == 1) list[bm + __popcll(mm & below)] = (unsigned)brick;
    // Free space: every voxel of the brick gets exactly (+32767, +1).  That is ONE integer add here instead of a 4 KB read +
    // 4 KB write by the update kernel; the counters are folded into the records before anything reads them (fold_free_kernel).
    if (cls == 2) atomicAdd(free_cnt + brick, 1u);
}

// records += count x (32767, 1) for every brick with a pending free-space count; the count returns to zero (exchanged, so a
// count that lands between the read and the reset cannot be lost).  One wave per brick, 16 B per lane.  Runs on the main
// stream before anything reads the TSDF channel (download, merge, extraction, weight check), i.e. once per scan, not per frame.
__global__ __launch_bounds__(256) void fold_free_kernel(int2 *__restrict__ grid, unsigned *__restrict__ free_cnt, unsigned nbricks) {
    const int lane = threadIdx.x & 63;
    for (unsigned b = blockIdx.x * 4u + (threadIdx.x >> 6); b < nbricks; b += gridDim.x * 4u) {
        if (__builtin_amdgcn_readfirstlane(free_cnt[b]) == 0u) continue;
        unsigned c = 0;
        if (lane == 0) c = atomicExch(free_cnt + b, 0u);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c == 0u) continue;
        int4 *__restrict__ recs = reinterpret_cast<int4 *>(grid + ((size_t)b << 9));
        const int dq = (int)(c * 32767u), dw = (int)c;
        int4 r0 = recs[lane], r1 = recs[64 + lane], r2 = recs[128 + lane], r3 = recs[192 + lane];
        r0.x += dq; r0.y += dw; r0.z += dq; r0.w += dw;
        r1.x += dq; r1.y += dw; r1.z += dq; r1.w += dw;
        r2.x += dq; r2.y += dw; r2.z += dq; r2.w += dw;
        r3.x += dq; r3.y += dw; r3.z += dq; r3.w += dw;
        recs[lane] = r0; recs[64 + lane] = r1; recs[128 + lane] = r2; recs[192 + lane] = r3;
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

// min / max over the 8 lanes of a group (lane bits 0-2) in three DPP steps, no LDS traffic: quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], then row_half_mirror (lane i <- lane 7 - i of its 8-lane half: the other quad, which is uniform by then)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group8_min(float v) {
    v = fminf(v, dpp_mov<0xB1>(v));
    v = fminf(v, dpp_mov<0x4E>(v));
    return fminf(v, dpp_mov<0x141>(v));
}
__device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    return fmaxf(v, dpp_mov<0x141>(v));
}

// bits 8 s of a ballot (one per 8-lane group) -> bits s
__device__ __forceinline__ unsigned group_bits(unsigned long long b) {
    unsigned m = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) m |= (unsigned)((b >> (8 * s)) & 1ull) << s;
    return m;
}

// Classes of the eight 4x4x4 sub-bricks of brick (bx, by, bz) in one frame: bit s of `mixed` / `free_` (sub-brick s =
// (x >> 2) | (y >> 2) << 1 | (z >> 2) << 2; neither bit: skip).  Lane l handles extreme voxel centre l & 7 of sub-brick l >> 3
// with the per-voxel position sequence (the 8 extreme centres span the box of all 64: a projective map keeps convexity while
// every depth is positive, so all 64 project inside the 8 projections' pixel box), then the 8 lanes of a sub-brick share
// <= 8 pyramid tiles that cover the box (finest level at which it spans <= 8 tiles).
__device__ __forceinline__ void classify_subbricks(const Cam &cam, const Grid &g, const Pyramid &py, const PoseF &pose,
                                                   const float4 *__restrict__ tiles, int bx, int by, int bz, int lane, unsigned &mixed,
                                                   unsigned &free_) {
    const int s = lane >> 3, cn = lane & 7;
    const int i = bx * 8 + (s & 1) * 4 + ((cn & 1) ? 3 : 0);
    const int j = by * 8 + ((s >> 1) & 1) * 4 + ((cn & 2) ? 3 : 0);
    const int k = bz * 8 + (s >> 2) * 4 + ((cn & 4) ? 3 : 0);
    const float px = fmaf((float)i + 0.5f, g.vs, g.ox), py_ = fmaf((float)j + 0.5f, g.vs, g.oy), pz = fmaf((float)k + 0.5f, g.vs, g.oz);
    const float x = fmaf(pose.r[0], px, fmaf(pose.r[1], py_, fmaf(pose.r[2], pz, pose.t[0])));
    const float y = fmaf(pose.r[3], px, fmaf(pose.r[4], py_, fmaf(pose.r[5], pz, pose.t[1])));
    const float z = fmaf(pose.r[6], px, fmaf(pose.r[7], py_, fmaf(pose.r[8], pz, pose.t[2])));
    const float zmin = group8_min(z), zmax = group8_max(z);
    int cls = 1;
    if (zmin > 1e-3f) {
        const float iz = __builtin_amdgcn_rcpf(z);                 // 1 ulp; the 1.5 px margin absorbs it
        const float u = cam.fx * x * iz + cam.cx, v = cam.fy * y * iz + cam.cy;
        const float umin = group8_min(u) - 1.5f, umax = group8_max(u) + 1.5f;
        const float vmin = group8_min(v) - 1.5f, vmax = group8_max(v) + 1.5f;
        if (umax < 0.0f || vmax < 0.0f || umin > (float)(cam.W - 1) || vmin > (float)(cam.H - 1)) {
            cls = 0;
        } else {
            const bool inside = umin >= 0.0f && vmin >= 0.0f && umax <= (float)(cam.W - 1) && vmax <= (float)(cam.H - 1);
            const int pu0 = max(0, (int)floorf(umin)), pu1 = min(cam.W - 1, (int)ceilf(umax));
            const int pv0 = max(0, (int)floorf(vmin)), pv1 = min(cam.H - 1, (int)ceilf(vmax));
            // finest level at which the box spans at most 8 tiles; level geometry by arithmetic (no table lookups per lane)
            int L = 0, off = 0, ntx = py.ntx[0], nty = py.nty[0];
            int tu0, tv0, nu, nv;
            for (;;) {
                tu0 = pu0 >> (TILE0_SHIFT + L); tv0 = pv0 >> (TILE0_SHIFT + L);
                nu = (pu1 >> (TILE0_SHIFT + L)) - tu0 + 1; nv = (pv1 >> (TILE0_SHIFT + L)) - tv0 + 1;
                if (nu * nv <= 8 || L >= py.nlev - 1) break;
                off += ntx * nty;
                ntx = (ntx + 1) >> 1; nty = (nty + 1) >> 1;
                ++L;
            }
            const int q = cn < nu * nv ? cn : 0;
            const int qr = (int)(((float)q + 0.5f) * __builtin_amdgcn_rcpf((float)nu));      // q / nu for 0 <= q < 8, 1 <= nu <= 8
            const float4 b = tiles[off + (tv0 + qr) * ntx + (tu0 + (q - qr * nu))];
            const float4 a = make_float4(group8_min(b.x), group8_max(b.y), group8_min(b.z), 0.0f);
            cls = classify_box(g, a, zmin, zmax, inside);
        }
    }
    mixed = group_bits(__ballot(cls == 1));
    free_ = group_bits(__ballot(cls == 2));
}

// Prep kernel 4: sub-brick masks of the listed bricks, one wave per brick; and, for the second frame of a prepared pair
// (blockIdx.y == 1, pair != 0), the compact list of its bricks that the first frame does not list (lane-parallel scan of the
// list, one atomic per workgroup).  Runs behind brick_cull_kernel of BOTH frames on the same stream.
__global__ __launch_bounds__(256) void subbrick_classify_kernel(Cam cam, Grid g, Pyramid py, PrepFrames P, int pair) {
    const PrepFrame &F = P.f[blockIdx.y];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    const unsigned n = min(F.list_counts[0], nbricks);
    if (pair && blockIdx.y == 1) {
        __shared__ unsigned s_cnt[4], s_base;
        const unsigned char *__restrict__ cls_prev = P.f[0].cls;
        for (unsigned t0 = blockIdx.x * 256u; t0 < n; t0 += gridDim.x * 256u) {
            const unsigned t = t0 + threadIdx.x;
            unsigned brick = 0;
            bool solo = false;
            if (t < n) {
                brick = min(F.list[t], nbricks - 1u);
                solo = cls_prev[brick] != 1;
            }
            const unsigned long long m = __ballot(solo);
            if (lane == 0) s_cnt[wid] = (unsigned)__popcll(m);
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
                s_base = tot ? atomicAdd(F.list_counts + 2, tot) : 0u;
            }
            __syncthreads();
            unsigned base = s_base;
            for (int w = 0; w < wid; ++w) base += s_cnt[w];
            if (solo) F.solo[base + __popcll(m & ((1ull << lane) - 1ull))] = brick;
            __syncthreads();
        }
    }
    for (unsigned t = blockIdx.x * 4u + wid; t < n; t += gridDim.x * 4u) {
        const unsigned brick = min((unsigned)__builtin_amdgcn_readfirstlane((int)F.list[t]), nbricks - 1u);
        const int bx = (int)(brick % (unsigned)g.nbx), by = (int)((brick / (unsigned)g.nbx) % (unsigned)g.nby), bz = (int)(brick / (unsigned)(g.nbx * g.nby));
        unsigned mixed, free_;
        classify_subbricks(cam, g, py, F.pose, F.tiles, bx, by, bz, lane, mixed, free_);
        if (lane == 0) F.sub[brick] = (unsigned short)(mixed | (free_ << 8));
    }
}

// One frame (NF = 1) or two consecutive frames (NF = 2) per visit of a brick.  The gathers are per frame; the records need
// not be: a brick that both frames see near a surface is read and written ONCE for both (integer sums: the same grid bit for
// bit).  Tasks = frame A's MIXED list, then frame B's; a brick of A's list that is MIXED in B too (B's class map) takes both
// frames' gathers before its one read-modify-write; a brick of B's list that was on A's is skipped.
// EXP (experiments flavour of the library only; results incomplete): bit 0 no depth gathers, bit 1 no record accesses, bit 2 no
// sub-brick classification (every sub-brick of a listed brick is treated as MIXED)
template <bool COUNT, int NF, typename DT, int EXP = 0>
__global__ __launch_bounds__(256) void tsdf_update_kernel(Cam cam, Grid g, UpdFrame A, UpdFrame B, int xcd_group,
                                                          int2 *__restrict__ grid, unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    // the trip counts come from device memory: clamp them to what a list can hold, so that no ordering mistake upstream can
    // ever turn into an unbounded loop or an out-of-range list read
    const unsigned nA = min(A.counts[0], nbricks), nB = NF == 2 ? min(B.counts[2], nbricks) : 0u;      // B: its solo list
    unsigned nread = 0, nwritten = 0;
    // Blocks with the same blockIdx % 8 share an XCD and its L2 (a placement habit of the dispatcher: a speed choice, never
    // a correctness one).  Each such group takes one contiguous eighth of BOTH lists -- the classification emits bricks in
    // grid order, so an eighth is a slab of the volume and its depth lookups stay in one region of the image
    const bool grouped = XCD_GROUPS > 1 && (gridDim.x % XCD_GROUPS) == 0 && xcd_group != 0;
    const unsigned ngrp = grouped ? XCD_GROUPS : 1u;
    const unsigned grp = grouped ? blockIdx.x % XCD_GROUPS : 0u, bi = grouped ? blockIdx.x / XCD_GROUPS : blockIdx.x;
    const unsigned perA = (nA + ngrp - 1u) / ngrp, perB = (nB + ngrp - 1u) / ngrp;
    const unsigned a_beg = min(nA, grp * perA), a_len = min(nA, a_beg + perA) - a_beg;
    const unsigned b_beg = min(nB, grp * perB), b_len = min(nB, b_beg + perB) - b_beg;
    const unsigned lstep = (gridDim.x / ngrp) * 4u;
    const DT *__restrict__ depthA = static_cast<const DT *>(A.depth);
    const DT *__restrict__ depthB = static_cast<const DT *>(B.depth);

    // mA / fA / mB / fB: the brick's MIXED and FREE sub-brick masks in frame A and B (wave-uniform; zero for a frame that does
    // not list the brick)
    auto process = [&](unsigned brick, unsigned mA, unsigned fA, unsigned mB, unsigned fB) {
        const int bx = (int)(brick % (unsigned)g.nbx), by = (int)((brick / (unsigned)g.nbx) % (unsigned)g.nby), bz = (int)(brick / (unsigned)(g.nbx * g.nby));
        if (EXP & 4) {
            if (mA | fA) { mA = 0xffu; fA = 0u; }
            if (mB | fB) { mB = 0xffu; fB = 0u; }
        }
        const unsigned any = mA | fA | mB | fB;
        if (any == 0u) return;
        int2 *__restrict__ recs = grid + ((size_t)brick << 9);
        // the lane's voxel in sub-brick s: (x, y, z) = (4 (s & 1) + (lane & 3), 4 (s >> 1 & 1) + (lane >> 2 & 3), 4 (s >> 2) + (lane >> 4)),
        // record s * 64 + lane; world coordinates: two values per axis
        float wx[2], wy[2], wz[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            wx[h] = fmaf((float)(bx * 8 + 4 * h + (lane & 3)) + 0.5f, g.vs, g.ox);
            wy[h] = fmaf((float)(by * 8 + 4 * h + ((lane >> 2) & 3)) + 0.5f, g.vs, g.oy);
            wz[h] = fmaf((float)(bz * 8 + 4 * h + (lane >> 4)) + 0.5f, g.vs, g.oz);
        }
        // phase 1: project the lane's voxel of every MIXED sub-brick of every frame, issue all depth gathers back to back
        float zc[NF][8], dv[NF][8];
        unsigned okm[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            okm[f] = 0u;
            const unsigned m = f ? mB : mA;
            if (m == 0u) continue;
            const PoseF &pose = f ? B.pose : A.pose;
            const TsdfConst &c = f ? B.c : A.c;
            const DT *__restrict__ depth = f ? depthB : depthA;
            float ax[2][2], ay[2][2], az[2][2];                    // [z half][y half]: the two inner fma levels of the position chain
#pragma unroll
            for (int hz = 0; hz < 2; ++hz)
#pragma unroll
                for (int hy = 0; hy < 2; ++hy) {
                    ax[hz][hy] = fmaf(pose.r[1], wy[hy], fmaf(pose.r[2], wz[hz], pose.t[0]));
                    ay[hz][hy] = fmaf(pose.r[4], wy[hy], fmaf(pose.r[5], wz[hz], pose.t[1]));
                    az[hz][hy] = fmaf(pose.r[7], wy[hy], fmaf(pose.r[8], wz[hz], pose.t[2]));
                }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (!((m >> s) & 1u)) continue;
                const float xc = fmaf(pose.r[0], wx[s & 1], ax[s >> 2][(s >> 1) & 1]);
                const float yc = fmaf(pose.r[3], wx[s & 1], ay[s >> 2][(s >> 1) & 1]);
                zc[f][s] = fmaf(pose.r[6], wx[s & 1], az[s >> 2][(s >> 1) & 1]);
                int pix;
                if (tsdf_project(cam, c, xc, yc, zc[f][s], pix)) okm[f] |= 1u << s;
                dv[f][s] = (EXP & 1) ? 1.0f + 1e-6f * (float)(pix & 1023) : ld_depth(depth, (size_t)pix);
            }
        }
        // phase 2: decide; a FREE sub-brick adds (32767, 1) to every one of its voxels
        int qs[8], ws[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            qs[s] = 0; ws[s] = 0;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const unsigned m = f ? mB : mA, fr = f ? fB : fA;
                if ((m >> s) & 1u) {
                    int q;
                    if (tsdf_finish(g, f ? B.c : A.c, (okm[f] >> s) & 1u, dv[f][s], zc[f][s], q)) { qs[s] += q; ws[s] += 1; }
                } else if ((fr >> s) & 1u) {
                    qs[s] += 32767; ws[s] += 1;
                }
            }
        }
        // phase 3: load the records that change (8 B per lane, a sub-brick = one 512-B run); phase 4: add and store them
        // (experiments, EXP bits 3-4: which lanes move a record -- 0: the lanes whose voxel changes; 8: all 64 lanes of a sub-brick in
        //  which any voxel changes; 16: all 16 lanes of a 128-B line in which any voxel changes)
        constexpr int RMW = (EXP >> 3) & 3;
        int2 rec[8];
        bool mv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            mv[s] = false;
            if (!((any >> s) & 1u)) continue;
            if (RMW == 0) mv[s] = ws[s] != 0;
            else {
                const unsigned long long bal = __ballot(ws[s] != 0);
                mv[s] = RMW == 1 ? bal != 0ull : ((bal >> (lane & 48)) & 0xffffull) != 0ull;
            }
            if (mv[s]) rec[s] = (EXP & 2) ? make_int2(qs[s] ^ lane, s) : recs[s * 64 + lane];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (mv[s]) {
                rec[s].x += qs[s];
                rec[s].y += ws[s];
                if (EXP & 2) { if (rec[s].x == cam.W * 7919 + lane) recs[0] = rec[s]; }      // keep the values alive, (practically) never store
                else recs[s * 64 + lane] = rec[s];
                if (COUNT) { nread += 1; nwritten += 1; }
            }
    };
    // one list entry per wave and trip, A's list then B's solo list; the entry of the NEXT trip (brick, its sub-brick masks and,
    // for A's entries, its class and masks in frame B) is fetched before this trip's brick is processed
    {
        const unsigned ntot = a_len + b_len;
        auto fetch = [&](unsigned t, unsigned &brick, unsigned &sa, unsigned &sb) {
            const bool fromA = t < a_len;
            brick = min((unsigned)__builtin_amdgcn_readfirstlane((int)(fromA ? A.list[a_beg + t] : B.list[b_beg + (t - a_len)])), nbricks - 1u);
            sa = fromA ? (unsigned)__builtin_amdgcn_readfirstlane((int)A.sub[brick]) : 0u;
            sb = 0u;
            if (NF == 2) {
                const bool listedB = fromA ? __builtin_amdgcn_readfirstlane((int)B.cls[brick]) == 1 : true;
                const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)B.sub[brick]);
                sb = listedB ? v : 0u;
            }
        };
        unsigned t = bi * 4u + wid;
        unsigned brick_n = 0, sa_n = 0, sb_n = 0;
        if (t < ntot) fetch(t, brick_n, sa_n, sb_n);
        for (; t < ntot; t += lstep) {
            const unsigned brick = brick_n, sa = sa_n, sb = sb_n;
            if (t + lstep < ntot) fetch(t + lstep, brick_n, sa_n, sb_n);
            process(brick, sa & 0xffu, sa >> 8, sb & 0xffu, sb >> 8);
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
            unsigned long long vis = 0, fre = 0;
            {
                const unsigned f = min(A.counts[1], nbricks - nA);
                vis += nA + f; fre += f;
            }
            if (NF == 2) {
                const unsigned lb = min(B.counts[0], nbricks), f = min(B.counts[1], nbricks - lb);
                vis += lb + f; fre += f;
            }
            atomicAdd(counters + 4, vis);
            atomicAdd(counters + 5, fre);
            atomicAdd(counters + 6, fre);
        }
    }
}

static Pyramid make_pyramid(const Cam &cam) {
    Pyramid p;
    memset(&p, 0, sizeof(p));
    int nx = (cam.W + TILE0 - 1) / TILE0, ny = (cam.H + TILE0 - 1) / TILE0, off = 0, L = 0;
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
    return 256 + pyramid_tiles(p) * sizeof(float4) + 2 * (nbricks + 64) * sizeof(unsigned) + ((nbricks + 255) & ~(size_t)255) +
           ((nbricks * sizeof(unsigned short) + 255) & ~(size_t)255);
}

struct TsdfScratch {
    unsigned *list_counts;       // [0] mixed (listed), [1] free (counted)
    float4 *tiles;
    unsigned *list, *solo;
    unsigned char *cls;          // [nbricks] class of every brick (0 skip, 1 mixed, 2 free)
    unsigned short *sub;         // [nbricks] sub-brick masks of the listed bricks
    Pyramid py;
};

static TsdfScratch carve(const Cam &cam, const Grid &g, void *scratch) {
    TsdfScratch t;
    t.py = make_pyramid(cam);
    // scratch layout: [list_counts (256 B)] [tile pyramid] [brick list] [solo list] [brick classes] [sub-brick masks]
    const size_t nbricks = (size_t)g.nbx * g.nby * g.nbz;
    t.list_counts = reinterpret_cast<unsigned *>(scratch);
    t.tiles = reinterpret_cast<float4 *>(reinterpret_cast<char *>(scratch) + 256);
    t.list = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(scratch) + 256 + pyramid_tiles(t.py) * sizeof(float4));
    t.solo = t.list + (nbricks + 64);
    t.cls = reinterpret_cast<unsigned char *>(t.solo + (nbricks + 64));
    t.sub = reinterpret_cast<unsigned short *>(t.cls + ((nbricks + 255) & ~(size_t)255));
    return t;
}

static TsdfConst make_const(const Cam &cam, float scale, float mind, float maxd) {
    TsdfConst c;
    c.mind = mind;
    c.maxd = maxd;
    c.sc = scale;
    c.wlim = (float)cam.W - 0.5f;
    c.hlim = (float)cam.H - 0.5f;
    return c;
}

// depth tiles + pyramid + brick classification + sub-brick masks -> compact brick list in scratch, for one frame or for two
// in the same four launches (n = 2: both depth images of one kind; the second frame then also gets its solo list and may be
// updated together with the first)
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
        P.f[i].sub = t.sub;
        P.f[i].solo = t.solo;
        P.f[i].c = make_const(cam, scale[i], mind, maxd);
        P.f[i].pose = p[i];
    }
    const int nrx = (cam.W + REGION - 1) / REGION, nry = (cam.H + REGION - 1) / REGION;
    const int nreg = nrx * nry;
    if (depth_u16)
        hipLaunchKernelGGL(depth_tiles_kernel<uint16_t>, dim3(nreg < 1024 ? nreg : 1024, n), dim3(256), 0, s, cam, P, py, nrx, nry);
    else
        hipLaunchKernelGGL(depth_tiles_kernel<float>, dim3(nreg < 1024 ? nreg : 1024, n), dim3(256), 0, s, cam, P, py, nrx, nry);
    TL3D_HIP(hipGetLastError());
    if (py.nlev > 3) {
        hipLaunchKernelGGL(tile_pyramid_kernel, dim3(n), dim3(256), 0, s, py, P);
        TL3D_HIP(hipGetLastError());
    }
    const int ncells = ((g.nbx + 3) / 4) * ((g.nby + 3) / 4) * ((g.nbz + 3) / 4);
    hipLaunchKernelGGL(brick_cull_kernel, dim3((ncells + 3) / 4, n), dim3(256), 0, s, cam, g, P, fr, py, free_cnt);
    TL3D_HIP(hipGetLastError());
    // sub-brick masks of the listed bricks (their number is known only on the device: a fixed grid strides over the list)
    const int nbricks = g.nbx * g.nby * g.nbz;
    const int ncb = (nbricks + 3) / 4 < 1024 ? (nbricks + 3) / 4 : 1024;
    hipLaunchKernelGGL(subbrick_classify_kernel, dim3(ncb, n), dim3(256), 0, s, cam, g, py, P, n == 2 ? 1 : 0);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_fold_free(hipStream_t s, const Grid &g, int2 *grid, unsigned *free_cnt) {
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    const unsigned nb = (nbricks + 3u) / 4u < 2048u ? (nbricks + 3u) / 4u : 2048u;
    hipLaunchKernelGGL(fold_free_kernel, dim3(nb ? nb : 1), dim3(256), 0, s, grid, free_cnt, nbricks);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// the dominant kernel: read-modify-write of the listed bricks of one prepared frame (n = 1) or of two that were prepared
// TOGETHER by one launch_tsdf_prepare call (n = 2: the second frame's solo list exists)
int launch_tsdf_update(hipStream_t s, const Cam &cam, const Grid &g, int n, const PoseF *p, const void *const *depth, bool depth_u16,
                       const float *scale, float mind, float maxd, int2 *grid, void *const *scratch, unsigned long long *counters, bool count,
                       int max_blocks, int xcd_group) {
    UpdFrame F[2];
    memset(F, 0, sizeof(F));
    for (int i = 0; i < n; ++i) {
        const TsdfScratch t = carve(cam, g, scratch[i]);
        F[i].pose = p[i];
        F[i].c = make_const(cam, scale[i], mind, maxd);
        F[i].depth = depth[i];
        F[i].list = i == 0 ? t.list : t.solo;
        F[i].counts = t.list_counts;
        F[i].cls = t.cls;
        F[i].sub = t.sub;
    }
    if (n == 1) F[1] = F[0];
    const int nbricks = g.nbx * g.nby * g.nbz;
    // 6 workgroups per CU by default: the update saturates from 4 per CU upwards, and the prep kernels of later frames,
    // which run beside it, then get just the slots they need
    int nblk = (nbricks + 3) / 4;
    if (nblk > max_blocks) nblk = max_blocks;
#define TL3D_LAUNCH_UPD(C_, N_, T_, E_) \
    hipLaunchKernelGGL((tsdf_update_kernel<C_, N_, T_, E_>), dim3(nblk), dim3(256), 0, s, cam, g, F[0], F[1], xcd_group, grid, counters)
#define TL3D_LAUNCH_UPD_E(E_)                                                                                   \
    do {                                                                                                        \
        if (n == 2) { if (depth_u16) TL3D_LAUNCH_UPD(false, 2, uint16_t, E_); else TL3D_LAUNCH_UPD(false, 2, float, E_); } \
        else { if (depth_u16) TL3D_LAUNCH_UPD(false, 1, uint16_t, E_); else TL3D_LAUNCH_UPD(false, 1, float, E_); }        \
    } while (0)
#ifdef TL3D_EXPERIMENTS
    static const int exp_mode = getenv("TL3D_TSDF_EXP") ? atoi(getenv("TL3D_TSDF_EXP")) : 0;     // timing ablations: results incomplete
#else
    constexpr int exp_mode = 0;
#endif
    if (count) {
        if (n == 2) { if (depth_u16) TL3D_LAUNCH_UPD(true, 2, uint16_t, 0); else TL3D_LAUNCH_UPD(true, 2, float, 0); }
        else { if (depth_u16) TL3D_LAUNCH_UPD(true, 1, uint16_t, 0); else TL3D_LAUNCH_UPD(true, 1, float, 0); }
    }
#ifdef TL3D_EXPERIMENTS
    else if (exp_mode == 1) TL3D_LAUNCH_UPD_E(1);
    else if (exp_mode == 2) TL3D_LAUNCH_UPD_E(2);
    else if (exp_mode == 3) TL3D_LAUNCH_UPD_E(3);
    else if (exp_mode == 4) TL3D_LAUNCH_UPD_E(4);
    else if (exp_mode == 5) TL3D_LAUNCH_UPD_E(5);
    else if (exp_mode == 6) TL3D_LAUNCH_UPD_E(6);
    else if (exp_mode == 8) TL3D_LAUNCH_UPD_E(8);
    else if (exp_mode == 9) TL3D_LAUNCH_UPD_E(9);
    else if (exp_mode == 16) TL3D_LAUNCH_UPD_E(16);
    else if (exp_mode == 17) TL3D_LAUNCH_UPD_E(17);
#endif
    else TL3D_LAUNCH_UPD_E(0);
    (void)exp_mode;
#undef TL3D_LAUNCH_UPD_E
#undef TL3D_LAUNCH_UPD
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

}  // namespace tl3d
