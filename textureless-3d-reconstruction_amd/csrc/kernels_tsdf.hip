// kernels_tsdf.hip -- row a11: TSDF integration of depth frames into the brick-major grid.
// No reference code exists for this row; the convention is the oracle's
// (oracle/tl3d_oracle.c: orc_tsdf_integrate) and both sides evaluate the same f32 sequence, so the integer
// grid {sum of rint(tsdf*32767), weight} is bit-identical.
//
// tl3d_integrate collects frames into BATCHES of up to 32 (TL3D_TSDF_MAXBATCH).  Per batch, five launches; the per-frame
// arguments (pose, depth image, scratch pointers) of all its frames sit in one descriptor array in device memory and
// blockIdx.y / a bit index picks the frame:
//   1. depth_tiles_kernel     (min, max, all-valid) of the valid scaled depth over 8x8-, 16x16- and 32x32-pixel tiles
//                             (pyramid levels 0-2) of every frame, 4 B/pixel read; the 8x8 level once more as 8-B records
//                             for the update kernel's per-voxel test
//   2. tile_pyramid_kernel    1 workgroup per frame: 2x2 reductions of level 2 up to a single tile; one valid pixel of the frame
//   3. brick_cull_kernel      per frame, one lane per 8^3 brick (one wave per 4x4x4-brick cell) classifies it from <= 16
//                             pyramid lookups:
//        SKIP   outside the view, no valid depth under it, or more than trunc behind every surface it can see
//        FREE   wholly inside the image, every pixel under it valid, and at least trunc in front of every
//               surface: every voxel gets exactly tsdf = 1 (q = 32767): ONE add to the brick's free-space counter
//        MIXED  everything else (near a surface, on the image border, straddling the camera plane): the frame's compact
//               list, its bit in the brick's frame mask, and -- if it is the first frame of the batch to list the brick --
//               the batch's brick list
//   4. subbrick_classify_kernel  per frame, one wave per listed brick: its eight 4x4x4 SUB-BRICKS by the same three rules
//        (8 lanes per sub-brick: one extreme voxel centre each, <= 8 pyramid lookups) -> a (mixed, free) bit mask.  On the
//        headline sequence 3.4 of the 8 sub-bricks of a listed brick stay MIXED.
//   5. tsdf_update_kernel     ONE launch per batch, one 64-lane wave per brick of the batch list: for every frame that lists
//        the brick (frame mask), the voxels of its MIXED sub-bricks are projected; the 8x8-pixel tile under a voxel's pixel
//        decides most of them (free / behind everything), the rest gather their depth value; FREE sub-bricks add (32767, 1);
//        the increments of all frames are summed in registers and the brick's records are read and written ONCE per batch
//        (integer sums commute: the grid is the one-frame-at-a-time grid bit for bit).  A sub-brick's 64 records are
//        contiguous (tl3d_internal.h: in_brick_index): every record access of a wave is one 512-B run.
// Every classification is conservative with respect to the per-voxel rule, so the result is the oracle's bit
// for bit whatever the view.
// Algorithmic bytes per launch = 8 B x (records read + written), counted by the kernel in counting mode
// ("counted, never estimated"; F = frames of the batch is reported), + 8 B per counted free-space
// brick + the depth images of the batch's frames.
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
constexpr int TICKET_GROUPS = 64;        // ticket counters of the update kernel (one per group of workgroups)
constexpr unsigned HDR_TICKETS = 64;     // batch header (unsigned words): [0] batch list length, [64 + 16 g] ticket counter of group g
constexpr unsigned HDR_HIST = HDR_TICKETS + 16 * TICKET_GROUPS;      // [33] bricks of the batch list by the number of frames that list them ...
constexpr unsigned HDR_CURSOR = HDR_HIST + 64;                       // [33] ... and the cursors of the counting sort (batch_order_kernel)
constexpr unsigned HDR_WORDS = HDR_CURSOR + 64;

struct Pyramid {
    int nlev;
    int ntx[MAX_LEVELS], nty[MAX_LEVELS], off[MAX_LEVELS];     // per level: tiles in x / y, offset into the float4 array
};

// One frame of a batch.  The array of a batch's descriptors is written by the host and copied to the device in front of the
// batch's first kernel; every kernel indexes it with a wave-uniform frame number (scalar loads).
struct FrameDesc {
    PoseF pose;                          // [0, 48)
    TsdfConst c;                         // [48, 68)
    unsigned valid_pix;                  // [68] index of ONE pixel of the frame that is valid (written by tile_pyramid_kernel); the update
                                         //      kernel sends the lanes whose depth value cannot matter there
    const void *depth;                   // [72] f32 metres, or the 16-bit millimetre image (one kind per batch)
    float2 *vtile;                       // [80] level-0 tiles, 8 B each, as the update kernel's per-voxel test wants them (below)
    float4 *tiles;                       // the frame's tile pyramid
    unsigned *list;                      // its listed (MIXED) bricks, for the sub-brick classification
    unsigned *counts;                    // [0] listed bricks, [1] free-space bricks (counted)
    unsigned short *sub;                 // [nbricks] sub-brick masks of its listed bricks: bits 0-7 mixed, bits 8-15 free
    unsigned *cells;                     // the 4x4x4-brick cells whose bounding sphere meets the view (the cull kernel's work list); count: counts[CELL_COUNT]
};
static_assert(sizeof(FrameDesc) == 128, "FrameDesc: 16 of them travel as one kernel argument block");

// what all frames of a batch share
struct BatchBufs {
    const FrameDesc *frames;             // [n_frames]
    unsigned *hdr;                       // [0] length of the batch list (reset by the tiles kernel)
    unsigned *list;                      // bricks that at least one frame of the batch lists, in order of first listing
    unsigned *order;                     // the same bricks, dearest first (by the number of frames that list them): what the update walks
    unsigned *framemask;                 // [nbricks] bit f: frame f lists the brick; all zero between batches (the update re-arms it)
    int n_frames;
};

// The descriptors do not change while a kernel that reads them runs: read through the constant address space, a wave-uniform
// index turns into scalar loads (pose and limits live in SGPRs) -- through a plain pointer the compiler must assume that the
// kernel's own stores could alias them and uses vector loads, 17 VGPRs per frame in flight.
typedef const FrameDesc __attribute__((address_space(4))) *DescPtr;
__device__ __forceinline__ DescPtr const_descs(const BatchBufs &B) { return (DescPtr)(B.frames); }
__device__ __forceinline__ PoseF desc_pose(DescPtr d) {
    PoseF p;
#pragma unroll
    for (int i = 0; i < 9; ++i) p.r[i] = d->pose.r[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) p.t[i] = d->pose.t[i];
    return p;
}
__device__ __forceinline__ TsdfConst desc_const(DescPtr d) {
    TsdfConst c;
    c.mind = d->c.mind; c.maxd = d->c.maxd; c.sc = d->c.sc; c.wlim = d->c.wlim; c.hlim = d->c.hlim;
    return c;
}

// The descriptors travel to the device as the ARGUMENTS of a tiny kernel (captured at launch: no staging buffer whose lifetime
// the host would have to track), 16 per launch (kernel arguments are limited to 4 KB).
struct DescChunk { FrameDesc d[16]; };
__global__ __launch_bounds__(256) void desc_upload_kernel(DescChunk c, FrameDesc *__restrict__ dst, int n) {
    const unsigned *src = reinterpret_cast<const unsigned *>(&c);
    unsigned *out = reinterpret_cast<unsigned *>(dst);
    const int words = n * (int)(sizeof(FrameDesc) / 4);
    for (int i = threadIdx.x; i < words; i += 256) out[i] = src[i];
}

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
// One 32x32-pixel region per WAVE.  Lane (q4 = lane & 7, rq = lane >> 3) holds the 4x4-pixel block at columns 4 q4 .. 4 q4 + 3, rows
// 4 rq .. 4 rq + 3: load i of a lane is row 4 rq + i, so every load instruction of the wave covers eight whole 128-byte row
// segments, and the lane's four 16-B loads are issued before the first is looked at.  The block is reduced inside the lane; an 8x8
// tile is 2 x 2 lanes (lane bits 0 and 3), a 16x16 tile 4 x 4 (+ bits 1 and 4), the region all of them (+ bits 2 and 5): SIX
// butterfly steps for the three levels.  (Round 3 gave a lane four rows 8 apart: every row of 8x8 tiles was a 16-lane
// reduction of its own, 19 steps; the kernel's 33 M vector instructions per 32 frames were a sixth of the batch's -- and the batch is
// bound by exactly those)  min / max over the VALID pixels: an invalid one enters as NaN, which v_min_f32 / v_max_f32
// pass over.  No LDS, no workgroup barrier.
template <typename DT>
__global__ __launch_bounds__(256) void depth_tiles_kernel(Cam cam, BatchBufs B, Pyramid py, int nrx, int nry) {
    const DescPtr F = const_descs(B) + blockIdx.y;
    const TsdfConst c = desc_const(F);
    const DT *__restrict__ depth = static_cast<const DT *>(F->depth);
    float4 *__restrict__ tiles = F->tiles;
    float2 *__restrict__ vtile = F->vtile;
    if (blockIdx.x == 0 && threadIdx.x < 2) F->counts[threadIdx.x] = 0u;        // reset the frame's list cursors
    if (blockIdx.x == 0 && blockIdx.y == 0) {                                    // and the batch list's, the update's ticket counters, the ordering pass's bins
        if (threadIdx.x == 2) B.hdr[0] = 0u;
        if (threadIdx.x >= 64 && threadIdx.x < 64 + TICKET_GROUPS) B.hdr[HDR_TICKETS + 16u * (threadIdx.x - 64)] = 0u;
        if (threadIdx.x >= 128) B.hdr[HDR_HIST + (threadIdx.x - 128)] = 0u;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int q4 = lane & 7, rq = lane >> 3;
    const bool vec = (cam.W & 3) == 0;                              // rows are 16-B aligned
    const float qnan = __uint_as_float(0x7fc00000u);
    for (int region = blockIdx.x * 4 + wid; region < nrx * nry; region += gridDim.x * 4) {
        const int rx = region % nrx, ry = region / nrx;
        const int u0 = rx * REGION + q4 * 4, v0 = ry * REGION + rq * 4;
        const int nv = min(4, cam.W - u0);
        float dd[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dd[i][0] = dd[i][1] = dd[i][2] = dd[i][3] = 0.0f;
            if (v0 + i < cam.H && u0 < cam.W) {
                if (vec) {
                    ld_depth4(depth, (size_t)(v0 + i) * cam.W + u0, dd[i]);
                } else {
                    for (int k = 0; k < 4; ++k) dd[i][k] = (k < nv) ? ld_depth(depth, (size_t)(v0 + i) * cam.W + u0 + k) : 0.0f;
                }
            }
        }
        // the lane's block: lowest / highest VALID scaled depth, and whether a pixel of it (inside the image) is not valid
        float mn = qnan, mx = qnan;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = dd[i][k] * c.sc;
                const bool in_img = (v0 + i < cam.H) && (k < nv) && (u0 < cam.W);
                const bool ok = d > c.mind && d < c.maxd;
                t[k] = (ok && in_img) ? d : qnan;
                bad = bad || (in_img && !ok);
            }
            mn = fminf(fminf(mn, t[0]), fminf(t[1], fminf(t[2], t[3])));        // (NaN operands are passed over)
            mx = fmaxf(fmaxf(mx, t[0]), fmaxf(t[1], fmaxf(t[2], t[3])));
        }
        // from here on: +inf / -inf for "no valid pixel" (the format of the pyramid), the flag as a number
        mn = (mn == mn) ? mn : INFINITY;
        mx = (mx == mx) ? mx : -INFINITY;
        int bd = bad ? 1 : 0;
        auto combine = [&](int d) {
            mn = fminf(mn, __shfl_xor(mn, d));
            mx = fmaxf(mx, __shfl_xor(mx, d));
            bd |= __shfl_xor(bd, d);
        };
        combine(1); combine(8);                                     // level 0: 8x8 pixels
        if ((lane & 9) == 0) {
            const int tx = rx * 4 + (q4 >> 1), ty = ry * 4 + (rq >> 1);
            if (tx < py.ntx[0] && ty < py.nty[0]) {
                tiles[py.off[0] + ty * py.ntx[0] + tx] = make_float4(mn, mx, bd ? 0.0f : 1.0f, 0.0f);
                // (lowest depth if EVERY pixel is valid, else -inf: "never surely free"; highest valid depth, -inf if none: "skip")
                vtile[ty * py.ntx[0] + tx] = make_float2(bd ? -INFINITY : mn, mx);
            }
        }
        combine(2); combine(16);                                    // level 1: 16x16
        if ((lane & 27) == 0 && py.nlev > 1) {
            const int tx = rx * 2 + (q4 >> 2), ty = ry * 2 + (rq >> 2);
            if (tx < py.ntx[1] && ty < py.nty[1]) tiles[py.off[1] + ty * py.ntx[1] + tx] = make_float4(mn, mx, bd ? 0.0f : 1.0f, 0.0f);
        }
        combine(4); combine(32);                                    // level 2: the region
        if (lane == 0 && py.nlev > 2) tiles[py.off[2] + ry * py.ntx[2] + rx] = make_float4(mn, mx, bd ? 0.0f : 1.0f, 0.0f);
    }
}

// ---- 2. pyramid: levels 3 .. from level 2 ------------------------------------------------------------------
// Also finds ONE PIXEL OF THE FRAME THAT IS VALID (the top-left pixel of an all-valid tile; counts[VALID_PIXEL]): where the update
// kernel sends the lanes whose depth value cannot matter (below).
constexpr int VALID_PIXEL = 32;          // word of a frame's counts block
constexpr int CELL_COUNT = 48;           // word of a frame's counts block: length of the frame's cell list
__device__ __forceinline__ bool sphere_in_view(const Frustum &fr, float x, float y, float z, float rad) {
    return (z + rad > 0.0f) && (fr.lx * x + fr.lz * z >= -rad) && (fr.rx * x + fr.rz * z >= -rad) &&
           (fr.ty * y + fr.tz * z >= -rad) && (fr.by * y + fr.bz * z >= -rad);
}

// ... and the frame's CELL LIST: the 4x4x4-brick cells whose bounding sphere meets the view (one in five on the headline orbit), so
// that the brick classification starts waves only for those.
__global__ __launch_bounds__(256) void tile_pyramid_kernel(Cam cam, Grid g, Frustum fr, Pyramid py, BatchBufs B) {
    float4 *__restrict__ tiles = const_descs(B)[blockIdx.x].tiles;
    __shared__ unsigned s_found, s_ncell;
    if (threadIdx.x == 0) { s_found = 0xffffffffu; s_ncell = 0u; }
    __syncthreads();
    {
        const DescPtr F = const_descs(B) + blockIdx.x;
        const PoseF pose = desc_pose(F);
        unsigned *__restrict__ cells = F->cells;
        const int ncx = (g.nbx + 3) >> 2, ncy = (g.nby + 3) >> 2, ncz = (g.nbz + 3) >> 2;
        const float crad = 27.712812f * g.vs * 1.01f;               // half diagonal of a 32^3-voxel cell, +1 % (the cull kernel's own test)
        for (int c0 = 0; c0 < ncx * ncy * ncz; c0 += 256) {
            const int cell = c0 + (int)threadIdx.x;
            bool in = false;
            if (cell < ncx * ncy * ncz) {
                const int ccx = cell % ncx, ccy = (cell / ncx) % ncy, ccz = cell / (ncx * ncy);
                const float qx = fmaf((float)(ccx * 32 + 16), g.vs, g.ox), qy = fmaf((float)(ccy * 32 + 16), g.vs, g.oy);
                const float qz = fmaf((float)(ccz * 32 + 16), g.vs, g.oz);
                const float ex = pose.r[0] * qx + pose.r[1] * qy + pose.r[2] * qz + pose.t[0];
                const float ey = pose.r[3] * qx + pose.r[4] * qy + pose.r[5] * qz + pose.t[1];
                const float ez = pose.r[6] * qx + pose.r[7] * qy + pose.r[8] * qz + pose.t[2];
                in = sphere_in_view(fr, ex, ey, ez, crad);
            }
            const unsigned long long m = __ballot(in);
            unsigned base = 0;
            if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(&s_ncell, (unsigned)__popcll(m));
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (in) cells[base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (unsigned)cell;
        }
        __syncthreads();
        if (threadIdx.x == 0) F->counts[CELL_COUNT] = s_ncell;
    }
    for (int L = min(2, py.nlev - 1); L >= 0; L -= 2) {           // 32-pixel tiles first; only a frame full of holes needs the 8-pixel ones
        const int n = py.ntx[L] * py.nty[L];
        for (int i = threadIdx.x; i < n; i += 256)
            if (tiles[py.off[L] + i].z > 0.5f) {
                const unsigned x = (unsigned)(i % py.ntx[L]) << (TILE0_SHIFT + L), y = (unsigned)(i / py.ntx[L]) << (TILE0_SHIFT + L);
                atomicMin(&s_found, y * (unsigned)cam.W + x);
                break;
            }
        __syncthreads();
        if (s_found != 0xffffffffu) break;                        // (uniform: read after the barrier)
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned vp = s_found == 0xffffffffu ? 0u : s_found;
        const_descs(B)[blockIdx.x].counts[VALID_PIXEL] = vp;
        const_cast<FrameDesc *>(B.frames)[blockIdx.x].valid_pix = vp;      // (the update kernel of this batch reads it as a scalar)
    }
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
// One wave per LISTED cell of 4x4x4 bricks (the frame's cell list: cells whose bounding sphere meets the view), one lane per brick; the four waves of a workgroup pool their survivors so the list
// cursors see one atomic per workgroup, not per wave (same-address returning atomics retire at ~90 per microsecond).
// Margins: 1.5 px on projected bounds, 1 % of a voxel on depths, 0.1 % on the truncation distance.
// The three rules on one box of voxels, given what the pyramid says about the pixels under it: a = (min, max, all-valid) of
// the valid scaled depth over a superset of the box's pixel footprint, [zmin, zmax] the camera depths of its voxel centres,
// inside = the footprint (widened by 1.5 px) lies wholly in the image.  0 skip, 1 mixed, 2 free.
__device__ __forceinline__ int classify_box(const Grid &g, float4 a, float zmin, float zmax, bool inside) {
    const float m = 0.01f * g.vs;
    if (!(a.y > -INFINITY) || (zmin - m > a.y + g.trunc)) return 0;      // no valid depth under the box, or > trunc behind all it can see
    if (inside && a.z > 0.5f && (a.x - (zmax + m) >= g.trunc * 1.001f)) return 2;   // every voxel: in image, valid depth, sdf >= trunc => tsdf == 1 exactly
    return 1;
}

__global__ __launch_bounds__(256) void brick_cull_kernel(Cam cam, Grid g, BatchBufs B, Frustum fr, Pyramid py,
                                                         unsigned *__restrict__ free_cnt) {
    const DescPtr F = const_descs(B) + blockIdx.y;
    const PoseF pose = desc_pose(F);
    const float4 *__restrict__ tiles = F->tiles;
    unsigned *__restrict__ list = F->list;
    __shared__ unsigned s_cnt[4][3];
    __shared__ unsigned s_base[3];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int ncx = (g.nbx + 3) >> 2, ncy = (g.nby + 3) >> 2, ncz = (g.nbz + 3) >> 2;
    // the frame's cell list (tile_pyramid_kernel): four cells per workgroup and trip
    const unsigned ncell = min(F->counts[CELL_COUNT], (unsigned)(ncx * ncy * ncz));
    const unsigned *__restrict__ cells = F->cells;
    for (unsigned quad = blockIdx.x; quad * 4u < ncell; quad += gridDim.x) {
    const unsigned li = quad * 4u + (unsigned)wid;
    const int cell = li < ncell ? (int)min(cells[li], (unsigned)(ncx * ncy * ncz - 1)) : -1;
    int cls = 0;                                                  // 0 skip, 1 mixed, 2 free
    int brick = 0;
    if (cell >= 0) {
        const int ccx = cell % ncx, ccy = (cell / ncx) % ncy, ccz = cell / (ncx * ncy);
        const int bx = ccx * 4 + (lane & 3), by = ccy * 4 + ((lane >> 2) & 3), bz = ccz * 4 + (lane >> 4);
        const bool in_grid = bx < g.nbx && by < g.nby && bz < g.nbz;
        brick = (bz * g.nby + by) * g.nbx + bx;
        const float crad = 27.712812f * g.vs * 1.01f;               // half diagonal of a 32^3-voxel cell, +1 %
        const float qx = fmaf((float)(ccx * 32 + 16), g.vs, g.ox), qy = fmaf((float)(ccy * 32 + 16), g.vs, g.oy);
        const float qz = fmaf((float)(ccz * 32 + 16), g.vs, g.oz);
        const float ex = pose.r[0] * qx + pose.r[1] * qy + pose.r[2] * qz + pose.t[0];
        const float ey = pose.r[3] * qx + pose.r[4] * qy + pose.r[5] * qz + pose.t[1];
        const float ez = pose.r[6] * qx + pose.r[7] * qy + pose.r[8] * qz + pose.t[2];
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
                        // (A SPARSE grid classifies exactly as a dense one.  Until round 3 a brick that every VALID pixel under it
                        // put in free space, but whose footprint had holes or left the image, was skipped there -- "records for free
                        // space only" -- and the voxels of such a brick that came within the band in another frame lost those
                        // observations.  Such bricks take a pool slot now and the sparse grid is the oracle's grid bit for bit.)
                    }
                }
            }
        }
    }
    // MIXED: the brick's bit in the frame mask; the first frame of the batch to set a bit also puts the brick on the batch list
    bool first = false;
    if (cls == 1) {
        first = atomicOr(B.framemask + brick, 1u << blockIdx.y) == 0u;
        if (first) (void)brick_slot_ensure(g.tsdf_tab, g.cursors, g.tsdf_cap, (unsigned)brick);     // sparse grid: records on first touch
    }
    // Free space: every voxel of the brick gets exactly (+32767, +1).  That is ONE integer add here instead of a 4 KB read +
    // 4 KB write by the update kernel; the counters are folded into the records before anything reads them (fold_free_kernel).
    if (cls == 2) atomicAdd(free_cnt + brick, 1u);
    const unsigned long long mm = __ballot(cls == 1), mf = __ballot(cls == 2), m1 = __ballot(first);
    if (lane == 0) { s_cnt[wid][0] = (unsigned)__popcll(mm); s_cnt[wid][1] = (unsigned)__popcll(mf); s_cnt[wid][2] = (unsigned)__popcll(m1); }
    __syncthreads();
    if (threadIdx.x < 3) {
        const unsigned tot = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
        unsigned *cur = threadIdx.x < 2 ? F->counts + threadIdx.x : B.hdr;      // [1]: free-space bricks, counted only
        s_base[threadIdx.x] = tot ? atomicAdd(cur, tot) : 0u;
    }
    __syncthreads();
    unsigned bm = s_base[0], b1 = s_base[2];
    for (int w = 0; w < wid; ++w) { bm += s_cnt[w][0]; b1 += s_cnt[w][2]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (cls == 1) list[bm + __popcll(mm & below)] = (unsigned)brick;
    if (first) B.list[b1 + __popcll(m1 & below)] = (unsigned)brick;
    __syncthreads();                                               // s_cnt / s_base are reused by the next trip
    }
}

// records += count x (32767, 1) for every brick with a pending free-space count; the count returns to zero (exchanged, so a
// count that lands between the read and the reset cannot be lost).  One wave per brick, 16 B per lane.  Runs on the main
// stream before anything reads the TSDF channel (download, merge, extraction, weight check), i.e. once per scan, not per frame.
// A brick WITHOUT records (sparse grid: free space nobody ever saw a surface in) keeps its count: readers add it (tsdf_record).
__global__ __launch_bounds__(256) void fold_free_kernel(Grid g, int2 *__restrict__ grid, unsigned *__restrict__ free_cnt, unsigned nbricks) {
    const int lane = threadIdx.x & 63;
    for (unsigned b = blockIdx.x * 4u + (threadIdx.x >> 6); b < nbricks; b += gridDim.x * 4u) {
        if (__builtin_amdgcn_readfirstlane(free_cnt[b]) == 0u) continue;
        const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane((int)brick_slot(g.tsdf_tab, b));
        if (slot >= SLOT_FULL) continue;
        unsigned c = 0;
        if (lane == 0) c = atomicExch(free_cnt + b, 0u);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c == 0u) continue;
        int4 *__restrict__ recs = reinterpret_cast<int4 *>(grid + ((size_t)slot << 9));
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
__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b_uniform, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}
__device__ __forceinline__ bool tsdf_project(const Cam &cam, const TsdfConst &c, float xc, float yc, float zc, int ntx0, int pix_bytes, unsigned &pix, unsigned &tix) {
    bool ok = zc > 0.0f;
    // 1 / zc.  v_rcp_f32 + one Newton step equals the IEEE quotient for EVERY float with 2^-126 <= zc < 2^126 (exhaustive:
    // tools/ubench_rcp.hip, profiles/r03_ubench_rcp.txt); the division proper (~10 instructions) runs only for lanes outside
    // that range (a voxel centre within 1e-38 m of the camera plane, or absurdly far) -- the value is the oracle's either way.
    float inv = __builtin_amdgcn_rcpf(zc);
    inv = fmaf(fmaf(-zc, inv, 1.0f), inv, inv);
    if (__builtin_expect(!(zc >= 1.17549435e-38f && zc < 8.5e37f), 0)) inv = 1.0f / zc;
    const float uf = fmaf(cam.fx * xc, inv, cam.cx);
    const float vf = fmaf(cam.fy * yc, inv, cam.cy);
    ok = ok && (uf >= -0.5f && uf < c.wlim && vf >= -0.5f && vf < c.hlim);
    int u = (int)floorf(uf + 0.5f), v = (int)floorf(vf + 0.5f);
    u = min(max(u, 0), cam.W - 1);          // always a legal address, so the gather needs no branch
    v = min(max(v, 0), cam.H - 1);
    // (u, v, W < 2^16: 24-bit multiplies are exact and full rate)
    // BYTE offsets of the pixel in the image and of the 8-B record of the 8x8-pixel tile that holds it
    // (v_mad_u32_u24 spelled out: every operand is below 2^24, the instruction is full rate, and the optimiser, left to
    // itself, folds these into 64-bit multiply-adds of the address; as opaque 32-bit values they become the 32-bit offset of a
    // load from a wave-uniform base)
    pix = mad_u24((unsigned)v, (unsigned)(cam.W * pix_bytes), (unsigned)u * (unsigned)pix_bytes);
    tix = mad_u24((unsigned)v >> TILE0_SHIFT, (unsigned)ntx0 << 3, (unsigned)u & ~7u);
    return ok;
}

__device__ __forceinline__ bool tsdf_finish(const Grid &g, float sc, float mind, float maxd, float draw, float zc, int &q) {
    const float d = draw * sc;
    bool ok = d > mind && d < maxd;
    const float sdf = d - zc;
    ok = ok && (sdf >= -g.trunc);
    const float tsdf = fminf(1.0f, sdf * g.inv_trunc);
    q = (int)rintf(tsdf * 32767.0f);
    return ok;
}

// Class of ONE 4x4x4 sub-brick (the lane's): its 8 extreme voxel centres by the per-voxel position sequence (they span the box
// of all 64: a projective map keeps convexity while every depth is positive, so all 64 project inside the 8 projections' pixel
// box), then <= 8 pyramid tiles that cover the box (finest level at which it spans <= 8 tiles).  0 skip, 1 mixed, 2 free.
__device__ __forceinline__ int classify_subbrick(const Cam &cam, const Grid &g, const Pyramid &py, const PoseF &pose,
                                                 const float4 *__restrict__ tiles, int i0, int j0, int k0) {
    float wx[2], wy[2], wz[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        wx[h] = fmaf((float)(i0 + 3 * h) + 0.5f, g.vs, g.ox);
        wy[h] = fmaf((float)(j0 + 3 * h) + 0.5f, g.vs, g.oy);
        wz[h] = fmaf((float)(k0 + 3 * h) + 0.5f, g.vs, g.oz);
    }
    float x[8], y[8], z[8];
    float zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
    for (int hz = 0; hz < 2; ++hz) {
        const float ax0 = fmaf(pose.r[2], wz[hz], pose.t[0]), ay0 = fmaf(pose.r[5], wz[hz], pose.t[1]), az0 = fmaf(pose.r[8], wz[hz], pose.t[2]);
#pragma unroll
        for (int hy = 0; hy < 2; ++hy) {
            const float ax = fmaf(pose.r[1], wy[hy], ax0), ay = fmaf(pose.r[4], wy[hy], ay0), az = fmaf(pose.r[7], wy[hy], az0);
#pragma unroll
            for (int hx = 0; hx < 2; ++hx) {
                const int c = hz * 4 + hy * 2 + hx;
                x[c] = fmaf(pose.r[0], wx[hx], ax);
                y[c] = fmaf(pose.r[3], wx[hx], ay);
                z[c] = fmaf(pose.r[6], wx[hx], az);
                zmin = fminf(zmin, z[c]);
                zmax = fmaxf(zmax, z[c]);
            }
        }
    }
    if (!(zmin > 1e-3f)) return 1;                                  // touches the camera plane: per-voxel rule decides
    float umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float iz = __builtin_amdgcn_rcpf(z[c]);                // 1 ulp; the 1.5 px margin absorbs it
        const float u = cam.fx * x[c] * iz + cam.cx, v = cam.fy * y[c] * iz + cam.cy;
        umin = fminf(umin, u); umax = fmaxf(umax, u);
        vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
    }
    umin -= 1.5f; umax += 1.5f; vmin -= 1.5f; vmax += 1.5f;
    if (umax < 0.0f || vmax < 0.0f || umin > (float)(cam.W - 1) || vmin > (float)(cam.H - 1)) return 0;
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
    const float4 *__restrict__ lv = tiles + off;
    float4 a = make_float4(INFINITY, -INFINITY, 1.0f, 0.0f);
    int du = 0, dv = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {                                   // tiles (du, dv) row by row; beyond the box: the last one again (harmless)
        const float4 b = lv[(tv0 + dv) * ntx + (tu0 + du)];
        a.x = fminf(a.x, b.x);
        a.y = fmaxf(a.y, b.y);
        a.z = fminf(a.z, b.z);
        if (du + 1 < nu) ++du;
        else if (dv + 1 < nv) { du = 0; ++dv; }
    }
    return classify_box(g, a, zmin, zmax, inside);
}

// Prep kernel 4: sub-brick masks of the listed bricks of every frame; one lane per sub-brick, 8 bricks per wave.
__global__ __launch_bounds__(256) void subbrick_classify_kernel(Cam cam, Grid g, Pyramid py, BatchBufs B) {
    const DescPtr F = const_descs(B) + blockIdx.y;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    const unsigned n = min(F->counts[0], nbricks);
    const PoseF pose = desc_pose(F);
    const float4 *__restrict__ tiles = F->tiles;
    const unsigned *__restrict__ list = F->list;
    unsigned short *__restrict__ sub = F->sub;
    const int j = lane >> 3, sb = lane & 7;
    for (unsigned base = (blockIdx.x * 4u + (unsigned)wid) * 8u; base < n; base += gridDim.x * 32u) {
        const unsigned t = base + (unsigned)j;
        int cls = 0;
        unsigned brick = 0;
        if (t < n) {
            brick = min(list[t], nbricks - 1u);
            const int bx = (int)(brick % (unsigned)g.nbx), by = (int)((brick / (unsigned)g.nbx) % (unsigned)g.nby), bz = (int)(brick / (unsigned)(g.nbx * g.nby));
            cls = classify_subbrick(cam, g, py, pose, tiles, bx * 8 + (sb & 1) * 4, by * 8 + ((sb >> 1) & 1) * 4, bz * 8 + (sb >> 2) * 4);
        }
        const unsigned long long mm = __ballot(cls == 1), mf = __ballot(cls == 2);
        if (sb == 0 && t < n) sub[brick] = (unsigned short)(((mm >> (8 * j)) & 0xffull) | (((mf >> (8 * j)) & 0xffull) << 8));
    }
}

// ---- 4b. the batch list, dearest bricks first -------------------------------------------------------------------------------
// A brick costs the update one pipeline step per (frame, MIXED sub-brick): 1 ... 256.  The list comes out of the classification
// in order of first listing, a wave takes ~5 bricks of it, and the launch lasts as long as its unluckiest wave: the longest wave
// ran 1.9 x the mean (stamps, experiments flavour), i.e. half of the chip idled through the second half of the launch.  Ordered
// by cost, dearest first, the tickets hand the cheap bricks out last and the waves end together.  Cost proxy: the number of frames
// that list the brick (1 ... 32, known from the frame mask); a counting sort in two small launches of the prep chain:
// histogram, then scatter (every workgroup reserves its range of each bin with one atomic per bin).
constexpr int ORDER_BLOCKS = 64;
__device__ __forceinline__ unsigned brick_cost_bin(const unsigned *__restrict__ framemask, unsigned brick) {
    const unsigned c = (unsigned)__popc(framemask[brick]);
    return c > 32u ? 32u : c;                                          // bin 0: nobody lists it any more (cannot happen; kept last)
}
__global__ __launch_bounds__(256) void batch_hist_kernel(Grid g, BatchBufs B) {
    __shared__ unsigned h[33];
    if (threadIdx.x < 33) h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    const unsigned n = min(B.hdr[0], nbricks);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        atomicAdd(&h[brick_cost_bin(B.framemask, min(B.list[i], nbricks - 1u))], 1u);
    __syncthreads();
    if (threadIdx.x < 33 && h[threadIdx.x]) atomicAdd(B.hdr + HDR_HIST + threadIdx.x, h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void batch_order_kernel(Grid g, BatchBufs B) {
    __shared__ unsigned h[33], base[33];
    if (threadIdx.x < 33) h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    const unsigned n = min(B.hdr[0], nbricks);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        atomicAdd(&h[brick_cost_bin(B.framemask, min(B.list[i], nbricks - 1u))], 1u);
    __syncthreads();
    if (threadIdx.x < 33) {
        // bins in descending order of cost: 32, 31, ..., 1, then 0; this workgroup's range inside its bins
        unsigned start = 0;
        for (unsigned c = 32u; c > threadIdx.x; --c) start += B.hdr[HDR_HIST + c];
        if (threadIdx.x == 0) { start = 0; for (unsigned c = 1u; c <= 32u; ++c) start += B.hdr[HDR_HIST + c]; }
        base[threadIdx.x] = start + (h[threadIdx.x] ? atomicAdd(B.hdr + HDR_CURSOR + threadIdx.x, h[threadIdx.x]) : 0u);
    }
    __syncthreads();
    if (threadIdx.x < 33) h[threadIdx.x] = 0u;
    __syncthreads();
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const unsigned b = min(B.list[i], nbricks - 1u);
        const unsigned c = brick_cost_bin(B.framemask, b);
        const unsigned pos = base[c] + atomicAdd(&h[c], 1u);
        if (pos < nbricks + 64u) B.order[pos] = b;                     // (always: the bins partition the list)
    }
}

// ---- 5. the update: one launch per batch ---------------------------------------------------------------------------------
// A wave takes one brick of the batch list per trip.  For every frame whose bit is set in the brick's frame mask it projects the
// lane's voxel of each MIXED sub-brick and adds the quantised tsdf to the lane's eight running sums; FREE sub-bricks add
// (32767, 1).  After the last frame the records that changed are read, added to and written: once per batch, whatever the
// number of frames.
//
// What bounds this kernel is the RATE OF SCATTERED DEPTH READS the memory system sustains (56 G 64-B requests per second that
// miss the L2, whatever the occupancy: tools/ubench_gather.hip), so a voxel asks for its depth pixel only when the answer is
// open.  Per voxel and frame, in three pipelined stages:
//   A  project (the oracle's sequence) -> pixel, and load the 8-B record of the 8x8-PIXEL TILE that holds the pixel
//      (lowest depth if all 64 pixels are valid, highest valid depth; 260 KB per frame: these loads hit the L2);
//   B  the tile decides most voxels: highest - zc < -trunc  => sdf < -trunc whatever the pixel says: no update;
//                                    lowest - zc >= 1.001 trunc => tsdf == 1 exactly: (+32767, +1);
//      (rounding is monotone: d >= lowest => fl(d - zc) >= fl(lowest - zc), so both decisions are the per-voxel rule's own);
//      only the voxels in between -- a third of the lanes of a MIXED sub-brick -- gather their depth pixel, the others read
//      pixel 0 (one shared line);
//   C  depth value -> increment.
// Stage A of frame k+2, B of k+1 and C of k are issued back to back: every load has a stage or two of arithmetic to hide behind.
// ALWAYS eight loads per stage, so that the number of loads in flight is known wherever a value is awaited (a count that depended
// on the masks would turn every wait into "all loads", the newest included).
// EXP (experiments flavour of the library only; results incomplete): bit 0 no tile / depth loads, bit 1 no record accesses,
// bit 2 no software pipeline.
#ifdef TL3D_EXPERIMENTS        // the frame-major kernel of round 3: kept in the experiments flavour for A/B runs on one box
template <typename DT> struct RawDepth { typedef float type; };
template <> struct RawDepth<uint16_t> { typedef unsigned short type; };
__device__ __forceinline__ float depth_value(float raw) { return raw; }
__device__ __forceinline__ float depth_value(unsigned short raw) { return mm_to_m(raw); }

template <typename DT>
struct FrameStage {             // one frame's pass over one brick, in flight
    float zc[8];
    unsigned pix[8];            // A -> B: byte offset of the pixel in the frame's image
    float2 tl[8];               // A -> B, as loaded
    typename RawDepth<DT>::type dv[8];   // B -> C, as loaded: nothing touches a value before its stage (the loads stay in flight)
    unsigned m, fr;             // the frame's MIXED and FREE sub-brick masks of this brick (wave-uniform)
    unsigned valid_pix;         // byte offset of a pixel of the frame that is valid (wave-uniform)
    float sc, mind, maxd;       // of the frame (wave-uniform)
    const char *depth;
};

template <bool COUNT, typename DT, int EXP = 0>
#ifndef TL3D_UPD_WAVES
#define TL3D_UPD_WAVES 4
#endif
__global__ __launch_bounds__(256, TL3D_UPD_WAVES) void tsdf_update_kernel(Cam cam, Grid g, BatchBufs B, int xcd_group, int2 *__restrict__ grid,
                                                          unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform, and known to be
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    // the trip count comes from device memory: clamp it to what the list can hold, so that no ordering mistake upstream can
    // ever turn into an unbounded loop or an out-of-range list read
    const unsigned ntask = min(B.hdr[0], nbricks);
    unsigned nread = 0, nwritten = 0;
    // Work distribution.  A brick costs one trip of the frame loop per frame that lists it (1 .. 32), so a static share of the
    // list leaves waves idle for a third of the launch.  Tasks are handed out by TICKET instead: the workgroups form up to 64
    // groups (blockIdx % groups), group g owns list entries g, g + groups, g + 2 groups, ... (every group sees the same mix of
    // cheap and dear bricks) and a ticket counter of its own (64-B apart; a few hundred returning atomics per counter and
    // launch, handed out two tasks ahead of their use, so nobody waits for one).  A wave's first task needs no ticket.
    const unsigned ngrp = min((unsigned)TICKET_GROUPS, gridDim.x);
    const unsigned grp = blockIdx.x % ngrp, bi = blockIdx.x / ngrp;
    const unsigned waves_in_group = ((gridDim.x - grp + ngrp - 1u) / ngrp) * 4u;
    unsigned *__restrict__ ticket = B.hdr + HDR_TICKETS + 16u * grp;
    // A group's sequence runs over chunks of `chunk` consecutive list entries (the bricks of one 4x4x4 cell are listed together
    // and look at the same pixels): group g owns chunks g, g + groups, ...; its workgroups share an XCD (blockIdx % 8), so a
    // chunk's depth lines meet in one L2
    const unsigned chunk = xcd_group > 0 ? (unsigned)xcd_group : 1u;
    auto entry_of = [&](unsigned k) -> unsigned {               // k-th task of this group -> list position (saturating)
        if (k >= 0x2000000u) return 0xffffffffu;
        return ((k / chunk) * ngrp + grp) * chunk + k % chunk;
    };
    const DescPtr frames = const_descs(B);
    const int ntx0 = (cam.W + TILE0 - 1) >> TILE0_SHIFT;
    const float trunc_free = g.trunc * 1.001f;

    // the next brick of this wave, its frame mask and (lane f) its sub-brick masks in frame f: fetched one trip ahead
    auto fetch = [&](unsigned t, unsigned &brick, unsigned &fm, unsigned &subv, unsigned &slot) {
        brick = min((unsigned)__builtin_amdgcn_readfirstlane((int)B.list[t]), nbricks - 1u);
        fm = (unsigned)__builtin_amdgcn_readfirstlane((int)B.framemask[brick]);
        slot = (unsigned)__builtin_amdgcn_readfirstlane((int)brick_slot(g.tsdf_tab, brick));     // where the brick's records live (set by the classification)
        subv = 0u;
        if (lane < B.n_frames && ((fm >> lane) & 1u)) subv = frames[lane].sub[brick];
    };

    auto take_ticket = [&]() -> unsigned {                      // position in the group's sequence of the task after next
        unsigned k = 0;
        if (lane == 0) k = atomicAdd(ticket, 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)k) + waves_in_group;
    };
    unsigned t = entry_of(bi * 4u + (unsigned)wid);                // list entry of this trip
    unsigned brick_n = 0, fm_n = 0, subv_n = 0, slot_n = 0;
    if (t < ntask) fetch(t, brick_n, fm_n, subv_n, slot_n);
    unsigned k_next = t < ntask ? take_ticket() : 0u;             // the entry after this one: its ticket is drawn a trip ahead of its fetch
    while (t < ntask) {
        const unsigned brick = brick_n, subv = subv_n, slot = slot_n;
        unsigned fm = fm_n;
        if (lane == 0) B.framemask[brick] = 0u;                       // re-armed for the next batch that uses this buffer
        // saturating: a ticket far beyond the list must not wrap into it
        const unsigned t_next = entry_of(k_next);
        if (t_next < ntask) {
            fetch(t_next, brick_n, fm_n, subv_n, slot_n);
            k_next = take_ticket();
        }
        t = t_next;
        if (B.n_frames < 32) fm &= (1u << B.n_frames) - 1u;
        const int bx = (int)(brick % (unsigned)g.nbx), by = (int)((brick / (unsigned)g.nbx) % (unsigned)g.nby), bz = (int)(brick / (unsigned)(g.nbx * g.nby));
        // the lane's voxel in sub-brick s: (x, y, z) = (4 (s & 1) + (lane & 3), 4 (s >> 1 & 1) + (lane >> 2 & 3), 4 (s >> 2) + (lane >> 4)),
        // record s * 64 + lane; world coordinates: two values per axis
        float wx[2], wy[2], wz[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            wx[h] = fmaf((float)(bx * 8 + 4 * h + (lane & 3)) + 0.5f, g.vs, g.ox);
            wy[h] = fmaf((float)(by * 8 + 4 * h + ((lane >> 2) & 3)) + 0.5f, g.vs, g.oy);
            wz[h] = fmaf((float)(bz * 8 + 4 * h + (lane >> 4)) + 0.5f, g.vs, g.oz);
        }
        // running sums of the lane's eight voxels, packed: bits 0-20 sum of (q + 32768) (<= 32 x 65535 < 2^21), bits 21-26 weight
        unsigned acc[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] = 0u;
        unsigned touched = 0u;                                      // sub-bricks any frame adds to (wave-uniform)

        // stage A: project the lane's voxel of every MIXED sub-brick of frame f, load the tiles of the pixels
        auto stage_a = [&](int f, FrameStage<DT> &S) {
            const unsigned sm = (unsigned)__builtin_amdgcn_readlane((int)subv, f);
            S.m = sm & 0xffu; S.fr = sm >> 8;
            touched |= S.m | S.fr;
            const PoseF pose = desc_pose(frames + f);
            const TsdfConst c = desc_const(frames + f);
            S.sc = c.sc; S.mind = c.mind; S.maxd = c.maxd;
            S.depth = static_cast<const char *>(frames[f].depth);
            const char *__restrict__ vt = reinterpret_cast<const char *>(frames[f].vtile);
            S.valid_pix = ((const unsigned __attribute__((address_space(4))) *)(frames[f].counts))[VALID_PIXEL] * (unsigned)sizeof(DT);
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
                unsigned pix = 0u, tix = 0u;
                S.zc[s] = INFINITY;
                if ((S.m >> s) & 1u) {
                    const float xc = fmaf(pose.r[0], wx[s & 1], ax[s >> 2][(s >> 1) & 1]);
                    const float yc = fmaf(pose.r[3], wx[s & 1], ay[s >> 2][(s >> 1) & 1]);
                    const float zc = fmaf(pose.r[6], wx[s & 1], az[s >> 2][(s >> 1) & 1]);
                    // a voxel that does not project into the image: zc = +inf, "behind everything" for stages B and C
                    S.zc[s] = tsdf_project(cam, c, xc, yc, zc, ntx0, (int)sizeof(DT), pix, tix) ? zc : INFINITY;
                }
                S.pix[s] = pix;                                        // (32-bit byte offsets from a wave-uniform base: one address register per load)
                if (EXP & 1) S.tl[s] = make_float2(-INFINITY, INFINITY);
                else S.tl[s] = *reinterpret_cast<const float2 *>(vt + tix);
            }
        };
        // stage B: what the tile leaves open gathers its depth pixel.  The decided lanes read the frame's valid pixel (one shared
        // line) and carry their decision in zc: +inf = behind everything (sdf = -inf: no update), -inf = in front of everything
        // (sdf = +inf: tsdf = 1 exactly, and the depth read IS valid) -- stage C needs no flags.
        auto stage_b = [&](FrameStage<DT> &S) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                unsigned pix = S.valid_pix;
                {   // (no branch around a sub-brick that is not MIXED -- its zc is +inf, "behind everything", and stays so: 7 instructions
                    // for all eight sub-bricks cost less than 8 wave-uniform branches, 66.2 k -> 68.2 k frames/s; the same in stage C,
                    // 13 instructions each, costs more: 65.2 k)
                    const float zc = S.zc[s];
                    const bool fre = S.tl[s].x - zc >= trunc_free;
                    const bool skp = !(S.tl[s].y - zc >= -g.trunc);
                    S.zc[s] = fre ? -INFINITY : (skp ? INFINITY : zc);
                    pix = (fre || skp) ? pix : S.pix[s];
                }
                if (EXP & 1) S.dv[s] = (typename RawDepth<DT>::type)(1 + (pix & 1023));
                else S.dv[s] = *reinterpret_cast<const typename RawDepth<DT>::type *>(S.depth + pix);
            }
        };
        // stage C: depth values -> increments
        auto stage_c = [&](const FrameStage<DT> &S) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if ((S.m >> s) & 1u) {
                    int q;
                    if (tsdf_finish(g, S.sc, S.mind, S.maxd, depth_value(S.dv[s]), S.zc[s], q)) acc[s] += (unsigned)(q + 32768) + (1u << 21);
                }
                acc[s] += ((S.fr >> s) & 1u) ? 65535u + (1u << 21) : 0u;      // (a FREE sub-brick: one add of a wave-uniform value, no second branch)
            }
        };
        if (EXP & 4) {                                          // experiments: one frame at a time, no software pipeline
            FrameStage<DT> S0;
            for (unsigned rest = fm; rest; rest &= rest - 1u) {
                stage_a(__builtin_ctz(rest), S0);
                stage_b(S0);
                stage_c(S0);
            }
        } else if (fm) {
            // Steady state of a trip: A(k+1) issues its tile loads, C(k) awaits the depth values of frame k (issued a trip ago; the
            // new tile loads are younger and stay in flight), B(k+1) awaits the tiles and issues the gathers.
            // The loop body is written out twice so that every stage names its registers.
            FrameStage<DT> S0, S1;
            unsigned rest = fm;
#define TL3D_POP(f_) const int f_ = __builtin_ctz(rest); rest &= rest - 1u
            TL3D_POP(f0);
            stage_a(f0, S0); stage_b(S0);
            for (;;) {
                if (!rest) { stage_c(S0); break; }
                { TL3D_POP(f); stage_a(f, S1); stage_c(S0); stage_b(S1); }
                if (!rest) { stage_c(S1); break; }
                { TL3D_POP(f); stage_a(f, S0); stage_c(S1); stage_b(S0); }
            }
#undef TL3D_POP
        }
        if (touched == 0u) continue;
        // load the records that change (8 B per lane, a sub-brick = one 512-B run), add, store
        if (slot >= SLOT_FULL) continue;                              // (sparse grid whose pool ran out: counted when the slot was refused)
        int2 *__restrict__ recs = grid + ((size_t)slot << 9);
        int2 rec[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (((touched >> s) & 1u) && (acc[s] >> 21)) rec[s] = (EXP & 2) ? make_int2((int)acc[s] ^ lane, s) : recs[s * 64 + lane];
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (((touched >> s) & 1u) && (acc[s] >> 21)) {
                const int w = (int)(acc[s] >> 21);
                rec[s].x += (int)(acc[s] & 0x1fffffu) - 32768 * w;
                rec[s].y += w;
                if (EXP & 2) { if (rec[s].x == cam.W * 7919 + lane) recs[0] = rec[s]; }      // keep the values alive, (practically) never store
                else recs[s * 64 + lane] = rec[s];
                if (COUNT) { nread += 1; nwritten += 1; }
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
            for (int f = 0; f < B.n_frames; ++f) {
                const unsigned l = min(frames[f].counts[0], nbricks), fc = min(frames[f].counts[1], nbricks - l);
                vis += l + fc; fre += fc;
            }
            atomicAdd(counters + 4, vis);
            atomicAdd(counters + 5, fre);
            atomicAdd(counters + 6, fre);
            atomicAdd(counters + 7, (unsigned long long)ntask);
        }
    }
}

#endif  // TL3D_EXPERIMENTS

// ---- 5b. the update, PAIR form (what the library launches) -----------------------------------------------------------------
// Same task as above -- one wave per brick of the batch list, every listed (frame, MIXED sub-brick) of the brick projected, the
// records read and written once per batch -- in a different shape.  The frame-major kernel above carries eight sub-bricks per lane
// through every stage, chooses among them with wave-uniform branches (~40 taken or not per frame step: each costs this kernel
// about five instruction slots, the branches together as much as half of its vector work), keeps the per-sub-brick conditions as
// sixteen 64-bit lane masks in scalar registers (spilled to vector lanes) and needs 128 vector registers, i.e. three or four
// waves per SIMD.  Here the unit of the pipeline is ONE PAIR (frame, sub-brick): 64 lanes = the 64 voxels of that sub-brick.
//   * the brick's pairs are walked in frame-major order (frame f's scalars -- pose, scale, image and tile-record base -- are
//     loaded when its first pair comes up), sub-brick s of the pair is a scalar: the lane's world coordinates are chosen by
//     three selects, everything else is straight-line code;
//   * the running sums live in LDS: acc[s][lane], wave-private, one ds_add per pair (a register array indexed by a scalar would
//     need eight predicated adds);  FREE sub-bricks are counted per sub-brick over the frames with ballots and enter the sums
//     once per brick;
//   * three stages as before -- A project + tile record, B tile test + depth gather, C quantise + add -- software-pipelined
//     over the pairs: step k runs A(k), C(k-2), B(k-1); a tile record has a full step, a depth value two thirds of one, before
//     anything waits for it (and seven other waves of the SIMD have work meanwhile).  P pairs take P + 2 steps; the empty slots at either end run with zc = +inf ("behind
//     everything": no update) and legal addresses, so the steady loop has no branch but its back edge and the frame switch
//     (stages skipped by wave-uniform branches at either end instead: the compiler's wait counts at the joins turn conservative and
//     the launch takes a quarter longer: 61 k against 76 k frames/s);
//   * every load of a stage is a scalar base + a 32-bit per-lane offset.
// About 45 vector registers: eight waves per SIMD hide what the three of the form above could not.
// The arithmetic of a voxel is tsdf_project / tsdf_finish, as above: the grid is the oracle's bit for bit.
__device__ __forceinline__ int clamp_i32(int x, int hi_uniform) {        // min(max(x, 0), hi): one instruction (hi >= 0)
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi_uniform));
    return r;
}

// loads from a wave-uniform base + a 32-bit per-lane byte offset (scalar base, vector offset: one address register per load)
template <typename DT> struct PixLoad;
template <> struct PixLoad<float> {
    typedef float raw;
    static __device__ __forceinline__ raw load(const char *base, unsigned off) { return *reinterpret_cast<const float *>(base + off); }
    static __device__ __forceinline__ float value(raw v) { return v; }
};
template <> struct PixLoad<uint16_t> {
    typedef unsigned short raw;
    static __device__ __forceinline__ raw load(const char *base, unsigned off) { return *reinterpret_cast<const unsigned short *>(base + off); }
    static __device__ __forceinline__ float value(raw v) { return mm_to_m(v); }
};
__device__ __forceinline__ int cvt_flr_i32(float x) {                    // (int)floorf(x), one instruction; saturates, NaN -> 0
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ unsigned keep_scalar(unsigned x) {             // x, wave-uniform, is computed HERE and lives in one scalar register
    asm volatile("" : "+s"(x));
    return x;
}

#ifdef TL3D_EXPERIMENTS
__device__ unsigned long long g_upd_span[16384][2];        // experiments: start / end (100 MHz real-time clock) of every wave of the last stamped launch
#endif
constexpr int UPD_WAVES = 8;             // waves per SIMD the pair kernel is built for (64 vector registers)

// EXP (experiments flavour only, results incomplete): bit 0 tile records read at lane * 8 (coalesced), bit 1 every lane reads the frame's
// valid pixel (no scattered depth reads), bit 2 no LDS add, bit 3 no record accesses
template <bool COUNT, typename DT, int EXP = 0>
__global__ __launch_bounds__(256, UPD_WAVES) void tsdf_update_pairs_kernel(Cam cam, Grid g, BatchBufs B, float mind, float maxd,
                                                                           int2 *__restrict__ grid, unsigned long long *__restrict__ counters) {
    __shared__ unsigned s_acc[4 * 512];                           // per wave: [8 sub-bricks][64 lanes] packed running sums
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned *const acc = s_acc + wid * 512 + lane;               // + 64 s
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    const unsigned ntask = min(B.hdr[0], nbricks);                // (clamped: see the kernel above)
    unsigned nread = 0, nwritten = 0;
    // tasks by ticket, exactly as above: groups of workgroups, each with its own counter, entries interleaved between the groups
    const unsigned ngrp = min((unsigned)TICKET_GROUPS, gridDim.x);
    const unsigned grp = blockIdx.x % ngrp, bi = blockIdx.x / ngrp;
    const unsigned waves_in_group = ((gridDim.x - grp + ngrp - 1u) / ngrp) * 4u;
    unsigned *__restrict__ ticket = B.hdr + HDR_TICKETS + 16u * grp;
    auto entry_of = [&](unsigned k) -> unsigned { return k >= 0x2000000u ? 0xffffffffu : k * ngrp + grp; };
    const DescPtr frames = const_descs(B);
    const int ntx0 = (cam.W + TILE0 - 1) >> TILE0_SHIFT;
    const float trunc_free = g.trunc * 1.001f;
    const unsigned row_bytes = (unsigned)cam.W * (unsigned)sizeof(DT);

    auto fetch = [&](unsigned t, unsigned &brick, unsigned &fm, unsigned &subv, unsigned &slot) {
        brick = min((unsigned)__builtin_amdgcn_readfirstlane((int)B.order[t]), nbricks - 1u);
        fm = (unsigned)__builtin_amdgcn_readfirstlane((int)B.framemask[brick]);
        slot = (unsigned)__builtin_amdgcn_readfirstlane((int)brick_slot(g.tsdf_tab, brick));
        subv = 0u;
        if (lane < B.n_frames && ((fm >> lane) & 1u)) subv = frames[lane].sub[brick];
    };
    auto take_ticket = [&]() -> unsigned {
        unsigned k = 0;
        if (lane == 0) k = atomicAdd(ticket, 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)k) + waves_in_group;
    };
    unsigned t = entry_of(bi * 4u + (unsigned)wid);
    unsigned brick_n = 0, fm_n = 0, subv_n = 0, slot_n = 0;
    if (t < ntask) fetch(t, brick_n, fm_n, subv_n, slot_n);
    unsigned k_next = t < ntask ? take_ticket() : 0u;
    unsigned long long pf_t0 = 0, pf_setup = 0, pf_loop = 0, pf_rmw = 0, pf_bricks = 0, pf_steps = 0, pf_a = 0, pf_b = 0;
    unsigned long long pf_rt0 = 0;
    if (EXP & 16) pf_t0 = pf_a = __builtin_readcyclecounter();
    if (EXP & 32) pf_rt0 = __builtin_amdgcn_s_memrealtime();      // light mode: only the start and the end of every wave (the heavy stamps of bit 4 slow the kernel threefold)
    while (t < ntask) {
        const unsigned brick = brick_n, slot = slot_n;
        unsigned subv = subv_n;
        if (lane == 0) B.framemask[brick] = 0u;                   // re-armed for the next batch that uses this buffer
        const unsigned t_next = entry_of(k_next);
        if (t_next < ntask) {
            fetch(t_next, brick_n, fm_n, subv_n, slot_n);
            k_next = take_ticket();
        }
        t = t_next;
        if (lane >= B.n_frames) subv = 0u;
        const int bx = (int)(brick % (unsigned)g.nbx), by = (int)((brick / (unsigned)g.nbx) % (unsigned)g.nby), bz = (int)(brick / (unsigned)(g.nbx * g.nby));
        // the lane's voxel in sub-brick s is (4 (s & 1) + (lane & 3), 4 (s >> 1 & 1) + (lane >> 2 & 3), 4 (s >> 2) + (lane >> 4)); its
        // centre along an axis: fma(i + 0.5, voxel, origin) -- two values per axis (named scalars: an array indexed by a bit of s
        // would be moved to LDS by the compiler)
        const float wx0 = fmaf((float)(bx * 8 + (lane & 3)) + 0.5f, g.vs, g.ox), wx1 = fmaf((float)(bx * 8 + 4 + (lane & 3)) + 0.5f, g.vs, g.ox);
        const float wy0 = fmaf((float)(by * 8 + ((lane >> 2) & 3)) + 0.5f, g.vs, g.oy), wy1 = fmaf((float)(by * 8 + 4 + ((lane >> 2) & 3)) + 0.5f, g.vs, g.oy);
        const float wz0 = fmaf((float)(bz * 8 + (lane >> 4)) + 0.5f, g.vs, g.oz), wz1 = fmaf((float)(bz * 8 + 4 + (lane >> 4)) + 0.5f, g.vs, g.oz);
        // the brick's pairs: lane f holds frame f's masks.  Pairs = set MIXED bits; FREE sub-bricks are counted per sub-brick over
        // the frames and start the running sums; touched = sub-bricks anything adds to
        unsigned npairs = 0, touched = 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const unsigned long long bm = __ballot((subv >> s) & 1u), bf = __ballot((subv >> (8 + s)) & 1u);
            npairs += (unsigned)__popcll(bm);
            touched |= (bm | bf) ? (1u << s) : 0u;
            acc[64 * s] = (unsigned)__popcll(bf) * (65535u + (1u << 21));
        }
        npairs = keep_scalar(npairs);                                          // (computed now: the sixteen ballots die here)
        touched = keep_scalar(touched);
        unsigned rest = (unsigned)__ballot((subv & 0xffu) != 0u);              // frames with at least one MIXED sub-brick (lanes 0..31)
        if (EXP & 16) { pf_b = __builtin_readcyclecounter(); pf_setup += pf_b - pf_a; pf_bricks += 1; pf_steps += npairs; }
        if (npairs) {
            // ---- scalars of the frame stage A works in, of the pair stage B works on, of the three pairs on their way to stage C
            float r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0, r8 = 0, t0 = 0, t1 = 0, t2 = 0, a_sc = 0;
            unsigned a_vp = 0, m = 0;
            const char *a_vt = nullptr, *a_img = nullptr;
            auto switch_frame = [&]() {
                const int f = __builtin_ctz(rest);
                rest &= rest - 1u;
                m = (unsigned)__builtin_amdgcn_readlane((int)subv, f) & 0xffu;
                const DescPtr F = frames + f;
                r0 = F->pose.r[0]; r1 = F->pose.r[1]; r2 = F->pose.r[2]; r3 = F->pose.r[3]; r4 = F->pose.r[4]; r5 = F->pose.r[5];
                r6 = F->pose.r[6]; r7 = F->pose.r[7]; r8 = F->pose.r[8]; t0 = F->pose.t[0]; t1 = F->pose.t[1]; t2 = F->pose.t[2];
                a_sc = F->c.sc;
                a_vp = F->valid_pix * (unsigned)sizeof(DT);
                a_vt = reinterpret_cast<const char *>(F->vtile);
                a_img = static_cast<const char *>(F->depth);
            };
            switch_frame();
            const char *b_img = a_img;
            unsigned b_vp = a_vp;
            float c_sc1 = a_sc, c_sc2 = a_sc;
            unsigned c_s1 = 0, c_s2 = 0;
            bool c_open2 = false;                                              // pair k - 2 left some voxel to its depth pixel (wave-uniform)
            // ---- pipeline registers, by the parity of the pair: A -> B (zc or +inf, pixel offset, tile record), B -> C (zc / +-inf, depth)
            float zcA[2] = {INFINITY, INFINITY}, zcB[2] = {INFINITY, INFINITY};
            unsigned pixA[2] = {0u, 0u};
            float2 tlA[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
            typename PixLoad<DT>::raw dvB[2] = {0, 0};
            const unsigned nsteps = npairs + 2u;
            auto step = [&](unsigned k, const int par) {
                // ---- A(k): the lane's voxel of sub-brick s in the current frame -> pixel, tile record.  An empty slot (k >= npairs)
                // runs with the camera plane at infinity: nothing lies in front of it
                const bool have = k < npairs;
                const unsigned s = have ? (unsigned)__builtin_ctz(m) : 0u;
                const float zmin = have ? 0.0f : INFINITY;
                m &= m - 1u;                                                   // (0 stays 0)
                {
                    const float wxs = (s & 1u) ? wx1 : wx0, wys = (s & 2u) ? wy1 : wy0, wzs = (s & 4u) ? wz1 : wz0;
                    const float xc = fmaf(r0, wxs, fmaf(r1, wys, fmaf(r2, wzs, t0)));
                    const float yc = fmaf(r3, wxs, fmaf(r4, wys, fmaf(r5, wzs, t1)));
                    const float zc = fmaf(r6, wxs, fmaf(r7, wys, fmaf(r8, wzs, t2)));
                    float inv = __builtin_amdgcn_rcpf(zc);                     // + one Newton step == the IEEE quotient (tsdf_project)
                    inv = fmaf(fmaf(-zc, inv, 1.0f), inv, inv);
                    if (__builtin_expect(!(zc >= 1.17549435e-38f && zc < 8.5e37f), 0)) inv = 1.0f / zc;
                    const float uf = fmaf(cam.fx * xc, inv, cam.cx);
                    const float vf = fmaf(cam.fy * yc, inv, cam.cy);
                    // nearest pixel floor(uf + 0.5) in ONE conversion (v_cvt_flr_i32_f32 == v_floor_f32 + v_cvt_i32_f32 on every bit
                    // pattern), and the window  -0.5 <= uf < W - 0.5  as ONE unsigned compare of it  (the same decision for every float
                    // and every width: tools/ubench_flr.hip, exhaustive)
                    const int ui = cvt_flr_i32(uf + 0.5f), vi = cvt_flr_i32(vf + 0.5f);
                    const bool ok = (zc > zmin) & ((unsigned)ui <= (unsigned)(cam.W - 1)) & ((unsigned)vi <= (unsigned)(cam.H - 1));
                    const int u = clamp_i32(ui, cam.W - 1), v = clamp_i32(vi, cam.H - 1);
                    pixA[par] = mad_u24((unsigned)v, row_bytes, (unsigned)u * (unsigned)sizeof(DT));
                    const unsigned tix = mad_u24((unsigned)v >> TILE0_SHIFT, (unsigned)ntx0 << 3, (unsigned)u & ~7u);
                    zcA[par] = ok ? zc : INFINITY;
                    tlA[par] = *reinterpret_cast<const float2 *>(a_vt + ((EXP & 1) ? (unsigned)lane * 8u : tix));
                }
                // the scalars pair k hands down the pipeline, then -- its frame used up -- the next frame's (the loads have stages C
                // and B to arrive in)
                const char *n_img = a_img;
                const unsigned n_vp = a_vp, n_s = s;
                const float n_sc = a_sc;
                if (m == 0u && rest != 0u) switch_frame();
                // ---- C(k - 2): depth value -> increment of the running sum.  A pair whose every voxel the tiles decided (two in five)
                // has only "in front of everything" lanes (zc = -inf: tsdf = 1 exactly, and the pixel they read is valid) and "behind
                // everything" lanes: no arithmetic
                {
                    const float zc = zcB[par];
                    unsigned inc;
                    if (c_open2) {
                        const float d = PixLoad<DT>::value(dvB[par]) * c_sc2;
                        const float sdf = d - zc;
                        const bool ok = (d > mind && d < maxd) && (sdf >= -g.trunc);
                        const float tsdf = fminf(1.0f, sdf * g.inv_trunc);
                        const int q = (int)rintf(tsdf * 32767.0f);
                        inc = ok ? (unsigned)(q + 32768) + (1u << 21) : 0u;
                    } else {
                        inc = (zc < 0.0f) ? 65535u + (1u << 21) : 0u;
                    }
                    if (!(EXP & 4) || inc == 0x12345u) (void)__hip_atomic_fetch_add(acc + 64u * c_s2, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                // ---- B(k - 1): what the tile leaves open gathers its pixel; the decided lanes read the frame's valid pixel
                bool open1;
                {
                    const float zc = zcA[par ^ 1];
                    const bool fre = tlA[par ^ 1].x - zc >= trunc_free;
                    const bool skp = !(tlA[par ^ 1].y - zc >= -g.trunc);
                    zcB[par ^ 1] = fre ? -INFINITY : (skp ? INFINITY : zc);
                    const unsigned pix = ((EXP & 2) || fre || skp) ? b_vp : pixA[par ^ 1];
                    dvB[par ^ 1] = PixLoad<DT>::load(b_img, pix);
                    open1 = __ballot(!(fre || skp)) != 0ull;
                }
                b_img = n_img; b_vp = n_vp;
                c_sc2 = c_sc1; c_s2 = c_s1; c_open2 = open1; c_sc1 = n_sc; c_s1 = n_s;
            };
            for (unsigned k = 0; k < nsteps; k += 2u) {                         // (an odd count: one more empty step)
                step(k, 0);
                step(k + 1u, 1);
            }
        }
        if (EXP & 16) { pf_a = __builtin_readcyclecounter(); pf_loop += pf_a - pf_b; }
        if (touched == 0u || slot >= SLOT_FULL) continue;                       // (a full pool: counted when the slot was refused)
        if ((EXP & 8) && acc[0] != 0x7654321u) continue;
        // the records that change: read, add, write -- once per batch
        int2 *__restrict__ recs = grid + ((size_t)slot << 9);
        unsigned a[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) a[s] = ((touched >> s) & 1u) ? acc[64 * s] : 0u;
        int2 rec[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (a[s] >> 21) rec[s] = recs[s * 64 + lane];
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (a[s] >> 21) {
                const int w = (int)(a[s] >> 21);
                rec[s].x += (int)(a[s] & 0x1fffffu) - 32768 * w;
                rec[s].y += w;
                recs[s * 64 + lane] = rec[s];
                if (COUNT) { nread += 1; nwritten += 1; }
            }
        if (EXP & 16) { const unsigned long long now = __builtin_readcyclecounter(); pf_rmw += now - pf_a; pf_a = now; }
    }
#ifdef TL3D_EXPERIMENTS
    if ((EXP & 32) && lane == 0) {
        const unsigned wv = blockIdx.x * 4u + (unsigned)wid;
        if (wv < 16384u) { g_upd_span[wv][0] = pf_rt0; g_upd_span[wv][1] = __builtin_amdgcn_s_memrealtime(); }
        if (wv == 0) atomicAdd(counters + 14, 1ull);
    }
#endif
    if ((EXP & 16) && lane == 0) {                             // experiments: where a wave's time goes (s_memtime ticks)
        atomicAdd(counters + 8, pf_setup); atomicAdd(counters + 9, pf_loop); atomicAdd(counters + 10, pf_rmw);
        atomicAdd(counters + 11, __builtin_readcyclecounter() - pf_t0); atomicAdd(counters + 12, pf_bricks); atomicAdd(counters + 13, pf_steps);
        atomicAdd(counters + 14, 1ull);
        atomicMax(counters + 15, __builtin_readcyclecounter() - pf_t0);
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
            for (int f = 0; f < B.n_frames; ++f) {
                const unsigned l = min(frames[f].counts[0], nbricks), fc = min(frames[f].counts[1], nbricks - l);
                vis += l + fc; fre += fc;
            }
            atomicAdd(counters + 4, vis);
            atomicAdd(counters + 5, fre);
            atomicAdd(counters + 6, fre);
            atomicAdd(counters + 7, (unsigned long long)ntask);
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

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// Scratch of ONE batch in flight (the context keeps two and alternates):
//   [descriptors (max_frames)] [header 256 B] [batch list] [frame masks] then per frame [counts 256 B] [tile pyramid] [list] [sub-brick masks] [8-B level-0 tiles]
struct BatchLayout {
    size_t off_desc, off_hdr, off_list, off_order, off_mask, off_frames, per_frame, f_counts, f_tiles, f_list, f_sub, f_vt, f_cells, total;
};
static BatchLayout batch_layout(const Cam &cam, const Grid &g, int max_frames) {
    const Pyramid p = make_pyramid(cam);
    const size_t nbricks = (size_t)g.nbx * g.nby * g.nbz;
    BatchLayout L;
    L.off_desc = 0;
    L.off_hdr = up256((size_t)max_frames * sizeof(FrameDesc));
    L.off_list = L.off_hdr + up256(HDR_WORDS * sizeof(unsigned));
    L.off_order = L.off_list + up256((nbricks + 64) * sizeof(unsigned));
    L.off_mask = L.off_order + up256((nbricks + 64) * sizeof(unsigned));
    L.off_frames = L.off_mask + up256(nbricks * sizeof(unsigned));
    L.f_counts = 0;
    L.f_tiles = 256;
    L.f_list = L.f_tiles + up256(pyramid_tiles(p) * sizeof(float4));
    L.f_sub = L.f_list + up256((nbricks + 64) * sizeof(unsigned));
    L.f_vt = L.f_sub + up256(nbricks * sizeof(unsigned short));
    L.f_cells = L.f_vt + up256((size_t)p.ntx[0] * p.nty[0] * sizeof(float2));
    const size_t ncells = (size_t)((g.nbx + 3) / 4) * ((g.nby + 3) / 4) * ((g.nbz + 3) / 4);
    L.per_frame = L.f_cells + up256((ncells + 4) * sizeof(unsigned));
    L.total = L.off_frames + L.per_frame * (size_t)max_frames;
    return L;
}

size_t tsdf_batch_scratch_bytes(const Cam &cam, const Grid &g, int max_frames) { return batch_layout(cam, g, max_frames).total; }

// bytes of a batch scratch that must be zero before its first use (the frame masks; the update kernel re-arms them)
void tsdf_batch_scratch_zero_range(const Cam &cam, const Grid &g, int max_frames, size_t *off, size_t *bytes) {
    const BatchLayout L = batch_layout(cam, g, max_frames);
    *off = L.off_mask;
    *bytes = L.off_frames - L.off_mask;
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

static BatchBufs batch_bufs(const BatchLayout &L, void *scratch, int n) {
    char *base = static_cast<char *>(scratch);
    BatchBufs B;
    B.frames = reinterpret_cast<const FrameDesc *>(base + L.off_desc);
    B.hdr = reinterpret_cast<unsigned *>(base + L.off_hdr);
    B.list = reinterpret_cast<unsigned *>(base + L.off_list);
    B.order = reinterpret_cast<unsigned *>(base + L.off_order);
    B.framemask = reinterpret_cast<unsigned *>(base + L.off_mask);
    B.n_frames = n;
    return B;
}

// The whole prep of a batch of n frames (one depth kind) on stream s: descriptor upload, depth tiles, pyramid, brick
// classification, sub-brick masks.  `scratch` is a batch scratch of at least n frames (tsdf_batch_scratch_bytes).
int launch_tsdf_prepare(hipStream_t s, const Cam &cam, const Grid &g, int n, int max_frames, const PoseF *p, const Frustum &fr,
                        const void *const *depth, bool depth_u16, const float *scale, float mind, float maxd, void *scratch, unsigned *free_cnt,
                        bool classify_only) {
    if (n < 1 || n > max_frames || n > TL3D_TSDF_MAXBATCH) return set_err(TL3D_E_INVALID, "bad batch size %d", n);
    const BatchLayout L = batch_layout(cam, g, max_frames);
    const Pyramid py = make_pyramid(cam);
    char *base = static_cast<char *>(scratch);
    FrameDesc d[TL3D_TSDF_MAXBATCH];
    memset(d, 0, sizeof(d));
    for (int i = 0; i < n; ++i) {
        char *fb = base + L.off_frames + L.per_frame * (size_t)i;
        d[i].pose = p[i];
        d[i].c = make_const(cam, scale[i], mind, maxd);
        d[i].depth = depth[i];
        d[i].counts = reinterpret_cast<unsigned *>(fb + L.f_counts);
        d[i].tiles = reinterpret_cast<float4 *>(fb + L.f_tiles);
        d[i].list = reinterpret_cast<unsigned *>(fb + L.f_list);
        d[i].sub = reinterpret_cast<unsigned short *>(fb + L.f_sub);
        d[i].vtile = reinterpret_cast<float2 *>(fb + L.f_vt);
        d[i].cells = reinterpret_cast<unsigned *>(fb + L.f_cells);
    }
    for (int i0 = 0; i0 < n; i0 += 16) {
        DescChunk ch;
        const int m = n - i0 < 16 ? n - i0 : 16;
        memcpy(ch.d, d + i0, (size_t)m * sizeof(FrameDesc));
        if (m < 16) memset(ch.d + m, 0, (size_t)(16 - m) * sizeof(FrameDesc));
        hipLaunchKernelGGL(desc_upload_kernel, dim3(1), dim3(256), 0, s, ch, reinterpret_cast<FrameDesc *>(base + L.off_desc) + i0, m);
        TL3D_HIP(hipGetLastError());
    }
    const BatchBufs B = batch_bufs(L, scratch, n);
    const int nrx = (cam.W + REGION - 1) / REGION, nry = (cam.H + REGION - 1) / REGION;
    const int nreg = nrx * nry;
    if (depth_u16)
        hipLaunchKernelGGL(depth_tiles_kernel<uint16_t>, dim3((nreg + 3) / 4 < 512 ? (nreg + 3) / 4 : 512, n), dim3(256), 0, s, cam, B, py, nrx, nry);
    else
        hipLaunchKernelGGL(depth_tiles_kernel<float>, dim3((nreg + 3) / 4 < 512 ? (nreg + 3) / 4 : 512, n), dim3(256), 0, s, cam, B, py, nrx, nry);
    TL3D_HIP(hipGetLastError());
    hipLaunchKernelGGL(tile_pyramid_kernel, dim3(n), dim3(256), 0, s, cam, g, fr, py, B);
    TL3D_HIP(hipGetLastError());
    const int ncells = ((g.nbx + 3) / 4) * ((g.nby + 3) / 4) * ((g.nbz + 3) / 4);
    const int nquad = (ncells + 3) / 4;                 // the cells in view are known only on the device: a fixed grid strides over each frame's list
    hipLaunchKernelGGL(brick_cull_kernel, dim3(nquad < 256 ? nquad : 256, n), dim3(256), 0, s, cam, g, B, fr, py, free_cnt);
    TL3D_HIP(hipGetLastError());
    if (classify_only) return TL3D_OK;                    // (tl3d_count_bricks: which bricks WOULD get records is all that is asked)
    // The batch list ordered by cost, dearest bricks first (a one-frame batch: every brick costs the same; the update walks the list).
    // A brick's cost is the number of frames that list it -- the frame masks, which the cull has just finished -- so the two small
    // kernels go BEFORE the sub-brick classification, not behind it: beside an update kernel the classification stretches over the
    // whole update period and ends when the update drains, and whatever follows it in the chain sits in front of the next update
    // (12-37 us per 400 us step in the timeline of round 4).  Measured: 78.6-79.0 k frames/s either way (three
    // interleaved runs each) -- the step is bound by the sum of the work, not by that gap; the shorter tail is kept.
    if (n > 1) {
        hipLaunchKernelGGL(batch_hist_kernel, dim3(ORDER_BLOCKS), dim3(256), 0, s, g, B);
        hipLaunchKernelGGL(batch_order_kernel, dim3(ORDER_BLOCKS), dim3(256), 0, s, g, B);
        TL3D_HIP(hipGetLastError());
    }
    // sub-brick masks of the listed bricks (their number is known only on the device: a fixed grid strides over each list)
    const int nbricks = g.nbx * g.nby * g.nbz;
    int ncb = (nbricks + 31) / 32;                       // 8 bricks per wave, 4 waves per workgroup
    const int cap = n >= 8 ? 128 : 512;
    if (ncb > cap) ncb = cap;
    hipLaunchKernelGGL(subbrick_classify_kernel, dim3(ncb, n), dim3(256), 0, s, cam, g, py, B);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_fold_free(hipStream_t s, const Grid &g, int2 *grid, unsigned *free_cnt) {
    const unsigned nbricks = (unsigned)(g.nbx * g.nby * g.nbz);
    const unsigned nb = (nbricks + 3u) / 4u < 2048u ? (nbricks + 3u) / 4u : 2048u;
    hipLaunchKernelGGL(fold_free_kernel, dim3(nb ? nb : 1), dim3(256), 0, s, g, grid, free_cnt, nbricks);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// the dominant kernel: ONE read-modify-write pass over the bricks the n prepared frames of the batch list
int launch_tsdf_update(hipStream_t s, const Cam &cam, const Grid &g, int n, int max_frames, bool depth_u16, float mind, float maxd, int2 *grid,
                       void *scratch, unsigned long long *counters, bool count, int max_blocks, int xcd_group) {
    const BatchLayout L = batch_layout(cam, g, max_frames);
    BatchBufs B = batch_bufs(L, scratch, n);
    if (n == 1) B.order = B.list;
    const int nbricks = g.nbx * g.nby * g.nbz;
    int nblk = (nbricks + 3) / 4;
    if (nblk > max_blocks) nblk = max_blocks;
#ifdef TL3D_EXPERIMENTS
    static const int no_order = getenv("TL3D_NO_ORDER") ? atoi(getenv("TL3D_NO_ORDER")) : 0;     // walk the list as the classification left it
    if (no_order) B.order = B.list;
    static const int exp_mode = getenv("TL3D_TSDF_EXP") ? atoi(getenv("TL3D_TSDF_EXP")) : 0;     // timing ablations: results incomplete
    static const int upd_lds = getenv("TL3D_UPD_LDS") ? atoi(getenv("TL3D_UPD_LDS")) : 0;        // unused LDS per workgroup: caps the waves per SIMD
    static const int upd_old = getenv("TL3D_UPD_OLD") ? atoi(getenv("TL3D_UPD_OLD")) : 0;        // the frame-major kernel of round 3 (A/B on one box)
    if (upd_old || exp_mode) {
        int ob = max_blocks;
        if (getenv("TL3D_UPD_OLD_BLOCKS")) ob = atoi(getenv("TL3D_UPD_OLD_BLOCKS"));
        if (nblk > ob) nblk = ob;
#define TL3D_LAUNCH_UPD(C_, T_, E_) \
    hipLaunchKernelGGL((tsdf_update_kernel<C_, T_, E_>), dim3(nblk), dim3(256), upd_lds, s, cam, g, B, xcd_group, grid, counters)
#define TL3D_LAUNCH_UPD_E(E_)                                                                          \
    do {                                                                                               \
        if (depth_u16) TL3D_LAUNCH_UPD(false, uint16_t, E_); else TL3D_LAUNCH_UPD(false, float, E_);   \
    } while (0)
        if (count) {
            if (depth_u16) TL3D_LAUNCH_UPD(true, uint16_t, 0); else TL3D_LAUNCH_UPD(true, float, 0);
        }
        else if (exp_mode == 1) TL3D_LAUNCH_UPD_E(1);
        else if (exp_mode == 2) TL3D_LAUNCH_UPD_E(2);
        else if (exp_mode == 3) TL3D_LAUNCH_UPD_E(3);
        else if (exp_mode == 4) TL3D_LAUNCH_UPD_E(4);
        else if (exp_mode == 5) TL3D_LAUNCH_UPD_E(5);
        else TL3D_LAUNCH_UPD_E(0);
#undef TL3D_LAUNCH_UPD_E
#undef TL3D_LAUNCH_UPD
        TL3D_HIP(hipGetLastError());
        return TL3D_OK;
    }
#endif
    (void)xcd_group;
#ifdef TL3D_EXPERIMENTS
    static const int pexp = getenv("TL3D_PAIRS_EXP") ? atoi(getenv("TL3D_PAIRS_EXP")) : 0;
#define TL3D_LAUNCH_PEXP(E_) \
    hipLaunchKernelGGL((tsdf_update_pairs_kernel<false, float, E_>), dim3(nblk), dim3(256), 0, s, cam, g, B, mind, maxd, grid, counters)
    if (pexp && !count && !depth_u16) {
        switch (pexp) {
            case 1: TL3D_LAUNCH_PEXP(1); break;
            case 2: TL3D_LAUNCH_PEXP(2); break;
            case 3: TL3D_LAUNCH_PEXP(3); break;
            case 4: TL3D_LAUNCH_PEXP(4); break;
            case 7: TL3D_LAUNCH_PEXP(7); break;
            case 8: TL3D_LAUNCH_PEXP(8); break;
            case 16: TL3D_LAUNCH_PEXP(16); break;
            case 32: TL3D_LAUNCH_PEXP(32); break;
            default: TL3D_LAUNCH_PEXP(15); break;
        }
        TL3D_HIP(hipGetLastError());
        return TL3D_OK;
    }
#endif
#define TL3D_LAUNCH_PAIRS(C_, T_) \
    hipLaunchKernelGGL((tsdf_update_pairs_kernel<C_, T_>), dim3(nblk), dim3(256), 0, s, cam, g, B, mind, maxd, grid, counters)
    if (count) {
        if (depth_u16) TL3D_LAUNCH_PAIRS(true, uint16_t); else TL3D_LAUNCH_PAIRS(true, float);
    } else {
        if (depth_u16) TL3D_LAUNCH_PAIRS(false, uint16_t); else TL3D_LAUNCH_PAIRS(false, float);
    }
#undef TL3D_LAUNCH_PAIRS
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

#ifdef TL3D_EXPERIMENTS
// when the waves of the last stamped launch ended (TL3D_PAIRS_EXP=32)
void tsdf_debug_print_spans(int nwaves) {
    static unsigned long long h[16384][2];
    if (nwaves > 16384) nwaves = 16384;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_upd_span), sizeof(h)) != hipSuccess) return;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < nwaves; ++i) if (h[i][0]) { if (h[i][0] < t0) t0 = h[i][0]; if (h[i][1] > t1) t1 = h[i][1]; }
    if (t1 <= t0) return;
    const double span = (double)(t1 - t0);
    int he[10] = {0};
    for (int i = 0; i < nwaves; ++i) if (h[i][0]) { int b = (int)(10.0 * (double)(h[i][1] - t0) / span); he[b > 9 ? 9 : b]++; }
    fprintf(stderr, "[tl3d exp] launch %.1f us; waves ENDING in each tenth of it:", span * 0.01);
    for (int i = 0; i < 10; ++i) fprintf(stderr, " %d", he[i]);
    fprintf(stderr, "\n");
}
#endif

}  // namespace tl3d
