// kernels_tsdf.hip -- row a11: TSDF integration of one depth frame into the brick-major grid.
// No reference code exists for this row; the convention is the oracle's
// (oracle/tl3d_oracle.c: orc_tsdf_integrate) and both sides evaluate the same f32 sequence, so the integer
// grid {sum of rint(tsdf*32767), weight} is bit-identical.
//
// Mapping: one 64-lane wave per 8x8x8 brick (4 KB of records, contiguous).  Each lane owns 4 x 16 B = 8 voxels;
// a wave-instruction touches 1 KB contiguous.  Bricks whose bounding sphere is outside the (1-pixel widened)
// view frustum are skipped before any record is touched.  Records are loaded only by lanes that update and
// stored only when changed, so HBM traffic = 16 B x (lanes that update); the depth gathers (8.3 MB frame) are
// served by L2 / Infinity Cache.  Algorithmic bytes per launch = 8 B x (records read + records written), counted
// by the kernel itself in counting mode (SURVEY.md section 8d "counted, never estimated").
#include "tl3d_internal.h"

namespace tl3d {

struct TsdfConst {
    float mind, maxd, sc, wlim, hlim;
};

__device__ __forceinline__ bool tsdf_voxel(const Cam &cam, const Grid &g, const TsdfConst &c, const float *__restrict__ depth,
                                           float xc, float yc, float zc, int &q) {
    if (!(zc > 0.0f)) return false;
    const float inv = 1.0f / zc;
    const float uf = fmaf(cam.fx * xc, inv, cam.cx);
    const float vf = fmaf(cam.fy * yc, inv, cam.cy);
    if (!(uf >= -0.5f && uf < c.wlim && vf >= -0.5f && vf < c.hlim)) return false;
    int u = (int)floorf(uf + 0.5f), v = (int)floorf(vf + 0.5f);
    u = min(u, cam.W - 1);
    v = min(v, cam.H - 1);
    const float d = depth[(size_t)v * cam.W + u] * c.sc;
    if (!(d > c.mind && d < c.maxd)) return false;
    const float sdf = d - zc;
    if (!(sdf >= -g.trunc)) return false;
    const float tsdf = fminf(1.0f, sdf * g.inv_trunc);
    q = (int)rintf(tsdf * 32767.0f);
    return true;
}

template <bool COUNT>
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(Cam cam, Grid g, PoseF pose, Frustum fr, TsdfConst c,
                                                             const float *__restrict__ depth, int2 *__restrict__ grid,
                                                             unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    // XCD-aware block remap: blocks that share an XCD (blockIdx % 8) walk one contiguous eighth of each window of
    // bricks, so the depth tiles they gather stay in that XCD's L2.  Speed only; any placement is correct.
    const int nblk = gridDim.x;
    const int per = nblk >> 3;                                   // launcher guarantees nblk % 8 == 0
    const int vblock = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int nbricks = g.nbx * g.nby * g.nbz;
    const float rad = 6.9282032f * g.vs * 1.01f;                 // half diagonal of a brick: 4*sqrt(3)*voxel, +1 %
    unsigned nread = 0, nwritten = 0, nvisited = 0;

    for (int brick0 = vblock * 4; brick0 < nbricks; brick0 += nblk * 4) {
        const int brick = __builtin_amdgcn_readfirstlane(brick0 + wid);
        if (brick >= nbricks) break;
        const int bx = brick % g.nbx;
        const int by = (brick / g.nbx) % g.nby;
        const int bz = brick / (g.nbx * g.nby);
        {   // conservative frustum cull on the brick's bounding sphere
            const float wx = fmaf((float)(bx * 8 + 4), g.vs, g.ox);
            const float wy = fmaf((float)(by * 8 + 4), g.vs, g.oy);
            const float wz = fmaf((float)(bz * 8 + 4), g.vs, g.oz);
            const float bxc = pose.r[0] * wx + pose.r[1] * wy + pose.r[2] * wz + pose.t[0];
            const float byc = pose.r[3] * wx + pose.r[4] * wy + pose.r[5] * wz + pose.t[1];
            const float bzc = pose.r[6] * wx + pose.r[7] * wy + pose.r[8] * wz + pose.t[2];
            if (bzc + rad <= 0.0f) continue;
            if (fr.lx * bxc + fr.lz * bzc < -rad) continue;
            if (fr.rx * bxc + fr.rz * bzc < -rad) continue;
            if (fr.ty * byc + fr.tz * bzc < -rad) continue;
            if (fr.by * byc + fr.bz * bzc < -rad) continue;
        }
        if (COUNT) ++nvisited;
        int4 *__restrict__ recs = reinterpret_cast<int4 *>(grid + ((size_t)brick << 9));
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
            const float px0 = fmaf((float)i + 0.5f, g.vs, g.ox);
            const float px1 = fmaf((float)(i + 1) + 0.5f, g.vs, g.ox);
            int q0 = 0, q1 = 0;
            const bool u0 = tsdf_voxel(cam, g, c, depth, fmaf(pose.r[0], px0, ax), fmaf(pose.r[3], px0, ay),
                                       fmaf(pose.r[6], px0, az), q0);
            const bool u1 = tsdf_voxel(cam, g, c, depth, fmaf(pose.r[0], px1, ax), fmaf(pose.r[3], px1, ay),
                                       fmaf(pose.r[6], px1, az), q1);
            if (u0 | u1) {
                int4 rec = recs[pr];
                if (u0) { rec.x += q0; rec.y += 1; }
                if (u1) { rec.z += q1; rec.w += 1; }
                recs[pr] = rec;
                if (COUNT) { nread += 2; nwritten += 2; }
            }
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
            atomicAdd(counters + 4, (unsigned long long)nvisited);
        }
    }
}

int launch_tsdf_integrate(hipStream_t s, const Cam &cam, const Grid &g, const PoseF &p, const Frustum &fr, const float *depth,
                          float scale, float mind, float maxd, int2 *grid, unsigned long long *counters, bool count) {
    TsdfConst c;
    c.mind = mind;
    c.maxd = maxd;
    c.sc = scale;
    c.wlim = (float)cam.W - 0.5f;
    c.hlim = (float)cam.H - 0.5f;
    const int nbricks = g.nbx * g.nby * g.nbz;
    int nblk = (nbricks + 3) / 4;
    if (nblk > 2048) nblk = 2048;
    nblk = (nblk + 7) & ~7;
    if (count)
        hipLaunchKernelGGL(tsdf_integrate_kernel<true>, dim3(nblk), dim3(256), 0, s, cam, g, p, fr, c, depth, grid, counters);
    else
        hipLaunchKernelGGL(tsdf_integrate_kernel<false>, dim3(nblk), dim3(256), 0, s, cam, g, p, fr, c, depth, grid, counters);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

}  // namespace tl3d
