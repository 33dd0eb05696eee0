// kernels_icp.hip -- row a10: vertex/normal map and pairwise point-to-plane ICP (projective association).
// The reference has no ICP (its poses come from SIFT + essential matrix, depth_to_reconstruction.py:144-215);
// this replaces that pose source and returns the relative pose in the reference's chaining convention
// (D2R:618-620).  Convention and f32 sequence = oracle/tl3d_oracle.c (orc_normals, icp_pass, solve6, se3_apply).
//
// Device-resident iteration: icp_reduce_kernel accumulates the 6x6 normal equations (21 + 6 + 3 sums, fp64)
// per lane, reduces them with 64-lane wave shuffles, then across the 4 waves through LDS, and writes one
// 32-double partial per workgroup; icp_solve_kernel sums the partials in block order (deterministic), solves
// the damped system by a 6x6 Jacobi eigen-decomposition (unobservable directions dropped) and updates T in device memory.  No host round trip inside the loop: the host
// replays one captured hipGraph per run (2 x (iters+1) kernels + the state copies) and reads the state once.  Bytes per iteration and sampled pixel: 4 (source
// depth) + 16 (target normal+depth gather) = 20 B.
#include "tl3d_internal.h"

namespace tl3d {

__device__ __forceinline__ bool load_vertex(const Cam &cam, const float *__restrict__ depth, int u, int v, float sc,
                                            float mind, float maxd, float p[3]) {
    const float d = depth[(size_t)v * cam.W + u] * sc;
    if (!(d > mind && d < maxd)) return false;
    p[0] = (((float)u - cam.cx) / cam.fx) * d;
    p[1] = (((float)v - cam.cy) / cam.fy) * d;
    p[2] = d;
    return true;
}

__global__ __launch_bounds__(256) void normals_kernel(Cam cam, const float *__restrict__ depth, float sc, float mind,
                                                      float maxd, float jump, float4 *__restrict__ nmap) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= cam.W || v >= cam.H) return;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    float p[3], l[3], r[3], up[3], dn[3];
    bool ok = (u >= 1 && v >= 1 && u <= cam.W - 2 && v <= cam.H - 2);
    ok = ok && load_vertex(cam, depth, u, v, sc, mind, maxd, p);
    ok = ok && load_vertex(cam, depth, u - 1, v, sc, mind, maxd, l);
    ok = ok && load_vertex(cam, depth, u + 1, v, sc, mind, maxd, r);
    ok = ok && load_vertex(cam, depth, u, v - 1, sc, mind, maxd, up);
    ok = ok && load_vertex(cam, depth, u, v + 1, sc, mind, maxd, dn);
    if (ok) {
        ok = fabsf(l[2] - p[2]) <= jump && fabsf(r[2] - p[2]) <= jump && fabsf(up[2] - p[2]) <= jump &&
             fabsf(dn[2] - p[2]) <= jump;
    }
    if (ok) {
        const float ax = r[0] - l[0], ay = r[1] - l[1], az = r[2] - l[2];
        const float bx = dn[0] - up[0], by = dn[1] - up[1], bz = dn[2] - up[2];
        float nx = fmaf(ay, bz, -(az * by));
        float ny = fmaf(az, bx, -(ax * bz));
        float nz = fmaf(ax, by, -(ay * bx));
        const float len2 = fmaf(nx, nx, fmaf(ny, ny, nz * nz));
        if (len2 > 1e-30f) {
            const float inv = 1.0f / sqrtf(len2);
            nx *= inv; ny *= inv; nz *= inv;
            const float dotv = fmaf(nx, p[0], fmaf(ny, p[1], nz * p[2]));
            if (dotv > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
            o = make_float4(nx, ny, nz, p[2]);
        }
    }
    nmap[(size_t)v * cam.W + u] = o;
}

__global__ __launch_bounds__(256) void icp_reduce_kernel(Cam cam, const IcpRun *__restrict__ run,
                                                         const IcpState *__restrict__ state, int final_pass,
                                                         double *__restrict__ slab) {
    if (!final_pass && state->done) return;
    const float *__restrict__ depth_s = run->depth_src;
    const float4 *__restrict__ nmap_t = run->nmap_tgt;
    const float sc = run->scale, mind = run->mind, maxd = run->maxd, md2 = run->md2;
    const int stride = run->stride, Ws = run->Ws, Hs = run->Hs;
    __shared__ double sm[4][ICP_SLAB];
    float r[9], t[3];
    r[0] = (float)state->T[0]; r[1] = (float)state->T[1]; r[2] = (float)state->T[2];  t[0] = (float)state->T[3];
    r[3] = (float)state->T[4]; r[4] = (float)state->T[5]; r[5] = (float)state->T[6];  t[1] = (float)state->T[7];
    r[6] = (float)state->T[8]; r[7] = (float)state->T[9]; r[8] = (float)state->T[10]; t[2] = (float)state->T[11];
    const float wlim = (float)cam.W - 0.5f, hlim = (float)cam.H - 0.5f;
    double acc[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) acc[i] = 0.0;
    const long long ns = (long long)Ws * Hs;
    for (long long s = (long long)blockIdx.x * 256 + threadIdx.x; s < ns; s += (long long)gridDim.x * 256) {
        const int vs = (int)(s / Ws), us = (int)(s - (long long)vs * Ws);
        const int u = us * stride, v = vs * stride;
        float ps[3];
        if (!load_vertex(cam, depth_s, u, v, sc, mind, maxd, ps)) continue;
        acc[29] += 1.0;
        const float px = fmaf(r[0], ps[0], fmaf(r[1], ps[1], fmaf(r[2], ps[2], t[0])));
        const float py = fmaf(r[3], ps[0], fmaf(r[4], ps[1], fmaf(r[5], ps[2], t[1])));
        const float pz = fmaf(r[6], ps[0], fmaf(r[7], ps[1], fmaf(r[8], ps[2], t[2])));
        if (!(pz > 0.0f)) continue;
        const float inv = 1.0f / pz;
        const float uf = fmaf(cam.fx * px, inv, cam.cx);
        const float vf = fmaf(cam.fy * py, inv, cam.cy);
        if (!(uf >= -0.5f && uf < wlim && vf >= -0.5f && vf < hlim)) continue;
        int ut = (int)floorf(uf + 0.5f), vt = (int)floorf(vf + 0.5f);
        ut = min(ut, cam.W - 1);
        vt = min(vt, cam.H - 1);
        const float4 nd = nmap_t[(size_t)vt * cam.W + ut];
        const float dt = nd.w;
        if (!(dt > 0.0f)) continue;
        const float qx = (((float)ut - cam.cx) / cam.fx) * dt;
        const float qy = (((float)vt - cam.cy) / cam.fy) * dt;
        const float dx = px - qx, dy = py - qy, dz = pz - dt;
        const float dist2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        if (!(dist2 <= md2)) continue;
        const float res = fmaf(dx, nd.x, fmaf(dy, nd.y, dz * nd.z));
        const double J[6] = {(double)fmaf(py, nd.z, -(pz * nd.y)), (double)fmaf(pz, nd.x, -(px * nd.z)),
                             (double)fmaf(px, nd.y, -(py * nd.x)), (double)nd.x, (double)nd.y, (double)nd.z};
        const double rr = (double)res;
        int m = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = a; b < 6; ++b) { acc[m] += J[a] * J[b]; ++m; }
            acc[21 + a] += J[a] * rr;
        }
        acc[27] += rr * rr;
        acc[28] += 1.0;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 30; ++i) {
        double v = acc[i];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
        if (lane == 0) sm[wid][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < ICP_SLAB) {
        const int i = threadIdx.x;
        slab[(size_t)blockIdx.x * ICP_SLAB + i] = (i < 30) ? ((sm[0][i] + sm[1][i]) + sm[2][i]) + sm[3][i] : 0.0;
    }
}

// Same algorithm and the same arithmetic order as oracle/tl3d_oracle.c: solve6 (cyclic Jacobi, 12 sweeps, relative
// eigenvalue cutoff), spread over the lanes of one wave: lane k (< 6) keeps row k of A and row k of V in registers; a
// sweep's rotations run three at a time on disjoint index pairs (see the loop), partners swap rows through shuffles.
// All 64 lanes execute every shuffle (EXEC full); lanes >= 6 carry zeros.
// Returns 0 on success; x[k] is valid on every lane (k < 6).
__device__ __forceinline__ int solve6_wave(const double *__restrict__ a21, const double *__restrict__ b, double damping,
                                           double eig_rel, double x[6]) {
    const int lane = threadIdx.x & 63;
    double a[6], v[6];
    double tr = 0.0;
    {
        int m = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) {
                const double e = a21[m++];
                if (lane == i) a[j] = e;
                if (lane == j) a[i] = e;
            }
    }
    tr = ((((a21[0] + a21[6]) + a21[11]) + a21[15]) + a21[18]) + a21[20];      // A00+A11+...+A55 in index order
    if (!(tr > 0.0)) return 1;
    const double lam = damping * (tr / 6.0);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        if (lane >= 6) a[j] = 0.0;
        if (lane == j) a[j] += lam;
        v[j] = (lane == j) ? 1.0 : 0.0;
    }
    // Round-robin order: a sweep is 5 rounds of 3 rotations on disjoint pairs (the oracle's RR table).  Lane i < 6 computes
    // the angle of the pair it belongs to, so a round costs one chain of fp64 div/sqrt instead of three; the (c, s) of the
    // three pairs are then broadcast, every lane rotates its row's columns, and partners exchange rows.
    for (int sweep = 0; sweep < 12; ++sweep) {
        double offmax = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k != lane && fabs(a[k]) > offmax) offmax = fabs(a[k]);          // lanes >= 6 hold zeros
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) offmax = fmax(offmax, __shfl_xor(offmax, d));
        offmax = __shfl(offmax, 0);
        if (!(offmax > 1e-15 * tr)) break;                          // wave-uniform; same rule as the oracle
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            // pairs of round r as compile-time constants
            constexpr int RP[5][3] = {{0, 1, 2}, {0, 3, 1}, {0, 2, 1}, {0, 1, 4}, {0, 2, 3}};
            constexpr int RQ[5][3] = {{5, 4, 3}, {4, 5, 2}, {3, 4, 5}, {2, 3, 5}, {1, 5, 4}};
            int partner = lane;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (lane == RP[r][u]) partner = RQ[r][u];
                if (lane == RQ[r][u]) partner = RP[r][u];
            }
            const bool isp = lane < partner;
            double my_diag = 0.0, my_off = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (k == lane) my_diag = a[k];
                if (k == partner && partner != lane) my_off = a[k];
            }
            const double partner_diag = __shfl(my_diag, partner);
            const double off_p = __shfl(my_off, partner);               // the lower lane's A[p][q] is the pivot for both
            const double app = isp ? my_diag : partner_diag, aqq = isp ? partner_diag : my_diag;
            const double apq = isp ? my_off : off_p;
            double c = 1.0, sn = 0.0;
            if (lane < 6 && !(fabs(apq) < 1e-300)) {
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                c = 1.0 / sqrt(t * t + 1.0);
                sn = t * c;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {                               // columns of every row
                const double cu = __shfl(c, RP[r][u]), su = __shfl(sn, RP[r][u]);
                const double akp = a[RP[r][u]], akq = a[RQ[r][u]];
                a[RP[r][u]] = cu * akp - su * akq;
                a[RQ[r][u]] = su * akp + cu * akq;
                const double vkp = v[RP[r][u]], vkq = v[RQ[r][u]];
                v[RP[r][u]] = cu * vkp - su * vkq;
                v[RQ[r][u]] = su * vkp + cu * vkq;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {                               // rows: partners exchange
                const double other = __shfl(a[k], partner);
                const double np_ = c * a[k] - sn * other;               // this lane is p: c*row_p - s*row_q
                const double nq_ = sn * other + c * a[k];               // this lane is q: s*row_p + c*row_q
                if (partner != lane) a[k] = isp ? np_ : nq_;
            }
        }
    }
    double lam_e[6], lmax = 0.0;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        lam_e[e] = __shfl(a[e], e);
        if (lam_e[e] > lmax) lmax = lam_e[e];
    }
    if (!(lmax > 0.0)) return 1;
    double xk = 0.0;
    int used = 0;
    const double bk = (lane < 6) ? b[lane] : 0.0;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const double l = lam_e[e];
        if (!(l > eig_rel * lmax) || !(l > 0.0)) continue;            // wave-uniform
        // proj = sum_k V[k][e] * b[k], k = 0..5 in order (as the oracle)
        double proj = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) proj += __shfl(v[e] * bk, k);
        const double coef = -proj / l;
        xk += coef * v[e];
        ++used;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = __shfl(xk, k);
    return used == 0;
}

__device__ void se3_apply(const double x[6], double *T) {
    const double wx = x[0], wy = x[1], wz = x[2];
    const double th2 = wx * wx + wy * wy + wz * wz;
    const double th = sqrt(th2);
    double a, bq;
    if (th < 1e-12) { a = 1.0; bq = 0.5; } else { a = sin(th) / th; bq = (1.0 - cos(th)) / th2; }
    const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double K2[9], dR[9], Tn[16];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j];
            K2[3 * i + j] = s;
        }
    for (int i = 0; i < 9; ++i) dR[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * K[i] + bq * K2[i];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += dR[3 * i + k] * T[4 * k + j];
            Tn[4 * i + j] = s;
        }
        Tn[4 * i + 3] += x[3 + i];
    }
    Tn[12] = 0; Tn[13] = 0; Tn[14] = 0; Tn[15] = 1;
    for (int i = 0; i < 16; ++i) T[i] = Tn[i];
}

__global__ __launch_bounds__(256) void icp_solve_kernel(const double *__restrict__ slab, int nblocks, IcpState *state,
                                                        const IcpRun *__restrict__ run, int final_pass) {
    if (!final_pass && state->done) return;
    const double damping = run->damping, eps = run->eps, eig_rel = run->eig_rel;
    __shared__ double part[8][ICP_SLAB];
    __shared__ double sums[ICP_SLAB];
    const int t = threadIdx.x;
    {   // slab reduction: 8 groups x 32 components, loads batched 8 deep, combined in a fixed order (deterministic)
        const int comp = t & 31, grp = t >> 5;
        double s = 0.0;
        int b = grp;
        for (; b + 56 < nblocks; b += 64) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = slab[(size_t)(b + 8 * k) * ICP_SLAB + comp];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; b < nblocks; b += 8) s += slab[(size_t)b * ICP_SLAB + comp];
        part[grp][comp] = s;
    }
    __syncthreads();
    if (t < ICP_SLAB) {
        double s = part[0][t];
#pragma unroll
        for (int g = 1; g < 8; ++g) s += part[g][t];
        sums[t] = s;
        state->sums[t] = s;
    }
    __syncthreads();
    if (final_pass || t >= 64) return;                     // wave 0 solves
    double x[6];
    const int fail = (sums[28] < 6.0) ? 1 : solve6_wave(sums, sums + 21, damping, eig_rel, x);
    if (t != 0) return;
    if (fail) {
        state->done = 1;
        state->status = 2;
        return;
    }
    se3_apply(x, state->T);
    state->iters_run += 1;
    double mx = 0.0;
    for (int i = 0; i < 6; ++i) mx = fmax(mx, fabs(x[i]));
    if (mx < eps) {
        state->done = 1;
        state->status = 1;
    }
}

int launch_normals(hipStream_t s, const Cam &cam, const float *depth, float scale, float mind, float maxd, float jump,
                   float4 *nmap) {
    dim3 grid((cam.W + 63) / 64, (cam.H + 3) / 4);
    hipLaunchKernelGGL(normals_kernel, grid, dim3(256), 0, s, cam, depth, scale, mind, maxd, jump, nmap);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_icp_iteration(hipStream_t s, const Cam &cam, const IcpRun *run, int final_pass, double *slab, IcpState *state, int nblocks) {
    hipLaunchKernelGGL(icp_reduce_kernel, dim3(nblocks), dim3(256), 0, s, cam, run, state, final_pass, slab);
    TL3D_HIP(hipGetLastError());
    hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(256), 0, s, slab, nblocks, state, run, final_pass);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

}  // namespace tl3d
