// kernels_icp.hip -- row a10: vertex/normal map and pairwise point-to-plane ICP (projective association).
// The reference has no ICP (its poses come from SIFT + essential matrix, depth_to_reconstruction.py:144-215);
// this replaces that pose source and returns the relative pose in the reference's chaining convention
// (D2R:618-620).  Convention and f32 sequence = oracle/tl3d_oracle.c (orc_normals, icp_pass, solve6, se3_apply).
//
// Device-resident iteration, ONE launch per iteration (icp_iter_kernel): every workgroup accumulates the 6x6 normal
// equations (21 + 6 + 3 sums, fp64) per lane, reduces them with 64-lane wave shuffles, then across its 4 waves through
// LDS, and publishes one 32-double partial (write-through stores, drained, then an agent-scope ticket).  The workgroup whose
// ticket is the last one sums the partials in block order (deterministic), solves the damped system by a 6x6 Jacobi
// eigen-decomposition (unobservable directions dropped), updates T in device memory and re-arms the ticket.  A dependent
// two-kernel chain (reduce, then a one-workgroup solve) cost a second launch boundary and a second ramp per iteration:
// 38 us -> see DESIGN.md section 7.  No host round trip inside the loop: the host replays one captured hipGraph per run
// ((iters+1) kernels + the state copies) and reads the state once.  Bytes per iteration and sampled pixel: 4 (source
// depth) + 16 (target normal+depth gather) = 20 B.
//
// Batched form (icp_batch_kernel, tl3d_icp_batch_*): many pairs, each through all its coarse-to-fine levels and all its
// iterations, in ONE persistent launch; the workgroups that share a pair meet at a per-pair barrier in device memory after every
// pass, and every workgroup works on two (pair, member) slots in turn.  Same accumulate / solve code as the per-iteration
// kernel; what the pipeline uses.
// Maps: normal maps and window-averaged depth live in PHASE-MAJOR ROWS (tl3d_internal.h: pm_index); a raw frame is row-major.
#include "tl3d_internal.h"

namespace tl3d {

// PM: `depth` is kept in phase-major rows (pm_index, tl3d_internal.h: the window-averaged depth), else row-major (the frame itself)
template <bool PM>
__device__ __forceinline__ bool load_vertex(const Cam &cam, const float *__restrict__ depth, int u, int v, float sc,
                                            float mind, float maxd, float p[3]) {
    const float d = depth[PM ? pm_index(u, v, pm_w4(cam.W)) : (size_t)v * cam.W + u] * sc;
    if (!(d > mind && d < maxd)) return false;
    p[0] = (((float)u - cam.cx) / cam.fx) * d;
    p[1] = (((float)v - cam.cy) / cam.fy) * d;
    p[2] = d;
    return true;
}

// Noise-robust normals, first half (oracle: orc_normals_smooth): the depth averaged over the (2 radius + 1)^2 window of every pixel
// -- over the pixels that are valid and within `jump` of the centre pixel, summed in row-major order in f32 (the oracle's order);
// out stays in the units of `depth`, 0 = invalid centre.  1 mm of depth noise turns central differences over one pixel (0.6 mm
// apart at 1 m, f = 1719) into noise; normals_kernel then takes its tangent vectors `radius` pixels to either side of this map.
__global__ __launch_bounds__(256) void smooth_depth_kernel(Cam cam, const float *__restrict__ depth, float sc, float mind, float maxd, float jump,
                                                           int radius, float *__restrict__ out) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= cam.W || v >= cam.H) return;
    const float d0 = depth[(size_t)v * cam.W + u] * sc;
    float res = 0.0f;
    if (d0 > mind && d0 < maxd) {
        // mean of the INVERSE depth (linear in the pixel coordinates on a plane, however oblique: a plain mean of the depth is
        // biased on a floor seen at a grazing angle, 0.06 mm per frame of drift on config 4) over the centre and the pixel PAIRS
        // (u + du, v + dv), (u - du, v - dv) that are both in the image, valid and within the jump of the centre -- a symmetric
        // set: the mean of a linear function over it is its centre value, at the edge of a surface too.  The oracle's order.
        float sum = 1.0f / depth[(size_t)v * cam.W + u];
        int n = 1;
        for (int dv = 0; dv <= radius; ++dv)
            for (int du = (dv == 0 ? 1 : -radius); du <= radius; ++du) {
                const int ua = u + du, va = v + dv, ub = u - du, vb = v - dv;
                if (ua < 0 || ua >= cam.W || va < 0 || va >= cam.H || ub < 0 || ub >= cam.W || vb < 0 || vb >= cam.H) continue;
                const float ra = depth[(size_t)va * cam.W + ua], rb = depth[(size_t)vb * cam.W + ub];
                const float da = ra * sc, db = rb * sc;
                if (!(da > mind && da < maxd) || !(db > mind && db < maxd)) continue;
                if (!(fabsf(da - d0) <= jump) || !(fabsf(db - d0) <= jump)) continue;
                sum += 1.0f / ra;
                sum += 1.0f / rb;
                n += 2;
            }
        res = (float)n / sum;
    }
    out[pm_index(u, v, pm_w4(cam.W))] = res;              // phase-major rows: what the registration samples at its stride
}

// step: the tangent vectors come from the pixels `step` to either side (1: plain central differences).  The map is written in
// phase-major rows.
template <bool PM>
__global__ __launch_bounds__(256) void normals_kernel(Cam cam, const float *__restrict__ depth, float sc, float mind,
                                                      float maxd, float jump, int step, float4 *__restrict__ nmap) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= cam.W || v >= cam.H) return;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    float p[3], l[3], r[3], up[3], dn[3];
    bool ok = (u >= step && v >= step && u <= cam.W - 1 - step && v <= cam.H - 1 - step);
    ok = ok && load_vertex<PM>(cam, depth, u, v, sc, mind, maxd, p);
    ok = ok && load_vertex<PM>(cam, depth, u - step, v, sc, mind, maxd, l);
    ok = ok && load_vertex<PM>(cam, depth, u + step, v, sc, mind, maxd, r);
    ok = ok && load_vertex<PM>(cam, depth, u, v - step, sc, mind, maxd, up);
    ok = ok && load_vertex<PM>(cam, depth, u, v + step, sc, mind, maxd, dn);
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
    nmap[pm_index(u, v, pm_w4(cam.W))] = o;
}

// accumulate this workgroup's share of the sums and leave the workgroup total in sm_out[0..31] (valid for threads < 32
// after the function's last barrier)
// `member` of `members` workgroups share one registration: member m takes samples m*256 + tid, + members*256, ...
// ray_tab (batched kernel; null: computed in place): the pixel-ray factors ((float)u - cx) / fx for u < W, then ((float)v - cy) / fy
// for v < H, in LDS -- the very quotients the expression gives (filled with that expression), looked up instead of divided
// out four times per sample and pass (an IEEE f32 division is ~10 vector instructions; they were 40 of the ~115 per sample).
template <bool SCALE, bool TAB>
__device__ __forceinline__ void icp_accumulate_core(const Cam &cam, const float *__restrict__ depth_s, const float4 *__restrict__ nmap_t,
                                                    float sc, float mind, float maxd, float md2, int stride, int Ws, int Hs,
                                                    const float r[9], const float t[3], int member, int members, int src_pm,
                                                    double (*sm)[ICP_SLAB], double *__restrict__ sm_out, unsigned long long *stamp = nullptr,
                                                    const float *ray_tab = nullptr) {
    const float wlim = (float)cam.W - 0.5f, hlim = (float)cam.H - 0.5f;
    // Both maps are device memory: say so.  The batched kernel reads the pointers from a table in memory, where the compiler only
    // knows a generic pointer and issues FLAT loads -- which also count on the LDS counter, so that every wait for an LDS read (the
    // ray tables below) would wait for every gather in flight and undo the software pipeline.
    typedef const __attribute__((address_space(1))) float *gfloat_p;
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) v4f *gfloat4_p;
    const gfloat_p depth_g = (gfloat_p)depth_s;
    const gfloat4_p nmap_g = (gfloat4_p)nmap_t;
    // the target map is kept in phase-major rows, the source depth too when it is a window-averaged one (src_pm); a raw frame is
    // row-major: (mask, shift, phase length, row length) make one index expression of both (wave-uniform values)
    const int w4 = pm_w4(cam.W);
    const int s_mask = src_pm ? 3 : 0, s_shift = src_pm ? 2 : 0, s_row = src_pm ? 4 * w4 : cam.W;
    const float *xtab = ray_tab, *ytab = ray_tab + cam.W;
    auto xray = [&](int u) { return TAB ? xtab[u] : ((float)u - cam.cx) / cam.fx; };
    auto yray = [&](int v) { return TAB ? ytab[v] : ((float)v - cam.cy) / cam.fy; };
    double acc[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) acc[i] = 0.0;
    double accs[8];                                      // scale column (Sim(3) runs): sum J_a J_alpha (6), J_alpha^2, J_alpha r
#pragma unroll
    for (int i = 0; i < 8; ++i) accs[i] = 0.0;
    // Samples are walked with 32-bit pixel coordinates kept per thread (a sample index divided by the level's width cost a 64-bit
    // division per sample): sample s = vs * Ws + us; the next one of this thread is `step` = members * 256 samples on.
    const int step = members * 256;
    const int dvs = step / Ws, dus = step - dvs * Ws;        // wave-uniform
    struct Samp { float px, py, pz; int ut, vt; bool src_ok, ok; };
    auto prep = [&](float draw, int u, int v) {
        Samp q;
        const float d = draw * sc;
        q.src_ok = (d > mind && d < maxd);
        const float p0 = xray(u) * d;
        const float p1 = yray(v) * d;
        q.px = fmaf(r[0], p0, fmaf(r[1], p1, fmaf(r[2], d, t[0])));
        q.py = fmaf(r[3], p0, fmaf(r[4], p1, fmaf(r[5], d, t[1])));
        q.pz = fmaf(r[6], p0, fmaf(r[7], p1, fmaf(r[8], d, t[2])));
        bool ok = q.src_ok && (q.pz > 0.0f);
        // 1 / pz: v_rcp_f32 + one Newton step IS the IEEE quotient for every positive float in [2^-126, 2^126) (exhaustive check:
        // tools/ubench_rcp.hip, profiles/r03_ubench_rcp.txt); lanes outside take the division; pz <= 0 is rejected whatever it gives
        float inv = __builtin_amdgcn_rcpf(q.pz);
        inv = fmaf(fmaf(-q.pz, inv, 1.0f), inv, inv);
        if (__builtin_expect(ok && !(q.pz >= 1.17549435e-38f && q.pz < 8.5e37f), 0)) inv = 1.0f / q.pz;
        const float uf = fmaf(cam.fx * q.px, inv, cam.cx);
        const float vf = fmaf(cam.fy * q.py, inv, cam.cy);
        ok = ok && (uf >= -0.5f && uf < wlim && vf >= -0.5f && vf < hlim);
        int ut = (int)floorf(uf + 0.5f), vt = (int)floorf(vf + 0.5f);
        ut = min(ut, cam.W - 1);
        vt = min(vt, cam.H - 1);
        q.ut = ok ? ut : 0;
        q.vt = ok ? vt : 0;
        q.ok = ok;
        return q;
    };
    auto accum = [&](const Samp &q, const v4f nd) {
        if (q.src_ok) acc[29] += 1.0;
        const float dt = nd.w;
        if (!(q.ok && dt > 0.0f)) return;
        const float px = q.px, py = q.py, pz = q.pz;
        const float qx = xray(q.ut) * dt;
        const float qy = yray(q.vt) * dt;
        const float dx = px - qx, dy = py - qy, dz = pz - dt;
        const float dist2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        if (!(dist2 <= md2)) return;
        const float res = fmaf(dx, nd.x, fmaf(dy, nd.y, dz * nd.z));
        const double J[6] = {(double)fmaf(py, nd.z, -(pz * nd.y)), (double)fmaf(pz, nd.x, -(px * nd.z)),
                             (double)fmaf(px, nd.y, -(py * nd.x)), (double)nd.x, (double)nd.y, (double)nd.z};
        const double rr = (double)res;
        // every factor is an f32 value, so every product is EXACT in fp64 (48 significant bits) and fma(a, b, s) rounds the very
        // sum s + a * b the oracle's multiply-then-add rounds: one instruction instead of two, the same bits
        int m = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = a; b < 6; ++b) { acc[m] = fma(J[a], J[b], acc[m]); ++m; }
            acc[21 + a] = fma(J[a], rr, acc[21 + a]);
        }
        acc[27] = fma(rr, rr, acc[27]);
        acc[28] += 1.0;
        if (SCALE) {
            // sigma <- sigma exp(alpha) moves q = R sigma p_hat + t by alpha (q - t):  J_alpha = n . (q - t)
            const double ja = (double)fmaf(nd.x, px - t[0], fmaf(nd.y, py - t[1], nd.z * (pz - t[2])));
#pragma unroll
            for (int a = 0; a < 6; ++a) accs[a] = fma(J[a], ja, accs[a]);
            accs[6] = fma(ja, ja, accs[6]);
            accs[7] = fma(ja, rr, accs[7]);
        }
    };
    // Four samples per trip, software-pipelined over the trips: while trip i is added up, the normal-map gathers of trip i + 1 and
    // the source depths of trip i + 2 are in flight (the trace of round 4 showed a trip of the plain loop -- depths, wait, gathers,
    // wait, sums -- at 2.7-4 us, two exposed memory latencies, against ~0.8 us of arithmetic: with two workgroups per CU there is
    // one other wave per SIMD to fill them).  Loads past a thread's last sample go to element 0 and count as depth 0 (rejected
    // as any invalid depth); the per-thread order of the sums is the sample order, as ever.
    constexpr int NS = 4;
    struct Src { float d[NS]; int uu[NS], vv[NS]; };
    int us, vs;                                              // the next sample this thread fetches
    {
        const int s0 = member * 256 + (int)threadIdx.x;
        vs = s0 / Ws;
        us = s0 - vs * Ws;
    }
    auto fetch_src = [&]() {
        Src x;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const bool in = vs < Hs;
            x.uu[k] = in ? us * stride : 0;
            x.vv[k] = in ? vs * stride : 0;
            const float dv = depth_g[(size_t)x.vv[k] * s_row + ((x.uu[k] & s_mask) * w4 + (x.uu[k] >> s_shift))];
            x.d[k] = in ? dv : 0.0f;
            us += dus;
            vs += dvs;
            if (us >= Ws) { us -= Ws; vs += 1; }
        }
        return x;
    };
    const int vs_first = vs;
    Src src = fetch_src();                                  // trip 0's depths
    Samp q[NS];
    v4f nd[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) q[k] = prep(src.d[k], src.uu[k], src.vv[k]);
#pragma unroll
    for (int k = 0; k < NS; ++k) nd[k] = nmap_g[pm_index(q[k].ut, q[k].vt, w4)];      // trip 0's gathers
    int vs_cur = vs_first, vs_next = vs;                    // row of the first sample of the trip in q / of the trip in src
    src = fetch_src();                                      // trip 1's depths
    // Two trips per turn of the loop, the two sets of registers (q, nd) and (qb, ndb) changing roles: a one-trip body ends with
    // "the next trip becomes the current one", 36 register moves per trip (7 % of its vector instructions) that unrolling by hand
    // makes disappear.  The loop is left after whichever half finds no trip of its own in hand.
    Samp qb[NS];
    v4f ndb[NS];
    for (;;) {
        if (!(vs_cur < Hs)) break;                          // the trip in (q, nd) holds no sample of this thread
#pragma unroll
        for (int k = 0; k < NS; ++k) qb[k] = prep(src.d[k], src.uu[k], src.vv[k]);
#pragma unroll
        for (int k = 0; k < NS; ++k) ndb[k] = nmap_g[pm_index(qb[k].ut, qb[k].vt, w4)];      // next trip's gathers
        vs_cur = vs_next;
        vs_next = vs;
        src = fetch_src();                                  // the depths of the trip after the next
#pragma unroll
        for (int k = 0; k < NS; ++k) accum(q[k], nd[k]);    // this trip's gathers were issued a trip ago
        if (!(vs_cur < Hs)) break;                          // ... and the same with the roles of the two register sets exchanged
#pragma unroll
        for (int k = 0; k < NS; ++k) q[k] = prep(src.d[k], src.uu[k], src.vv[k]);
#pragma unroll
        for (int k = 0; k < NS; ++k) nd[k] = nmap_g[pm_index(q[k].ut, q[k].vt, w4)];
        vs_cur = vs_next;
        vs_next = vs;
        src = fetch_src();
#pragma unroll
        for (int k = 0; k < NS; ++k) accum(qb[k], ndb[k]);
    }
    if (stamp) stamp[0] = wall_clock64();
    // Wave reduction of the 30 sums.  A shuffle tree per sum is 30 x 6 dependent 64-bit shuffles (6.6 us measured, a third of
    // an iteration); instead the lanes split the sums between them while they add: at distance 32 the lower half of the wave
    // keeps sums 0..15 and the upper half 16..31, at distance 16 each quarter keeps 8 of those, ... -- 16 + 8 + 4 + 2 + 1 + 1
    // shuffles.  Every sum is still added over the same tree (lane l with l + 32, then with l + 16, ...; IEEE addition commutes),
    // so the totals are bit for bit those of the shuffle tree; the total of sum c ends in lanes 2c and 2c + 1.
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double v[32];
#pragma unroll
    for (int i = 0; i < 30; ++i) v[i] = acc[i];
    v[30] = 0.0;
    v[31] = 0.0;
#pragma unroll
    for (int half = 16, dist = 32; half >= 1; half >>= 1, dist >>= 1) {
        const bool up = (lane & dist) != 0;
#pragma unroll
        for (int k = 0; k < half; ++k) {
            const double send = up ? v[k] : v[k + half];
            const double keep = up ? v[k + half] : v[k];
            v[k] = keep + __shfl_xor(send, dist);
        }
    }
    v[0] += __shfl_xor(v[0], 1);
    if (!(lane & 1)) sm[wid][lane >> 1] = v[0];
    if (SCALE) {                                         // the 8 sums of the scale column: plain shuffle trees
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double w = accs[k];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) w += __shfl_xor(w, d);
            if (lane == 0) sm[wid][32 + k] = w;
        }
    } else if (lane < 8) {
        sm[wid][32 + lane] = 0.0;
    }
    if (stamp) stamp[1] = wall_clock64();
    __syncthreads();
    if (threadIdx.x < ICP_SLAB) {
        const int i = threadIdx.x;
        sm_out[i] = ((sm[0][i] + sm[1][i]) + sm[2][i]) + sm[3][i];            // (sums 30 and 31 are zero; 32..39 too unless the scale is estimated)
    }
}

__device__ __forceinline__ void icp_accumulate(const Cam &cam, const IcpRun *__restrict__ run, const IcpState *state,
                                               double (*sm)[ICP_SLAB], double *__restrict__ sm_out) {
    float r[9], t[3];
    r[0] = (float)state->T[0]; r[1] = (float)state->T[1]; r[2] = (float)state->T[2];  t[0] = (float)state->T[3];
    r[3] = (float)state->T[4]; r[4] = (float)state->T[5]; r[5] = (float)state->T[6];  t[1] = (float)state->T[7];
    r[6] = (float)state->T[8]; r[7] = (float)state->T[9]; r[8] = (float)state->T[10]; t[2] = (float)state->T[11];
    const float sc = (float)state->scale;                  // the caller's scale, or the running estimate of a Sim(3) run
    if (run->est_scale)
        icp_accumulate_core<true, false>(cam, run->depth_src, run->nmap_tgt, sc, run->mind, run->maxd, run->md2, run->stride, run->Ws, run->Hs,
                                  r, t, (int)blockIdx.x, (int)gridDim.x, run->src_pm, sm, sm_out);
    else
        icp_accumulate_core<false, false>(cam, run->depth_src, run->nmap_tgt, sc, run->mind, run->maxd, run->md2, run->stride, run->Ws, run->Hs,
                                   r, t, (int)blockIdx.x, (int)gridDim.x, run->src_pm, sm, sm_out);
}

// 1/x and 1/sqrt(x) to ~1e-16 relative: hardware seed + two Newton steps (no IEEE division / square-root sequence)
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ double fast_rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-x * y, y, 1.0);
    return fma(0.5 * y, e, y);
}

// Direct path of oracle/tl3d_oracle.c: solve6_direct, same sequence.  Every lane runs it redundantly (no shuffles, all
// indices static): ~250 dependent-ish fp64 operations instead of the eigen-decomposition's ~30 rounds of cross-lane
// exchanges (18 of the 34 us of an iteration).  Returns 1 and x when it applies; 0 -> the caller runs solve6_wave.
__device__ __forceinline__ int solve6_direct(const double *__restrict__ a21, const double *__restrict__ b, double lam,
                                             double eig_rel, double x[6]) {
    double A[6][6], L[6][6], M[6][6], d[6];
    {
        int m = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) { const double e = a21[m++]; A[i][j] = e; A[j][i] = e; }
#pragma unroll
        for (int i = 0; i < 6; ++i) A[i][i] += lam;
    }
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) tr += A[i][i];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double s = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= (L[j][k] * L[j][k]) * d[k];
        ok = ok && (s > 0.0);
        d[j] = s;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= (L[i][k] * L[j][k]) * d[k];
            L[i][j] = t / s;
        }
    }
    if (!ok) return 0;                                       // uniform: every lane holds the same numbers
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) {
            double t = L[i][j];
#pragma unroll
            for (int k = j + 1; k < i; ++k) t += L[i][k] * M[k][j];
            M[i][j] = -t;
        }
    double tinv = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double s = 1.0 / d[j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) s += (M[i][j] * M[i][j]) / d[i];
        tinv += s;
    }
    if (!(tinv > 0.0) || !(1.0 / tinv > eig_rel * tr)) return 0;
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double t = -b[i];
#pragma unroll
        for (int j = 0; j < i; ++j) t += M[i][j] * -b[j];
        y[i] = t / d[i];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double t = y[j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) t += M[i][j] * y[i];
        x[j] = t;
    }
    return 1;
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
    if (solve6_direct(a21, b, lam, eig_rel, x)) return 0;                     // well conditioned: nothing would be truncated
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
            // Rotation (c, s) of the pair.  A Jacobi sweep only needs c^2 + s^2 = 1 to rounding and an angle close to the
            // annihilating one (its error is squared away by the next sweep), so the reciprocals and square roots are the
            // hardware approximations + two Newton steps (~1e-16 relative) instead of IEEE sequences: the dependent chain of
            // a round shrinks from ~150 to ~30 fp64 instructions (the solve was 18 of the 34 us of an iteration).  The
            // eigenpairs the iteration converges to are the same to rounding; the oracle keeps exact divisions.
            double c = 1.0, sn = 0.0;
            if (lane < 6 && !(fabs(apq) < 1e-300)) {
                const double theta = (aqq - app) * fast_rcp(2.0 * apq);
                const double h2 = fma(theta, theta, 1.0);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) * fast_rcp(fabs(theta) + h2 * fast_rsq(h2));
                c = fast_rsq(fma(t, t, 1.0));
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

// The solve for n <= 7 unknowns (Sim(3): pose + log-scale), on ONE lane: oracle/tl3d_oracle.c solve_n, same sequence (direct
// LDL^T path when no eigen-direction can be truncated, else row-cyclic Jacobi with the relative eigenvalue cutoff).
__device__ int solve_n_serial(int n, const double *ap, const double *b, double damping, double eig_rel, double *x) {
    double A[7][7], V[7][7], L[7][7], M[7][7], d[7];
    int m = 0;
    double tr = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) { A[i][j] = A[j][i] = ap[m++]; }
    for (int i = 0; i < n; ++i) tr += A[i][i];
    if (!(tr > 0.0)) return 1;
    const double lam = damping * (tr / (double)n);
    for (int i = 0; i < n; ++i) A[i][i] += lam;
    tr = 0.0;
    for (int i = 0; i < n; ++i) tr += A[i][i];
    {
        int ok = 1;
        for (int j = 0; j < n && ok; ++j) {
            double s = A[j][j];
            for (int k = 0; k < j; ++k) s -= (L[j][k] * L[j][k]) * d[k];
            if (!(s > 0.0)) { ok = 0; break; }
            d[j] = s;
            for (int i = j + 1; i < n; ++i) {
                double t = A[i][j];
                for (int k = 0; k < j; ++k) t -= (L[i][k] * L[j][k]) * d[k];
                L[i][j] = t / s;
            }
        }
        if (ok) {
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < i; ++j) {
                    double t = L[i][j];
                    for (int k = j + 1; k < i; ++k) t += L[i][k] * M[k][j];
                    M[i][j] = -t;
                }
            double tinv = 0.0;
            for (int j = 0; j < n; ++j) {
                double s = 1.0 / d[j];
                for (int i = j + 1; i < n; ++i) s += (M[i][j] * M[i][j]) / d[i];
                tinv += s;
            }
            if (tinv > 0.0 && 1.0 / tinv > eig_rel * tr) {
                double y[7];
                for (int i = 0; i < n; ++i) {
                    double t = -b[i];
                    for (int j = 0; j < i; ++j) t += M[i][j] * -b[j];
                    y[i] = t / d[i];
                }
                for (int j = 0; j < n; ++j) {
                    double t = y[j];
                    for (int i = j + 1; i < n; ++i) t += M[i][j] * y[i];
                    x[j] = t;
                }
                return 0;
            }
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 16; ++sweep) {
        double offmax = 0.0;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < n; ++k)
                if (k != i && fabs(A[i][k]) > offmax) offmax = fabs(A[i][k]);
        if (!(offmax > 1e-15 * tr)) break;
        for (int pp = 0; pp < n - 1; ++pp)
            for (int q = pp + 1; q < n; ++q) {
                const double apq = A[pp][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[pp][pp]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k][pp], akq = A[k][q];
                    A[k][pp] = cs * akp - sn * akq;
                    A[k][q] = sn * akp + cs * akq;
                    const double vkp = V[k][pp], vkq = V[k][q];
                    V[k][pp] = cs * vkp - sn * vkq;
                    V[k][q] = sn * vkp + cs * vkq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[pp][k], aqk = A[q][k];
                    A[pp][k] = cs * apk - sn * aqk;
                    A[q][k] = sn * apk + cs * aqk;
                }
            }
    }
    double lmax = 0.0;
    for (int i = 0; i < n; ++i) if (A[i][i] > lmax) lmax = A[i][i];
    if (!(lmax > 0.0)) return 1;
    for (int i = 0; i < n; ++i) x[i] = 0.0;
    int used = 0;
    for (int e = 0; e < n; ++e) {
        const double l = A[e][e];
        if (!(l > eig_rel * lmax) || !(l > 0.0)) continue;
        double proj = 0.0;
        for (int k = 0; k < n; ++k) proj += V[k][e] * b[k];
        const double coef = -proj / l;
        for (int k = 0; k < n; ++k) x[k] += coef * V[k][e];
        ++used;
    }
    return used == 0;
}

// the 7x7 system from the sums (layout of IcpState::sums): rows 0..5 = the pose block with the scale column appended
__device__ int solve7_serial(const double *sums, double damping, double eig_rel, double x[7]) {
    double ap[28], b[7];
    int m = 0, k = 0;
    for (int i = 0; i < 6; ++i) {
        for (int j = i; j < 6; ++j) ap[k++] = sums[m++];
        ap[k++] = sums[32 + i];
    }
    ap[k++] = sums[38];
    for (int i = 0; i < 6; ++i) b[i] = sums[21 + i];
    b[6] = sums[39];
    return solve_n_serial(7, ap, b, damping, eig_rel, x);
}

// State words another workgroup of the same launch reads next (batched runs): write-through (agent-scope) stores, so that
// they need no L2 write-back before the hand-off; the reader acquires as before.
template <typename V> __device__ __forceinline__ void st_agent(V *p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename V> __device__ __forceinline__ V ld_agent(const V *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ void se3_apply(const double x[6], double *T, float *T_f32 = nullptr) {
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
    for (int i = 0; i < 16; ++i) st_agent(T + i, Tn[i]);
    if (T_f32)
        for (int i = 0; i < 12; ++i) T_f32[i] = (float)Tn[i];
}

// The last workgroup's part: fixed-order sum of the published partials, solve, pose update.
// Returns (thread 0 only) 1 when the run converged, 2 when it failed in this pass, else 0.  T_f32: where thread 0 leaves the
// new pose as the 12 floats a pass uses (LDS; untouched when the pose did not change).
__device__ __forceinline__ int icp_finish(const double *slab, int nblocks, IcpState *state, double damping, double eps, double eig_rel,
                                          int final_pass, double (*part)[ICP_SLAB], double *sums, int est_scale, float *T_f32 = nullptr,
                                          int updates_so_far = -1) {
    const int t = threadIdx.x;
    {   // slab reduction: 8 groups x 32 components, loads batched 8 deep, combined in a fixed order (deterministic).
        // The partials were written by other CUs in THIS launch: agent-scope (sc1) loads, never served from this CU's L1
        const int comp = t & 31, grp = t >> 5;
        double s = 0.0;
        int b = grp;
        for (; b + 56 < nblocks; b += 64) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __hip_atomic_load(slab + (size_t)(b + 8 * k) * ICP_SLAB + comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; b < nblocks; b += 8) s += __hip_atomic_load(slab + (size_t)b * ICP_SLAB + comp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        part[grp][comp] = s;
    }
    if (est_scale && t < 8) {                              // the scale column's 8 sums: member order, one thread each
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += __hip_atomic_load(slab + (size_t)b * ICP_SLAB + 32 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sums[32 + t] = s;
        st_agent(&state->sums[32 + t], s);                 // write-through, as every state word: see the note at sums[t] below
    }
    __syncthreads();
    if (t < 32) {
        double s = part[0][t];
#pragma unroll
        for (int g = 1; g < 8; ++g) s += part[g][t];
        sums[t] = s;
        // WRITE-THROUGH (agent scope), not a plain store: in a batched launch the sums of consecutive passes of one pair are written
        // by different workgroups, i.e. from different XCDs, and a plain store sits in its XCD's L2 until that writes it back --
        // the line of pass n - 1 could reach memory AFTER the line of the final pass (both flushed at the end of the launch, in no
        // order): the statistics the host read (rmse, fitness, correspondences) were then those of the last-but-one pose.  Seen as
        // run-to-run differences of rmse / n_corr of the LAST pairs of a batch with bit-identical poses (tools/bench_icp.py, round 4);
        // the pose words have always been written through.
        st_agent(&state->sums[t], s);
    }
    __syncthreads();
    if (final_pass || t >= 64) return 0;                   // wave 0 solves
    double x[7];
    x[6] = 0.0;
    int fail = 0;
    if (est_scale) {                                       // Sim(3): the 7x7 system, on one lane (wave-uniform branch)
        if (t == 0) fail = (sums[28] < 6.0) ? 1 : solve7_serial(sums, damping, eig_rel, x);
    } else {
        fail = (sums[28] < 6.0) ? 1 : solve6_wave(sums, sums + 21, damping, eig_rel, x);
    }
    if (t != 0) return 0;
    if (fail) {
        st_agent(&state->done, 1);
        st_agent(&state->status, 2);
        return 2;
    }
    se3_apply(x, state->T, T_f32);
    if (est_scale) {
        const double sn = ld_agent(&state->scale) * exp(x[6]);
        st_agent(&state->scale, sn);
        if (T_f32) T_f32[12] = (float)sn;
    }
    st_agent(&state->iters_run, (updates_so_far >= 0 ? updates_so_far : ld_agent(&state->iters_run)) + 1);    // (the load is a round trip)
    double mx = 0.0;
    for (int i = 0; i < (est_scale ? 7 : 6); ++i) mx = fmax(mx, fabs(x[i]));
    if (mx < eps) {
        st_agent(&state->done, 1);
        st_agent(&state->status, 1);
        return 1;
    }
    return 0;
}

// One ICP iteration in one launch.  Hand-off between workgroups inside the launch (MI355X guide, Guideline 16 / the
// split-K "arrival ticket, last arriver combines" form): partial = 8-byte agent-scope (write-through) stores, the storing
// wave drains them, the workgroup's barrier, ONE lane takes an agent-scope ticket; the workgroup whose ticket is the last
// one acquires (one lane, then wait + barrier) and reads every partial with agent-scope loads.  Nothing depends on
// dispatch order or placement; every workgroup takes exactly one ticket per launch, so the grid always drains.
__global__ __launch_bounds__(256) void icp_iter_kernel(Cam cam, const IcpRun *__restrict__ run, IcpState *state, int final_pass,
                                                       double *slab, unsigned *ticket) {
    if (!final_pass && state->done) return;                // set by an EARLIER launch only: uniform over the grid
    __shared__ double sm[8][ICP_SLAB];
    __shared__ double tot[ICP_SLAB];
    __shared__ int s_last;
    icp_accumulate(cam, run, state, sm, tot);
    if (threadIdx.x < ICP_SLAB) {
        __hip_atomic_store(slab + (size_t)blockIdx.x * ICP_SLAB + threadIdx.x, tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the storing wave drains before anyone signals for it
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (tk == gridDim.x - 1u);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    (void)icp_finish(slab, (int)gridDim.x, state, run->damping, run->eps, run->eig_rel, final_pass, sm, tot, final_pass ? 0 : run->est_scale);
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-arm for the next launch
}

// A batch of registrations, each through all its levels and iterations, in ONE launch.  `members` workgroups share a
// pair; after every pass they meet at the pair's barrier in device memory: partial sums as agent-scope stores (drained),
// one arrival ticket per workgroup; the last arriver acquires, sums the partials in member order, solves, updates the
// pose (write-through stores, drained) and publishes the pair's 64-B generation line: words 1..12 = the new pose as the
// twelve floats the next pass uses, then word 0 = pass count + the done / over / failed flags.  The others poll that line with ONE
// wave instruction (13 lanes, one 64-B request) that returns the flags and the pose together: no fence and no load on
// the waiting side.  Publishing and polling are read-modify-write atomics, which execute at the memory side: polls by
// agent-scope (sc1) LOADS were seen to return a word their XCD had cached before the store for seconds (88 of 113
// members of a pair stuck, the 25 on the publisher's XCD gone ahead), whatever the allocation flags.  Short sleeps between
// polls: polls of one line from a hundred CUs at full rate saturate its channel and starve the arrivals.
// Arrival counter and generation line of a pair are different lines; consecutive pairs' lines are > 4 KB apart.
// Waiting needs the pair's other workgroups to be running.  The launch is therefore PERSISTENT: at most as many workgroups as
// the chip holds at once, each taking (pair, member) TICKETS in a loop -- one when it starts, the next when its pair is through
// all its passes.  The tickets handed out at any moment form a window of consecutive (pair, member) slots held by running
// workgroups: every pair that lies wholly inside the window completes, its workgroups come back for tickets, and the pair at
// the window's upper end gets its missing members from them (members <= 32 against >= 256 running workgroups: the grid is
// what the occupancy query says the chip holds, never less than one workgroup per CU).  Nothing
// depends on WHEN the hardware starts a workgroup.  (Until round 3 the grid was one workgroup per slot, tickets taken at
// the start: the last pair of a 128-pair batch of 720p frames then waited for the grid's last two or three workgroups, which
// the dispatcher now and then -- one batch in eight -- started seconds late: the in-kernel time-out of config 4.)
// Every wait is bounded in time; a time-out raises the error word and ends the launch.
constexpr unsigned long long ICP_WAIT_LIMIT_TICKS = 200000000ull;          // 2 s of the 100 MHz wall clock
__device__ __forceinline__ size_t icp_sync_line(int pair, int rows) {        // 16-word line of `pair`: neighbours are rows*64 B (> 4 KB) apart
    return ((size_t)(pair & 63) * (size_t)rows + (size_t)(pair >> 6)) * 16;
}
// SCALE_OK: the launch may hold levels that estimate the source depth's scale (Sim(3)).  The common instantiation (false) carries
// neither the scale column's sums nor the 7-unknown solver: 256 registers instead of 512, i.e. TWO workgroups per CU instead of one
// -- twice the pairs in flight (a waiting member holds its slot).
// SEVERAL SLOTS PER WORKGROUP (round 4; ICP_SLOTS).  The trace of a 512-pair launch (tools/bench_icp.py, TL3D_ICP_TRACE) showed a pass
// of a pair at ~21 us of which a member accumulated for 9 and then sat out 10-12: the stragglers' arrivals, the last arriver's sum +
// solve + pose update (5-7 us on one wave) and the poll.  A workgroup therefore holds several (pair, member) slots and runs whichever
// has its pose, the oldest first: while slot A's pair is being solved, slot B accumulates.  With two slots a pair's pass stretches to
// 33 us (a member's workgroup may be busy with its other slot for a pass when the pose arrives) and the workgroups still idle a
// third of the time; with four they were busy and the launch no faster: what remained was the cost per member and pass, and the
// answer to that fewer members per pair (tl3d_internal.h: ICP_BATCH_SAMPLES_PER_MEMBER).  Two slots it is.
// Nothing blocks on ONE slot: a workgroup with a ready slot runs it, a workgroup with none polls the generation lines of all its
// waiting slots in turn -- so every pair whose tickets are all handed out still completes whatever the other slots of its workgroups
// hold (the argument above, per slot), and the slots it frees take the next tickets.  Slot s only takes a ticket once every workgroup
// of the grid has had the chance of s earlier ones (tickets handed out >= s * gridDim.x): a launch with fewer slots than the chip
// holds workgroups spreads over all CUs as before.
// TAB: the pixel-ray factors of the W columns and H rows sit in (dynamic) LDS, filled once per workgroup (icp_accumulate_core);
// frames too large for that (W + H > ICP_RAY_TAB_MAX) run the instantiation that divides.
constexpr int ICP_RAY_TAB_MAX = 12288;                     // floats: 48 KB
#ifndef TL3D_ICP_SLOTS
#define TL3D_ICP_SLOTS 2
#endif
constexpr int ICP_SLOTS = TL3D_ICP_SLOTS;                  // (pair, member) slots a workgroup works on in turn
enum { ICP_PH_NEED = 0, ICP_PH_INIT = 1, ICP_PH_READY = 2, ICP_PH_WAITING = 3, ICP_PH_EMPTY = 4 };
struct IcpSlot {                                           // LDS; written by thread 0 only, read by all after a barrier
    int phase, ticket, pair, member;
    int lv, it, done, over, failed_run, final_pass;
    unsigned gen, polls;
    unsigned long long wait_t0;
};
template <bool SCALE_OK, bool TAB>
__global__ __launch_bounds__(256, SCALE_OK ? 1 : 2) void icp_batch_kernel(Cam cam, IcpBatchArgs a) {
    __shared__ double sm[8][ICP_SLAB];
    __shared__ double tot[ICP_SLAB];
    __shared__ float sT[ICP_SLOTS][16];                    // per slot: the pose as 12 floats + the source depth's scale
    __shared__ IcpSlot sl[ICP_SLOTS];
    __shared__ int s_flag[4];                              // [1] last arriver, [3] leave the launch (a wait timed out)
    extern __shared__ float ray_tab_lds[];                 // TAB: [W] x factors, then [H] y factors
    const float *ray_tab = ray_tab_lds;
    const int tid = threadIdx.x;
    if (TAB) {
        for (int i = tid; i < cam.W; i += 256) ray_tab_lds[i] = ((float)i - cam.cx) / cam.fx;
        for (int i = tid; i < cam.H; i += 256) ray_tab_lds[cam.W + i] = ((float)i - cam.cy) / cam.fy;
    }
    if (tid == 0) {
        for (int s = 0; s < ICP_SLOTS; ++s) sl[s].phase = ICP_PH_NEED;
        s_flag[3] = 0;
    }
    __syncthreads();
    const size_t gen_base = (size_t)64 * a.sync_rows * 16;
    // what the pair of a pass just completed goes on with, from the generation word (thread 0)
    auto advance = [&](int s, unsigned word) {
        IcpSlot &S = sl[s];
        S.gen += 1u;
        S.done = (int)(word & 1u);
        S.over = (int)((word >> 1) & 1u);
        S.failed_run = (int)((word >> 2) & 1u);
        if (S.final_pass) {
            if (S.over || S.lv + 1 >= a.n_levels) {
                S.phase = ICP_PH_NEED;                     // the pair is through: the slot takes the next ticket
            } else {
                S.lv += 1; S.it = 0; S.done = 0; S.over = 0;
                S.phase = ICP_PH_READY;
            }
        } else {
            S.it += 1;
            S.phase = ICP_PH_READY;
        }
    };
    for (;;) {
        // ---- tickets and initial poses ---------------------------------------------------------------------------------------
        if (tid == 0) {
            for (int s = 0; s < ICP_SLOTS; ++s) {
                if (sl[s].phase != ICP_PH_NEED) continue;
                if (s > 0 && __hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s * gridDim.x) continue;
                const int ticket = (int)__hip_atomic_fetch_add(a.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int pair = ticket / a.members;
                if (pair >= a.n_pairs) { sl[s].phase = ICP_PH_EMPTY; continue; }     // every slot has been handed out
                IcpSlot &S = sl[s];
                S.phase = ICP_PH_INIT; S.ticket = ticket; S.pair = pair; S.member = ticket - pair * a.members;
                S.lv = 0; S.it = 0; S.done = 0; S.over = 0; S.failed_run = 0; S.final_pass = 0; S.gen = 0u; S.polls = 0u; S.wait_t0 = 0ull;
                if (a.stage) { unsigned *stage = a.stage + (size_t)ticket * 4; stage[0] = 1u; stage[1] = (unsigned)blockIdx.x; stage[3] = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20); }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < ICP_SLOTS; ++s)
            if (sl[s].phase == ICP_PH_INIT) {              // the initial pose and scale: uploaded before the launch
                const IcpState *st0 = a.states + sl[s].pair;
                if (tid < 12) sT[s][tid] = (float)st0->T[tid];
                if (tid == 12) sT[s][12] = (float)st0->scale;
            }
        __syncthreads();
        if (tid == 0)
            for (int s = 0; s < ICP_SLOTS; ++s)
                if (sl[s].phase == ICP_PH_INIT) sl[s].phase = ICP_PH_READY;
        __syncthreads();
        int run = -1, n_wait = 0;
#pragma unroll
        for (int s = 0; s < ICP_SLOTS; ++s) {
            const int ph = sl[s].phase;
            if (ph == ICP_PH_READY && (run < 0 || sl[s].ticket < sl[run].ticket)) run = s;      // the oldest pair first
            n_wait += (ph == ICP_PH_WAITING);
        }
        // Every wave has taken its decision from the same slot states before anybody changes them: wave 0's poll below may move a
        // slot to READY, and a wave that read the states only then would run a pass while the others poll (barriers out of step).
        __syncthreads();
        // Nothing in hand: slot 0, which takes a ticket whenever it needs one, is neither ready nor waiting, so it has found every
        // ticket handed out -- and so would the later slots that have not tried yet.
        if (run < 0 && n_wait == 0) return;
        if (run < 0) {
            // ---- no slot has its pose: wave 0 polls the generation lines of the waiting ones ----------------------------------
            // One wave instruction per line (14 lanes, one 64-B request) returns the flags and the pose together.  Polling is a
            // read-modify-write atomic, which executes at the memory side: polls by agent-scope (sc1) LOADS were seen to return a
            // word their XCD had cached before the store for seconds, whatever the allocation flags.
            if (tid < 64) {
                bool got = false;
#pragma unroll
                for (int s = 0; s < ICP_SLOTS; ++s) {
                    if (sl[s].phase != ICP_PH_WAITING) continue;
                    unsigned *genl = a.sync + gen_base + icp_sync_line(sl[s].pair, a.sync_rows);
                    unsigned w = 0;
                    if (tid < 14) w = __hip_atomic_fetch_add(genl + tid, (unsigned)a.zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned word = __builtin_amdgcn_readfirstlane(w);
                    if ((word >> 3) != sl[s].gen) {
                        if (tid >= 1 && tid <= 13) sT[s][tid - 1] = __uint_as_float(w);
                        if (tid == 0) {
                            if (a.dbg && sl[s].pair == 0 && sl[s].gen < 16u) a.dbg[((size_t)sl[s].member * 16 + sl[s].gen) * 8 + 6] = wall_clock64();
                            advance(s, word);
                        }
                        got = true;
                    } else if (tid == 0) {
                        const unsigned polls = ++sl[s].polls;
                        if ((polls & 63u) == 0u &&
                            (wall_clock64() - sl[s].wait_t0 > ICP_WAIT_LIMIT_TICKS || __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                            if (__hip_atomic_exchange(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                                unsigned *arrive = a.sync + icp_sync_line(sl[s].pair, a.sync_rows);
                                a.ctl[4] = (unsigned)sl[s].pair; a.ctl[5] = (unsigned)sl[s].member; a.ctl[6] = sl[s].gen; a.ctl[7] = polls;      // the first time-out, for the host's message
                                a.ctl[8] = __hip_atomic_fetch_add(arrive, (unsigned)a.zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                a.ctl[9] = word;
                                a.ctl[10] = (unsigned)sl[s].lv; a.ctl[11] = (unsigned)sl[s].it;
                            }
                            s_flag[3] = 1;
                        }
                    }
                }
                if (!got) __builtin_amdgcn_s_sleep(16);    // polls of one line from a hundred CUs at full rate saturate its channel and starve the arrivals
            }
            __syncthreads();
            if (s_flag[3]) return;                         // a wait timed out: the error word is set, the host reports it
            continue;
        }
        // ---- one pass of slot `run` ----------------------------------------------------------------------------------------------
        const int pair = sl[run].pair, member = sl[run].member, lv = sl[run].lv, it = sl[run].it;
        const unsigned gen = sl[run].gen;
        const int failed_run = sl[run].failed_run;
        const IcpBatchPair pr = a.pairs[pair];
        IcpState *st = a.states + pair;
        const IcpLevel L = a.lv[lv];
        const int done = sl[run].done;
        const int final_pass = (done || it >= L.iters);
        unsigned *stage = a.stage ? a.stage + (size_t)sl[run].ticket * 4 : nullptr;      // experiments: how far this slot got
        unsigned *arrive = a.sync + icp_sync_line(pair, a.sync_rows);
        unsigned *genl = a.sync + gen_base + icp_sync_line(pair, a.sync_rows);
        double *slab = a.slab + (size_t)pair * a.members * ICP_SLAB;
        unsigned long long *dbg = (a.dbg && pair == 0 && tid == 0) ? a.dbg + (size_t)member * 16 * 8 : nullptr;
#define ICP_STAMP(k_) do { if (dbg && gen < 16u) dbg[gen * 8 + (k_)] = wall_clock64(); } while (0)
        ICP_STAMP(0);
        {
            // the pose is the same in every lane: scalar registers (an LDS read lands in vector registers, twelve of them, live
            // through the whole accumulation)
            auto uni = [](float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); };
            const float *P = sT[run];
            float r[9], t[3];
            r[0] = uni(P[0]); r[1] = uni(P[1]); r[2] = uni(P[2]);  t[0] = uni(P[3]);
            r[3] = uni(P[4]); r[4] = uni(P[5]); r[5] = uni(P[6]);  t[1] = uni(P[7]);
            r[6] = uni(P[8]); r[7] = uni(P[9]); r[8] = uni(P[10]); t[2] = uni(P[11]);
            const float sc_u = uni(P[12]);
            ICP_STAMP(1);
            const int est = SCALE_OK ? (final_pass ? 0 : L.est_scale) : 0;
            if (est)
                icp_accumulate_core<true, TAB>(cam, pr.depth_src, pr.nmap_tgt, sc_u, a.mind, a.maxd, L.md2, L.stride, L.Ws, L.Hs, r, t, member, a.members, pr.src_pm, sm, tot,
                                               (dbg && gen < 16u) ? dbg + gen * 8 + 3 : nullptr, ray_tab);
            else
                icp_accumulate_core<false, TAB>(cam, pr.depth_src, pr.nmap_tgt, sc_u, a.mind, a.maxd, L.md2, L.stride, L.Ws, L.Hs, r, t, member, a.members, pr.src_pm, sm, tot,
                                                (dbg && gen < 16u) ? dbg + gen * 8 + 3 : nullptr, ray_tab);
            ICP_STAMP(2);
            if (stage && tid == 0) stage[0] = 2u + 16u * gen;
            if (tid < ICP_SLAB) {
                __hip_atomic_store(slab + (size_t)member * ICP_SLAB + tid, tot[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (tid == 0) {
                const unsigned arrived = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = (arrived == (unsigned)a.members - 1u);
                if (stage) { stage[0] = 3u + 16u * gen; stage[2] = arrived; }
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                s_flag[1] = last;
                sl[run].final_pass = final_pass;
                if (!last) { sl[run].phase = ICP_PH_WAITING; sl[run].polls = 0u; sl[run].wait_t0 = wall_clock64(); }
                if (dbg && gen < 16u) dbg[gen * 8 + 7] = (unsigned long long)last;
            }
            ICP_STAMP(5);
            __syncthreads();
            if (s_flag[1]) {                               // the last arriver: sum, solve, new pose, publish
                const int fin = icp_finish(slab, a.members, st, L.damping, L.eps, L.eig_rel, final_pass, sm, tot, est, sT[run], it);
                if (tid < 64) {                            // wave 0 publishes
                    // thread 0 may just have written the new pose to sT: make that visible to lanes 1..12 of this wave
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // pose words and the arrival counter of the next pass: one instruction, issued before the state stores have drained
                    // (returning forms: their data comes back only after the memory side has performed them)
                    unsigned old = 0;
                    if (tid >= 1 && tid <= 13) old = __hip_atomic_exchange(genl + tid, __float_as_uint(sT[run][tid - 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (tid == 14) old = __hip_atomic_exchange(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("" ::"v"(old));
                    unsigned word = 0;
                    if (tid == 0) {
                        int fl_done = (done || fin != 0), fl_fail = (failed_run || fin == 2), fl_over = 0;
                        if (final_pass) {                  // what the host did between levels: stop on failure or < 8 correspondences
                            if (lv == a.n_levels - 1 || fl_fail || tot[28] < 8.0) {
                                st_agent(&st->over, 1);
                                fl_over = 1;
                            } else {
                                st_agent(&st->done, 0);
                                st_agent(&st->status, 0);
                                st_agent(&st->iters_run, 0);
                                fl_done = 0;
                            }
                        }
                        word = ((gen + 1u) << 3) | (fl_fail ? 4u : 0u) | (fl_over ? 2u : 0u) | (fl_done ? 1u : 0u);
                    }
                    // state stores written through, pose words and arrival counter in place: only then the flags word says so
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (tid == 0) {
                        (void)__hip_atomic_exchange(genl, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (dbg && gen < 16u) dbg[gen * 8 + 6] = wall_clock64();
                        advance(run, word);
                    }
                }
            }
        }
#undef ICP_STAMP
        __syncthreads();                                   // slot state and the shared scratch are rewritten by the next trip
    }
}

int launch_icp_batch(hipStream_t s, const Cam &cam, const IcpBatchArgs &a) {
    bool scale = false;
    for (int l = 0; l < a.n_levels; ++l) scale = scale || a.lv[l].est_scale != 0;
    const bool tab = cam.W + cam.H <= ICP_RAY_TAB_MAX;
    const size_t lds = tab ? (size_t)(cam.W + cam.H) * sizeof(float) : 0;
    using Kern = void (*)(Cam, IcpBatchArgs);
    const Kern kern = scale ? (tab ? (Kern)icp_batch_kernel<true, true> : (Kern)icp_batch_kernel<true, false>)
                            : (tab ? (Kern)icp_batch_kernel<false, true> : (Kern)icp_batch_kernel<false, false>);
    static int resident[2][2] = {{0, 0}, {0, 0}};          // workgroups the chip holds at once, per instantiation
    static size_t resident_lds[2][2] = {{0, 0}, {0, 0}};   // ... for this much dynamic LDS
    if (!resident[scale][tab] || resident_lds[scale][tab] != lds) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds);
        if (e != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
        resident[scale][tab] = per_cu * cus;
        resident_lds[scale][tab] = lds;
    }
    const unsigned slots = (unsigned)a.n_pairs * (unsigned)a.members;
    const unsigned grid = slots < (unsigned)resident[scale][tab] ? slots : (unsigned)resident[scale][tab];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, cam, a);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// a normal map (phase-major rows) as the row-major [H][W] image the caller of tl3d_download_normals gets
__global__ __launch_bounds__(256) void nmap_rowmajor_kernel(int W, int H, const float4 *__restrict__ nmap, float4 *__restrict__ out) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u < W && v < H) out[(size_t)v * W + u] = nmap[pm_index(u, v, pm_w4(W))];
}

int launch_nmap_rowmajor(hipStream_t s, const Cam &cam, const float4 *nmap, float4 *out) {
    hipLaunchKernelGGL(nmap_rowmajor_kernel, dim3((cam.W + 63) / 64, (cam.H + 3) / 4), dim3(256), 0, s, cam.W, cam.H, nmap, out);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

// radius 0: central differences of the depth image itself; radius >= 1: the window-averaged depth goes to `sdepth` first and the
// normals come from it with a `radius`-pixel step
int launch_normals(hipStream_t s, const Cam &cam, const float *depth, float scale, float mind, float maxd, float jump, int radius, float *sdepth,
                   float4 *nmap) {
    dim3 grid((cam.W + 63) / 64, (cam.H + 3) / 4);
    if (radius >= 1) {
        hipLaunchKernelGGL(smooth_depth_kernel, grid, dim3(256), 0, s, cam, depth, scale, mind, maxd, jump, radius, sdepth);
        TL3D_HIP(hipGetLastError());
        hipLaunchKernelGGL(normals_kernel<true>, grid, dim3(256), 0, s, cam, sdepth, scale, mind, maxd, jump, radius, nmap);
    } else {
        hipLaunchKernelGGL(normals_kernel<false>, grid, dim3(256), 0, s, cam, depth, scale, mind, maxd, jump, 1, nmap);
    }
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

int launch_icp_iteration(hipStream_t s, const Cam &cam, const IcpRun *run, int final_pass, double *slab, IcpState *state, int nblocks,
                         unsigned *ticket) {
    hipLaunchKernelGGL(icp_iter_kernel, dim3(nblocks), dim3(256), 0, s, cam, run, state, final_pass, slab, ticket);
    TL3D_HIP(hipGetLastError());
    return TL3D_OK;
}

}  // namespace tl3d
