/*
 * tl3d_oracle.c -- CPU ORACLE (plain C), TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; the product path never does.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).
 *
 * What it restates, and how each part is pinned:
 *   orc_backproject / orc_centroid_*   the reference's per-pixel back-projection
 *        (depth_to_reconstruction.py:328-384, depth_enhanced_reconstruction.py:554-613) in fp64, feeding
 *        voxel-centroid accumulators with the semantics of Open3D voxel_down_sample as called at
 *        depth_to_reconstruction.py:406-410.  The back-projection is PINNED through oracle/ref_numpy.py,
 *        which is bit-checked against vectors captured from the reference (tests/golden/); the voxel
 *        centroid is checked against ref_numpy.voxel_centroid_open3d (Open3D itself: parity unpinned,
 *        absent third-party dependency, see ref_numpy.py header).
 *   orc_tsdf_integrate, orc_normals, orc_icp, orc_extract (TSDF mode)
 *        PARITY UNPINNED at the reference: the reference contains no ICP and no TSDF (SURVEY.md
 *        section 0.2, rows a10/a11).  These functions DEFINE the convention (DESIGN.md "Arithmetic
 *        contracts"); they are pinned only by analytic known-answer scenes in tests/.
 *
 * Arithmetic contract shared with the HIP kernels (bit-exact integer grids):
 *   - f32 per-voxel / per-pixel math with explicit fmaf() where written, no other contraction
 *     (-ffp-contract=off here, -ffp-contract=off in the HIP build), IEEE division and sqrt;
 *   - accumulators are integers (order-free); ICP normal equations accumulate fp64 products of f32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QSCALE 32767.0f
#define FRAC_BITS 12
#define FRAC_ONE 4096.0

typedef struct orc_cfg {
    int32_t width, height;
    double fx, fy, cx, cy;
    double min_depth, max_depth;
    int32_t nx, ny, nz;
    double origin[3];
    double voxel_size;
    double sdf_trunc;
} orc_cfg;

typedef struct orc_icp_params {
    int32_t iters, stride;
    double max_dist, damping, eps, eig_rel;
    int32_t estimate_scale, pad;            /* 1: the metric scale of the source depth is a 7th unknown (Sim(3)) */
} orc_icp_params;

typedef struct orc_icp_result {
    double T[16];
    double fitness, rmse;
    int64_t n_corr, n_src;
    int32_t iters_run, status;
    double scale;                           /* scale of the source depth at the end (= scale_s unless estimated) */
} orc_icp_result;

int orc_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* n > 0: number of OpenMP threads the parallel loops use from now on (the host may grant fewer CPUs than it shows) */
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* brick-major record index: 8x8x8 voxels per brick, bricks x fastest; inside a brick eight 4x4x4 sub-bricks
 * (x >> 2 | (y >> 2) << 1 | (z >> 2) << 2) of 64 contiguous records each, x fastest inside a sub-brick */
static inline size_t vox_index(int i, int j, int k, int nbx, int nby) {
    size_t b = ((size_t)(k >> 3) * (size_t)nby + (size_t)(j >> 3)) * (size_t)nbx + (size_t)(i >> 3);
    return (b << 9) + (size_t)(((k & 4) << 6) | ((j & 4) << 5) | ((i & 4) << 4) | ((k & 3) << 4) | ((j & 3) << 2) | (i & 3));
}

size_t orc_vox_index(const orc_cfg *c, int i, int j, int k) { return vox_index(i, j, k, c->nx >> 3, c->ny >> 3); }

/* ------------------------------------------------------------------------------------------------
 * a11  TSDF integration of one frame.  Convention (Curless-Levoy, projective signed distance along z):
 *   voxel centre -> camera -> nearest pixel; sdf = d - z_c; update iff sdf >= -trunc;
 *   tsdf = min(1, sdf/trunc); grid.sum += rint(tsdf*32767); grid.w += 1.
 * grid: int32 [nvox][2] brick-major.
 * ------------------------------------------------------------------------------------------------ */
void orc_tsdf_integrate(const orc_cfg *c, const float *depth, const double R[9], const double t[3],
                        double scale, int32_t *grid) {
    const int W = c->width, H = c->height;
    const int nbx = c->nx >> 3, nby = c->ny >> 3;
    const float fx = (float)c->fx, fy = (float)c->fy, cx = (float)c->cx, cy = (float)c->cy;
    const float ox = (float)c->origin[0], oy = (float)c->origin[1], oz = (float)c->origin[2];
    const float vs = (float)c->voxel_size;
    const float trunc = (float)c->sdf_trunc, inv_trunc = 1.0f / trunc;
    const float mind = (float)c->min_depth, maxd = (float)c->max_depth, sc = (float)scale;
    float r[9], tt[3];
    for (int i = 0; i < 9; ++i) r[i] = (float)R[i];
    for (int i = 0; i < 3; ++i) tt[i] = (float)t[i];
    const float wlim = (float)W - 0.5f, hlim = (float)H - 0.5f;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < c->nz; ++k) {
        const float pz = fmaf((float)k + 0.5f, vs, oz);
        for (int j = 0; j < c->ny; ++j) {
            const float py = fmaf((float)j + 0.5f, vs, oy);
            for (int i = 0; i < c->nx; ++i) {
                const float px = fmaf((float)i + 0.5f, vs, ox);
                const float xc = fmaf(r[0], px, fmaf(r[1], py, fmaf(r[2], pz, tt[0])));
                const float yc = fmaf(r[3], px, fmaf(r[4], py, fmaf(r[5], pz, tt[1])));
                const float zc = fmaf(r[6], px, fmaf(r[7], py, fmaf(r[8], pz, tt[2])));
                if (!(zc > 0.0f)) continue;
                const float inv = 1.0f / zc;
                const float uf = fmaf(fx * xc, inv, cx);
                const float vf = fmaf(fy * yc, inv, cy);
                if (!(uf >= -0.5f && uf < wlim && vf >= -0.5f && vf < hlim)) continue;
                int u = (int)floorf(uf + 0.5f), v = (int)floorf(vf + 0.5f);
                if (u > W - 1) u = W - 1;
                if (v > H - 1) v = H - 1;
                const float d = depth[(size_t)v * W + u] * sc;
                if (!(d > mind && d < maxd)) continue;
                const float sdf = d - zc;
                if (!(sdf >= -trunc)) continue;
                const float tsdf = fminf(1.0f, sdf * inv_trunc);
                const int q = (int)rintf(tsdf * QSCALE);
                int32_t *rec = grid + 2 * vox_index(i, j, k, nbx, nby);
                rec[0] += q;
                rec[1] += 1;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * a4/a5 in C (fp64 intermediates exactly as the reference: factors (u-cx)/fx in fp64, times z, R^T P - R^T t,
 * cast to f32).  Returns 1 and fills p[3] if the pixel survives the mask, else 0.
 * flags bit0: depth*scale and the compares in fp64 (np.float64 scale), else f32.  bit1: no pose.
 * ------------------------------------------------------------------------------------------------ */
static inline int bp_pixel(const orc_cfg *c, const float *depth, int u, int v, const double R[9], const double ct[3],
                           double scale, uint32_t flags, double min_d, double max_d, float p[3]) {
    const float d32 = depth[(size_t)v * c->width + u];
    double z;
    if (flags & 1u) {
        const double d = (double)d32 * scale;
        if (!(d > min_d && d < max_d && isfinite(d))) return 0;
        z = d;
    } else {
        const float d = d32 * (float)scale;
        if (!(d > (float)min_d && d < (float)max_d && isfinite(d))) return 0;
        z = (double)d;
    }
    const double x = (((double)u - c->cx) / c->fx) * z;
    const double y = (((double)v - c->cy) / c->fy) * z;
    if (flags & 2u) {
        p[0] = (float)x; p[1] = (float)y; p[2] = (float)z;
    } else {
        p[0] = (float)(((R[0] * x + R[3] * y) + R[6] * z) - ct[0]);
        p[1] = (float)(((R[1] * x + R[4] * y) + R[7] * z) - ct[1]);
        p[2] = (float)(((R[2] * x + R[5] * y) + R[8] * z) - ct[2]);
    }
    return 1;
}

static inline void rt_t(const double R[9], const double t[3], double ct[3]) {
    for (int i = 0; i < 3; ++i) ct[i] = (R[0 + i] * t[0] + R[3 + i] * t[1]) + R[6 + i] * t[2];   /* (R^T t)_i */
}

int64_t orc_backproject(const orc_cfg *c, const float *depth, const uint8_t *bgr, const double R[9], const double t[3],
                        double scale, uint32_t flags, int subsample, double min_d, double max_d,
                        float *out_xyz, uint8_t *out_rgb) {
    double ct[3] = {0, 0, 0};
    if (!(flags & 2u)) rt_t(R, t, ct);
    int64_t n = 0;
    for (int v = 0; v < c->height; v += subsample)
        for (int u = 0; u < c->width; u += subsample) {
            float p[3];
            if (!bp_pixel(c, depth, u, v, R, ct, scale, flags, min_d, max_d, p)) continue;
            out_xyz[3 * n + 0] = p[0]; out_xyz[3 * n + 1] = p[1]; out_xyz[3 * n + 2] = p[2];
            const uint8_t *px = bgr ? bgr + 3 * ((size_t)v * c->width + u) : NULL;
            out_rgb[3 * n + 0] = px ? px[2] : 0; out_rgb[3 * n + 1] = px ? px[1] : 0; out_rgb[3 * n + 2] = px ? px[0] : 0;
            ++n;
        }
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * a7 fusion half: voxel-centroid accumulators.  index = floor((p - origin)/voxel) in fp64 from the f32
 * point (Open3D VoxelDownSample); in-voxel offset quantised to voxel/4096 (floor).
 * record: u64[4] = { sx | sy<<32, sz | n<<32, sr | sg<<32, sb }.
 * ------------------------------------------------------------------------------------------------ */
static inline int centroid_add(const orc_cfg *c, const float p[3], const uint8_t rgb[3], uint64_t *grid) {
    int idx[3];
    uint64_t q[3];
    const int dims[3] = {c->nx, c->ny, c->nz};
    for (int a = 0; a < 3; ++a) {
        const double rc = ((double)p[a] - c->origin[a]) / c->voxel_size;
        const double fl = floor(rc);
        if (!(fl >= 0.0 && fl < (double)dims[a])) return 0;
        idx[a] = (int)fl;
        int qq = (int)((rc - fl) * FRAC_ONE);
        if (qq > (1 << FRAC_BITS) - 1) qq = (1 << FRAC_BITS) - 1;
        q[a] = (uint64_t)qq;
    }
    uint64_t *rec = grid + 4 * vox_index(idx[0], idx[1], idx[2], c->nx >> 3, c->ny >> 3);
    rec[0] += q[0] | (q[1] << 32);
    rec[1] += q[2] | (1ull << 32);
    rec[2] += (uint64_t)rgb[0] | ((uint64_t)rgb[1] << 32);
    rec[3] += (uint64_t)rgb[2];
    return 1;
}

void orc_centroid_accumulate(const orc_cfg *c, const float *depth, const uint8_t *bgr, const double R[9],
                             const double t[3], double scale, uint32_t flags, int subsample, double min_d,
                             double max_d, uint64_t *grid, uint64_t *n_acc, uint64_t *n_drop) {
    double ct[3] = {0, 0, 0};
    if (!(flags & 2u)) rt_t(R, t, ct);
    uint64_t acc = 0, drop = 0;
    for (int v = 0; v < c->height; v += subsample)
        for (int u = 0; u < c->width; u += subsample) {
            float p[3];
            if (!bp_pixel(c, depth, u, v, R, ct, scale, flags, min_d, max_d, p)) continue;
            uint8_t rgb[3] = {0, 0, 0};
            if (bgr) {
                const uint8_t *px = bgr + 3 * ((size_t)v * c->width + u);
                rgb[0] = px[2]; rgb[1] = px[1]; rgb[2] = px[0];
            }
            if (centroid_add(c, p, rgb, grid)) ++acc; else ++drop;
        }
    if (n_acc) *n_acc += acc;
    if (n_drop) *n_drop += drop;
}

void orc_centroid_points(const orc_cfg *c, const float *xyz, const uint8_t *rgb, int64_t n, uint64_t *grid,
                         uint64_t *n_acc, uint64_t *n_drop) {
    uint64_t acc = 0, drop = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (centroid_add(c, xyz + 3 * i, rgb + 3 * i, grid)) ++acc; else ++drop;
    }
    if (n_acc) *n_acc += acc;
    if (n_drop) *n_drop += drop;
}

/* ------------------------------------------------------------------------------------------------
 * N4  extraction.  Output order: ascending record index (brick-major).
 * ------------------------------------------------------------------------------------------------ */
static inline void rec_coords(size_t idx, int nbx, int nby, int *i, int *j, int *k) {
    const size_t b = idx >> 9;
    const int l = (int)(idx & 511);
    const int bx = (int)(b % (size_t)nbx), by = (int)((b / (size_t)nbx) % (size_t)nby), bz = (int)(b / ((size_t)nbx * (size_t)nby));
    *i = (bx << 3) | ((l >> 4) & 4) | (l & 3);
    *j = (by << 3) | ((l >> 5) & 4) | ((l >> 2) & 3);
    *k = (bz << 3) | ((l >> 6) & 4) | ((l >> 4) & 3);
}

int64_t orc_extract(const orc_cfg *c, int mode, int min_count, int min_weight, double max_abs_tsdf,
                    const int32_t *tsdf, const uint64_t *cen, float *out_xyz, uint8_t *out_rgb, int64_t cap) {
    const int nbx = c->nx >> 3, nby = c->ny >> 3;
    const size_t nvox = (size_t)c->nx * c->ny * c->nz;
    const int dims[3] = {c->nx, c->ny, c->nz};
    int64_t n_out = 0;
    if (min_count < 1) min_count = 1;
    for (size_t idx = 0; idx < nvox; ++idx) {
        int ijk[3];
        rec_coords(idx, nbx, nby, &ijk[0], &ijk[1], &ijk[2]);
        if (mode == 0) {
            if (!cen) return -1;
            const uint64_t *rec = cen + 4 * idx;
            const uint64_t n = rec[1] >> 32;
            if (n < (uint64_t)min_count) continue;
            if (tsdf && min_weight > 0) {
                const int32_t w = tsdf[2 * idx + 1];
                if (w < min_weight) continue;
                const double mean = (double)tsdf[2 * idx] / ((double)w * 32767.0);
                if (!(fabs(mean) <= max_abs_tsdf)) continue;
            }
            if (n_out < cap) {
                const uint64_t s[3] = {rec[0] & 0xffffffffull, rec[0] >> 32, rec[1] & 0xffffffffull};
                for (int a = 0; a < 3; ++a) {
                    const double f = ((double)s[a] + 0.5 * (double)n) / ((double)n * FRAC_ONE);
                    out_xyz[3 * n_out + a] = (float)(c->origin[a] + ((double)ijk[a] + f) * c->voxel_size);
                }
                out_rgb[3 * n_out + 0] = (uint8_t)((rec[2] & 0xffffffffull) / n);
                out_rgb[3 * n_out + 1] = (uint8_t)((rec[2] >> 32) / n);
                out_rgb[3 * n_out + 2] = (uint8_t)((rec[3] & 0xffffffffull) / n);
            }
            ++n_out;
        } else {
            if (!tsdf) return -1;
            const int32_t wa = tsdf[2 * idx + 1];
            const int mw = min_weight < 1 ? 1 : min_weight;
            if (wa < mw) continue;
            const double ta = (double)tsdf[2 * idx] / ((double)wa * 32767.0);
            if (!(fabs(ta) < 0.98)) continue;
            for (int e = 0; e < 3; ++e) {
                int nb[3] = {ijk[0], ijk[1], ijk[2]};
                nb[e] += 1;
                if (nb[e] >= dims[e]) continue;
                const size_t jdx = vox_index(nb[0], nb[1], nb[2], nbx, nby);
                const int32_t wb = tsdf[2 * jdx + 1];
                if (wb < mw) continue;
                const double tb = (double)tsdf[2 * jdx] / ((double)wb * 32767.0);
                if (!(fabs(tb) < 0.98)) continue;
                if (!(ta * tb < 0.0)) continue;
                if (n_out < cap) {
                    const double r0 = fabs(ta), r1 = fabs(tb);
                    const double frac = r0 / (r0 + r1);
                    for (int a = 0; a < 3; ++a) {
                        const double cc = c->origin[a] + ((double)ijk[a] + 0.5) * c->voxel_size;
                        out_xyz[3 * n_out + a] = (float)(a == e ? cc + frac * c->voxel_size : cc);
                    }
                    uint8_t col[3] = {128, 128, 128};
                    if (cen) {
                        const size_t first = (r0 <= r1) ? idx : jdx, second = (r0 <= r1) ? jdx : idx;
                        const size_t cand[2] = {first, second};
                        for (int q = 0; q < 2; ++q) {
                            const uint64_t *rec = cen + 4 * cand[q];
                            const uint64_t n = rec[1] >> 32;
                            if (n > 0) {
                                col[0] = (uint8_t)((rec[2] & 0xffffffffull) / n);
                                col[1] = (uint8_t)((rec[2] >> 32) / n);
                                col[2] = (uint8_t)((rec[3] & 0xffffffffull) / n);
                                break;
                            }
                        }
                    }
                    out_rgb[3 * n_out + 0] = col[0]; out_rgb[3 * n_out + 1] = col[1]; out_rgb[3 * n_out + 2] = col[2];
                }
                ++n_out;
            }
        }
    }
    return n_out;
}

/* ------------------------------------------------------------------------------------------------
 * a10  vertex / normal map (f32).  nmap[v][u] = (nx, ny, nz, d); d = 0 marks an invalid pixel.
 *   V(u,v) = (((float)u - cx)/fx * d, ((float)v - cy)/fy * d, d),  d = depth*scale
 *   n = normalize( (V(u+1,v) - V(u-1,v)) x (V(u,v+1) - V(u,v-1)) ), flipped to face the camera;
 *   a pixel is valid iff it and its 4 neighbours are valid and |d_nb - d| <= depth_jump.
 * ------------------------------------------------------------------------------------------------ */
static inline int load_vertex(const orc_cfg *c, const float *depth, int u, int v, float sc, float mind, float maxd,
                              float fx, float fy, float cx, float cy, float p[3]) {
    const float d = depth[(size_t)v * c->width + u] * sc;
    if (!(d > mind && d < maxd)) return 0;
    p[0] = (((float)u - cx) / fx) * d;
    p[1] = (((float)v - cy) / fy) * d;
    p[2] = d;
    return 1;
}

/* step: the tangent vectors come from the pixels `step` to either side (1: plain central differences) */
static void normals_from(const orc_cfg *c, const float *depth, double scale, double depth_jump, int step, float *nmap);

void orc_normals(const orc_cfg *c, const float *depth, double scale, double depth_jump, float *nmap) {
    normals_from(c, depth, scale, depth_jump, 1, nmap);
}

/* Noise-robust variant: the depth is first averaged over the (2 radius + 1)^2 window of every pixel -- over the pixels that are
 * valid and within depth_jump of the centre pixel (so that no average runs across a depth edge), summed in row-major order in
 * f32; sdepth stays in the units of `depth` (0 = invalid centre) -- and the normals are taken from that map with the tangent
 * vectors `radius` pixels to either side.  1 mm of depth noise turns central differences over one pixel (0.6 mm apart at 1 m,
 * f = 1719) into noise; a 5 x 5 window and a 2-pixel step bring the normal error to a few degrees.  nmap.w = smoothed depth. */
void orc_normals_smooth(const orc_cfg *c, const float *depth, double scale, double depth_jump, int radius, float *sdepth, float *nmap) {
    const int W = c->width, H = c->height;
    const float mind = (float)c->min_depth, maxd = (float)c->max_depth, sc = (float)scale, jump = (float)depth_jump;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            const float d0 = depth[(size_t)v * W + u] * sc;
            float out = 0.0f;
            if (d0 > mind && d0 < maxd) {
                /* mean of the INVERSE depth (linear in the pixel coordinates on a plane, however oblique) over the centre and the
                   pixel PAIRS (u + du, v + dv), (u - du, v - dv) that are both in the image, valid and within the jump of the
                   centre (a symmetric set: the mean of a linear function over it is its centre value) */
                float sum = 1.0f / depth[(size_t)v * W + u];
                int n = 1;
                for (int dv = 0; dv <= radius; ++dv)
                    for (int du = (dv == 0 ? 1 : -radius); du <= radius; ++du) {
                        const int ua = u + du, va = v + dv, ub = u - du, vb = v - dv;
                        if (ua < 0 || ua >= W || va < 0 || va >= H || ub < 0 || ub >= W || vb < 0 || vb >= H) continue;
                        const float ra = depth[(size_t)va * W + ua], rb = depth[(size_t)vb * W + ub];
                        const float da = ra * sc, db = rb * sc;
                        if (!(da > mind && da < maxd) || !(db > mind && db < maxd)) continue;
                        if (!(fabsf(da - d0) <= jump) || !(fabsf(db - d0) <= jump)) continue;
                        sum += 1.0f / ra;
                        sum += 1.0f / rb;
                        n += 2;
                    }
                out = (float)n / sum;
            }
            sdepth[(size_t)v * W + u] = out;
        }
    normals_from(c, sdepth, scale, depth_jump, radius < 1 ? 1 : radius, nmap);
}

static void normals_from(const orc_cfg *c, const float *depth, double scale, double depth_jump, int step, float *nmap) {
    const int W = c->width, H = c->height;
    const float fx = (float)c->fx, fy = (float)c->fy, cx = (float)c->cx, cy = (float)c->cy;
    const float mind = (float)c->min_depth, maxd = (float)c->max_depth, sc = (float)scale, jump = (float)depth_jump;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            float *o = nmap + 4 * ((size_t)v * W + u);
            o[0] = o[1] = o[2] = o[3] = 0.0f;
            if (u < step || v < step || u > W - 1 - step || v > H - 1 - step) continue;
            float p[3], l[3], r[3], up[3], dn[3];
            if (!load_vertex(c, depth, u, v, sc, mind, maxd, fx, fy, cx, cy, p)) continue;
            if (!load_vertex(c, depth, u - step, v, sc, mind, maxd, fx, fy, cx, cy, l)) continue;
            if (!load_vertex(c, depth, u + step, v, sc, mind, maxd, fx, fy, cx, cy, r)) continue;
            if (!load_vertex(c, depth, u, v - step, sc, mind, maxd, fx, fy, cx, cy, up)) continue;
            if (!load_vertex(c, depth, u, v + step, sc, mind, maxd, fx, fy, cx, cy, dn)) continue;
            if (!(fabsf(l[2] - p[2]) <= jump && fabsf(r[2] - p[2]) <= jump && fabsf(up[2] - p[2]) <= jump &&
                  fabsf(dn[2] - p[2]) <= jump)) continue;
            const float ax = r[0] - l[0], ay = r[1] - l[1], az = r[2] - l[2];
            const float bx = dn[0] - up[0], by = dn[1] - up[1], bz = dn[2] - up[2];
            float nx = fmaf(ay, bz, -(az * by));
            float ny = fmaf(az, bx, -(ax * bz));
            float nz = fmaf(ax, by, -(ay * bx));
            const float len2 = fmaf(nx, nx, fmaf(ny, ny, nz * nz));
            if (!(len2 > 1e-30f)) continue;
            const float inv = 1.0f / sqrtf(len2);
            nx *= inv; ny *= inv; nz *= inv;
            const float dotv = fmaf(nx, p[0], fmaf(ny, p[1], nz * p[2]));
            if (dotv > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
            o[0] = nx; o[1] = ny; o[2] = nz; o[3] = p[2];
        }
}

/* ------------------------------------------------------------------------------------------------
 * a10  point-to-plane ICP with projective association.  T maps source-camera to target-camera coordinates.
 * Per iteration: for every source pixel on the stride grid with valid depth
 *    p = R p_s + t (f32, R,t = (float)T); project to the target image (nearest pixel); (n, d_t) = nmap_t there;
 *    q = back-projected target vertex; gate |p-q|^2 <= max_dist^2; r = (p-q).n; J = [p x n, n];
 *    A += J J^T, b += J r, e += r^2  (fp64 sums of fp64 products of the f32 values).
 * Solve (A + damping*trace(A)/6*I) x = -b by eigen-decomposition with a relative eigenvalue cutoff (solve6);
 * T <- [exp(w) | tau] * T.  A final pass evaluates fitness/rmse at T.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { double a[21], b[6], e; int64_t cnt, nsrc; double c[6], cc, bc; } icp_sums;

/* est_scale: also the scale column.  The source point is sigma * p_hat (p_hat from depth * sc, sc = the current scale estimate);
 * a step sigma <- sigma * exp(alpha) moves the transformed point q = R sigma p_hat + t by alpha * (q - t), so
 * J_alpha = n . (q - t);  c[a] = sum J_a J_alpha, cc = sum J_alpha^2, bc = sum J_alpha r. */
static void icp_pass(const orc_cfg *c, const float *depth_s, float sc, const float *nmap_t, const double T[16],
                     int stride, float max_dist, int est_scale, icp_sums *s) {
    const int W = c->width, H = c->height;
    const float fx = (float)c->fx, fy = (float)c->fy, cx = (float)c->cx, cy = (float)c->cy;
    const float mind = (float)c->min_depth, maxd = (float)c->max_depth;
    const float wlim = (float)W - 0.5f, hlim = (float)H - 0.5f, md2 = max_dist * max_dist;
    float r[9], t[3];
    r[0] = (float)T[0]; r[1] = (float)T[1]; r[2] = (float)T[2]; t[0] = (float)T[3];
    r[3] = (float)T[4]; r[4] = (float)T[5]; r[5] = (float)T[6]; t[1] = (float)T[7];
    r[6] = (float)T[8]; r[7] = (float)T[9]; r[8] = (float)T[10]; t[2] = (float)T[11];
    memset(s, 0, sizeof(*s));
    for (int v = 0; v < H; v += stride)
        for (int u = 0; u < W; u += stride) {
            float ps[3];
            if (!load_vertex(c, depth_s, u, v, sc, mind, maxd, fx, fy, cx, cy, ps)) continue;
            s->nsrc++;
            const float px = fmaf(r[0], ps[0], fmaf(r[1], ps[1], fmaf(r[2], ps[2], t[0])));
            const float py = fmaf(r[3], ps[0], fmaf(r[4], ps[1], fmaf(r[5], ps[2], t[1])));
            const float pz = fmaf(r[6], ps[0], fmaf(r[7], ps[1], fmaf(r[8], ps[2], t[2])));
            if (!(pz > 0.0f)) continue;
            const float inv = 1.0f / pz;
            const float uf = fmaf(fx * px, inv, cx);
            const float vf = fmaf(fy * py, inv, cy);
            if (!(uf >= -0.5f && uf < wlim && vf >= -0.5f && vf < hlim)) continue;
            int ut = (int)floorf(uf + 0.5f), vt = (int)floorf(vf + 0.5f);
            if (ut > W - 1) ut = W - 1;
            if (vt > H - 1) vt = H - 1;
            const float *nd = nmap_t + 4 * ((size_t)vt * W + ut);
            const float dt = nd[3];
            if (!(dt > 0.0f)) continue;
            const float qx = (((float)ut - cx) / fx) * dt;
            const float qy = (((float)vt - cy) / fy) * dt;
            const float dx = px - qx, dy = py - qy, dz = pz - dt;
            const float dist2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
            if (!(dist2 <= md2)) continue;
            const float nx = nd[0], ny = nd[1], nz = nd[2];
            const float res = fmaf(dx, nx, fmaf(dy, ny, dz * nz));
            const float j0 = fmaf(py, nz, -(pz * ny));
            const float j1 = fmaf(pz, nx, -(px * nz));
            const float j2 = fmaf(px, ny, -(py * nx));
            const double J[6] = {j0, j1, j2, nx, ny, nz};
            const double rr = (double)res;
            int m = 0;
            for (int a = 0; a < 6; ++a) {
                for (int b = a; b < 6; ++b) s->a[m++] += J[a] * J[b];
                s->b[a] += J[a] * rr;
            }
            s->e += rr * rr;
            s->cnt++;
            if (est_scale) {
                const double ja = (double)fmaf(nx, px - t[0], fmaf(ny, py - t[1], nz * (pz - t[2])));
                for (int a = 0; a < 6; ++a) s->c[a] += J[a] * ja;
                s->cc += ja * ja;
                s->bc += ja * rr;
            }
        }
}

/* Solve (A + damping*trace(A)/6*I) x = -b through the eigen-decomposition of the symmetric 6x6 matrix (cyclic Jacobi,
 * fixed 12 sweeps), dropping directions whose eigenvalue is below eig_rel * (largest eigenvalue): degrees of freedom
 * the geometry does not observe (a plane slides, a cylinder spins, a sphere-on-plane rotates about its axis) are left
 * at the caller's prior instead of drifting (hard part H3).  Returns 0 on success. */
/* Direct path.  When NO eigenvalue can fall under the cutoff, the truncated eigen-solution is simply -A^-1 b, and that
 * can be known without eigenvalues: lambda_min >= 1 / trace(A^-1) and lambda_max <= trace(A), so
 * 1 / trace(A^-1) > eig_rel * trace(A) implies lambda_min > eig_rel * lambda_max.  LDL^T without pivoting (A is symmetric
 * positive definite here, or the test fails), the inverse of the unit-triangular factor gives diag(A^-1).  Returns 1 and x
 * when it applies, 0 otherwise (ill-conditioned or degenerate: the caller runs the eigen-decomposition).  Every loop is
 * in fixed index order; the device runs the same sequence. */
static int solve6_direct(double A[6][6], const double b[6], double eig_rel, double x[6]) {
    double L[6][6], M[6][6], d[6];
    double tr = 0.0;
    for (int i = 0; i < 6; ++i) tr += A[i][i];
    for (int j = 0; j < 6; ++j) {
        double s = A[j][j];
        for (int k = 0; k < j; ++k) s -= (L[j][k] * L[j][k]) * d[k];
        if (!(s > 0.0)) return 0;
        d[j] = s;
        for (int i = j + 1; i < 6; ++i) {
            double t = A[i][j];
            for (int k = 0; k < j; ++k) t -= (L[i][k] * L[j][k]) * d[k];
            L[i][j] = t / s;
        }
    }
    for (int i = 0; i < 6; ++i)                                  /* M = L^-1 (unit lower triangular) */
        for (int j = 0; j < i; ++j) {
            double t = L[i][j];
            for (int k = j + 1; k < i; ++k) t += L[i][k] * M[k][j];
            M[i][j] = -t;
        }
    double tinv = 0.0;                                           /* trace(A^-1) = sum_j sum_{i>=j} M[i][j]^2 / d[i] */
    for (int j = 0; j < 6; ++j) {
        double s = 1.0 / d[j];
        for (int i = j + 1; i < 6; ++i) s += (M[i][j] * M[i][j]) / d[i];
        tinv += s;
    }
    if (!(tinv > 0.0) || !(1.0 / tinv > eig_rel * tr)) return 0;
    double y[6];                                                 /* y = M (-b);  z = y / d;  x = M^T z */
    for (int i = 0; i < 6; ++i) {
        double t = -b[i];
        for (int j = 0; j < i; ++j) t += M[i][j] * -b[j];
        y[i] = t / d[i];
    }
    for (int j = 0; j < 6; ++j) {
        double t = y[j];
        for (int i = j + 1; i < 6; ++i) t += M[i][j] * y[i];
        x[j] = t;
    }
    return 1;
}

static int solve6(const double a21[21], const double b[6], double damping, double eig_rel, double x[6]) {
    double A[6][6], V[6][6];
    int m = 0;
    double tr = 0.0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { A[i][j] = A[j][i] = a21[m++]; }
    for (int i = 0; i < 6; ++i) tr += A[i][i];
    if (!(tr > 0.0)) return 1;
    const double lam = damping * (tr / 6.0);
    for (int i = 0; i < 6; ++i) {
        A[i][i] += lam;
        for (int j = 0; j < 6; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    }
    if (solve6_direct(A, b, eig_rel, x)) return 0;               /* well conditioned: nothing would be truncated */
    /* Cyclic Jacobi in round-robin order: a sweep is 5 rounds of 3 rotations on disjoint index pairs.  The three
     * angles of a round are taken from the same matrix, then all column rotations are applied, then all row rotations
     * (disjoint pairs commute, so this is the sequential sweep up to rounding) -- the order the device's wave executes
     * in parallel.  A sweep starts only while an off-diagonal entry is still > 1e-15 * trace. */
    static const int RR[5][3][2] = {{{0, 5}, {1, 4}, {2, 3}}, {{0, 4}, {3, 5}, {1, 2}}, {{0, 3}, {2, 4}, {1, 5}},
                                    {{0, 2}, {1, 3}, {4, 5}}, {{0, 1}, {2, 5}, {3, 4}}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        double offmax = 0.0;
        for (int i = 0; i < 6; ++i)
            for (int k = 0; k < 6; ++k)
                if (k != i && fabs(A[i][k]) > offmax) offmax = fabs(A[i][k]);
        if (!(offmax > 1e-15 * tr)) break;
        for (int r = 0; r < 5; ++r) {
            double c[3], sn[3];
            for (int u = 0; u < 3; ++u) {
                const int pp = RR[r][u][0], q = RR[r][u][1];
                const double apq = A[pp][q];
                if (fabs(apq) < 1e-300) { c[u] = 1.0; sn[u] = 0.0; continue; }
                const double theta = (A[q][q] - A[pp][pp]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                c[u] = 1.0 / sqrt(t * t + 1.0);
                sn[u] = t * c[u];
            }
            for (int u = 0; u < 3; ++u) {                       /* columns */
                const int pp = RR[r][u][0], q = RR[r][u][1];
                for (int k = 0; k < 6; ++k) {
                    const double akp = A[k][pp], akq = A[k][q];
                    A[k][pp] = c[u] * akp - sn[u] * akq;
                    A[k][q] = sn[u] * akp + c[u] * akq;
                    const double vkp = V[k][pp], vkq = V[k][q];
                    V[k][pp] = c[u] * vkp - sn[u] * vkq;
                    V[k][q] = sn[u] * vkp + c[u] * vkq;
                }
            }
            for (int u = 0; u < 3; ++u) {                       /* rows */
                const int pp = RR[r][u][0], q = RR[r][u][1];
                for (int k = 0; k < 6; ++k) {
                    const double apk = A[pp][k], aqk = A[q][k];
                    A[pp][k] = c[u] * apk - sn[u] * aqk;
                    A[q][k] = sn[u] * apk + c[u] * aqk;
                }
            }
        }
    }
    double lmax = 0.0;
    for (int i = 0; i < 6; ++i) if (A[i][i] > lmax) lmax = A[i][i];
    if (!(lmax > 0.0)) return 1;
    for (int i = 0; i < 6; ++i) x[i] = 0.0;
    int used = 0;
    for (int e = 0; e < 6; ++e) {
        const double l = A[e][e];
        if (!(l > eig_rel * lmax) || !(l > 0.0)) continue;
        double proj = 0.0;
        for (int k = 0; k < 6; ++k) proj += V[k][e] * b[k];
        const double coef = -proj / l;
        for (int k = 0; k < 6; ++k) x[k] += coef * V[k][e];
        ++used;
    }
    return used == 0;
}

/* The same solve for n <= 7 unknowns (Sim(3): the 6 of the pose + log-scale): packed upper triangle a[n (n + 1) / 2] (row by
 * row), damping * trace / n on the diagonal, direct LDL^T path when no eigen-direction can be truncated, else cyclic Jacobi
 * (row-cyclic order, <= 16 sweeps) with the relative eigenvalue cutoff.  The device runs this very sequence on one lane. */
static int solve_n(int n, const double *ap, const double *b, double damping, double eig_rel, double *x) {
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
    {   /* direct path */
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

/* the 7x7 system of the pose + log-scale from the sums: rows 0..5 = the pose block with the scale column appended */
static int solve7(const icp_sums *s, double damping, double eig_rel, double x[7]) {
    double ap[28], b[7];
    int m = 0, k = 0;
    for (int i = 0; i < 6; ++i) {
        for (int j = i; j < 6; ++j) ap[k++] = s->a[m++];
        ap[k++] = s->c[i];
    }
    ap[k++] = s->cc;
    for (int i = 0; i < 6; ++i) b[i] = s->b[i];
    b[6] = s->bc;
    return solve_n(7, ap, b, damping, eig_rel, x);
}

/* T <- [exp(w) | tau] * T */
static void se3_apply(const double x[6], double T[16]) {
    const double wx = x[0], wy = x[1], wz = x[2];
    const double th2 = wx * wx + wy * wy + wz * wz;
    const double th = sqrt(th2);
    double a, bq;
    if (th < 1e-12) { a = 1.0; bq = 0.5; } else { a = sin(th) / th; bq = (1.0 - cos(th)) / th2; }
    const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double K2[9], dR[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j];
            K2[3 * i + j] = s;
        }
    for (int i = 0; i < 9; ++i) dR[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * K[i] + bq * K2[i];
    double Tn[16];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += dR[3 * i + k] * T[4 * k + j];
            Tn[4 * i + j] = s;
        }
        Tn[4 * i + 3] += x[3 + i];
    }
    Tn[12] = 0; Tn[13] = 0; Tn[14] = 0; Tn[15] = 1;
    memcpy(T, Tn, sizeof(Tn));
}

int orc_icp(const orc_cfg *c, const float *depth_s, double scale_s, const float *nmap_t, const double T_init[16],
            const orc_icp_params *prm, orc_icp_result *out) {
    double T[16];
    memcpy(T, T_init, sizeof(T));
    icp_sums s;
    int status = 0, iters_run = 0;
    double scale = scale_s;
    const int est = prm->estimate_scale != 0;
    for (int it = 0; it < prm->iters; ++it) {
        icp_pass(c, depth_s, (float)scale, nmap_t, T, prm->stride, (float)prm->max_dist, est, &s);
        double x[7];
        x[6] = 0.0;
        if (s.cnt < 6 || (est ? solve7(&s, prm->damping, prm->eig_rel, x) : solve6(s.a, s.b, prm->damping, prm->eig_rel, x))) { status = 2; break; }
        se3_apply(x, T);
        if (est) scale *= exp(x[6]);
        ++iters_run;
        double mx = 0;
        for (int i = 0; i < (est ? 7 : 6); ++i) if (fabs(x[i]) > mx) mx = fabs(x[i]);
        if (mx < prm->eps) { status = 1; break; }
    }
    icp_pass(c, depth_s, (float)scale, nmap_t, T, prm->stride, (float)prm->max_dist, 0, &s);
    memcpy(out->T, T, sizeof(T));
    out->n_corr = s.cnt;
    out->n_src = s.nsrc;
    out->fitness = s.nsrc > 0 ? (double)s.cnt / (double)s.nsrc : 0.0;
    out->rmse = s.cnt > 0 ? sqrt(s.e / (double)s.cnt) : 0.0;
    out->iters_run = iters_run;
    out->status = status;
    out->scale = scale;
    return 0;
}

/* exposed for tests of the device reduction: one evaluation pass, sums out as 29 doubles + 2 counts */
void orc_icp_sums(const orc_cfg *c, const float *depth_s, double scale_s, const float *nmap_t, const double T[16],
                  int stride, double max_dist, double out29[29], int64_t *cnt, int64_t *nsrc) {
    icp_sums s;
    icp_pass(c, depth_s, (float)scale_s, nmap_t, T, stride, (float)max_dist, 0, &s);
    memcpy(out29, s.a, 21 * sizeof(double));
    memcpy(out29 + 21, s.b, 6 * sizeof(double));
    out29[27] = s.e;
    out29[28] = 0.0;
    *cnt = s.cnt;
    *nsrc = s.nsrc;
}

/* the same pass with the Sim(3) scale column (est_scale = 1): + c[6] = sum J_a J_alpha, cc = sum J_alpha^2, bc = sum J_alpha r;
 * exposed for tests/test_oracle_second_formulation.py */
void orc_icp_sums_scale(const orc_cfg *c, const float *depth_s, double scale_s, const float *nmap_t, const double T[16],
                        int stride, double max_dist, double out37[37], int64_t *cnt, int64_t *nsrc) {
    icp_sums s;
    icp_pass(c, depth_s, (float)scale_s, nmap_t, T, stride, (float)max_dist, 1, &s);
    memcpy(out37, s.a, 21 * sizeof(double));
    memcpy(out37 + 21, s.b, 6 * sizeof(double));
    out37[27] = s.e;
    out37[28] = 0.0;
    memcpy(out37 + 29, s.c, 6 * sizeof(double));
    out37[35] = s.cc;
    out37[36] = s.bc;
    *cnt = s.cnt;
    *nsrc = s.nsrc;
}
