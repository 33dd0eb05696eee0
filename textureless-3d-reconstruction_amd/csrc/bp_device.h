// bp_device.h -- the per-pixel back-projection shared by the point-list kernel (kernels_backproject.hip)
// and the fused centroid accumulation (kernels_centroid.hip).  One definition so both produce the same bits.
// Reference: depth_to_reconstruction.py:343-384, depth_enhanced_reconstruction.py:572-613.
#pragma once
#include "tl3d_internal.h"

namespace tl3d {

// uint16 millimetres -> float32 metres, bit-identical to numpy's `.astype(float32) / 1000.0` (D2R:90) for every uint16:
// product with fl(1/1000), exact remainder, one correction (checked against exact rational arithmetic for all 65 536
// inputs, and on the device by tests/test_gpu_backproject.py).  3 instructions instead of an IEEE division's ~10.
__device__ __forceinline__ float mm_to_m(uint16_t v) {
    const float f = (float)v;
    const float q = f * 0.001f;
    return fmaf(fmaf(-q, 1000.0f, f), 0.001f, q);
}


__device__ __forceinline__ bool bp_point(const Cam &cam, const BpArgs &a, const PoseD &p, float d32, int u, int v, float out[3]);

__device__ __forceinline__ bool bp_pixel(const Cam &cam, const BpArgs &a, const PoseD &p, const float *__restrict__ depth,
                                         int u, int v, float out[3]) {
    return bp_point(cam, a, p, depth[(size_t)v * cam.W + u], u, v, out);
}

// the same two steps on a depth value already in a register
__device__ __forceinline__ bool bp_valid_value(const BpArgs &a, float d32) {
    if (a.flags & TL3D_F_SCALE_F64) {
        const double d = (double)d32 * a.scale;
        return d > a.min_d && d < a.max_d;
    }
    const float d = d32 * (float)a.scale;
    return d > (float)a.min_d && d < (float)a.max_d;
}

// back-projection of one depth value given the pixel's projection factors xf = (u - cx) / fx, yf = (v - cy) / fy
__device__ __forceinline__ bool bp_point_f(const BpArgs &a, const PoseD &p, float d32, double xf, double yf, float out[3]) {
    double z;
    if (a.flags & TL3D_F_SCALE_F64) {
        const double d = (double)d32 * a.scale;
        if (!(d > a.min_d && d < a.max_d)) return false;
        z = d;
    } else {
        const float d = d32 * (float)a.scale;
        if (!(d > (float)a.min_d && d < (float)a.max_d)) return false;
        z = (double)d;
    }
    const double x = xf * z;
    const double y = yf * z;
    if (a.flags & TL3D_F_NO_POSE) {
        out[0] = (float)x; out[1] = (float)y; out[2] = (float)z;
    } else {
        out[0] = (float)(((p.r[0] * x + p.r[3] * y) + p.r[6] * z) - p.ct[0]);
        out[1] = (float)(((p.r[1] * x + p.r[4] * y) + p.r[7] * z) - p.ct[1]);
        out[2] = (float)(((p.r[2] * x + p.r[5] * y) + p.r[8] * z) - p.ct[2]);
    }
    return true;
}

__device__ __forceinline__ bool bp_point(const Cam &cam, const BpArgs &a, const PoseD &p, float d32, int u, int v, float out[3]) {
    return bp_point_f(a, p, d32, ((double)u - cam.cxd) / cam.fxd, ((double)v - cam.cyd) / cam.fyd, out);
}

__device__ __forceinline__ bool bp_valid_only(const Cam &cam, const BpArgs &a, const float *__restrict__ depth, int u, int v) {
    const float d32 = depth[(size_t)v * cam.W + u];
    if (a.flags & TL3D_F_SCALE_F64) {
        const double d = (double)d32 * a.scale;
        return d > a.min_d && d < a.max_d;
    }
    const float d = d32 * (float)a.scale;
    return d > (float)a.min_d && d < (float)a.max_d;
}

}  // namespace tl3d
