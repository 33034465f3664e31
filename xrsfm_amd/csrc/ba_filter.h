// Post-BA track filter on the GPU.
// Restates Point3dProcessor::FilterPoints3d / FilterPoint3d / Reprojection_Error / UpdateTrackAngle
// (/root/reference/src/geometry/track_processor.cc:19-26, 253-332) and colmap::CalculateTriangulationAngle
// (/root/reference/src/geometry/colmap/base/triangulation.cc:124-147): per observation the reprojection error
// (WorldToImage incl. the pinhole 2f quirk, no depth clamp) and the depth test 1e-3 <= z <= 1e3; per track the
// "all but one observation dropped" rule, the mean error of the kept observations and the maximum pairwise
// triangulation angle with the reference's early exit (pairs in frame-id order).
#pragma once
#include "ba_math.h"

namespace xba {

__global__ void k_cam_centres(const CamRec* __restrict__ cam, int n, double* __restrict__ centre) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    double M[9];
    const double q[4] = {cam[c].q[0], cam[c].q[1], cam[c].q[2], cam[c].q[3]};
    quat_to_mat(q, M);
    const double* t = cam[c].t;
    // -(R^T t)
    centre[3 * c + 0] = -(M[0] * t[0] + M[3] * t[1] + M[6] * t[2]);
    centre[3 * c + 1] = -(M[1] * t[0] + M[4] * t[1] + M[7] * t[2]);
    centre[3 * c + 2] = -(M[2] * t[0] + M[5] * t[1] + M[8] * t[2]);
}

__device__ __forceinline__ double tri_angle(const double* c1, const double* c2, const double* P) {
    double b2 = 0, r1 = 0, r2 = 0;
    for (int k = 0; k < 3; ++k) {
        const double b = c1[k] - c2[k], a1 = P[k] - c1[k], a2 = P[k] - c2[k];
        b2 += b * b; r1 += a1 * a1; r2 += a2 * a2;
    }
    const double den = 2.0 * sqrt(r1 * r2);
    if (den == 0.0) return 0.0;
    const double ang = fabs(acos((r1 + r2 - b2) / den));
    return fmin(ang, 3.14159265358979323846 - ang);
}

// one thread per track; observations of a track are contiguous and ordered by camera (= frame) index
__global__ void k_filter_tracks(const CamRec* __restrict__ cam, const int* __restrict__ cam_model, const double* __restrict__ centre,
                                const double* __restrict__ P, const int* __restrict__ pt_ptr, const int* __restrict__ obs_cam,
                                const double* __restrict__ obs_uv, int n_pts, double max_re, double min_angle,
                                unsigned char* __restrict__ obs_del, unsigned char* __restrict__ trk_out,
                                double* __restrict__ trk_err, double* __restrict__ trk_angle, int* __restrict__ counters) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pts) return;
    const int beg = pt_ptr[j], end = pt_ptr[j + 1], n = end - beg;
    trk_out[j] = 0; trk_err[j] = -1.0; trk_angle[j] = -1.0;
    if (n == 0) return;
    const double Pw[3] = {P[3 * (size_t)j], P[3 * (size_t)j + 1], P[3 * (size_t)j + 2]};
    int ndel = 0;
    double sum = 0.0;
    for (int o = beg; o < end; ++o) {
        const int c = obs_cam[o];
        const CamRec& cr = cam[c];
        const double q[4] = {cr.q[0], cr.q[1], cr.q[2], cr.q[3]};
        double M[9];
        quat_to_mat(q, M);
        const double X = M[0] * Pw[0] + M[1] * Pw[1] + M[2] * Pw[2] + cr.t[0];
        const double Y = M[3] * Pw[0] + M[4] * Pw[1] + M[5] * Pw[2] + cr.t[1];
        const double Z = M[6] * Pw[0] + M[7] * Pw[1] + M[8] * Pw[2] + cr.t[2];
        // WorldToImage on hnormalized() without the BA functor's depth clamp
        const double xn = X / Z, yn = Y / Z, r2 = xn * xn + yn * yn;
        const double* k = cr.intr;
        double fx, fy, cx, cy, du, dv;
        switch (cam_model[c]) {
        case 0: fx = k[0]; fy = k[0]; cx = k[1]; cy = k[2]; du = xn; dv = yn; break;
        case 1: fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3]; du = xn; dv = yn; break;
        case 2: fx = k[0]; fy = k[0]; cx = k[1]; cy = k[2]; du = xn * (k[3] * r2); dv = yn * (k[3] * r2); break;
        case 3: fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3]; du = xn * (k[4] * r2); dv = yn * (k[4] * r2); break;
        default: {
            fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3];
            const double rad = k[4] * r2 + k[5] * r2 * r2, xy = xn * yn;
            du = xn * rad + 2.0 * k[6] * xy + k[7] * (r2 + 2.0 * xn * xn);
            dv = yn * rad + 2.0 * k[7] * xy + k[6] * (r2 + 2.0 * yn * yn);
        } }
        const double e0 = fx * (xn + du) + cx - obs_uv[2 * (size_t)o], e1 = fy * (yn + dv) + cy - obs_uv[2 * (size_t)o + 1];
        const double re = sqrt(e0 * e0 + e1 * e1);
        const bool del = re > max_re || Z < 1e-3 || Z > 1e3;
        obs_del[o] = del ? 1 : 0;
        if (del) ++ndel; else sum += re;
    }
    if (ndel >= n - 1) {                       // obs_to_delete.size() >= observations_.size() - 1
        trk_out[j] = 1;
        atomicAdd(&counters[0], n);
        return;
    }
    atomicAdd(&counters[0], ndel);
    trk_err[j] = sum / (double)(n - ndel);
    double best = 0.0;
    bool done = false;
    for (int a = beg; a < end && !done; ++a) {
        if (obs_del[a]) continue;
        for (int b = a + 1; b < end; ++b) {
            if (obs_del[b]) continue;
            const double ang = tri_angle(centre + 3 * (size_t)obs_cam[a], centre + 3 * (size_t)obs_cam[b], Pw);
            if (ang > best) {
                best = ang;
                if (best > min_angle) { done = true; break; }
            }
        }
    }
    trk_angle[j] = best;
    if (best < min_angle) { trk_out[j] = 2; atomicAdd(&counters[1], 1); }
}

}  // namespace xba
