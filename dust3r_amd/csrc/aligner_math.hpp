// dust3r_amd -- closed-form forward/backward of the global aligner's objective, shared by the
// HIP kernels (aligner.hip) and by a host loop used only for CPU-side unit tests.
//
// Objective (reference dust3r/cloud_opt/optimizer.py:188-201, base_opt.py:143-195,
// commons.py:62-80; restated in SURVEY.md 8 "Aligner math"):
//   s~_e = exp(P_e[7]) * exp(log(base_scale) - mean_k P_k[7])          (norm_pw_scale)
//   R_e  = R(q_e/|q_e|), q = P_e[0:4] XYZW ; T_e = sign(t) expm1|t|, t = P_e[4:7]
//   M_e  = [ s~_e R_e diag(adapt_e) | s~_e T_e ]
//   X_i[p] = R_i ( exp(d_i[p]) ((u-ppx)/F_i, (v-ppy)/F_i, 1) ) + T_i ,  F_i = exp(f_i/focal_break)
//   L = sum_e sum_p w_i^e[p] |X_ei[p] - M_e pred_i^e[p]| / area_i + (same for j)
// The reference obtains gradients by autograd; here they are written out analytically so that
// forward, backward and the Adam step make ONE pass over the (E, A) tensors.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define D3R_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define D3R_HD inline
#endif

namespace d3r {

struct Quat { float x, y, z, w; };

// roma.unitquat_to_rotmat on the normalised quaternion (XYZW); R row-major [9]
D3R_HD void quat_to_rotmat(const float q[4], float R[9]) {
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, xw = x * w, yw = y * w, zw = z * w;
    R[0] = 1 - 2 * (yy + zz); R[1] = 2 * (xy - zw);     R[2] = 2 * (xz + yw);
    R[3] = 2 * (xy + zw);     R[4] = 1 - 2 * (xx + zz); R[5] = 2 * (yz - xw);
    R[6] = 2 * (xz - yw);     R[7] = 2 * (yz + xw);     R[8] = 1 - 2 * (xx + yy);
}

// dL/dq (un-normalised q) from GR = dL/dR (row-major [9]); double accumulators
D3R_HD void rotmat_grad_to_quat(const float q[4], const double GR[9], double gq[4]) {
    const double n = sqrt((double)q[0] * q[0] + (double)q[1] * q[1] + (double)q[2] * q[2] + (double)q[3] * q[3]);
    const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    double g[4];
    g[0] = 2 * (GR[1] * y + GR[2] * z + GR[3] * y - 2 * GR[4] * x - GR[5] * w + GR[6] * z + GR[7] * w - 2 * GR[8] * x);
    g[1] = 2 * (-2 * GR[0] * y + GR[1] * x + GR[2] * w + GR[3] * x + GR[5] * z - GR[6] * w + GR[7] * z - 2 * GR[8] * y);
    g[2] = 2 * (-2 * GR[0] * z - GR[1] * w + GR[2] * x + GR[3] * w - 2 * GR[4] * z + GR[5] * y + GR[6] * x + GR[7] * y);
    g[3] = 2 * (-GR[1] * z + GR[2] * y + GR[3] * z - GR[5] * x - GR[6] * y + GR[7] * x);
    // back through q -> q/|q| : (I - qn qn^T) g / |q|
    const double dot = g[0] * x + g[1] * y + g[2] * z + g[3] * w;
    gq[0] = (g[0] - dot * x) / n;
    gq[1] = (g[1] - dot * y) / n;
    gq[2] = (g[2] - dot * z) / n;
    gq[3] = (g[3] - dot * w) / n;
}

D3R_HD float signed_expm1f(float t) { return t > 0.f ? expm1f(t) : (t < 0.f ? -expm1f(-t) : 0.f); }
// d/dt [sign(t) expm1|t|] as autograd forms it: sign(t)^2 exp|t|  (0 at t == 0)
D3R_HD double signed_expm1_grad(float t) { return t != 0.f ? exp(fabs((double)t)) : 0.0; }

// Per-pixel residual of one edge side: accumulates loss, dL/dX (g) and -dL/dM partial sums.
// M row-major 3x4. gm[12] accumulates dL/dM = -gd (x) [pr;1].
D3R_HD void residual_accumulate(const float X[3], const float M[12], const float pr[3], float w, bool l2, float& loss, float g[3],
                                float gm[12]) {
    const float y0 = M[0] * pr[0] + M[1] * pr[1] + M[2] * pr[2] + M[3];
    const float y1 = M[4] * pr[0] + M[5] * pr[1] + M[6] * pr[2] + M[7];
    const float y2 = M[8] * pr[0] + M[9] * pr[1] + M[10] * pr[2] + M[11];
    const float r0 = X[0] - y0, r1 = X[1] - y1, r2 = X[2] - y2;
    const float n2 = r0 * r0 + r1 * r1 + r2 * r2;
    float coef;
    if (l2) {
        loss += w * n2;
        coef = 2.f * w;
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        // v_rsq_f32 (1 ulp) instead of IEEE sqrt + divide: ~16 VALU less per pixel in a VALU-bound kernel. Residuals below
        // 1e-15 (n2 < 1e-30, where v_rsq_f32 would see a denormal) count as zero, like the exact zero of the reference.
        const bool nz = n2 > 1e-30f;
        const float inv = nz ? __builtin_amdgcn_rsqf(n2) : 0.f;
        loss += w * (n2 * inv);
        coef = w * inv;
#else
        const float nrm = sqrtf(n2);
        loss += w * nrm;
        coef = n2 > 0.f ? w / nrm : 0.f;
#endif
    }
    const float g0 = coef * r0, g1 = coef * r1, g2 = coef * r2;
    g[0] += g0; g[1] += g1; g[2] += g2;
    gm[0] -= g0 * pr[0]; gm[1] -= g0 * pr[1]; gm[2] -= g0 * pr[2];  gm[3] -= g0;
    gm[4] -= g1 * pr[0]; gm[5] -= g1 * pr[1]; gm[6] -= g1 * pr[2];  gm[7] -= g1;
    gm[8] -= g2 * pr[0]; gm[9] -= g2 * pr[1]; gm[10] -= g2 * pr[2]; gm[11] -= g2;
}

// torch.optim.Adam single-tensor update (no weight decay, no amsgrad), fp32 state:
//   m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
struct AdamCoef { float b1, b2, eps, step_size, bc2_sqrt; };
D3R_HD float adam_update(float p, float g, float& m, float& v, const AdamCoef& c) {
    m = m + (g - m) * (1.f - c.b1);
    v = v * c.b2 + (1.f - c.b2) * g * g;
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    return p - c.step_size * (m / denom);
}

// ---- small-parameter chain rules ----------------------------------------------------------------
// Edge e: from GM = dL/dM_e (3x4, both sides summed) to grads of P_e[0:7] and dL/ds~_e.
D3R_HD void edge_chain(const float P[8], const float R[9], float stilde, const float adapt[3], const double GM[12], double gP[7],
                       double& g_stilde) {
    const float T[3] = {signed_expm1f(P[4]), signed_expm1f(P[5]), signed_expm1f(P[6])};
    double GR[9];
    double gs = 0.0;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            GR[r * 3 + c] = GM[r * 4 + c] * (double)stilde * (double)adapt[c];
            gs += GM[r * 4 + c] * (double)adapt[c] * (double)R[r * 3 + c];
        }
        gs += GM[r * 4 + 3] * (double)T[r];
    }
    double gq[4];
    rotmat_grad_to_quat(P, GR, gq);
    gP[0] = gq[0]; gP[1] = gq[1]; gP[2] = gq[2]; gP[3] = gq[3];
    for (int r = 0; r < 3; ++r) gP[4 + r] = GM[r * 4 + 3] * (double)stilde * signed_expm1_grad(P[4 + r]);
    g_stilde = gs;
}

// Image i: from GRi = sum g (x) cam (3x3), GT = sum g to grads of pose[0:7] and of the focal param.
D3R_HD void image_chain(const float P[7], const float R[9], float focal_break, const double GRi[9], const double GT[3], double gP[7],
                        double& g_focal) {
    double gq[4];
    rotmat_grad_to_quat(P, GRi, gq);
    gP[0] = gq[0]; gP[1] = gq[1]; gP[2] = gq[2]; gP[3] = gq[3];
    for (int r = 0; r < 3; ++r) gP[4 + r] = GT[r] * signed_expm1_grad(P[4 + r]);
    // dX/dF = R (-cam_x/F, -cam_y/F, 0); F = exp(f/fb) -> dL/df = -(1/fb) [ (R^T G)_00 + (R^T G)_11 ]
    double tr = 0.0;
    for (int r = 0; r < 3; ++r) tr += (double)R[r * 3 + 0] * GRi[r * 3 + 0] + (double)R[r * 3 + 1] * GRi[r * 3 + 1];
    g_focal = -tr / (double)focal_break;
}

}  // namespace d3r
