// dust3r_amd -- scene bootstrap: the one-shot initialisation of the global aligner on the GPU (gfx950, HBM-bound).
//
// What the reference does here (dust3r/cloud_opt/init_im_poses.py:67-287, post_process.py:12-60, pair_viewer.py:30-76) is a
// host-driven chain of small torch / roma / OpenCV calls over full-resolution pointmaps: one weighted similarity Procrustes per
// spanning-tree edge (each waiting for the previous one, because the target cloud of an edge is the transformed cloud of its
// parent), one more per graph edge for the pairwise poses, a 10-iteration Weiszfeld focal fit per image and a RANSAC-PnP per image
// that never led a tree edge. For 100 views / 600 edges that is ~700 passes over 196 608-point clouds issued one by one.
//
// MI355X design: nothing in that chain needs the big data sequentially.
//   * Weighted Umeyama is equivariant under a similarity of the TARGET cloud: reg(x -> G y) = G o reg(x -> y). Every world cloud of
//     the reference's walk is G_k applied to ONE raw pairwise pointmap, so every registration of the whole initialisation -- tree
//     edges and pairwise poses alike -- is a registration between two RAW maps followed by a 4x4 composition on the host. All of them
//     are independent: `similarity_moments_kernel` accumulates the 17 weighted moments of every job in ONE launch
//     (28 bytes per point pair read exactly once: ~3.8 GB for the 100-view case, < 1 ms of HBM time), the 3x3 SVDs stay on the host.
//   * `weiszfeld_focal_kernel`: one 1024-thread workgroup per image keeps its 2.4 MB map in L2 for the 11 passes.
//   * `anchor_depth_kernel`: log-depth of every image from its anchor map and one 1x4 row (no world cloud is ever materialised).
//   * PnP: hypotheses and 12x12 / 6x6 solves on the host (a few hundred flops); scoring every hypothesis against every masked point
//     (`pnp_score_kernel`) and the Gauss-Newton sums of the polish (`pnp_sums_kernel`) on the GPU, batched over all images that need
//     a pose.
// Sums are accumulated in fp32 per lane over at most 8 points, then in fp64 across lanes / workgroups in a fixed order (deterministic).
#include "../../include/dust3r_hip.h"
#include "kernels.hpp"

namespace d3r {

D3R_DEV double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Block-wide fixed-order sum of NV doubles per thread; result valid in thread 0 (returned in v[]). red: NW * NV doubles of LDS.
template <int NV, int NT>
D3R_DEV void block_sum_f64(double (&v)[NV], double* red) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum_f64(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave * NV + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += red[w * NV + k];
            v[k] = s;
        }
    }
    __syncthreads();
}

// ---- row means (edge confidence scores) --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_mean_kernel(const float* x, int cols, int ld, float* out) {
    __shared__ double red[4];
    const float* r = x + (size_t)blockIdx.x * ld;
    double acc[1] = {0.0};
    for (int c = threadIdx.x * 4; c < cols; c += 256 * 4) {
        if (c + 4 <= cols) {
            const float4 v = *reinterpret_cast<const float4*>(r + c);
            acc[0] += (double)((v.x + v.y) + (v.z + v.w));
        } else {
            for (int k = c; k < cols; ++k) acc[0] += (double)r[k];
        }
    }
    block_sum_f64<1, 256>(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(acc[0] / cols);
}

// ---- weighted similarity-registration moments ----------------------------------------------------------------------------
// job j: source cloud src[j] (npix[j] x 3, interleaved xyz), target cloud tgt[j], weights w[j] (npix[j]).
// out[j][17] = { W, Sx(3), Sy(3), Sxy(9, row-major x_a y_b), Sxx }   with S. = sum_p w_p (.)
static constexpr int MOM = 17;
static constexpr int MOM_CHUNK = 2048;   // points per workgroup: 256 threads x 8

__global__ __launch_bounds__(256) void similarity_moments_kernel(const float* const* src, const float* const* tgt, const float* const* wgt,
                                                                 const int* npix, int nchunk, double* partial) {
    __shared__ double red[4 * MOM];
    const int job = blockIdx.y, chunk = blockIdx.x;
    const int n = npix[job];
    const float* xs = src[job];
    const float* ys = tgt[job];
    const float* ws = wgt[job];
    double acc[MOM];
#pragma unroll
    for (int k = 0; k < MOM; ++k) acc[k] = 0.0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int p0 = chunk * MOM_CHUNK + half * 1024 + threadIdx.x * 4;   // 4 consecutive points = 3 float4 per cloud
        if (p0 >= n) continue;
        float x[4][3], y[4][3], w[4];
        if (p0 + 4 <= n) {
            const float4* xp = reinterpret_cast<const float4*>(xs + (size_t)p0 * 3);
            const float4* yp = reinterpret_cast<const float4*>(ys + (size_t)p0 * 3);
            const float4 a0 = xp[0], a1 = xp[1], a2 = xp[2], b0 = yp[0], b1 = yp[1], b2 = yp[2];
            const float4 wv = *reinterpret_cast<const float4*>(ws + p0);
            const float xa[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
            const float ya[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
            const float wa[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                w[q] = wa[q];
#pragma unroll
                for (int c = 0; c < 3; ++c) { x[q][c] = xa[q * 3 + c]; y[q][c] = ya[q * 3 + c]; }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = p0 + q < n;
                w[q] = ok ? ws[p0 + q] : 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) { x[q][c] = ok ? xs[(size_t)(p0 + q) * 3 + c] : 0.f; y[q][c] = ok ? ys[(size_t)(p0 + q) * 3 + c] : 0.f; }
            }
        }
        float f[MOM];
#pragma unroll
        for (int k = 0; k < MOM; ++k) f[k] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float wq = w[q];
            f[0] += wq;
            const float wx[3] = {wq * x[q][0], wq * x[q][1], wq * x[q][2]};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f[1 + c] += wx[c];
                f[4 + c] += wq * y[q][c];
#pragma unroll
                for (int d = 0; d < 3; ++d) f[7 + c * 3 + d] += wx[c] * y[q][d];
                f[16] += wx[c] * x[q][c];
            }
        }
#pragma unroll
        for (int k = 0; k < MOM; ++k) acc[k] += (double)f[k];
    }
    block_sum_f64<MOM, 256>(acc, red);
    if (threadIdx.x == 0) {
        double* o = partial + ((size_t)job * nchunk + chunk) * MOM;
#pragma unroll
        for (int k = 0; k < MOM; ++k) o[k] = acc[k];
    }
}

__global__ __launch_bounds__(64) void moments_reduce_kernel(const double* partial, const int* npix, int nchunk, int nv, int chunk_pts, double* out) {
    const int job = blockIdx.x, k = threadIdx.x;
    if (k >= nv) return;
    const int used = (npix[job] + chunk_pts - 1) / chunk_pts;
    double s = 0.0;
    for (int c = 0; c < used; ++c) s += partial[((size_t)job * nchunk + c) * nv + k];
    out[(size_t)job * nv + k] = s;
}

// ---- Weiszfeld focal (post_process.py:40-56, focal_mode='weiszfeld'; principal point = image centre) ---------------------------
__global__ __launch_bounds__(1024) void weiszfeld_focal_kernel(const float* const* maps, const int* hs, const int* ws, int iters, float* focal_out) {
    __shared__ double red[16 * 2];
    __shared__ float f_sh;
    const int job = blockIdx.x;
    const float* pts = maps[job];
    const int H = hs[job], W = ws[job], n = H * W;
    const float ppx = W * 0.5f, ppy = H * 0.5f;
    float focal = 0.f;
    for (int it = 0; it <= iters; ++it) {          // it == 0: the closed-form l2 initialisation (all weights 1)
        double acc[2] = {0.0, 0.0};
        for (int p = threadIdx.x; p < n; p += 1024) {
            const float X = pts[(size_t)p * 3], Y = pts[(size_t)p * 3 + 1], Z = pts[(size_t)p * 3 + 2];
            const int v = p / W, u = p - v * W;
            const float px = (float)u - ppx, py = (float)v - ppy;
            float a = X / Z, b = Y / Z;            // nan_to_num(posinf=0, neginf=0): non-finite -> 0
            a = (a - a == 0.f) ? a : 0.f;
            b = (b - b == 0.f) ? b : 0.f;
            const float dpx = a * px + b * py, dxx = a * a + b * b;
            float wgt = 1.f;
            if (it > 0) {
                const float ex = px - focal * a, ey = py - focal * b;
                wgt = 1.f / fmaxf(sqrtf(ex * ex + ey * ey), 1e-8f);
            }
            acc[0] += (double)(wgt * dpx);
            acc[1] += (double)(wgt * dxx);
        }
        block_sum_f64<2, 1024>(acc, red);
        if (threadIdx.x == 0) f_sh = (float)(acc[0] / acc[1]);
        __syncthreads();
        focal = f_sh;
        __syncthreads();
    }
    if (threadIdx.x == 0) focal_out[job] = fmaxf(focal, 0.f);   // clip(min = min_focal * base = 0, max = inf)
}

// ---- depth of every image from its anchor map: z = row . (p, 1); take_log: out = z > 0 ? log z : 0 (log().nan_to_num(neginf=0)) ---
__global__ __launch_bounds__(256) void anchor_depth_kernel(const float* const* maps, const float* rows, const int* npix, int max_area, int take_log, float* out) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= max_area) return;
    float r = 0.f;
    if (p < npix[img]) {
        const float* q = maps[img] + (size_t)p * 3;
        const float* m = rows + img * 4;
        const float z = m[0] * q[0] + m[1] * q[1] + m[2] * q[2] + m[3];
        r = !take_log ? z : (z > 0.f ? logf(z) : 0.f);   // z <= 0 or NaN: log gives -inf / NaN, which nan_to_num(neginf=0) maps to 0
    }
    out[(size_t)img * max_area + p] = r;
}

// ---- PnP support -----------------------------------------------------------------------------------------------------------
// Image job j: world points X = G_j (3x4) applied to the raw map maps[j] (H x W x 3), pixel grid (u, v), mask = conf[j] > thr,
// intrinsics (f, ppx, ppy). Poses are world -> camera (R | t), 12 floats row-major [R0 t0; R1 t1; R2 t2].
struct PnpJob {
    const float* map; const float* conf; float G[12]; float f, ppx, ppy, thr; int H, W;
};

D3R_DEV bool pnp_point(const PnpJob& j, int p, float (&X)[3], float& u, float& v) {
    if (!(j.conf[p] > j.thr)) return false;
    const float* q = j.map + (size_t)p * 3;
    const float x = q[0], y = q[1], z = q[2];
    X[0] = j.G[0] * x + j.G[1] * y + j.G[2] * z + j.G[3];
    X[1] = j.G[4] * x + j.G[5] * y + j.G[6] * z + j.G[7];
    X[2] = j.G[8] * x + j.G[9] * y + j.G[10] * z + j.G[11];
    const int vv = p / j.W;
    u = (float)(p - vv * j.W);
    v = (float)vv;
    return true;
}

// inlier counts of NH hypotheses per job: counts[job][h]. One pass over the map evaluates every hypothesis.
static constexpr int PNP_MAXH = 32;
__global__ __launch_bounds__(256) void pnp_score_kernel(const PnpJob* jobs, const float* hyp, int nh, float thr2, int* counts) {
    __shared__ float hs[PNP_MAXH * 12];
    __shared__ int cnt[PNP_MAXH];
    const PnpJob j = jobs[blockIdx.y];
    for (int k = threadIdx.x; k < PNP_MAXH * 12; k += 256) hs[k] = k < nh * 12 ? hyp[(size_t)blockIdx.y * PNP_MAXH * 12 + k] : 0.f;   // unused slots: zc = 0, never an inlier
    if (threadIdx.x < PNP_MAXH) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int n = j.H * j.W;
    int local[PNP_MAXH];
#pragma unroll
    for (int h = 0; h < PNP_MAXH; ++h) local[h] = 0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        float X[3], u, v;
        if (!pnp_point(j, p, X, u, v)) continue;
#pragma unroll
        for (int h = 0; h < PNP_MAXH; ++h) {
            const float* P = hs + h * 12;
            const float xc = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3];
            const float yc = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
            const float zc = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11];
            const float iz = 1.f / zc;
            const float du = j.f * xc * iz + j.ppx - u, dv = j.f * yc * iz + j.ppy - v;
            local[h] += (zc > 0.f && du * du + dv * dv < thr2) ? 1 : 0;
        }
    }
#pragma unroll
    for (int h = 0; h < PNP_MAXH; ++h) {
        int s = local[h];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&cnt[h], s);      // integer adds: order independent
    }
    __syncthreads();
    if (threadIdx.x < nh && cnt[threadIdx.x]) atomicAdd(&counts[blockIdx.y * PNP_MAXH + threadIdx.x], cnt[threadIdx.x]);
}

// Gauss-Newton sums over the inliers of pose[job] (reprojection error^2 < thr2, in front of the camera), unknowns = (rotation increment
// w about the camera origin, translation increment), plain least squares over the consensus set, like the refinement behind cv2.solvePnPRansac (round 6; rounds 2-5 weighted it Huber-wise, delta = 1 px: a different
// estimator, a few 1e-3 rad away from the reference's on noisy pointmaps): J^T W J (21, upper triangle row-major), J^T W r (6),
// cost, inlier count.
// (A DLT refit of the consensus set from fp32 moments was tried first: the 12x12 normal matrix is too ill-conditioned for it.)
static constexpr int PNP_NV = 29;
__global__ __launch_bounds__(256) void pnp_sums_kernel(const PnpJob* jobs, const float* pose, float thr2, int nchunk, double* partial) {
    __shared__ double red[4 * PNP_NV];
    const PnpJob j = jobs[blockIdx.y];
    const float* P = pose + blockIdx.y * 12;
    const int n = j.H * j.W;
    double acc[PNP_NV];
#pragma unroll
    for (int k = 0; k < PNP_NV; ++k) acc[k] = 0.0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        float X[3], u, v;
        if (!pnp_point(j, p, X, u, v)) continue;
        const float xc = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3];
        const float yc = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7];
        const float zc = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11];
        const float iz = 1.f / zc;
        const float ru = j.f * xc * iz + j.ppx - u, rv = j.f * yc * iz + j.ppy - v;
        if (!(zc > 0.f && ru * ru + rv * rv < thr2)) continue;
        {
            // d(proj)/d(Xc) and d(Xc)/d(w, t) with R <- exp([w]x) R:  dXc = -[Xc - t]x w + t'
            const float fx = j.f * iz, a0 = -j.f * xc * iz * iz, a1 = -j.f * yc * iz * iz;
            const float Xr[3] = {xc - P[3], yc - P[7], zc - P[11]};
            // rows of J (2 x 6): Ju = (fx, 0, a0) . [ -[Xr]x | I ],  Jv = (0, fx, a1) . [ -[Xr]x | I ];  -[Xr]x = [[0, z, -y], [-z, 0, x], [y, -x, 0]]
            const float S[3][3] = {{0.f, Xr[2], -Xr[1]}, {-Xr[2], 0.f, Xr[0]}, {Xr[1], -Xr[0], 0.f}};
            float Ju[6], Jv[6];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Ju[c] = fx * S[0][c] + a0 * S[2][c];
                Jv[c] = fx * S[1][c] + a1 * S[2][c];
            }
            Ju[3] = fx; Ju[4] = 0.f; Ju[5] = a0;
            Jv[3] = 0.f; Jv[4] = fx; Jv[5] = a1;
            // unit weight: the reference refines the RANSAC consensus set in plain least squares (init_im_poses.py:272-275 -> cv2.solvePnPRansac); the 5-pixel
            // band already bounds what a stray point can pull
            const float hw = 1.f;
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) { acc[k] += (double)(hw * (Ju[a] * Ju[b] + Jv[a] * Jv[b])); ++k; }
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] += (double)(hw * (Ju[a] * ru + Jv[a] * rv));
            acc[27] += (double)(ru * ru + rv * rv);
            acc[28] += 1.0;
        }
    }
    block_sum_f64<PNP_NV, 256>(acc, red);
    if (threadIdx.x == 0) {
        double* o = partial + ((size_t)blockIdx.y * nchunk + blockIdx.x) * PNP_NV;
#pragma unroll
        for (int k = 0; k < PNP_NV; ++k) o[k] = acc[k];
    }
}

__global__ __launch_bounds__(64) void pnp_reduce_kernel(const double* partial, int nchunk, double* out) {
    const int job = blockIdx.x, k = threadIdx.x;
    if (k >= PNP_NV) return;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += partial[((size_t)job * nchunk + c) * PNP_NV + k];
    out[(size_t)job * PNP_NV + k] = s;
}

}  // namespace d3r

using namespace d3r;
static inline int rc_of(hipError_t e) { return e == hipSuccess ? D3R_OK : 1000 + (int)e; }

extern "C" int d3r_row_means(const float* x, int rows, int cols, int ld, float* out, void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0 || ld < cols || (ld & 3)) return D3R_ERR_INVALID;
    hipLaunchKernelGGL(row_mean_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, cols, ld, out);
    return rc_of(hipGetLastError());
}

extern "C" size_t d3r_similarity_moments_workspace(int n_jobs, int max_points) {
    const int nchunk = (max_points + MOM_CHUNK - 1) / MOM_CHUNK;
    return (size_t)n_jobs * nchunk * MOM * sizeof(double);
}

extern "C" int d3r_similarity_moments(int n_jobs, const void* src_ptrs, const void* tgt_ptrs, const void* wgt_ptrs, const int* npix, int max_points,
                                      void* workspace, double* out, void* stream) {
    if (n_jobs <= 0 || !src_ptrs || !tgt_ptrs || !wgt_ptrs || !npix || !workspace || !out || max_points <= 0) return D3R_ERR_INVALID;
    const int nchunk = (max_points + MOM_CHUNK - 1) / MOM_CHUNK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(similarity_moments_kernel, dim3(nchunk, n_jobs), dim3(256), 0, st, (const float* const*)src_ptrs, (const float* const*)tgt_ptrs,
                       (const float* const*)wgt_ptrs, npix, nchunk, (double*)workspace);
    hipLaunchKernelGGL(moments_reduce_kernel, dim3(n_jobs), dim3(64), 0, st, (const double*)workspace, npix, nchunk, MOM, MOM_CHUNK, out);
    return rc_of(hipGetLastError());
}

extern "C" int d3r_weiszfeld_focals(int n_jobs, const void* map_ptrs, const int* heights, const int* widths, int iterations, float* focals, void* stream) {
    if (n_jobs <= 0 || !map_ptrs || !heights || !widths || !focals || iterations < 0) return D3R_ERR_INVALID;
    hipLaunchKernelGGL(weiszfeld_focal_kernel, dim3(n_jobs), dim3(1024), 0, (hipStream_t)stream, (const float* const*)map_ptrs, heights, widths, iterations, focals);
    return rc_of(hipGetLastError());
}

extern "C" int d3r_anchor_depth(int n_imgs, const void* map_ptrs, const float* rows, const int* npix, int max_area, int take_log, float* out, void* stream) {
    if (n_imgs <= 0 || !map_ptrs || !rows || !npix || !out || max_area <= 0) return D3R_ERR_INVALID;
    hipLaunchKernelGGL(anchor_depth_kernel, dim3((max_area + 255) / 256, n_imgs), dim3(256), 0, (hipStream_t)stream, (const float* const*)map_ptrs, rows, npix,
                       max_area, take_log, out);
    return rc_of(hipGetLastError());
}

static constexpr int PNP_GRID = 96;   // workgroups per image for the PnP passes (grid-stride over the map)

extern "C" int d3r_pnp_job_bytes(void) { return (int)sizeof(PnpJob); }
extern "C" int d3r_pnp_max_hypotheses(void) { return PNP_MAXH; }
extern "C" int d3r_pnp_sum_count(void) { return PNP_NV; }
extern "C" size_t d3r_pnp_workspace(int n_jobs) { return (size_t)n_jobs * PNP_GRID * PNP_NV * sizeof(double); }

extern "C" int d3r_pnp_score(int n_jobs, const void* jobs, const float* hypotheses, int n_hyp, float reproj_err, int* counts, void* stream) {
    if (n_jobs <= 0 || !jobs || !hypotheses || !counts || n_hyp <= 0 || n_hyp > PNP_MAXH) return D3R_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, (size_t)n_jobs * PNP_MAXH * sizeof(int), st) != hipSuccess) return D3R_ERR_LAUNCH;
    hipLaunchKernelGGL(pnp_score_kernel, dim3(PNP_GRID, n_jobs), dim3(256), 0, st, (const PnpJob*)jobs, hypotheses, n_hyp, reproj_err * reproj_err, counts);
    return rc_of(hipGetLastError());
}

extern "C" int d3r_pnp_sums(int n_jobs, const void* jobs, const float* poses, float reproj_err, void* workspace, double* out, void* stream) {
    if (n_jobs <= 0 || !jobs || !poses || !workspace || !out) return D3R_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pnp_sums_kernel, dim3(PNP_GRID, n_jobs), dim3(256), 0, st, (const PnpJob*)jobs, poses, reproj_err * reproj_err, PNP_GRID,
                       (double*)workspace);
    hipLaunchKernelGGL(pnp_reduce_kernel, dim3(n_jobs), dim3(64), 0, st, (const double*)workspace, PNP_GRID, out);
    return rc_of(hipGetLastError());
}
