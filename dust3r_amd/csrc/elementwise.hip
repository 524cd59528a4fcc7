// dust3r_amd -- bandwidth-bound kernels of the forward path (gfx950): LayerNorm, dtype
// conversion, patch gather, the standalone 2-D RoPE op, bilinear x2 upsampling and the
// post-processing epilogues. All of them move 8-16 bytes per lane per access.
#include <stdlib.h>

#include "kernels.hpp"

namespace d3r {

// ------------------------------------------------------------------------------ LayerNorm
// croco blocks use nn.LayerNorm(eps=1e-6) on the fp32 residual stream (oracle/croco_ref/models/
// croco.py); one wave per row, row kept in registers, two-pass mean / variance like ATen.
// X3IN: the input rows are split-fp16 rows (the residual stream of a folded-LayerNorm engine, kernels.hpp GF_X3RES) instead of fp32
template <int DT, bool X3IN = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ out, int rows,
                                                        int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * C);
    const int nch = C >> 2;
    constexpr int MAXV = 8;  // C <= 2048
    float4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            if constexpr (X3IN) v[i] = load4<D3R_F16X3>(x, (size_t)row * C + 4 * (size_t)c);
            else
            v[i] = xr[c];
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + b * b) + (cc * cc + d * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c];
            const float4 b = reinterpret_cast<const float4*>(beta)[c];
            store4<DT>(out, (size_t)row * C + 4 * c, (v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                       (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
        }
    }
}

hipError_t launch_layernorm_x3in(const void* x3rows, const float* gamma, const float* beta, void* out, int rows, int C, float eps, hipStream_t s) {
    if (C % 8 != 0 || C > 2048 || rows <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL((layernorm_kernel<D3R_F16X3, true>), dim3(cdiv(rows, 4)), dim3(256), 0, s, reinterpret_cast<const float*>(x3rows), gamma, beta, out, rows, C, eps);
    return hipGetLastError();
}
hipError_t launch_layernorm(int dt, const float* x, const float* gamma, const float* beta, void* out, int rows, int C,
                            float eps, hipStream_t s) {
    if (C % 4 != 0 || C > 2048 || rows <= 0) return hipErrorInvalidValue;
    dt = d3r_act_dt(dt);        // the 2.5-unit GEMMs read fp16 + fp8 activation rows
    const dim3 grid(cdiv(rows, 4)), block(256);
    switch (dt) {
        case D3R_BF16: hipLaunchKernelGGL(layernorm_kernel<D3R_BF16>, grid, block, 0, s, x, gamma, beta, out, rows, C, eps); break;
        case D3R_F16: hipLaunchKernelGGL(layernorm_kernel<D3R_F16>, grid, block, 0, s, x, gamma, beta, out, rows, C, eps); break;
        case D3R_F32: hipLaunchKernelGGL(layernorm_kernel<D3R_F32>, grid, block, 0, s, x, gamma, beta, out, rows, C, eps); break;
        case D3R_F16X3: hipLaunchKernelGGL(layernorm_kernel<D3R_F16X3>, grid, block, 0, s, x, gamma, beta, out, rows, C, eps); break;
        case D3R_F16F8:     // rows of 256-byte super-groups
            if (C % 64 != 0) return hipErrorInvalidValue;
            hipLaunchKernelGGL(layernorm_kernel<D3R_F16F8>, grid, block, 0, s, x, gamma, beta, out, rows, C, eps); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ convert
template <int DT> __global__ __launch_bounds__(256) void convert_kernel(const float* __restrict__ x, void* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        store4<DT>(out, 4 * i, v.x, v.y, v.z, v.w);
    }
}
hipError_t launch_convert(int dt, const float* x, void* out, size_t n, hipStream_t s) {
    if (n % 4 != 0) return hipErrorInvalidValue;
    dt = d3r_act_dt(dt);
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    switch (dt) {
        case D3R_BF16: hipLaunchKernelGGL(convert_kernel<D3R_BF16>, dim3(grid), dim3(256), 0, s, x, out, n4); break;
        case D3R_F16: hipLaunchKernelGGL(convert_kernel<D3R_F16>, dim3(grid), dim3(256), 0, s, x, out, n4); break;
        case D3R_F32: hipLaunchKernelGGL(convert_kernel<D3R_F32>, dim3(grid), dim3(256), 0, s, x, out, n4); break;
        case D3R_F16X3: hipLaunchKernelGGL(convert_kernel<D3R_F16X3>, dim3(grid), dim3(256), 0, s, x, out, n4); break;
        case D3R_F16F8:
            if (n % 64 != 0) return hipErrorInvalidValue;
            hipLaunchKernelGGL(convert_kernel<D3R_F16F8>, dim3(grid), dim3(256), 0, s, x, out, n4); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ patchify
// PatchEmbedDust3R (dust3r/patch_embed.py:19-29): conv k=s=ps is a GEMM over rows
// m = (b, ty, tx), k = (c, py, px) -- the order of proj.weight.view(D, -1).
template <int DT>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, void* __restrict__ out, int B, int H, int W,
                                                       int ps) {
    const int tw = W / ps, th = H / ps, q = ps / 4;
    const size_t total = (size_t)B * 3 * H * (W / 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t r = i;
        const int x4 = (int)(r % (W / 4)); r /= (W / 4);
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % 3);
        const int b = (int)(r / 3);
        const float4 v = *reinterpret_cast<const float4*>(img + (((size_t)b * 3 + c) * H + y) * W + 4 * x4);
        const int ty = y / ps, py = y - ty * ps, tx = x4 / q, px = (x4 - tx * q) * 4;
        const size_t m = ((size_t)b * th + ty) * tw + tx;
        store4<DT>(out, m * (size_t)(3 * ps * ps) + (size_t)c * ps * ps + py * ps + px, v.x, v.y, v.z, v.w);
    }
}
hipError_t launch_patchify(int dt, const float* img, void* out, int B, int H, int W, int ps, hipStream_t s) {
    if (ps % 4 != 0 || H % ps != 0 || W % ps != 0) return hipErrorInvalidValue;
    const size_t total = (size_t)B * 3 * H * (W / 4);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    switch (dt) {
        case D3R_BF16: hipLaunchKernelGGL(patchify_kernel<D3R_BF16>, dim3(grid), dim3(256), 0, s, img, out, B, H, W, ps); break;
        case D3R_F16: hipLaunchKernelGGL(patchify_kernel<D3R_F16>, dim3(grid), dim3(256), 0, s, img, out, B, H, W, ps); break;
        case D3R_F32: hipLaunchKernelGGL(patchify_kernel<D3R_F32>, dim3(grid), dim3(256), 0, s, img, out, B, H, W, ps); break;
        case D3R_F16X3: hipLaunchKernelGGL(patchify_kernel<D3R_F16X3>, dim3(grid), dim3(256), 0, s, img, out, B, H, W, ps); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ RoPE
// Standalone op with the signature of the reference's only native extension, croco's
// curope `rope_2d(tokens[B,N,H,D], positions[B,N,2] int64, base, F0)` (in place): first D/2
// dims rotate with y, second with x; inside a half, element i pairs with i + D/4, angle =
// pos * F0 / base^(i/(D/4)). The engine itself fuses RoPE into the projection epilogue
// (gemm.hip) from the cos/sin table below; this kernel is the drop-in for the op.
template <int DT>
__global__ __launch_bounds__(256) void rope2d_kernel(void* __restrict__ tokens, const int64_t* __restrict__ pos, int BN, int H,
                                                     int D, float base, float F0) {
    const int Q = D >> 2;
    const size_t total = (size_t)BN * H * 2 * Q;  // one thread per (token, head, half, i)
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        size_t r = idx;
        const int i = (int)(r % Q); r /= Q;
        const int half = (int)(r & 1); r >>= 1;
        const int h = (int)(r % H);
        const size_t bn = r / H;
        const float inv_freq = F0 / powf(base, (float)i / (float)Q);
        const float ang = (float)pos[bn * 2 + half] * inv_freq;
        float sn, cs;
        sincosf(ang, &sn, &cs);
        const size_t e = (bn * H + h) * (size_t)D + half * 2 * Q + i;
        const float u = load1<DT>(tokens, e), v = load1<DT>(tokens, e + Q);
        store1<DT>(tokens, e, u * cs - v * sn);
        store1<DT>(tokens, e + Q, v * cs + u * sn);
    }
}
hipError_t launch_rope2d(int dt, void* tokens, const int64_t* pos, int B, int N, int H, int D, float base, float F0,
                         hipStream_t s) {
    if (D % 4 != 0 || B <= 0 || N <= 0 || H <= 0) return hipErrorInvalidValue;
    const size_t total = (size_t)B * N * H * (D / 2);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    switch (dt) {
        case D3R_BF16: hipLaunchKernelGGL(rope2d_kernel<D3R_BF16>, dim3(grid), dim3(256), 0, s, tokens, pos, B * N, H, D, base, F0); break;
        case D3R_F16: hipLaunchKernelGGL(rope2d_kernel<D3R_F16>, dim3(grid), dim3(256), 0, s, tokens, pos, B * N, H, D, base, F0); break;
        case D3R_F32: hipLaunchKernelGGL(rope2d_kernel<D3R_F32>, dim3(grid), dim3(256), 0, s, tokens, pos, B * N, H, D, base, F0); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

__global__ void rope_table_kernel(float* table, int max_pos, float base, float F0) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // (pos, i) with i in [0,16)
    if (idx >= max_pos * 16) return;
    const int pos = idx >> 4, i = idx & 15;
    const float inv_freq = (float)((double)F0 / pow((double)base, (double)i / 16.0));
    const float ang = (float)pos * inv_freq;  // fp32 product, as the reference's torch fallback forms it
    table[2 * idx] = (float)cos((double)ang);
    table[2 * idx + 1] = (float)sin((double)ang);
}
hipError_t launch_rope_table(float* table, int max_pos, float base, float F0, hipStream_t s) {
    hipLaunchKernelGGL(rope_table_kernel, dim3(cdiv(max_pos * 16, 256)), dim3(256), 0, s, table, max_pos, base, F0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ bilinear x2
// F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) on NHWC, cropped to (Ho, Wo)
// (dpt_head.py:57 crops refinenet4's output to layer 3's size). ATen's source-index formula.
// One workgroup per output row (b, oy): the row's source lines and vertical weights are wave-uniform, the per-item index
// math is 32-bit (the first version decomposed a flat 64-bit index with three 64-bit divisions per item and moved 8 B per
// lane: 2x off the HBM rate at the head's 384x512 maps). NV groups of 4 channels per lane: 16 B accesses for 16-bit types.
// split-fp16 rows, 8 consecutive elements, with the non-temporal policy: the x2 maps (3.2 GB at the head's last stage) are written once and
// read by the NEXT launch only after the whole map has been written -- keeping them out of the L2's way of the four input rows being re-read
D3R_DEV void store8_x3_nt(void* base, size_t elem_off, const float (&v)[8]) {
    using TX = Traits<D3R_F16X3>;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    uint4 h, l;
    TX::split2(v[0], v[1], h.x, l.x); TX::split2(v[2], v[3], h.y, l.y);
    TX::split2(v[4], v[5], h.z, l.z); TX::split2(v[6], v[7], h.w, l.w);
    char* p = reinterpret_cast<char*>(base) + TX::boff(elem_off);
    const u32x4_t hv = {h.x, h.y, h.z, h.w}, lv = {l.x, l.y, l.z, l.w};
    __builtin_nontemporal_store(hv, reinterpret_cast<u32x4_t*>(p));
    __builtin_nontemporal_store(lv, reinterpret_cast<u32x4_t*>(p + 16));
}

template <int DT, int NV, bool NT = false>
__global__ __launch_bounds__(256) void upsample2x_kernel(const void* __restrict__ in, void* __restrict__ out, void* __restrict__ out_relu,
                                                         int Hi, int Wi, int C, int cstride, int Ho, int Wo) {
    const int cn = C / (4 * NV);
    const int b = blockIdx.x / Ho, oy = blockIdx.x - b * Ho;
    const float sh = Hi > 1 ? (float)(Hi - 1) / (float)(2 * Hi - 1) : 0.f;
    const float sw = Wi > 1 ? (float)(Wi - 1) / (float)(2 * Wi - 1) : 0.f;
    const float fy = sh * (float)oy;
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, hy = 1.f - ly;
    const size_t row0 = ((size_t)b * Hi + y0) * Wi, row1 = ((size_t)b * Hi + y1) * Wi;
    const size_t orow = ((size_t)b * Ho + oy) * Wo;
    const int items = Wo * cn;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int ox = i / cn, c = (i - ox * cn) * (4 * NV);
        const float fx = sw * (float)ox;
        const int x0 = (int)fx;
        const int x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
        const float lx = fx - (float)x0, hx = 1.f - lx;
        const size_t a00 = (row0 + x0) * (size_t)cstride + c, a01 = (row0 + x1) * (size_t)cstride + c;
        const size_t a10 = (row1 + x0) * (size_t)cstride + c, a11 = (row1 + x1) * (size_t)cstride + c;
        const size_t oo = (orow + ox) * (size_t)cstride + c;
        if constexpr (NV == 2) {   // cstride % 8 == 0 and c % 8 == 0: whole 16-byte accesses
            float v00[8], v01[8], v10[8], v11[8], o[8], orl[8];
            load8<DT>(in, a00, v00);
            load8<DT>(in, a01, v01);
            load8<DT>(in, a10, v10);
            load8<DT>(in, a11, v11);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                o[k] = hy * (hx * v00[k] + lx * v01[k]) + ly * (hx * v10[k] + lx * v11[k]);
                orl[k] = fmaxf(o[k], 0.f);
            }
            if constexpr (NT && DT == D3R_F16X3) {
                store8_x3_nt(out, oo, o);
                if (out_relu) store8_x3_nt(out_relu, oo, orl);
            } else {
                store8<DT>(out, oo, o);
                if (out_relu) store8<DT>(out_relu, oo, orl);
            }
        } else {
            const float4 v00 = load4<DT>(in, a00), v01 = load4<DT>(in, a01), v10 = load4<DT>(in, a10), v11 = load4<DT>(in, a11);
            const float o0 = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
            const float o1 = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
            const float o2 = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
            const float o3 = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
            store4<DT>(out, oo, o0, o1, o2, o3);
            if (out_relu) store4<DT>(out_relu, oo, fmaxf(o0, 0.f), fmaxf(o1, 0.f), fmaxf(o2, 0.f), fmaxf(o3, 0.f));
        }
    }
}
// x2 bilinear upsampling, a 2 x 2 block of output pixels x 8 channels per thread (round 3). The one-pixel kernel above issues 8 sixteen-byte
// loads per 2-4 stores and is bound by the CU's vector-memory issue path, not by HBM (2.7 TB/s on the 4 GB the head's last stage moves;
// waves waiting to issue 77 % of their cycles, profiles/r03_b/prof_summary.txt). With align_corners = True the source coordinate of output
// index o is o (n - 1) / (2 n - 1): outputs 2 c and 2 c + 1 need input columns {x0, x0 + 1} and {x0', x0' + 1} with x0' = x0 or x0 + 1, so a
// 2 x 2 output block reads a 3 x 3 input neighbourhood: 18 loads per 8 stores instead of 32 per 8. Every output pixel is evaluated with the
// expression of the one-pixel kernel (rows interpolated along x, then along y).
template <int DT, bool NT>
__global__ __launch_bounds__(256) void upsample2x_quad_kernel(const void* __restrict__ in, void* __restrict__ out, void* __restrict__ out_relu,
                                                              int Hi, int Wi, int C, int cstride, int Ho, int Wo) {
    const int cn = C / 8;
    const int hp = (Ho + 1) / 2, wp = (Wo + 1) / 2;
    const int b = blockIdx.x / hp, r = blockIdx.x - b * hp;
    const float sh = Hi > 1 ? (float)(Hi - 1) / (float)(2 * Hi - 1) : 0.f;
    const float sw = Wi > 1 ? (float)(Wi - 1) / (float)(2 * Wi - 1) : 0.f;
    // the two output rows of this block
    const int oyA = 2 * r, oyB = min(2 * r + 1, Ho - 1);
    const float fyA = sh * (float)oyA, fyB = sh * (float)oyB;
    const int yb = (int)fyA, y0B = (int)fyB;
    const float lyA = fyA - (float)yb, hyA = 1.f - lyA, lyB = fyB - (float)y0B, hyB = 1.f - lyB;
    const bool offY = y0B > yb;                                   // row B starts one input row further down
    const int ry0 = yb, ry1 = min(yb + 1, Hi - 1), ry2 = min(yb + 2, Hi - 1);
    const size_t row[3] = {((size_t)b * Hi + ry0) * Wi, ((size_t)b * Hi + ry1) * Wi, ((size_t)b * Hi + ry2) * Wi};
    const size_t orowA = ((size_t)b * Ho + oyA) * Wo, orowB = ((size_t)b * Ho + oyB) * Wo;
    const bool haveB = 2 * r + 1 < Ho;
    const int items = wp * cn;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int cpair = i / cn, c = (i - cpair * cn) * 8;
        const int oxA = 2 * cpair, oxB = min(2 * cpair + 1, Wo - 1);
        const float fxA = sw * (float)oxA, fxB = sw * (float)oxB;
        const int xb = (int)fxA, x0B = (int)fxB;
        const float lxA = fxA - (float)xb, hxA = 1.f - lxA, lxB = fxB - (float)x0B, hxB = 1.f - lxB;
        const bool offX = x0B > xb;
        const int cx0 = xb, cx1 = min(xb + 1, Wi - 1), cx2 = min(xb + 2, Wi - 1);
        // rows interpolated along x at the two output columns: hA[r][k], hB[r][k]
        float hA[3][8], hB[3][8];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            float v0[8], v1[8], v2[8];
            load8<DT>(in, (row[rr] + cx0) * (size_t)cstride + c, v0);
            load8<DT>(in, (row[rr] + cx1) * (size_t)cstride + c, v1);
            load8<DT>(in, (row[rr] + cx2) * (size_t)cstride + c, v2);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                hA[rr][k] = hxA * v0[k] + lxA * v1[k];
                const float b0 = offX ? v1[k] : v0[k], b1 = offX ? v2[k] : v1[k];
                hB[rr][k] = hxB * b0 + lxB * b1;
            }
        }
        float oAA[8], oAB[8], oBA[8], oBB[8];      // (row, column)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            oAA[k] = hyA * hA[0][k] + lyA * hA[1][k];
            oAB[k] = hyA * hB[0][k] + lyA * hB[1][k];
            const float a0 = offY ? hA[1][k] : hA[0][k], a1 = offY ? hA[2][k] : hA[1][k];
            const float b0 = offY ? hB[1][k] : hB[0][k], b1 = offY ? hB[2][k] : hB[1][k];
            oBA[k] = hyB * a0 + lyB * a1;
            oBB[k] = hyB * b0 + lyB * b1;
        }
        const bool haveXB = 2 * cpair + 1 < Wo;
        auto put = [&](size_t orow, int ox, const float (&o)[8]) __attribute__((always_inline)) {
            const size_t oo = (orow + ox) * (size_t)cstride + c;
            float orl[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) orl[k] = fmaxf(o[k], 0.f);
            if constexpr (NT && DT == D3R_F16X3) {
                store8_x3_nt(out, oo, o);
                if (out_relu) store8_x3_nt(out_relu, oo, orl);
            } else {
                store8<DT>(out, oo, o);
                if (out_relu) store8<DT>(out_relu, oo, orl);
            }
        };
        put(orowA, oxA, oAA);
        if (haveXB) put(orowA, oxB, oAB);
        if (haveB) {
            put(orowB, oxA, oBA);
            if (haveXB) put(orowB, oxB, oBB);
        }
    }
}

template <int DT> static void launch_upsample_t(const void* in, void* out, void* out_relu, int B, int Hi, int Wi, int C, int cstride, int Ho, int Wo,
                                                hipStream_t s) {
    // non-temporal stores of the x2 maps: measured 13.41 -> 12.75 ms of "other" kernels per step, forward 218.7 -> 219.6 pairs/s
    // (profiles/r02_f8/bench_upsample_nt.log); D3R_UPSAMPLE_NT=0: plain stores
    static const bool nt = [] { const char* e = probe_env("D3R_UPSAMPLE_NT"); return e ? e[0] != '0' : true; }();
    const char* e_v1 = getenv("D3R_UPSAMPLE_V1");            // 1: the one-output-pixel-per-thread kernel (A/B, parity tests); read per launch
    if (!(e_v1 && e_v1[0] == '1') && C % 8 == 0 && cstride % 8 == 0) {
        const int blocks = B * ((Ho + 1) / 2);
        if (DT == D3R_F16X3 && nt) hipLaunchKernelGGL((upsample2x_quad_kernel<DT, true>), dim3(blocks), dim3(256), 0, s, in, out, out_relu, Hi, Wi, C, cstride, Ho, Wo);
        else hipLaunchKernelGGL((upsample2x_quad_kernel<DT, false>), dim3(blocks), dim3(256), 0, s, in, out, out_relu, Hi, Wi, C, cstride, Ho, Wo);
        return;
    }
    if (DT == D3R_F16X3 && nt && C % 8 == 0 && cstride % 8 == 0)
        hipLaunchKernelGGL((upsample2x_kernel<DT, 2, true>), dim3(B * Ho), dim3(256), 0, s, in, out, out_relu, Hi, Wi, C, cstride, Ho, Wo);
    else if (C % 8 == 0 && cstride % 8 == 0) hipLaunchKernelGGL((upsample2x_kernel<DT, 2>), dim3(B * Ho), dim3(256), 0, s, in, out, out_relu, Hi, Wi, C, cstride, Ho, Wo);
    else hipLaunchKernelGGL((upsample2x_kernel<DT, 1>), dim3(B * Ho), dim3(256), 0, s, in, out, out_relu, Hi, Wi, C, cstride, Ho, Wo);
}
hipError_t launch_upsample2x(int dt, const void* in, void* out, void* out_relu, int B, int Hi, int Wi, int C, int cstride,
                             int Ho, int Wo, hipStream_t s) {
    if (C % 4 != 0 || Ho > 2 * Hi || Wo > 2 * Wi || B <= 0 || Ho <= 0 || Wo <= 0) return hipErrorInvalidValue;
    switch (dt) {
        case D3R_BF16: launch_upsample_t<D3R_BF16>(in, out, out_relu, B, Hi, Wi, C, cstride, Ho, Wo, s); break;
        case D3R_F16: launch_upsample_t<D3R_F16>(in, out, out_relu, B, Hi, Wi, C, cstride, Ho, Wo, s); break;
        case D3R_F32: launch_upsample_t<D3R_F32>(in, out, out_relu, B, Hi, Wi, C, cstride, Ho, Wo, s); break;
        case D3R_F16X3: launch_upsample_t<D3R_F16X3>(in, out, out_relu, B, Hi, Wi, C, cstride, Ho, Wo, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// postprocess_store (depth_mode / conf_mode of the reference's heads): common.hpp, shared with the GEMM kernel's fused head epilogue

// DPT head tail: Conv2d(last_dim, 4, 1) on the ReLU'd features + postprocess (dpt_head.py:63,
// croco dpt_block head[3:5]). 16 lanes per pixel, 8 channels per lane per step.
// Sums over the 16 lanes of a DPP row (= the 16 lanes that share a pixel), 4 values at once, every lane gets the totals. One asm
// block of DPP-fused adds (the builtin form compiles to mov/mov_dpp/add triples; __shfl_xor goes through the LDS crossbar).
D3R_DEV void row_sum4_dpp(float& a0, float& a1, float& a2, float& a3) {
#define D3R_DPP4(ctrl) \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl "\n\tv_add_f32_dpp %3, %3, %3 " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 D3R_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 D3R_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 D3R_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf")
                 D3R_DPP4("row_mirror row_mask:0xf bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
#undef D3R_DPP4
}

template <int DT>
__global__ __launch_bounds__(256) void head_final_kernel(const void* __restrict__ feat, int C, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ pts, float* __restrict__ conf,
                                                         size_t npix, int pstride, int cstride, PostMode post) {
    const int sub = threadIdx.x & 15;
    // this lane's slice of the 4 x C weight matrix stays in registers when C <= 128 (the DPT head: C = 128)
    float wr[4][8];
    const bool hoist = C <= 128;
    const int wbase = sub * 8 < C ? sub * 8 : 0;
    const float wsel = sub * 8 < C ? 1.f : 0.f;
    if (hoist) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) wr[o][e] = w[o * C + wbase + e] * wsel;   // C % 8 == 0: a lane's 8 channels are in or out together
    }
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2], b3 = bias[3];
    // all 16 groups of a workgroup iterate together (the DPP block needs every lane active): round the trip count up
    const size_t stride = (size_t)gridDim.x * 16;
    for (size_t base = (size_t)blockIdx.x * 16; base < npix; base += stride) {
        const size_t pix = base + (threadIdx.x >> 4);
        const bool live = pix < npix;
        const size_t lp = live ? pix : npix - 1;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (hoist) {
            if (sub * 8 < C) {
                float fv[8];
                load8<DT>(feat, lp * (size_t)C + sub * 8, fv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    a0 += fv[e] * wr[0][e];
                    a1 += fv[e] * wr[1][e];
                    a2 += fv[e] * wr[2][e];
                    a3 += fv[e] * wr[3][e];
                }
            }
        } else {
            for (int c = sub * 8; c < C; c += 128) {
                float fv[8];
                load8<DT>(feat, lp * (size_t)C + c, fv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    a0 += fv[e] * w[0 * C + c + e];
                    a1 += fv[e] * w[1 * C + c + e];
                    a2 += fv[e] * w[2 * C + c + e];
                    a3 += fv[e] * w[3 * C + c + e];
                }
            }
        }
        row_sum4_dpp(a0, a1, a2, a3);
        if (sub == 0 && live) postprocess_store(a0 + b0, a1 + b1, a2 + b2, a3 + b3, pts, conf, pix, pstride, cstride, post);
    }
}
hipError_t launch_head_final(int dt, const void* feat, int C, const float* w, const float* b, float* pts, float* conf,
                             size_t npix, int pstride, int cstride, PostMode post, hipStream_t s) {
    if (C % 8 != 0) return hipErrorInvalidValue;
    const int grid = (int)((npix + 15) / 16 < 65536 ? (npix + 15) / 16 : 65536);
    switch (dt) {
        case D3R_BF16: hipLaunchKernelGGL(head_final_kernel<D3R_BF16>, dim3(grid), dim3(256), 0, s, feat, C, w, b, pts, conf, npix, pstride, cstride, post); break;
        case D3R_F16: hipLaunchKernelGGL(head_final_kernel<D3R_F16>, dim3(grid), dim3(256), 0, s, feat, C, w, b, pts, conf, npix, pstride, cstride, post); break;
        case D3R_F32: hipLaunchKernelGGL(head_final_kernel<D3R_F32>, dim3(grid), dim3(256), 0, s, feat, C, w, b, pts, conf, npix, pstride, cstride, post); break;
        case D3R_F16X3: hipLaunchKernelGGL(head_final_kernel<D3R_F16X3>, dim3(grid), dim3(256), 0, s, feat, C, w, b, pts, conf, npix, pstride, cstride, post); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// LinearPts3d tail (dust3r/heads/linear_head.py:36-41): pixel_shuffle(ps) of the token-major
// projection (channel = c*ps*ps + py*ps + px) followed by postprocess.
__global__ __launch_bounds__(256) void linear_head_post_kernel(const float* __restrict__ feat, float* __restrict__ pts,
                                                               float* __restrict__ conf, int B, int th, int tw, int ps, int pstride, int cstride, PostMode post) {
    const int H = th * ps, W = tw * ps, pp = ps * ps;
    const size_t total = (size_t)B * H * W;
    for (size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (size_t)gridDim.x * 256) {
        size_t r = pix;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const int ty = y / ps, py = y - ty * ps, tx = x / ps, px = x - tx * ps;
        const float* f = feat + (((size_t)b * th + ty) * tw + tx) * (size_t)(4 * pp) + py * ps + px;
        postprocess_store(f[0], f[pp], f[2 * pp], f[3 * pp], pts, conf, pix, pstride, cstride, post);
    }
}
hipError_t launch_linear_head_post(const float* feat, float* pts, float* conf, int B, int th, int tw, int ps, int pstride, int cstride,
                                   PostMode post, hipStream_t s) {
    const size_t total = (size_t)B * th * tw * ps * ps;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(linear_head_post_kernel, dim3(grid), dim3(256), 0, s, feat, pts, conf, B, th, tw, ps, pstride, cstride, post);
    return hipGetLastError();
}

hipError_t launch_fill_zero(void* p, size_t bytes, hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s); }

// ---- weight packing (load time): fp32 checkpoint tensor (PyTorch layout, device) -> engine layout -------------
// One thread per SOURCE element; the destination buffers are zero-initialised at allocation, so the padding
// (rows up to n_pad, channels up to cin_pad / cout_pad) stays zero.
template <int DT> __global__ __launch_bounds__(256) void pack_weight_kernel(PackParams p) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.numel) return;
    float v = p.src[i];
    size_t d;
    if (p.kind == PACK_MAT) {                    // [rows][cols] -> [row_off + r][c]
        const size_t r = i / p.cols, c = i - r * p.cols;
        d = (r + p.row_off) * (size_t)p.dst_cols + c;
        if (p.kscale) v *= p.kscale[c];          // W diag(gamma): the LayerNorm in front of this nn.Linear is folded into it (kernels.hpp, GemmParams::ln_*)
    } else if (p.kind == PACK_CONV) {            // [Cout][Cin][k][k] -> [co][(ky*k+kx)*cin_pad + ci]
        const int kk = p.ksize * p.ksize;
        const size_t co = i / ((size_t)p.cin * kk);
        const int rem = (int)(i - co * (size_t)p.cin * kk);
        const int ci = rem / kk, t = rem - ci * kk;
        if (p.kslice_major) {                    // [co][(ci / KS) * (k*k*KS) + t * KS + ci % KS], KS = elements of one K step
            constexpr int KS = 128 / Traits<DT>::EB;
            d = co * (size_t)p.dst_cols + (size_t)(ci / KS) * ((size_t)kk * KS) + (size_t)t * KS + (ci % KS);
        } else {
            d = co * (size_t)p.dst_cols + (size_t)t * p.cin_pad + ci;
        }
    } else {                                     // PACK_CONVT: [Cin][Cout][k][k] -> [(ky*k+kx)*cout_pad + co][ci]
        const int kk = p.ksize * p.ksize;
        const size_t ci = i / ((size_t)p.cols * kk);
        const int rem = (int)(i - ci * (size_t)p.cols * kk);
        const int co = rem / kk, t = rem - co * kk;
        d = ((size_t)t * p.cout_pad + co) * (size_t)p.dst_cols + ci;
    }
    if constexpr (DT == D3R_F16F8) Traits<D3R_F16F8>::store1_wgt(p.dst, d, v);   // the weight encoding of the fp16 + fp8 rows
    else if constexpr (DT == D3R_F16X2F8) Traits<D3R_F16X2F8>::store1_wgt5(p.dst, d / p.dst_cols, d % p.dst_cols, p.dst_cols, v);   // 2.5-unit rows: five chunks per 128 k
    else store1<DT>(p.dst, d, v);
}
hipError_t launch_pack_weight(int dt, const PackParams& p, hipStream_t s) {
    if (p.numel == 0) return hipSuccess;
    const unsigned grid = (unsigned)((p.numel + 255) / 256);
    switch (dt) {
        case D3R_BF16: hipLaunchKernelGGL(pack_weight_kernel<D3R_BF16>, dim3(grid), dim3(256), 0, s, p); break;
        case D3R_F16: hipLaunchKernelGGL(pack_weight_kernel<D3R_F16>, dim3(grid), dim3(256), 0, s, p); break;
        case D3R_F32: hipLaunchKernelGGL(pack_weight_kernel<D3R_F32>, dim3(grid), dim3(256), 0, s, p); break;
        case D3R_F16X3: hipLaunchKernelGGL(pack_weight_kernel<D3R_F16X3>, dim3(grid), dim3(256), 0, s, p); break;
        case D3R_F16F8:     // nn.Linear matrices only (the convolutions of the DPT head stay split-fp16)
            if (p.kind != PACK_MAT || p.dst_cols % 64 != 0) return hipErrorInvalidValue;
            hipLaunchKernelGGL(pack_weight_kernel<D3R_F16F8>, dim3(grid), dim3(256), 0, s, p); break;
        case D3R_F16X2F8:   // nn.Linear matrices only, whole 128-k blocks of five chunks (5 bytes per element)
            if (p.kind != PACK_MAT || p.dst_cols % 128 != 0) return hipErrorInvalidValue;
            hipLaunchKernelGGL(pack_weight_kernel<D3R_F16X2F8>, dim3(grid), dim3(256), 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// ---- folded LayerNorm (kernels.hpp GemmParams::ln_*) ----------------------------------------------------------------------------------------
// rows x G partial (sum, sum of squares) pairs -> rstd, -mean rstd. 32 lanes per row (coalesced: a row's G pairs are one 8 G-byte run), lane g
// adds pairs g and g + 32 in fp64, then a 5-step xor butterfly inside the 32-lane half: a fixed tree, so the statistics of a row do not depend
// on how the producing GEMM was tiled, nor on the batch the row sits in. (First version: one thread per row walking its 256 bytes -- 64 cache
// lines per load instruction, ~10 us per launch at 49152 rows.)
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2* __restrict__ part, int rows, int G, float inv_c, float eps, float* __restrict__ rstd,
                                                          float* __restrict__ nmr) {
    const int g = threadIdx.x & 31;
    const int r = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int rc = r < rows ? r : rows - 1;
    const float2* pr = part + (size_t)rc * G;
    double s = 0.0, q = 0.0;
    if (g < G) { const float2 t = pr[g]; s = (double)t.x; q = (double)t.y; }
    if (g + 32 < G) { const float2 t = pr[g + 32]; s += (double)t.x; q += (double)t.y; }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (g == 0 && r < rows) {
        const double mean = s * (double)inv_c;
        double var = q * (double)inv_c - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float rs = (float)(1.0 / sqrt(var + (double)eps));
        rstd[r] = rs;
        nmr[r] = (float)(-mean) * rs;
    }
}
hipError_t launch_ln_finalize(const float* part, int rows, int C, float eps, float* rstd, float* nmr, hipStream_t s) {
    if (rows <= 0 || C % 32 != 0 || C > 2048) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((rows + 7) / 8), dim3(256), 0, s, reinterpret_cast<const float2*>(part), rows, C / 32, 1.0f / (float)C, eps, rstd, nmr);
    return hipGetLastError();
}
// one wave per weight row n: colsum[n] = sum_k r(gamma_k W_nk), bias_out[n] = bias_in[n] + sum_k beta_k W_nk (fp64 sums, lane-strided then butterfly)
template <int DT> __global__ __launch_bounds__(256) void ln_fold_vectors_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                                const float* __restrict__ bias_in, float* __restrict__ colsum, float* __restrict__ bias_out,
                                                                                int N, int K) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + (size_t)n * K;
    double s = 0.0, b = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float wv = w[k], wg = wv * gamma[k];       // the product the pack kernel rounds (same fp32 multiply)
        float r;
        if constexpr (DT == D3R_F16X3) {
            uint32_t hi, lo;
            Traits<D3R_F16X3>::split2(wg, 0.f, hi, lo);
            r = Traits<D3R_F16X3>::join_lo(hi, lo);      // hi + lo: exact in fp32 (two 11-bit significands 11 binades apart)
        } else if constexpr (DT == D3R_F32) {
            r = wg;
        } else {
            r = Traits<DT>::unpack_lo(Traits<DT>::pack2(wg, 0.f));
        }
        s += (double)r;
        b += (double)beta[k] * (double)wv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); b += __shfl_xor(b, o); }
    if (lane == 0) {
        colsum[n] = (float)s;
        bias_out[n] = (float)((bias_in ? (double)bias_in[n] : 0.0) + b);
    }
}
hipError_t launch_ln_fold_vectors(int dt, const float* W, const float* gamma, const float* beta, const float* bias_in, float* colsum, float* bias_out, int N, int K,
                                  hipStream_t s) {
    if (N <= 0 || K <= 0) return hipErrorInvalidValue;
    const dim3 grid((N + 3) / 4), block(256);
    switch (dt) {
        case D3R_F16X3: hipLaunchKernelGGL(ln_fold_vectors_kernel<D3R_F16X3>, grid, block, 0, s, W, gamma, beta, bias_in, colsum, bias_out, N, K); break;
        case D3R_F32: hipLaunchKernelGGL(ln_fold_vectors_kernel<D3R_F32>, grid, block, 0, s, W, gamma, beta, bias_in, colsum, bias_out, N, K); break;
        case D3R_BF16: hipLaunchKernelGGL(ln_fold_vectors_kernel<D3R_BF16>, grid, block, 0, s, W, gamma, beta, bias_in, colsum, bias_out, N, K); break;
        case D3R_F16: hipLaunchKernelGGL(ln_fold_vectors_kernel<D3R_F16>, grid, block, 0, s, W, gamma, beta, bias_in, colsum, bias_out, N, K); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ConvTranspose bias [Cout] -> fp32 [k*k][cout_pad]
__global__ __launch_bounds__(256) void pack_convt_bias_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cout_pad, int taps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cout * taps) return;
    const int t = i / cout, co = i - t * cout;
    dst[(size_t)t * cout_pad + co] = src[co];
}
hipError_t launch_pack_convt_bias(const float* src, float* dst, int cout, int cout_pad, int taps, hipStream_t s) {
    hipLaunchKernelGGL(pack_convt_bias_kernel, dim3((cout * taps + 255) / 256), dim3(256), 0, s, src, dst, cout, cout_pad, taps);
    return hipGetLastError();
}

}  // namespace d3r
