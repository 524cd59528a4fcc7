// dust3r_amd -- MFMA GEMM / implicit-GEMM convolution with fused epilogues (gfx950).
//
// One kernel serves every dense contraction of the DUSt3R forward (reference call sites):
//   Linear layers of croco Block / DecoderBlock (qkv, proj, fc1, fc2, projq/k/v) and
//   decoder_embed (dust3r/model.py:136-137,176-186), PatchEmbed's k16s16 conv as a GEMM
//   over pre-gathered patches (dust3r/patch_embed.py:19-29), the DPT head's 1x1 / 3x3 /
//   stride-2 convolutions and ConvTranspose k=s (dust3r/heads/dpt_head.py:34-65) as implicit
//   GEMMs over NHWC activations, and LinearPts3d.proj (dust3r/heads/linear_head.py:30-41).
//
// Shape: out[m][n] = sum_k act[m][k] * wgt[n][k]   ("NT": both operands K-contiguous).
// Tile:  128 (m) x 128 (n) x 128 bytes of K per step; 4 waves (2x2), each 64x64 = 4x4 MFMA
//        16x16 fragments. Operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR
//        round trip), double buffered; LDS image is lane-linear, the bank swizzle
//        (chunk ^= (row>>1)&7, conflict-free for ds_read_b128 on 128-byte rows) is applied on
//        the per-lane SOURCE address and again on the fragment read.
// MFMA operand roles: D[i][j] with 4 consecutive i per lane. Normally i = n (weights) so each
// lane owns 4 consecutive output columns of one row -> 8/16-byte stores; for V^T tiles of the
// attention projections the roles are swapped (i = m) so 4 consecutive TOKENS land together.
#include "kernels.hpp"

namespace d3r {

static constexpr int BM = 128, BN = 128, KTB = 128;  // KTB: bytes of K per tile row
static constexpr int STAGE_BYTES = (BM + BN) * KTB;  // 32 KiB
static constexpr int GEMM_LDS = 2 * STAGE_BYTES;     // 64 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int DT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TR = Traits<DT>;
    constexpr int EB = TR::EB;
    constexpr int KT = KTB / EB;  // elements of K per tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = p.n_pad / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / tiles_n, tn = lid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const bool swap = (p.epi == EPI_HEADS) && (p.head_kind[n0 / p.head_c] == HEAD_VT);

    // ---- staging addresses (per lane: 4 rows of each operand, one 16-byte chunk) -------------
    const int lrow = wave * 8 + (lane >> 3);                       // row inside a 32-row slab
    const int lchunk = (lane & 7) ^ (((lane >> 4) + wave * 4) & 7);  // logical chunk fetched by this lane
    const char* wsrc[4];
    const char* asrc[4];
    int iy0[4], ix0[4], ibase[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = q * 32 + lrow;
        wsrc[q] = reinterpret_cast<const char*>(p.wgt) + ((size_t)(n0 + r) * p.K) * EB + lchunk * 16;
        int m = m0 + r;
        m = m < p.M ? m : p.M - 1;
        if (p.amode == AMODE_LINEAR) {
            asrc[q] = reinterpret_cast<const char*>(p.act) + ((size_t)m * p.lda) * EB + lchunk * 16;
        } else {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            iy0[q] = oy * p.stride - p.pad;
            ix0[q] = ox * p.stride - p.pad;
            ibase[q] = b * p.Hin * p.Win;
            asrc[q] = nullptr;
        }
    }
    const char* zsrc = reinterpret_cast<const char*>(p.zero_page) + lchunk * 16;

    auto stage = [&](int kt, int buf) {
        char* sb = smem + buf * STAGE_BYTES;
        const size_t koff = (size_t)kt * KTB;
        if (p.amode == AMODE_LINEAR) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(asrc[q] + koff), (lptr_t)(sb + (q * 4 + wave) * 1024), 16, 0, 0);
        } else {
            const int kel = kt * KT;
            const int tap = kel / p.Cin, c0 = kel - tap * p.Cin;
            const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int iy = iy0[q] + ky, ix = ix0[q] + kx;
                const bool ok = (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
                const char* src = reinterpret_cast<const char*>(p.act) +
                                  ((size_t)(ibase[q] + iy * p.Win + ix) * p.cstride + c0) * EB + lchunk * 16;
                src = ok ? src : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sb + (q * 4 + wave) * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[q] + koff), (lptr_t)(sb + BM * KTB + (q * 4 + wave) * 1024), 16, 0, 0);
    };

    // ---- fragment read addresses ---------------------------------------------------------------
    const int wi = wave >> 1, wj = wave & 1;
    const int frow = lane & 15, fsw = (lane >> 1) & 7, fgrp = lane >> 4;
    // P tile supplies i (4 consecutive per lane), Q tile supplies j
    const int p_off = swap ? 0 : BM * KTB;  // activations live at 0, weights at BM*KTB
    const int q_off = swap ? BM * KTB : 0;

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / KT;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        __syncthreads();  // compiler drains vmcnt here: tile kt has landed; buf^1 is free again
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* sb = smem + buf * STAGE_BYTES;
        if constexpr (DT == D3R_F16X3) {
            // 128 bytes of a row = 32 logical k = 4 groups [hi x8][lo x8]; lane group fgrp owns group fgrp
            const int chi = ((2 * fgrp) ^ fsw) * 16, clo = ((2 * fgrp + 1) ^ fsw) * 16;
            uint4 pf[4], pl[4], qf[4], ql[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const char* pr = sb + p_off + (wi * 64 + f * 16 + frow) * KTB;
                const char* qr = sb + q_off + (wj * 64 + f * 16 + frow) * KTB;
                pf[f] = *reinterpret_cast<const uint4*>(pr + chi);
                pl[f] = *reinterpret_cast<const uint4*>(pr + clo);
                qf[f] = *reinterpret_cast<const uint4*>(qr + chi);
                ql[f] = *reinterpret_cast<const uint4*>(qr + clo);
            }
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                for (int fj = 0; fj < 4; ++fj) TR::mma16x3(acc[fi][fj], pf[fi], pl[fi], qf[fj], ql[fj]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int coff = ((ks * 4 + fgrp) ^ fsw) * 16;
                uint4 pf[4], qf[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    pf[f] = *reinterpret_cast<const uint4*>(sb + p_off + (wi * 64 + f * 16 + frow) * KTB + coff);
                    qf[f] = *reinterpret_cast<const uint4*>(sb + q_off + (wj * 64 + f * 16 + frow) * KTB + coff);
                }
#pragma unroll
                for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                    for (int fj = 0; fj < 4; ++fj) TR::mma16(acc[fi][fj], pf[fi], qf[fj]);
            }
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
    const int i4 = (lane >> 4) * 4;  // first of this lane's 4 consecutive i inside a fragment
    const int jl = lane & 15;
    if (!swap) {
        const int nb = n0 + wi * 64, mb = m0 + wj * 64;
        if (p.epi == EPI_HEADS) {
            // q / k projections: bias, 2-D RoPE on the fp32 accumulator, head-major store
            const int region = nb / p.head_c;
            const int h = (nb - region * p.head_c) >> 6;
            void* dst = p.head_dst[region];
            const bool rope = p.head_kind[region] == HEAD_ROPE;
            float4 bias[4];
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
                bias[fi] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nb + fi * 16 + i4) : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int fj = 0; fj < 4; ++fj) {
                const int m = mb + fj * 16 + jl;
                if (m >= p.M) continue;
                const int b = m / p.ntok, t = m - b * p.ntok;
                const int ty = t / p.tok_w, tx = t - ty * p.tok_w;
                const size_t obase = ((size_t)(b * p.heads + h) * p.ntok + t) * 64;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    f32x4_t u = acc[half * 2][fj], v = acc[half * 2 + 1][fj];
                    const float4 bu = bias[half * 2], bv = bias[half * 2 + 1];
                    float uu[4] = {u[0] + bu.x, u[1] + bu.y, u[2] + bu.z, u[3] + bu.w};
                    float vv[4] = {v[0] + bv.x, v[1] + bv.y, v[2] + bv.z, v[3] + bv.w};
                    if (rope) {
                        const int pos = half ? tx : ty;
                        const float4* cs = reinterpret_cast<const float4*>(p.rope_table + ((size_t)pos * 16 + i4) * 2);
                        const float4 c01 = cs[0], c23 = cs[1];  // (cos0,sin0,cos1,sin1), (cos2,sin2,cos3,sin3)
                        const float cc[4] = {c01.x, c01.z, c23.x, c23.z}, ss[4] = {c01.y, c01.w, c23.y, c23.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = uu[r], bq = vv[r];
                            uu[r] = a * cc[r] - bq * ss[r];
                            vv[r] = bq * cc[r] + a * ss[r];
                        }
                    }
                    store4<DT>(dst, obase + half * 32 + i4, uu[0], uu[1], uu[2], uu[3]);
                    store4<DT>(dst, obase + half * 32 + 16 + i4, vv[0], vv[1], vv[2], vv[3]);
                }
            }
            return;
        }
#pragma unroll
        for (int fi = 0; fi < 4; ++fi) {
            const int n = nb + fi * 16 + i4;
            if (n >= p.n_store) continue;
            const float4 bias = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int fj = 0; fj < 4; ++fj) {
                const int m = mb + fj * 16 + jl;
                if (m >= p.M) continue;
                const f32x4_t a = acc[fi][fj];
                float v0 = a[0] + bias.x, v1 = a[1] + bias.y, v2 = a[2] + bias.z, v3 = a[3] + bias.w;
                switch (p.epi) {
                    case EPI_F32: {
                        float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n;
                        if (p.res1) {
                            const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res1) + (size_t)m * p.ldr + n);
                            v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                        }
                        *reinterpret_cast<float4*>(o) = make_float4(v0, v1, v2, v3);
                        if (p.out2) store4<DT>(p.out2, (size_t)m * p.ldo2 + n, v0, v1, v2, v3);
                    } break;
                    case EPI_GELU:
                        store4<DT>(p.out, (size_t)m * p.ldo + n, gelu_erf(v0), gelu_erf(v1), gelu_erf(v2), gelu_erf(v3));
                        break;
                    case EPI_CONVT: {
                        // ConvTranspose2d(k == stride): column n = (tap, co); row m = input pixel
                        const int tap = n / p.ct_cout, co = n - tap * p.ct_cout;
                        const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
                        const int hw = p.Hin * p.Win;
                        const int b = m / hw, rem = m - b * hw;
                        const int y = rem / p.Win, x = rem - y * p.Win;
                        const size_t opix = ((size_t)b * p.Hin * p.ksize + (y * p.ksize + ky)) * (p.Win * p.ksize) + (x * p.ksize + kx);
                        store4<DT>(p.out, opix * p.ldo + co, v0, v1, v2, v3);  // bias is pre-expanded per (tap, co)
                    } break;
                    default: {  // EPI_T
                        if (p.res1) {
                            const float4 r = load4<DT>(p.res1, (size_t)m * p.ldr + n);
                            v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                        }
                        if (p.res2) {
                            const float4 r = load4<DT>(p.res2, (size_t)m * p.ldr + n);
                            v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                        }
                        if (p.flags & GF_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        store4<DT>(p.out, (size_t)m * p.ldo + n, v0, v1, v2, v3);
                        if (p.out2) store4<DT>(p.out2, (size_t)m * p.ldo2 + n, fmaxf(v0, 0.f), fmaxf(v1, 0.f), fmaxf(v2, 0.f), fmaxf(v3, 0.f));
                    } break;
                }
            }
        }
    } else {
        // V^T tiles: lane owns 4 consecutive tokens (i = m) of one feature (j = n)
        const int mb = m0 + wi * 64, nb = n0 + wj * 64;
        const int region = nb / p.head_c;
        const int h = (nb - region * p.head_c) >> 6;
        void* dst = p.head_dst[region];
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) {
            const int dd = fj * 16 + jl;
            const float bias = p.bias ? p.bias[nb + dd] : 0.f;
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const int m = mb + fi * 16 + i4;
                if (m >= p.M) continue;
                const f32x4_t a = acc[fi][fj];
                const int b = m / p.ntok, t = m - b * p.ntok;
                const size_t rowbase = ((size_t)(b * p.heads + h) * 64 + dd) * p.ldv;
                if (t + 3 < p.ntok && ((p.ldv | t) & 3) == 0) {
                    store4<DT>(dst, rowbase + t, a[0] + bias, a[1] + bias, a[2] + bias, a[3] + bias);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int mm = m + r;
                        if (mm >= p.M) break;
                        const int bb = mm / p.ntok, tt = mm - bb * p.ntok;
                        store1<DT>(dst, ((size_t)(bb * p.heads + h) * 64 + dd) * p.ldv + tt, a[r] + bias);
                    }
                }
            }
        }
    }
}

template <int DT> static hipError_t launch_t(const GemmParams& p, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr_set = true;
    }
    const int grid = cdiv(p.M, BM) * (p.n_pad / BN);
    hipLaunchKernelGGL(gemm_kernel<DT>, dim3(grid), dim3(256), GEMM_LDS, s, p);
    return hipGetLastError();
}

hipError_t launch_gemm(int dt, const GemmParams& p, hipStream_t s) {
    const int kt = KTB / (int)dt_bytes(dt);
    if (p.M <= 0 || p.n_pad % BN != 0 || p.K % kt != 0 || p.K <= 0) return hipErrorInvalidValue;
    if (p.amode == AMODE_CONV && (p.Cin % kt != 0 || p.zero_page == nullptr)) return hipErrorInvalidValue;
    switch (dt) {
        case D3R_BF16: return launch_t<D3R_BF16>(p, s);
        case D3R_F16: return launch_t<D3R_F16>(p, s);
        case D3R_F32: return launch_t<D3R_F32>(p, s);
        case D3R_F16X3: return launch_t<D3R_F16X3>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace d3r
