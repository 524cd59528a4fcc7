// dust3r_amd -- fused global-aligner iteration (gfx950, HBM-bound).
//
// Replaces the reference's hot loop #5: `global_alignment_iter` = PointCloudOptimizer.forward
// (~25 elementwise/bmm kernels over (E, A, 3) tensors) + autograd backward + torch Adam
// (reference dust3r/cloud_opt/base_opt.py:326-366, optimizer.py:188-201). One iteration here is
// three launches with no host synchronisation:
//   1. aligner_main_kernel   one pass over every (edge side, pixel): residuals, loss, dL/dX,
//                            per-pixel log-depth gradient + its Adam step, and per-wave partial
//                            sums of dL/dM_e (3x4) and of the per-image pose/focal terms.
//                            Image-major: a thread owns 4 pixels of image i and walks the
//                            edges incident to i, so each pred/weight byte is read exactly once
//                            (32 B per edge-pixel, the algorithmic minimum of SURVEY.md 8(d))
//                            and the depth gradient needs no atomics.
//   2. aligner_reduce_kernel fixed-order fp64 reduction of the partials (deterministic).
//   3. aligner_small_kernel  chain rule to (quaternion, log-translation, log-scale, log-focal),
//                            Adam on those few thousand parameters, loss bookkeeping, and the
//                            derived matrices (M_e, R_i, T_i, F_i) for the next iteration.
#include "aligner_math.hpp"
#include "kernels.hpp"

namespace d3r {

static constexpr int PPT = 4;            // pixels per thread
static constexpr int CHUNK = 256 * PPT;  // pixels per workgroup
static constexpr int PW = 16;            // floats per partial record

struct AlignerView {
    int n, E, maxA, nslot;  // nslot = workgroups (1024-pixel chunks) per image: one partial record per workgroup
    int img0;               // first image of this launch (d3r_aligner image range: one process per GPU owns a contiguous range of images)
    const int* img_w;       // [n]
    const int* img_area;    // [n]
    const int* adj_off;     // [n+1]
    const int* adj_es;      // [2E] entries e*2+side, grouped by projecting image
    const float* pred[2];   // PLANAR copies owned by the handle: [E][3][maxA] (x, y, z planes: unit-stride float4 loads)
    const float* wgt[2];    // [E][maxA]
    const float* inter;     // LAY == 1: [2][E][maxAp / 256][4][256] -- per (side, edge, 256-pixel wave chunk) the x, y, z and weight runs back to back (4 KiB)
    int maxAp;              // maxA rounded up to 256
    float* depth;           // [n][maxA] log-depth
    float* depth_m;
    float* depth_v;
    float* depth_grad;      // optional export (tests)
    const float* d_edge;    // derived [E][12]  M_e
    const float* d_img;     // derived [n][16]  R(9) T(3) F ppx ppy -
    float* part_edge;       // [2E][nslot][16]: gm(12) loss(1)
    float* part_img;        // [n][nslot][16]:  G = sum g (x) cam (9), sum g (3), sum g*exp(d) (3)
    float inv_area[2];
    int l2, update, use_dpp;
    AdamCoef adam;
};

// ---- wave reduction: result valid in lane 63 -------------------------------------------------------
D3R_DEV float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));  // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1,3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));  // row_bcast31 -> rows 2,3
    return v;
}
// The 13 per-edge sums (dL/dM 3x4 + loss) at once, results valid in lane 63. Written as ONE asm block of DPP-fused adds:
// from the builtin form above hipcc emits v_mov 0 / v_mov_dpp / v_pk_add groups (5 VALU per 2 values per step, ~190 per
// edge), and this kernel is VALU-bound (wave64 VALU = 4 cycles; ~440 VALU per thread-edge x 285 wave-edges per SIMD ~ 210 us
// of the 269 us launch). Fused: 78. Each value is read 13 instructions after it was written, so only the block's first DPP
// read needs the 2 wait states the hazard rules ask for after a VALU write (s_nop 1).
D3R_DEV void wave_sum13_dpp(float (&v)[13]) {
#define D3R_DPP13(ctrl)                                                                                                    \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl "\n\t"       \
    "v_add_f32_dpp %3, %3, %3 " ctrl "\n\tv_add_f32_dpp %4, %4, %4 " ctrl "\n\tv_add_f32_dpp %5, %5, %5 " ctrl "\n\t"       \
    "v_add_f32_dpp %6, %6, %6 " ctrl "\n\tv_add_f32_dpp %7, %7, %7 " ctrl "\n\tv_add_f32_dpp %8, %8, %8 " ctrl "\n\t"       \
    "v_add_f32_dpp %9, %9, %9 " ctrl "\n\tv_add_f32_dpp %10, %10, %10 " ctrl "\n\tv_add_f32_dpp %11, %11, %11 " ctrl "\n\t" \
    "v_add_f32_dpp %12, %12, %12 " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 D3R_DPP13("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 D3R_DPP13("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 D3R_DPP13("row_half_mirror row_mask:0xf bank_mask:0xf")
                 D3R_DPP13("row_mirror row_mask:0xf bank_mask:0xf")
                 D3R_DPP13("row_bcast:15 row_mask:0xa bank_mask:0xf")     // rows 1, 3 += lane 15 of the row before; rows 0, 2 keep their value
                 D3R_DPP13("row_bcast:31 row_mask:0xc bank_mask:0xf")     // rows 2, 3 += lane 31
                 "s_nop 1"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                   "+v"(v[10]), "+v"(v[11]), "+v"(v[12]));
#undef D3R_DPP13
}
D3R_DEV float wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// the pred / weight planes are read exactly once per iteration (1.29 GB at the BASELINE scene: nothing to keep in L2 / MALL)
typedef float f32x4v_t __attribute__((ext_vector_type(4)));
D3R_DEV float4 stream_ld4(const float* p) {
    const f32x4v_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v_t*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
#define D3R_STREAM_LD4(ptr) stream_ld4(ptr)
// NWV = waves per workgroup (4: 1024-pixel chunks; 8: 2048-pixel chunks = 8 KiB contiguous per plane and edge side, half the partial records;
// probe D3R_ALIGNER_NWV=8 at handle creation -- the partial sums are then grouped differently: same result to fp32 rounding, not bit for bit)
// PROBE (measurement aid, results INVALID; D3R_ALIGNER_PROBE=1|2 at handle creation, tools/aligner_probe.py): 1 = the pred / weight stream is loaded and summed,
// the per-edge residual math and wave reductions are skipped -- the speed at which this access pattern is delivered; 2 = math kept, wave reductions skipped
// LAY (round 5): 0 = four separate streams per edge side (x, y, z planes of the handle's planar copy + the caller's weight rows), a wave reads four 1 KiB
// runs megabytes apart; 1 = the handle's block-interleaved copy [side][edge][256-pixel chunk][x | y | z | w][256]: a wave reads ONE contiguous 4 KiB run,
// a workgroup 16 KiB per edge side (D3R_ALIGNER_LAYOUT at handle creation; same loads, same arithmetic, bit-identical results)
template <bool L2, int PF, int NWV = 4, int PROBE = 0, int LAY = 0>   // PF = prefetch distance of the pred / weight stream in edges (1 or 2)
__global__ __launch_bounds__(NWV * 64) void aligner_main_kernel(AlignerView a) {
    constexpr int NTH = NWV * 64, CHUNK_T = NTH * PPT;
    const int nchunk = a.nslot;
    const int img_l = blockIdx.x / nchunk, chunk = blockIdx.x - img_l * nchunk;
    const int img = a.img0 + img_l;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p0 = chunk * CHUNK_T + threadIdx.x * PPT;
    const int area = a.img_area[img], W = a.img_w[img];
    const bool active = p0 < a.maxA;  // maxA % 4 == 0: the 4 pixels are in or out together

    const float* di = a.d_img + img * 16;
    float R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = di[k];
    const float T0 = di[9], T1 = di[10], T2 = di[11], F = di[12], ppx = di[13], ppy = di[14];
    const float invF = 1.0f / F;

    float X[PPT][3], cam[PPT][3], g[PPT][3], ed[PPT];
    float4 dlog = make_float4(0, 0, 0, 0);
    if (active) dlog = *reinterpret_cast<const float4*>(a.depth + (size_t)img * a.maxA + p0);
    const float dl[4] = {dlog.x, dlog.y, dlog.z, dlog.w};
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = p0 + k;
        // padded pixels (p >= area) use grid (0,0) like the reference's zero padded _grid (optimizer.py:47-48)
        const int v = p < area ? p / W : 0, u = p < area ? p - v * W : 0;
        ed[k] = expf(dl[k]);
        cam[k][0] = ed[k] * ((float)u - ppx) * invF;
        cam[k][1] = ed[k] * ((float)v - ppy) * invF;
        cam[k][2] = ed[k];
        X[k][0] = R[0] * cam[k][0] + R[1] * cam[k][1] + R[2] * cam[k][2] + T0;
        X[k][1] = R[3] * cam[k][0] + R[4] * cam[k][1] + R[5] * cam[k][2] + T1;
        X[k][2] = R[6] * cam[k][0] + R[7] * cam[k][1] + R[8] * cam[k][2] + T2;
        g[k][0] = g[k][1] = g[k][2] = 0.f;
    }

    const int slot = chunk;
    const int a0 = a.adj_off[img], a1 = a.adj_off[img + 1];
    // Edge walk. The adjacency entries and the 3x4 matrices of up to EBATCH incident edge sides are staged in LDS first,
    // so that inside the walk the ONLY vector-memory traffic is the pred / weight stream (and the partial stores): any
    // other in-loop global load would force an in-order vmcnt wait that drains the prefetch. The stream is software
    // pipelined: the 64 bytes per lane of edge j+1 are requested before the math of edge j.
    constexpr int EBATCH = 64;
    __shared__ int sh_es[EBATCH];
    __shared__ __attribute__((aligned(16))) float sh_M[EBATCH][12];
    __shared__ __attribute__((aligned(16))) float sh_part[NWV][EBATCH][PW];   // per-wave sums, combined once per batch
    const int pl = active ? p0 : 0;   // inactive lanes (beyond maxA in the last chunk) stream pixel 0 and discard it
    for (int base = a0; base < a1; base += EBATCH) {
        const int nb = min(EBATCH, a1 - base);
        __syncthreads();
        for (int i = threadIdx.x; i < nb * 12; i += NTH) {
            const int j = i / 12, k = i - j * 12;
            const int es = a.adj_es[base + j];
            if (k == 0) sh_es[j] = es;
            sh_M[j][k] = a.d_edge[(es >> 1) * 12 + k];
        }
        __syncthreads();
        float4 nq0, nq1, nq2, nww, mq0, mq1, mq2, mww;   // edge j + 1 (and, PF == 2, edge j + 2) in flight
        const size_t ploff = (size_t)(pl >> 8) * 1024 + (pl & 255);     // LAY == 1: this lane's offset inside an edge side's interleaved block
        auto stream_edge = [&](int es, float4& d0, float4& d1, float4& d2, float4& dw) __attribute__((always_inline)) {
            if constexpr (LAY == 1) {
                const float* pp = a.inter + ((size_t)(es & 1) * a.E + (size_t)(es >> 1)) * 4 * (size_t)a.maxAp + ploff;
                d0 = D3R_STREAM_LD4(pp);
                d1 = D3R_STREAM_LD4(pp + 256);
                d2 = D3R_STREAM_LD4(pp + 512);
                dw = D3R_STREAM_LD4(pp + 768);
            } else {
                const float* pp = a.pred[es & 1] + (size_t)(es >> 1) * 3 * a.maxA + pl;
                d0 = D3R_STREAM_LD4(pp);
                d1 = D3R_STREAM_LD4(pp + a.maxA);
                d2 = D3R_STREAM_LD4(pp + 2 * (size_t)a.maxA);
                dw = D3R_STREAM_LD4(a.wgt[es & 1] + (size_t)(es >> 1) * a.maxA + pl);
            }
        };
        stream_edge(sh_es[0], nq0, nq1, nq2, nww);
        if (PF == 2) stream_edge(sh_es[nb > 1 ? 1 : 0], mq0, mq1, mq2, mww);
        for (int j = 0; j < nb; ++j) {
            const int es = sh_es[j];
            const int side = es & 1;
            const float4 q0 = nq0, q1 = nq1, q2 = nq2, ww = nww;
            {   // unconditional (index clamped): a branch here would make the compiler wait for the loads at its join
                const int jn = j + PF < nb ? j + PF : nb - 1;
                const int es2 = sh_es[jn];
                if (PF == 2) { nq0 = mq0; nq1 = mq1; nq2 = mq2; nww = mww; }
                float4& d0 = PF == 2 ? mq0 : nq0;
                float4& d1 = PF == 2 ? mq1 : nq1;
                float4& d2 = PF == 2 ? mq2 : nq2;
                float4& dw = PF == 2 ? mww : nww;
                stream_edge(es2, d0, d1, d2, dw);
            }
            float M[12];
            {
                const float4* Mp = reinterpret_cast<const float4*>(sh_M[j]);
                const float4 m0 = Mp[0], m1 = Mp[1], m2 = Mp[2];
                M[0] = m0.x; M[1] = m0.y; M[2] = m0.z; M[3] = m0.w; M[4] = m1.x; M[5] = m1.y; M[6] = m1.z; M[7] = m1.w;
                M[8] = m2.x; M[9] = m2.y; M[10] = m2.z; M[11] = m2.w;
            }
            float gm[12], loss = 0.f;
#pragma unroll
            for (int k = 0; k < 12; ++k) gm[k] = 0.f;
            {
                const float pr[PPT][3] = {{q0.x, q1.x, q2.x}, {q0.y, q1.y, q2.y}, {q0.z, q1.z, q2.z}, {q0.w, q1.w, q2.w}};   // q0 = x, q1 = y, q2 = z planes
                const float ia = active ? a.inv_area[side] : 0.f;   // zero weight: inactive lanes contribute nothing
                const float wv[PPT] = {ww.x * ia, ww.y * ia, ww.z * ia, ww.w * ia};
                if constexpr (PROBE == 1) {
#pragma unroll
                    for (int k = 0; k < PPT; ++k) { loss += pr[k][0] + pr[k][1] + pr[k][2] + wv[k] + M[k]; g[k][0] += pr[k][0]; }
                } else {
#pragma unroll
                for (int k = 0; k < PPT; ++k) residual_accumulate(X[k], M, pr[k], wv[k], L2, loss, g[k], gm);
                }
            }
            float red[13];
            if constexpr (PROBE != 0) {
#pragma unroll
                for (int k = 0; k < 12; ++k) red[k] = gm[k];
                red[12] = loss;
            } else
            if (a.use_dpp) {
#pragma unroll
                for (int k = 0; k < 12; ++k) red[k] = gm[k];
                red[12] = loss;
                wave_sum13_dpp(red);
            } else {
#pragma unroll
                for (int k = 0; k < 12; ++k) red[k] = wave_sum_shfl(gm[k]);
                red[12] = wave_sum_shfl(loss);
            }
            if (lane == 63) {
                float4* dst = reinterpret_cast<float4*>(sh_part[wave][j]);
                dst[0] = make_float4(red[0], red[1], red[2], red[3]);
                dst[1] = make_float4(red[4], red[5], red[6], red[7]);
                dst[2] = make_float4(red[8], red[9], red[10], red[11]);
                dst[3] = make_float4(red[12], 0.f, 0.f, 0.f);
            }
        }
        // the four waves' sums of every edge of the batch -> one record per (edge side, workgroup), fixed order
        __syncthreads();
        for (int i = threadIdx.x; i < nb * PW; i += NTH) {
            const int j = i / PW, v = i - j * PW;
            float t = (sh_part[0][j][v] + sh_part[1][j][v]) + (sh_part[2][j][v] + sh_part[3][j][v]);
            if constexpr (NWV == 8) t += (sh_part[4][j][v] + sh_part[5][j][v]) + (sh_part[6][j][v] + sh_part[7][j][v]);
            a.part_edge[((size_t)sh_es[j] * a.nslot + slot) * PW + v] = t;
        }
    }

    // ---- per-pixel log-depth gradient (+ Adam) and per-image partial sums ------------------------
    float pi[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) pi[k] = 0.f;
    float gd[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        // dX/dd = R cam = X - T
        gd[k] = g[k][0] * (X[k][0] - T0) + g[k][1] * (X[k][1] - T1) + g[k][2] * (X[k][2] - T2);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) pi[r * 3 + c] += g[k][r] * cam[k][c];
            pi[9 + r] += g[k][r];
            pi[12 + r] += g[k][r] * ed[k];
        }
    }
    if (active) {
        const size_t off = (size_t)img * a.maxA + p0;
        if (a.depth_grad) *reinterpret_cast<float4*>(a.depth_grad + off) = make_float4(gd[0], gd[1], gd[2], gd[3]);
        if (a.update) {
            float4 m4 = *reinterpret_cast<const float4*>(a.depth_m + off);
            float4 v4 = *reinterpret_cast<const float4*>(a.depth_v + off);
            float4 o;
            o.x = adam_update(dl[0], gd[0], m4.x, v4.x, a.adam);
            o.y = adam_update(dl[1], gd[1], m4.y, v4.y, a.adam);
            o.z = adam_update(dl[2], gd[2], m4.z, v4.z, a.adam);
            o.w = adam_update(dl[3], gd[3], m4.w, v4.w, a.adam);
            *reinterpret_cast<float4*>(a.depth + off) = o;
            *reinterpret_cast<float4*>(a.depth_m + off) = m4;
            *reinterpret_cast<float4*>(a.depth_v + off) = v4;
        }
    }
    float red[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) red[k] = a.use_dpp ? wave_sum_dpp(pi[k]) : wave_sum_shfl(pi[k]);
    __syncthreads();   // sh_part[.][0] is free again (the last batch's combine has been read)
    if (lane == 63) {
        float4* dst = reinterpret_cast<float4*>(sh_part[wave][0]);
        dst[0] = make_float4(red[0], red[1], red[2], red[3]);
        dst[1] = make_float4(red[4], red[5], red[6], red[7]);
        dst[2] = make_float4(red[8], red[9], red[10], red[11]);
        dst[3] = make_float4(red[12], red[13], red[14], 0.f);
    }
    __syncthreads();
    if (threadIdx.x < PW) {
        const int v = threadIdx.x;
        float t = (sh_part[0][0][v] + sh_part[1][0][v]) + (sh_part[2][0][v] + sh_part[3][0][v]);
        if constexpr (NWV == 8) t += (sh_part[4][0][v] + sh_part[5][0][v]) + (sh_part[6][0][v] + sh_part[7][0][v]);
        a.part_img[((size_t)img * a.nslot + slot) * PW + v] = t;
    }
}

// one-time re-layout at create: [E][maxA][3] (the reference's stacked pointmaps) -> [E][3][maxA]
__global__ __launch_bounds__(256) void aligner_planarize_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n_pix_total, int maxA) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_pix_total; i += (size_t)gridDim.x * 256) {
        const size_t e = i / maxA, p = i - e * maxA;
        const float x = in[i * 3], y = in[i * 3 + 1], z = in[i * 3 + 2];
        float* o = out + e * 3 * (size_t)maxA + p;
        o[0] = x; o[maxA] = y; o[2 * (size_t)maxA] = z;
    }
}

// one-time re-layout at create (LAY == 1): pred [E][maxA][3] + weights [E][maxA] of one side -> [E][maxAp / 256][x | y | z | w][256]
__global__ __launch_bounds__(256) void aligner_interleave_kernel(const float* __restrict__ pred, const float* __restrict__ wgt, float* __restrict__ out, int E, int maxA,
                                                                 int maxAp) {
    const size_t total = (size_t)E * maxAp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t e = i / maxAp;
        const int p = (int)(i - e * maxAp);
        float x = 0.f, y = 0.f, z = 0.f, w = 0.f;
        if (p < maxA) {
            const float* s = pred + (e * maxA + p) * 3;
            x = s[0]; y = s[1]; z = s[2];
            w = wgt[e * maxA + p];
        }
        float* o = out + e * 4 * (size_t)maxAp + (size_t)(p >> 8) * 1024 + (p & 255);
        o[0] = x; o[256] = y; o[512] = z; o[768] = w;
    }
}

// ---- fixed-order fp64 reduction of [entries][nslot][16] partial records --------------------------------
__global__ __launch_bounds__(256) void aligner_reduce_kernel(const float* __restrict__ part, double* __restrict__ out, int nslot) {
    __shared__ double sh[16][17];
    const int entry = blockIdx.x;
    const int val = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const float* base = part + (size_t)entry * nslot * PW;
    double acc = 0.0;
    for (int s = sl; s < nslot; s += 16) acc += (double)base[(size_t)s * PW + val];
    sh[sl][val] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][threadIdx.x];
        out[(size_t)entry * PW + threadIdx.x] = t;
    }
}

struct SmallView {
    int n, E;
    float* pw_poses;     // [E][8]
    float* pw_adaptors;  // [E][2] (updated when opt_adapt)
    float* im_poses;     // [n][7]
    float* im_focals;    // [n]
    float* im_pp;        // [n][2] (updated when opt_pp)
    const int* img_w; const int* img_h;
    float *pw_m, *pw_v, *imp_m, *imp_v, *foc_m, *foc_v, *pp_m, *pp_v, *pa_m, *pa_v;
    const double* red_edge;  // [2E][16]
    const double* red_img;   // [n][16]
    float* d_edge;           // [E][12]
    float* d_img;            // [n][16]
    double* scratch;         // [E][8]: gradient wrt P_e[0:7] from pass 1, and gs*s~ in slot 7
    float* loss_hist; int iter;
    float* g_pw; float* g_imp; float* g_foc; float* g_pp; float* g_pa;  // optional gradient export (tests)
    float base_scale, pw_break, focal_break;
    int norm_pw_scale, opt_poses, opt_focals, opt_pp, opt_adapt, update;
    AdamCoef adam;
};

D3R_DEV float pw_scale_factor(const SmallView& s, double mean_p7) {
    return s.norm_pw_scale ? expf(logf(s.base_scale) - (float)mean_p7) : 1.0f;
}

// fixed-order block sum (256 threads): wave butterfly in fp64, then the four wave totals in wave order
D3R_DEV double block_sum_f64(double v, double* sh4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
}

// single workgroup; E and n are a few hundred at most per call site (E <= 600). Every sum over edges is
// a block reduction (thread t owns edges t, t+256, ...): no thread-0 serial chains of dependent L2 loads.
__global__ __launch_bounds__(256) void aligner_small_kernel(SmallView s) {
    __shared__ double sh4[4];
    const int tid = threadIdx.x;
    if (s.update || s.g_pw) {
        // mean of P7 BEFORE the update defines the s~ the gradients were taken at
        double acc = 0.0;
        for (int e = tid; e < s.E; e += 256) acc += (double)s.pw_poses[e * 8 + 7];
        const double mean7 = block_sum_f64(acc, sh4) / (double)s.E;
        const float nf = pw_scale_factor(s, mean7);
        // pass 1: dL/ds~_e * s~_e (kept in scratch) and its sum over edges; loss = sum of the per-side partials
        double gsum = 0.0, lsum = 0.0;
        for (int e = tid; e < s.E; e += 256) {
            const float* P = s.pw_poses + e * 8;
            float R[9];
            quat_to_rotmat(P, R);
            const float st = expf(P[7]) * nf;
            float adapt[3];
            {
                const float a0 = s.pw_adaptors[e * 2], a1 = s.pw_adaptors[e * 2 + 1];
                const float mean = s.norm_pw_scale ? (2.f * a0 + a1) / 3.f : 0.f;
                adapt[0] = adapt[1] = expf((a0 - mean) / s.pw_break);
                adapt[2] = expf((a1 - mean) / s.pw_break);
            }
            double GM[12], gP[7], gs;
            for (int k = 0; k < 12; ++k) GM[k] = s.red_edge[(size_t)(2 * e) * PW + k] + s.red_edge[(size_t)(2 * e + 1) * PW + k];
            edge_chain(P, R, st, adapt, GM, gP, gs);
            for (int k = 0; k < 7; ++k) s.scratch[(size_t)e * 8 + k] = gP[k];   // reused by pass 2 (same thread, same e)
            s.scratch[(size_t)e * 8 + 7] = gs * (double)st;
            gsum += gs * (double)st;
            lsum += s.red_edge[(size_t)(2 * e) * PW + 12] + s.red_edge[(size_t)(2 * e + 1) * PW + 12];
        }
        const double sum_gs = block_sum_f64(gsum, sh4);
        const double loss = block_sum_f64(lsum, sh4);
        if (tid == 0 && s.loss_hist) s.loss_hist[s.iter] = (float)loss;
        // pass 2: the scale gradient needs the sum over ALL edges (norm_pw_scale couples them); then Adam on pairwise poses
        for (int e = tid; e < s.E; e += 256) {
            float* P = s.pw_poses + e * 8;
            double gP[8];
            for (int k = 0; k < 7; ++k) gP[k] = s.scratch[(size_t)e * 8 + k];
            gP[7] = s.scratch[(size_t)e * 8 + 7] - (s.norm_pw_scale ? sum_gs / (double)s.E : 0.0);
            if (s.g_pw)
                for (int k = 0; k < 8; ++k) s.g_pw[e * 8 + k] = (float)gP[k];
            if (s.g_pa || (s.update && s.opt_adapt)) {
                // pairwise adaptors (base_opt.py:143-149: adapt = exp((A - mean A) / pw_break), A = (a0, a0, a1)): M_e[r][c] =
                // s~ R[r][c] adapt_c, so dL/d adapt_c = s~ sum_r GM[r][c] R[r][c]; then through exp and the mean-centring
                float R[9];
                quat_to_rotmat(P, R);
                const float a0 = s.pw_adaptors[e * 2], a1 = s.pw_adaptors[e * 2 + 1];
                const float mean = s.norm_pw_scale ? (2.f * a0 + a1) / 3.f : 0.f;
                const double ad[3] = {exp((double)(a0 - mean) / s.pw_break), exp((double)(a0 - mean) / s.pw_break), exp((double)(a1 - mean) / s.pw_break)};
                const double st = exp((double)P[7]) * (double)nf;
                double gA[3], tot = 0.0;
                for (int c = 0; c < 3; ++c) {
                    double gc = 0.0;
                    for (int r = 0; r < 3; ++r)
                        gc += (s.red_edge[(size_t)(2 * e) * PW + r * 4 + c] + s.red_edge[(size_t)(2 * e + 1) * PW + r * 4 + c]) * (double)R[r * 3 + c];
                    gA[c] = gc * st * ad[c] / (double)s.pw_break;      // dL/d((A_c - mean) / b) chain through exp
                    tot += gA[c];
                }
                if (s.norm_pw_scale)
                    for (int c = 0; c < 3; ++c) gA[c] -= tot / 3.0;
                const double g0 = gA[0] + gA[1], g1 = gA[2];
                if (s.g_pa) { s.g_pa[e * 2] = (float)g0; s.g_pa[e * 2 + 1] = (float)g1; }
                if (s.update && s.opt_adapt) {
                    s.pw_adaptors[e * 2] = adam_update(a0, (float)g0, s.pa_m[e * 2], s.pa_v[e * 2], s.adam);
                    s.pw_adaptors[e * 2 + 1] = adam_update(a1, (float)g1, s.pa_m[e * 2 + 1], s.pa_v[e * 2 + 1], s.adam);
                }
            }
            if (s.update)
                for (int k = 0; k < 8; ++k) P[k] = adam_update(P[k], (float)gP[k], s.pw_m[e * 8 + k], s.pw_v[e * 8 + k], s.adam);
        }
        // image poses / focals
        for (int i = tid; i < s.n; i += 256) {
            float* P = s.im_poses + i * 7;
            float R[9];
            quat_to_rotmat(P, R);
            double G[9], GT[3], gP[7], gf;
            for (int k = 0; k < 9; ++k) G[k] = s.red_img[(size_t)i * PW + k];
            for (int k = 0; k < 3; ++k) GT[k] = s.red_img[(size_t)i * PW + 9 + k];
            image_chain(P, R, s.focal_break, G, GT, gP, gf);
            if (s.g_imp)
                for (int k = 0; k < 7; ++k) s.g_imp[i * 7 + k] = (float)gP[k];
            if (s.g_foc) s.g_foc[i] = (float)gf;
            if (s.update && s.opt_poses)
                for (int k = 0; k < 7; ++k) P[k] = adam_update(P[k], (float)gP[k], s.imp_m[i * 7 + k], s.imp_v[i * 7 + k], s.adam);
            const float focal_param_old = s.im_focals[i];
            if (s.update && s.opt_focals) s.im_focals[i] = adam_update(s.im_focals[i], (float)gf, s.foc_m[i], s.foc_v[i], s.adam);
            if (s.g_pp || (s.update && s.opt_pp)) {
                // principal point (optimizer.py:141-142: pp = (W/2, H/2) + 10 im_pp): dX/dppx = -exp(d)/F R[:,0], so
                // dL/d im_pp = -(10 / F) (R^T S)_{0,1} with S = sum_p g_p exp(d_p), the record's last three sums
                const double F = exp((double)focal_param_old / (double)s.focal_break);   // the focal the gradients were taken at
                const double S0 = s.red_img[(size_t)i * PW + 12], S1 = s.red_img[(size_t)i * PW + 13], S2 = s.red_img[(size_t)i * PW + 14];
                const double gx = -(10.0 / F) * ((double)R[0] * S0 + (double)R[3] * S1 + (double)R[6] * S2);
                const double gy = -(10.0 / F) * ((double)R[1] * S0 + (double)R[4] * S1 + (double)R[7] * S2);
                if (s.g_pp) { s.g_pp[i * 2] = (float)gx; s.g_pp[i * 2 + 1] = (float)gy; }
                if (s.update && s.opt_pp) {
                    s.im_pp[i * 2] = adam_update(s.im_pp[i * 2], (float)gx, s.pp_m[i * 2], s.pp_v[i * 2], s.adam);
                    s.im_pp[i * 2 + 1] = adam_update(s.im_pp[i * 2 + 1], (float)gy, s.pp_m[i * 2 + 1], s.pp_v[i * 2 + 1], s.adam);
                }
            }
        }
        __syncthreads();   // the updated P7 values are read by other threads below
    }
    // ---- derived quantities for the next main pass ---------------------------------------------------
    double acc2 = 0.0;
    for (int e = tid; e < s.E; e += 256) acc2 += (double)s.pw_poses[e * 8 + 7];
    const float nf = pw_scale_factor(s, block_sum_f64(acc2, sh4) / (double)s.E);
    for (int e = tid; e < s.E; e += 256) {
        const float* P = s.pw_poses + e * 8;
        float R[9];
        quat_to_rotmat(P, R);
        const float st = expf(P[7]) * nf;
        const float a0 = s.pw_adaptors[e * 2], a1 = s.pw_adaptors[e * 2 + 1];
        const float mean = s.norm_pw_scale ? (2.f * a0 + a1) / 3.f : 0.f;
        const float ad[3] = {expf((a0 - mean) / s.pw_break), expf((a0 - mean) / s.pw_break), expf((a1 - mean) / s.pw_break)};
        float* M = s.d_edge + e * 12;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) M[r * 4 + c] = st * R[r * 3 + c] * ad[c];
            M[r * 4 + 3] = st * signed_expm1f(P[4 + r]);
        }
    }
    for (int i = tid; i < s.n; i += 256) {
        const float* P = s.im_poses + i * 7;
        float* D = s.d_img + i * 16;
        quat_to_rotmat(P, D);
        D[9] = signed_expm1f(P[4]);
        D[10] = signed_expm1f(P[5]);
        D[11] = signed_expm1f(P[6]);
        D[12] = expf(s.im_focals[i] / s.focal_break);
        D[13] = 0.5f * (float)s.img_w[i] + 10.f * s.im_pp[i * 2];
        D[14] = 0.5f * (float)s.img_h[i] + 10.f * s.im_pp[i * 2 + 1];
        D[15] = 0.f;
    }
}


// ---- the same step with ONE edge and ONE image per thread (E, n <= NT) ----------------------------------------------------------
// The generic kernel above walks edges in strided loops, so everything that crosses a block sum goes through memory (scratch, the
// updated poses) and every phase starts with a dependent round trip to L2 / HBM: ~7 of them, 24.6 us per iteration at E = 190 on
// MI355X (9 % of the iteration). Here a thread owns its edge for the whole kernel: all operands are requested up front in one batch,
// gradients and the updated pose stay in registers across the three block sums, and the derived matrices are formed from registers.
template <int NT>
D3R_DEV void block_sum3_f64(double (&v)[3], double (*sh)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) sh[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double t = 0.0;
        for (int w = 0; w < NT / 64; ++w) t += sh[w][k];   // fixed order
        v[k] = t;
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void aligner_small1_kernel(SmallView s) {
    __shared__ double sh[NT / 64][3];
    const int tid = threadIdx.x;
    const bool he = tid < s.E, hi = tid < s.n;
    const bool grads = s.update || s.g_pw;
    const int e = he ? tid : 0, i = hi ? tid : 0;

    // ---- one batch of loads ------------------------------------------------------------------------------------------------
    float P[8], ad0, ad1, pm[8], pv[8], am[2], av[2];
    double GM[12], lside = 0.0;
    {
        const float4 p0 = *reinterpret_cast<const float4*>(s.pw_poses + e * 8), p1 = *reinterpret_cast<const float4*>(s.pw_poses + e * 8 + 4);
        P[0] = p0.x; P[1] = p0.y; P[2] = p0.z; P[3] = p0.w; P[4] = p1.x; P[5] = p1.y; P[6] = p1.z; P[7] = p1.w;
        const float2 a2 = *reinterpret_cast<const float2*>(s.pw_adaptors + e * 2);
        ad0 = a2.x; ad1 = a2.y;
    }
    if (s.update) {
        const float4 m0 = *reinterpret_cast<const float4*>(s.pw_m + e * 8), m1 = *reinterpret_cast<const float4*>(s.pw_m + e * 8 + 4);
        const float4 v0 = *reinterpret_cast<const float4*>(s.pw_v + e * 8), v1 = *reinterpret_cast<const float4*>(s.pw_v + e * 8 + 4);
        pm[0] = m0.x; pm[1] = m0.y; pm[2] = m0.z; pm[3] = m0.w; pm[4] = m1.x; pm[5] = m1.y; pm[6] = m1.z; pm[7] = m1.w;
        pv[0] = v0.x; pv[1] = v0.y; pv[2] = v0.z; pv[3] = v0.w; pv[4] = v1.x; pv[5] = v1.y; pv[6] = v1.z; pv[7] = v1.w;
        if (s.opt_adapt) { am[0] = s.pa_m[e * 2]; am[1] = s.pa_m[e * 2 + 1]; av[0] = s.pa_v[e * 2]; av[1] = s.pa_v[e * 2 + 1]; }
    }
    if (grads) {
        const double* r0 = s.red_edge + (size_t)(2 * e) * PW;
        const double* r1 = r0 + PW;
#pragma unroll
        for (int k = 0; k < 12; ++k) GM[k] = r0[k] + r1[k];
        lside = r0[12] + r1[12];
    }
    float Q[7], foc, pp0, pp1, qm[7], qv[7], fm = 0.f, fv = 0.f, ppm[2], ppv[2];
    double RI[15];
#pragma unroll
    for (int k = 0; k < 7; ++k) Q[k] = s.im_poses[i * 7 + k];
    foc = s.im_focals[i];
    pp0 = s.im_pp[i * 2]; pp1 = s.im_pp[i * 2 + 1];
    const int iw = s.img_w[i], ih = s.img_h[i];
    if (s.update) {
#pragma unroll
        for (int k = 0; k < 7; ++k) { qm[k] = s.imp_m[i * 7 + k]; qv[k] = s.imp_v[i * 7 + k]; }
        fm = s.foc_m[i]; fv = s.foc_v[i];
        if (s.opt_pp) { ppm[0] = s.pp_m[i * 2]; ppm[1] = s.pp_m[i * 2 + 1]; ppv[0] = s.pp_v[i * 2]; ppv[1] = s.pp_v[i * 2 + 1]; }
    }
    if (grads) {
#pragma unroll
        for (int k = 0; k < 15; ++k) RI[k] = s.red_img[(size_t)i * PW + k];
    }

    if (grads) {
        // ---- edges: chain rule at the parameters the main pass used ----------------------------------------------------------
        double v3[3] = {he ? (double)P[7] : 0.0, 0.0, 0.0};
        block_sum3_f64<NT>(v3, sh);
        const float nf = pw_scale_factor(s, v3[0] / (double)s.E);
        float R[9];
        quat_to_rotmat(P, R);
        const float st = expf(P[7]) * nf;
        const float amean = s.norm_pw_scale ? (2.f * ad0 + ad1) / 3.f : 0.f;
        const float adapt[3] = {expf((ad0 - amean) / s.pw_break), expf((ad0 - amean) / s.pw_break), expf((ad1 - amean) / s.pw_break)};
        double gP[8], gs;
        edge_chain(P, R, st, adapt, GM, gP, gs);
        gP[7] = gs * (double)st;
        double w3[3] = {he ? gP[7] : 0.0, he ? lside : 0.0, 0.0};
        block_sum3_f64<NT>(w3, sh);
        if (tid == 0 && s.loss_hist) s.loss_hist[s.iter] = (float)w3[1];
        if (s.norm_pw_scale) gP[7] -= w3[0] / (double)s.E;
        if (he) {
            if (s.g_pw)
                for (int k = 0; k < 8; ++k) s.g_pw[e * 8 + k] = (float)gP[k];
            if (s.g_pa || (s.update && s.opt_adapt)) {
                const double ad[3] = {exp((double)(ad0 - amean) / s.pw_break), exp((double)(ad0 - amean) / s.pw_break), exp((double)(ad1 - amean) / s.pw_break)};
                const double std_ = exp((double)P[7]) * (double)nf;
                double gA[3], tot = 0.0;
                for (int c = 0; c < 3; ++c) {
                    double gc = 0.0;
                    for (int r = 0; r < 3; ++r) gc += GM[r * 4 + c] * (double)R[r * 3 + c];
                    gA[c] = gc * std_ * ad[c] / (double)s.pw_break;
                    tot += gA[c];
                }
                if (s.norm_pw_scale)
                    for (int c = 0; c < 3; ++c) gA[c] -= tot / 3.0;
                const double g0 = gA[0] + gA[1], g1 = gA[2];
                if (s.g_pa) { s.g_pa[e * 2] = (float)g0; s.g_pa[e * 2 + 1] = (float)g1; }
                if (s.update && s.opt_adapt) {
                    ad0 = adam_update(ad0, (float)g0, am[0], av[0], s.adam);
                    ad1 = adam_update(ad1, (float)g1, am[1], av[1], s.adam);
                    *reinterpret_cast<float2*>(s.pw_adaptors + e * 2) = make_float2(ad0, ad1);
                    s.pa_m[e * 2] = am[0]; s.pa_m[e * 2 + 1] = am[1]; s.pa_v[e * 2] = av[0]; s.pa_v[e * 2 + 1] = av[1];
                }
            }
            if (s.update) {
#pragma unroll
                for (int k = 0; k < 8; ++k) P[k] = adam_update(P[k], (float)gP[k], pm[k], pv[k], s.adam);
                float4* d = reinterpret_cast<float4*>(s.pw_poses + e * 8);
                d[0] = make_float4(P[0], P[1], P[2], P[3]); d[1] = make_float4(P[4], P[5], P[6], P[7]);
                float4* dm = reinterpret_cast<float4*>(s.pw_m + e * 8);
                dm[0] = make_float4(pm[0], pm[1], pm[2], pm[3]); dm[1] = make_float4(pm[4], pm[5], pm[6], pm[7]);
                float4* dv = reinterpret_cast<float4*>(s.pw_v + e * 8);
                dv[0] = make_float4(pv[0], pv[1], pv[2], pv[3]); dv[1] = make_float4(pv[4], pv[5], pv[6], pv[7]);
            }
        }
        // ---- images ------------------------------------------------------------------------------------------------------------
        if (hi) {
            float R2[9];
            quat_to_rotmat(Q, R2);
            double gQ[7], gf;
            image_chain(Q, R2, s.focal_break, RI, RI + 9, gQ, gf);
            if (s.g_imp)
                for (int k = 0; k < 7; ++k) s.g_imp[i * 7 + k] = (float)gQ[k];
            if (s.g_foc) s.g_foc[i] = (float)gf;
            if (s.g_pp || (s.update && s.opt_pp)) {
                const double F = exp((double)foc / (double)s.focal_break);   // the focal the gradients were taken at
                const double gx = -(10.0 / F) * ((double)R2[0] * RI[12] + (double)R2[3] * RI[13] + (double)R2[6] * RI[14]);
                const double gy = -(10.0 / F) * ((double)R2[1] * RI[12] + (double)R2[4] * RI[13] + (double)R2[7] * RI[14]);
                if (s.g_pp) { s.g_pp[i * 2] = (float)gx; s.g_pp[i * 2 + 1] = (float)gy; }
                if (s.update && s.opt_pp) {
                    pp0 = adam_update(pp0, (float)gx, ppm[0], ppv[0], s.adam);
                    pp1 = adam_update(pp1, (float)gy, ppm[1], ppv[1], s.adam);
                    s.im_pp[i * 2] = pp0; s.im_pp[i * 2 + 1] = pp1;
                    s.pp_m[i * 2] = ppm[0]; s.pp_m[i * 2 + 1] = ppm[1]; s.pp_v[i * 2] = ppv[0]; s.pp_v[i * 2 + 1] = ppv[1];
                }
            }
            if (s.update && s.opt_poses) {
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    Q[k] = adam_update(Q[k], (float)gQ[k], qm[k], qv[k], s.adam);
                    s.im_poses[i * 7 + k] = Q[k]; s.imp_m[i * 7 + k] = qm[k]; s.imp_v[i * 7 + k] = qv[k];
                }
            }
            if (s.update && s.opt_focals) {
                foc = adam_update(foc, (float)gf, fm, fv, s.adam);
                s.im_focals[i] = foc; s.foc_m[i] = fm; s.foc_v[i] = fv;
            }
        }
    }
    // ---- derived quantities for the next main pass, from the registers ---------------------------------------------------------
    double u3[3] = {he ? (double)P[7] : 0.0, 0.0, 0.0};
    block_sum3_f64<NT>(u3, sh);
    const float nf2 = pw_scale_factor(s, u3[0] / (double)s.E);
    if (he) {
        float R[9];
        quat_to_rotmat(P, R);
        const float st = expf(P[7]) * nf2;
        const float mean = s.norm_pw_scale ? (2.f * ad0 + ad1) / 3.f : 0.f;
        const float ad[3] = {expf((ad0 - mean) / s.pw_break), expf((ad0 - mean) / s.pw_break), expf((ad1 - mean) / s.pw_break)};
        float4* M = reinterpret_cast<float4*>(s.d_edge + e * 12);
#pragma unroll
        for (int r = 0; r < 3; ++r)
            M[r] = make_float4(st * R[r * 3] * ad[0], st * R[r * 3 + 1] * ad[1], st * R[r * 3 + 2] * ad[2], st * signed_expm1f(P[4 + r]));
    }
    if (hi) {
        float D[16];
        quat_to_rotmat(Q, D);
        D[9] = signed_expm1f(Q[4]); D[10] = signed_expm1f(Q[5]); D[11] = signed_expm1f(Q[6]);
        D[12] = expf(foc / s.focal_break);
        D[13] = 0.5f * (float)iw + 10.f * pp0;
        D[14] = 0.5f * (float)ih + 10.f * pp1;
        D[15] = 0.f;
        float4* dst = reinterpret_cast<float4*>(s.d_img + i * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_float4(D[4 * k], D[4 * k + 1], D[4 * k + 2], D[4 * k + 3]);
    }
}

static void launch_small(const SmallView& s, hipStream_t st, bool generic) {
    // generic (D3R_ALIGNER_OPT_GENERIC_SMALL) pins the strided-loop kernel, which takes any E, n: parity tests run both
    const int need = s.E > s.n ? s.E : s.n;
    if (!generic && need <= 256) hipLaunchKernelGGL(aligner_small1_kernel<256>, dim3(1), dim3(256), 0, st, s);
    else if (!generic && need <= 1024) hipLaunchKernelGGL(aligner_small1_kernel<1024>, dim3(1), dim3(1024), 0, st, s);
    else hipLaunchKernelGGL(aligner_small_kernel, dim3(1), dim3(256), 0, st, s);
}

}  // namespace d3r

// =====================================================================================================
// C-ABI (include/dust3r_hip.h)
// =====================================================================================================
#include <cstdint>
#include <vector>
#include <new>
#include "../../include/dust3r_hip.h"

using namespace d3r;

struct d3r_aligner {
    int n = 0, E = 0, maxA = 0, nslot = 0;
    int nwv = 4;              // waves per workgroup of the main kernel (D3R_ALIGNER_NWV=8 at creation: 2048-pixel chunks)
    int probe = 0;            // D3R_ALIGNER_PROBE at creation (measurement aid, results INVALID)
    int layout = 0;           // D3R_ALIGNER_LAYOUT at creation: 1 = block-interleaved stream copy (aligner_main_kernel LAY)
    float* inter = nullptr;   // [2][E][maxAp / 256][4][256]
    int maxAp = 0;
    std::vector<int> h_w, h_h, h_area;
    int *d_w = nullptr, *d_h = nullptr, *d_area = nullptr, *d_adj_off = nullptr, *d_adj_es = nullptr;
    const float *pred[2] = {nullptr, nullptr}, *wgt[2] = {nullptr, nullptr};
    float* planar = nullptr;  // [2][E][3][maxA] re-laid-out copies of pred_i / pred_j
    float *pw_poses = nullptr, *pw_adaptors = nullptr, *im_poses = nullptr, *im_depth = nullptr, *im_focals = nullptr, *im_pp = nullptr;
    float* state = nullptr;  // one arena: Adam moments, derived matrices, partials
    float *depth_m, *depth_v, *pw_m, *pw_v, *imp_m, *imp_v, *foc_m, *foc_v, *pp_m, *pp_v, *pa_m, *pa_v, *d_edge, *d_img, *part_edge, *part_img, *loss_hist, *g_scratch;
    double *red_edge, *red_img, *scratch;
    size_t state_bytes = 0;
    float base_scale = 0.5f, pw_break = 20.f, focal_break = 20.f, inv_area[2] = {0, 0};
    int l2 = 0, norm_pw_scale = 1, opt_poses = 1, opt_focals = 1, opt_pp = 0, opt_adapt = 0, use_dpp = 1;
    long step = 0;
    // Image range (d3r_aligner_set_image_range; default: all images): the main kernel runs over the images [img0, img0 + imgc) only. A partial record belongs to
    // exactly one image -- the one whose pixels an edge side is compared with -- so with one process per GPU owning a contiguous range of images, the reduced
    // sums of the other images' records are exact zeros here, and an all-reduce (sum) of the reduced buffer across the ranks between d3r_aligner_step_begin and
    // d3r_aligner_step_end gives every rank bit for bit the sums of the single-GPU iteration; the pose / focal step then runs replicated.
    int img0 = 0, imgc = -1;
    bool part_clear_pending = false;
    bool reset_pending = false;   // D3R_ALIGNER_OPT_RESET_ADAM: cleared on the next run's stream
    bool generic_small = false;   // D3R_ALIGNER_OPT_GENERIC_SMALL
    int loss_cap = 0;
    // create() enqueues its clear + re-layout on the caller's stream and returns without synchronising: a later call on a DIFFERENT
    // stream first waits for this event (same stream: already ordered, no wait issued)
    hipStream_t create_stream = nullptr;
    hipEvent_t ev_ready = nullptr;
};

// orders `st` behind the work d3r_aligner_create left in flight on its own stream
static void aligner_wait_ready(d3r_aligner* a, hipStream_t st) {
    if (a->ev_ready && st != a->create_stream) (void)hipStreamWaitEvent(st, a->ev_ready, 0);
}

#define HIPCHK(x)                                  \
    do {                                           \
        hipError_t e_ = (x);                       \
        if (e_ != hipSuccess) return (int)e_ + 1000; \
    } while (0)

extern "C" int d3r_aligner_create(d3r_aligner** out, int n_imgs, int n_edges, const int* ei, const int* ej, const int* img_h,
                                  const int* img_w, int max_area, const float* pred_i, const float* pred_j, const float* w_i,
                                  const float* w_j, float* pw_poses, float* pw_adaptors, float* im_poses, float* im_depth,
                                  float* im_focals, float* im_pp, float base_scale, float pw_break, float focal_break, int dist_l2,
                                  int norm_pw_scale, int opt_poses, int opt_focals, int max_iters, void* stream) {
    if (!out || n_imgs <= 0 || n_edges <= 0 || max_area <= 0 || max_area % 4 != 0) return D3R_ERR_INVALID;
    // vector accesses: pw_poses rows as 2 x float4, im_depth / pred / weight rows as float4, pw_adaptors rows as float2
    if (((uintptr_t)pw_poses & 15) || ((uintptr_t)im_depth & 15) || ((uintptr_t)pw_adaptors & 7)) return D3R_ERR_INVALID;
    d3r_aligner* a = new (std::nothrow) d3r_aligner();
    if (!a) return D3R_ERR_ALLOC;
    a->n = n_imgs; a->E = n_edges; a->maxA = max_area;
    { const char* e = probe_env("D3R_ALIGNER_NWV"); a->nwv = (e && e[0] == '8') ? 8 : 4; }
    { const char* e = probe_env("D3R_ALIGNER_PROBE"); a->probe = (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0; }
    // default since round 5: the block-interleaved copy (same-process A/B, tools/aligner_probe.py, profiles/r05_k: 3998 -> 4085 it/s at 190 edges, 2142 -> 2161 at 380,
    // bit-identical losses); D3R_ALIGNER_LAYOUT=0: the planar copy + the caller's weight rows (rounds 1-4)
    { const char* e = probe_env("D3R_ALIGNER_LAYOUT"); a->layout = (!(e && e[0] == '0') && a->nwv == 4 && !a->probe) ? 1 : 0; }
    a->maxAp = (max_area + 255) / 256 * 256;
    a->nslot = cdiv(max_area, a->nwv * 64 * PPT);
    a->h_w.assign(img_w, img_w + n_imgs);
    a->h_h.assign(img_h, img_h + n_imgs);
    a->h_area.resize(n_imgs);
    double ta[2] = {0, 0};
    for (int i = 0; i < n_imgs; ++i) {
        a->h_area[i] = img_h[i] * img_w[i];
        if (a->h_area[i] > max_area || a->h_area[i] % 4 != 0) { delete a; return D3R_ERR_INVALID; }
    }
    // adjacency (CSR by projecting image): side 0 entries project onto ei, side 1 onto ej
    std::vector<int> off(n_imgs + 1, 0), es(2 * (size_t)n_edges);
    for (int e = 0; e < n_edges; ++e) {
        if (ei[e] < 0 || ei[e] >= n_imgs || ej[e] < 0 || ej[e] >= n_imgs) { delete a; return D3R_ERR_INVALID; }
        off[ei[e] + 1]++; off[ej[e] + 1]++;
        ta[0] += a->h_area[ei[e]]; ta[1] += a->h_area[ej[e]];
    }
    for (int i = 0; i < n_imgs; ++i) off[i + 1] += off[i];
    {
        std::vector<int> cur(off.begin(), off.end() - 1);
        for (int e = 0; e < n_edges; ++e) { es[cur[ei[e]]++] = 2 * e; es[cur[ej[e]]++] = 2 * e + 1; }
    }
    a->inv_area[0] = (float)(1.0 / ta[0]);
    a->inv_area[1] = (float)(1.0 / ta[1]);
    a->wgt[0] = w_i; a->wgt[1] = w_j;   // (pred_i / pred_j are copied into a planar layout below)
    a->pw_poses = pw_poses; a->pw_adaptors = pw_adaptors; a->im_poses = im_poses; a->im_depth = im_depth;
    a->im_focals = im_focals; a->im_pp = im_pp;
    a->base_scale = base_scale; a->pw_break = pw_break; a->focal_break = focal_break;
    a->l2 = dist_l2; a->norm_pw_scale = norm_pw_scale; a->opt_poses = opt_poses; a->opt_focals = opt_focals;
    a->loss_cap = max_iters > 0 ? max_iters : 1;

    const size_t nA = (size_t)n_imgs * max_area;
    size_t fl = 0;
    auto take = [&](size_t cnt) { size_t o = fl; fl += (cnt + 3) & ~(size_t)3; return o; };
    const size_t o_dm = take(nA), o_dv = take(nA), o_pwm = take((size_t)n_edges * 8), o_pwv = take((size_t)n_edges * 8),
                 o_im = take((size_t)n_imgs * 7), o_iv = take((size_t)n_imgs * 7), o_fm = take(n_imgs), o_fv = take(n_imgs),
                 o_pm = take((size_t)n_imgs * 2), o_pv = take((size_t)n_imgs * 2), o_am = take((size_t)n_edges * 2), o_av = take((size_t)n_edges * 2),
                 o_de = take((size_t)n_edges * 12), o_di = take((size_t)n_imgs * 16),
                 o_pe = take((size_t)2 * n_edges * a->nslot * PW), o_pi = take((size_t)n_imgs * a->nslot * PW),
                 o_lh = take(a->loss_cap), o_gs = take((size_t)n_edges * 8);
    const size_t dbl = ((size_t)2 * n_edges * PW + (size_t)n_imgs * PW + (size_t)n_edges * 8 + 8);
    a->state_bytes = fl * sizeof(float) + dbl * sizeof(double) + 64;
    if (hipMalloc((void**)&a->state, a->state_bytes) != hipSuccess) { delete a; return D3R_ERR_ALLOC; }
    hipStream_t st = (hipStream_t)stream;   // the clear and the re-layout below are ordered on the caller's stream, like every later call
    // D3R_ALIGNER_POISON=1 (stress harness, tools/c4_stress.py): every allocation of the handle is filled with 0xFF bytes (fp32 / fp64 NaN,
    // int -1) before its real initialisation, so that any read of a byte the create path failed to initialise shows up as NaN / a fault
    static const bool poison = [] { const char* e = probe_env("D3R_ALIGNER_POISON"); return e && e[0] == '1'; }();
    if (poison && hipMemsetAsync(a->state, 0xFF, a->state_bytes, st) != hipSuccess) { (void)hipFree(a->state); delete a; return D3R_ERR_LAUNCH; }
    if (hipMemsetAsync(a->state, 0, a->state_bytes, st) != hipSuccess) { (void)hipFree(a->state); delete a; return D3R_ERR_LAUNCH; }
    float* b = a->state;
    a->depth_m = b + o_dm; a->depth_v = b + o_dv; a->pw_m = b + o_pwm; a->pw_v = b + o_pwv; a->imp_m = b + o_im;
    a->imp_v = b + o_iv; a->foc_m = b + o_fm; a->foc_v = b + o_fv; a->pp_m = b + o_pm; a->pp_v = b + o_pv; a->pa_m = b + o_am; a->pa_v = b + o_av; a->d_edge = b + o_de; a->d_img = b + o_di;
    a->part_edge = b + o_pe; a->part_img = b + o_pi; a->loss_hist = b + o_lh; a->g_scratch = b + o_gs;
    double* db = reinterpret_cast<double*>(reinterpret_cast<char*>(b) + ((fl * sizeof(float) + 63) & ~(size_t)63));
    a->red_edge = db; a->red_img = db + (size_t)2 * n_edges * PW; a->scratch = a->red_img + (size_t)n_imgs * PW;

    const size_t ib = (3 * (size_t)n_imgs + (n_imgs + 1) + 2 * (size_t)n_edges) * sizeof(int);
    if (hipMalloc((void**)&a->d_w, ib) != hipSuccess) { (void)hipFree(a->state); delete a; return D3R_ERR_ALLOC; }
    a->d_h = a->d_w + n_imgs; a->d_area = a->d_h + n_imgs; a->d_adj_off = a->d_area + n_imgs; a->d_adj_es = a->d_adj_off + n_imgs + 1;
    if (poison) { (void)hipMemsetAsync(a->d_w, 0xFF, ib, st); (void)hipStreamSynchronize(st); }
    // blocking copies (pageable host vectors): complete in device memory when they return, whatever stream the caller works on
    (void)hipMemcpy(a->d_w, a->h_w.data(), n_imgs * sizeof(int), hipMemcpyHostToDevice);
    (void)hipMemcpy(a->d_h, a->h_h.data(), n_imgs * sizeof(int), hipMemcpyHostToDevice);
    (void)hipMemcpy(a->d_area, a->h_area.data(), n_imgs * sizeof(int), hipMemcpyHostToDevice);
    (void)hipMemcpy(a->d_adj_off, off.data(), (n_imgs + 1) * sizeof(int), hipMemcpyHostToDevice);
    (void)hipMemcpy(a->d_adj_es, es.data(), 2 * (size_t)n_edges * sizeof(int), hipMemcpyHostToDevice);
    if (a->layout != 1) {
        const size_t npix = (size_t)n_edges * max_area;
        if (hipMalloc((void**)&a->planar, 2 * npix * 3 * sizeof(float)) != hipSuccess) { (void)hipFree(a->state); (void)hipFree(a->d_w); delete a; return D3R_ERR_ALLOC; }
        const int grid = (int)((npix + 255) / 256 < 65536 ? (npix + 255) / 256 : 65536);
        if (poison) (void)hipMemsetAsync(a->planar, 0xFF, 2 * npix * 3 * sizeof(float), st);   // (layout 1: every float of the interleaved copy is written by its kernel, padding included)
        hipLaunchKernelGGL(aligner_planarize_kernel, dim3(grid), dim3(256), 0, st, pred_i, a->planar, npix, max_area);
        hipLaunchKernelGGL(aligner_planarize_kernel, dim3(grid), dim3(256), 0, st, pred_j, a->planar + npix * 3, npix, max_area);
        if (hipGetLastError() != hipSuccess) { (void)hipFree(a->planar); (void)hipFree(a->state); (void)hipFree(a->d_w); delete a; return D3R_ERR_LAUNCH; }
        a->pred[0] = a->planar; a->pred[1] = a->planar + npix * 3;
    }
    if (a->layout == 1) {
        const size_t per_side = (size_t)n_edges * 4 * a->maxAp;
        if (hipMalloc((void**)&a->inter, 2 * per_side * sizeof(float)) != hipSuccess) { (void)hipFree(a->planar); (void)hipFree(a->state); (void)hipFree(a->d_w); delete a; return D3R_ERR_ALLOC; }
        const size_t total = (size_t)n_edges * a->maxAp;
        const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        hipLaunchKernelGGL(aligner_interleave_kernel, dim3(grid), dim3(256), 0, st, pred_i, w_i, a->inter, n_edges, max_area, a->maxAp);
        hipLaunchKernelGGL(aligner_interleave_kernel, dim3(grid), dim3(256), 0, st, pred_j, w_j, a->inter + per_side, n_edges, max_area, a->maxAp);
        if (hipGetLastError() != hipSuccess) { (void)hipFree(a->inter); (void)hipFree(a->planar); (void)hipFree(a->state); (void)hipFree(a->d_w); delete a; return D3R_ERR_LAUNCH; }
    }
    a->create_stream = st;
    if (hipEventCreateWithFlags(&a->ev_ready, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(a->ev_ready, st);
    else a->ev_ready = nullptr;
    *out = a;
    return D3R_OK;
}

extern "C" int d3r_aligner_destroy(d3r_aligner* a) {
    if (!a) return D3R_OK;
    if (a->ev_ready) (void)hipEventDestroy(a->ev_ready);
    (void)hipFree(a->planar);
    if (a->inter) (void)hipFree(a->inter);
    (void)hipFree(a->state);
    (void)hipFree(a->d_w);
    delete a;
    return D3R_OK;
}

static AdamCoef adam_coef(double lr, long step) {
    const double b1 = 0.9, b2 = 0.9;  // base_opt.py:337 betas=(0.9, 0.9)
    AdamCoef c;
    c.b1 = (float)b1; c.b2 = (float)b2; c.eps = 1e-8f;
    c.step_size = (float)(lr / (1.0 - pow(b1, (double)step)));
    c.bc2_sqrt = (float)sqrt(1.0 - pow(b2, (double)step));
    return c;
}

// phase 0: the whole iteration; 1: [derived matrices] + main kernel + reduction (d3r_aligner_step_begin); 2: pose / focal step + step count (d3r_aligner_step_end)
static int aligner_pass(d3r_aligner* a, bool update, double lr, int hist_idx, float* g_pw, float* g_imp, float* g_depth, float* g_foc,
                        bool refresh_derived_first, hipStream_t st, float* g_pp = nullptr, float* g_pa = nullptr, int phase = 0) {
    SmallView s;
    s.n = a->n; s.E = a->E; s.pw_poses = a->pw_poses; s.pw_adaptors = a->pw_adaptors; s.im_poses = a->im_poses;
    s.im_focals = a->im_focals; s.im_pp = a->im_pp; s.img_w = a->d_w; s.img_h = a->d_h;
    s.pw_m = a->pw_m; s.pw_v = a->pw_v; s.imp_m = a->imp_m; s.imp_v = a->imp_v; s.foc_m = a->foc_m; s.foc_v = a->foc_v;
    s.pp_m = a->pp_m; s.pp_v = a->pp_v; s.g_pp = nullptr; s.opt_pp = a->opt_pp;
    s.pa_m = a->pa_m; s.pa_v = a->pa_v; s.g_pa = nullptr; s.opt_adapt = a->opt_adapt;
    s.red_edge = a->red_edge; s.red_img = a->red_img; s.d_edge = a->d_edge; s.d_img = a->d_img; s.scratch = a->scratch;
    s.loss_hist = a->loss_hist; s.iter = hist_idx; s.g_pw = nullptr; s.g_imp = nullptr; s.g_foc = nullptr;
    s.base_scale = a->base_scale; s.pw_break = a->pw_break; s.focal_break = a->focal_break;
    s.norm_pw_scale = a->norm_pw_scale; s.opt_poses = a->opt_poses; s.opt_focals = a->opt_focals;
    s.update = 0; s.adam = adam_coef(lr, a->step + 1);
    if (phase == 2) {
        s.update = update ? 1 : 0;
        s.g_pw = g_pw; s.g_imp = g_imp; s.g_foc = g_foc; s.g_pp = g_pp; s.g_pa = g_pa;
        launch_small(s, st, a->generic_small);
        if (update) a->step++;
        return hipGetLastError() == hipSuccess ? D3R_OK : D3R_ERR_LAUNCH;
    }
    if (refresh_derived_first) {
        SmallView s0 = s;
        s0.loss_hist = nullptr;
        launch_small(s0, st, a->generic_small);
    }
    const int imgc = a->imgc < 0 ? a->n : a->imgc;
    if (a->part_clear_pending) {        // records of images outside the range must read as zero (they are never written here)
        HIPCHK(hipMemsetAsync(a->part_edge, 0, (size_t)(2 * a->E + a->n) * a->nslot * PW * sizeof(float), st));
        a->part_clear_pending = false;
    }
    AlignerView v;
    v.img0 = a->img0;
    v.n = a->n; v.E = a->E; v.maxA = a->maxA; v.nslot = a->nslot; v.img_w = a->d_w; v.img_area = a->d_area;
    v.adj_off = a->d_adj_off; v.adj_es = a->d_adj_es; v.pred[0] = a->pred[0]; v.pred[1] = a->pred[1]; v.inter = a->inter; v.maxAp = a->maxAp;
    v.wgt[0] = a->wgt[0]; v.wgt[1] = a->wgt[1]; v.depth = a->im_depth; v.depth_m = a->depth_m; v.depth_v = a->depth_v;
    v.depth_grad = g_depth; v.d_edge = a->d_edge; v.d_img = a->d_img; v.part_edge = a->part_edge; v.part_img = a->part_img;
    v.inv_area[0] = a->inv_area[0]; v.inv_area[1] = a->inv_area[1]; v.l2 = a->l2; v.update = update ? 1 : 0;
    v.use_dpp = a->use_dpp; v.adam = s.adam;
    // D3R_ALIGNER_PF=2: two edges of the stream in flight per wave (probe; 16 more VGPRs, 3 instead of 4 waves per SIMD)
    static const int pf = [] { const char* e = probe_env("D3R_ALIGNER_PF"); return (e && e[0] == '2') ? 2 : 1; }();
    const dim3 grid(imgc * a->nslot);
    if (imgc > 0) {
    // default build: the block-interleaved layout only (a->layout == 1 whenever no probe switch is read); the planar / 512-thread / two-edges-in-flight /
    // ablation instances are compiled in probe builds (-DD3R_PROBES)
    bool launched = false;
    if constexpr (kProbes) {
        launched = true;
        if (a->probe && !a->l2) {
            if (a->probe == 1 && pf == 2) hipLaunchKernelGGL((aligner_main_kernel<false, 2, 4, 1>), grid, dim3(256), 0, st, v);
            else if (a->probe == 1) hipLaunchKernelGGL((aligner_main_kernel<false, 1, 4, 1>), grid, dim3(256), 0, st, v);
            else hipLaunchKernelGGL((aligner_main_kernel<false, 1, 4, 2>), grid, dim3(256), 0, st, v);
        } else if (a->layout == 1) {
            launched = false;
        } else if (a->nwv == 8) {
            if (a->l2) hipLaunchKernelGGL((aligner_main_kernel<true, 1, 8>), grid, dim3(512), 0, st, v);
            else hipLaunchKernelGGL((aligner_main_kernel<false, 1, 8>), grid, dim3(512), 0, st, v);
        } else if (a->l2) {
            if (pf == 2) hipLaunchKernelGGL((aligner_main_kernel<true, 2>), grid, dim3(256), 0, st, v);
            else hipLaunchKernelGGL((aligner_main_kernel<true, 1>), grid, dim3(256), 0, st, v);
        } else {
            if (pf == 2) hipLaunchKernelGGL((aligner_main_kernel<false, 2>), grid, dim3(256), 0, st, v);
            else hipLaunchKernelGGL((aligner_main_kernel<false, 1>), grid, dim3(256), 0, st, v);
        }
    }
    (void)pf;
    if (!launched) {
        if (a->l2) hipLaunchKernelGGL((aligner_main_kernel<true, 1, 4, 0, 1>), grid, dim3(256), 0, st, v);
        else hipLaunchKernelGGL((aligner_main_kernel<false, 1, 4, 0, 1>), grid, dim3(256), 0, st, v);
    }
    }
    // part_edge | part_img and red_edge | red_img are contiguous: one launch reduces the 2E + n entries
    hipLaunchKernelGGL(aligner_reduce_kernel, dim3(2 * a->E + a->n), dim3(256), 0, st, a->part_edge, a->red_edge, a->nslot);
    if (phase == 1) return hipGetLastError() == hipSuccess ? D3R_OK : D3R_ERR_LAUNCH;
    s.update = update ? 1 : 0;
    s.g_pw = g_pw; s.g_imp = g_imp; s.g_foc = g_foc; s.g_pp = g_pp; s.g_pa = g_pa;
    launch_small(s, st, a->generic_small);
    if (update) a->step++;
    return hipGetLastError() == hipSuccess ? D3R_OK : D3R_ERR_LAUNCH;
}

extern "C" int d3r_aligner_set_option(d3r_aligner* a, int option, int value) {
    if (!a) return D3R_ERR_INVALID;
    switch (option) {
        case D3R_ALIGNER_OPT_DPP_REDUCE: a->use_dpp = value; return D3R_OK;
        case D3R_ALIGNER_OPT_OPTIMIZE_PP: a->opt_pp = value != 0; return D3R_OK;
        case D3R_ALIGNER_OPT_OPTIMIZE_ADAPTORS: a->opt_adapt = value != 0; return D3R_OK;
        case D3R_ALIGNER_OPT_GENERIC_SMALL: a->generic_small = value != 0; return D3R_OK;
        case D3R_ALIGNER_OPT_RESET_ADAM:
            // the moments are cleared on the stream of the NEXT d3r_aligner_run / loss_grad call (ordered against the iterations that
            // are still in flight there), not on the legacy NULL stream
            a->reset_pending = true;
            a->step = 0;
            return D3R_OK;
    }
    return D3R_ERR_INVALID;
}

// niter iterations of global_alignment_iter; lr follows the reference schedule evaluated at
// t = (iter0 + k) / niter_total (base_opt.py:352-366, commons.py:83-90). losses (device or null).
extern "C" int d3r_aligner_run(d3r_aligner* a, int niter, int iter0, int niter_total, float lr_base, float lr_min, int schedule,
                               float* losses_out_device, void* stream) {
    if (!a || niter <= 0 || niter > a->loss_cap || niter_total <= 0) return D3R_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    aligner_wait_ready(a, st);
    if (a->reset_pending) {
        HIPCHK(hipMemsetAsync(a->depth_m, 0, (size_t)((char*)a->d_edge - (char*)a->depth_m), st));
        a->reset_pending = false;
    }
    for (int k = 0; k < niter; ++k) {
        const double t = (double)(iter0 + k) / (double)niter_total;
        const double lr = schedule == D3R_SCHEDULE_COSINE ? (double)lr_min + ((double)lr_base - (double)lr_min) * (1.0 + cos(t * M_PI)) / 2.0
                                                          : (double)lr_base + ((double)lr_min - (double)lr_base) * t;
        const int rc = aligner_pass(a, true, lr, k, nullptr, nullptr, nullptr, nullptr, k == 0, st);
        if (rc != D3R_OK) return rc;
    }
    if (losses_out_device) HIPCHK(hipMemcpyAsync(losses_out_device, a->loss_hist, niter * sizeof(float), hipMemcpyDeviceToDevice, st));
    return D3R_OK;
}

// ---- one iteration in two calls, for one process per GPU (include/dust3r_hip.h) ------------------------------------------------------------------
static double sched_lr(int k, int iter0, int niter_total, float lr_base, float lr_min, int schedule) {
    const double t = (double)(iter0 + k) / (double)niter_total;
    return schedule == D3R_SCHEDULE_COSINE ? (double)lr_min + ((double)lr_base - (double)lr_min) * (1.0 + cos(t * M_PI)) / 2.0
                                           : (double)lr_base + ((double)lr_min - (double)lr_base) * t;
}

extern "C" int d3r_aligner_set_image_range(d3r_aligner* a, int first, int count) {
    if (!a || first < 0 || count < 0 || first + count > a->n) return D3R_ERR_INVALID;
    a->img0 = first;
    a->imgc = (first == 0 && count == a->n) ? -1 : count;
    a->part_clear_pending = true;
    return D3R_OK;
}

extern "C" int d3r_aligner_step_begin(d3r_aligner* a, int k, int iter0, int niter_total, float lr_base, float lr_min, int schedule, void* stream) {
    if (!a || k < 0 || k >= a->loss_cap || niter_total <= 0) return D3R_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    aligner_wait_ready(a, st);
    if (a->reset_pending) {
        HIPCHK(hipMemsetAsync(a->depth_m, 0, (size_t)((char*)a->d_edge - (char*)a->depth_m), st));
        a->reset_pending = false;
    }
    return aligner_pass(a, true, sched_lr(k, iter0, niter_total, lr_base, lr_min, schedule), k, nullptr, nullptr, nullptr, nullptr, k == 0, st, nullptr, nullptr, 1);
}

extern "C" int d3r_aligner_step_end(d3r_aligner* a, int k, int iter0, int niter_total, float lr_base, float lr_min, int schedule, void* stream) {
    if (!a || k < 0 || k >= a->loss_cap || niter_total <= 0) return D3R_ERR_INVALID;
    return aligner_pass(a, true, sched_lr(k, iter0, niter_total, lr_base, lr_min, schedule), k, nullptr, nullptr, nullptr, nullptr, false, (hipStream_t)stream, nullptr, nullptr, 2);
}

extern "C" int d3r_aligner_reduced_sums(d3r_aligner* a, void** ptr, long long* count) {
    if (!a || !ptr || !count) return D3R_ERR_INVALID;
    *ptr = a->red_edge;                               // red_edge | red_img: contiguous fp64 [2E + n][16]
    *count = (long long)(2 * a->E + a->n) * PW;
    return D3R_OK;
}

extern "C" int d3r_aligner_read_losses(d3r_aligner* a, int niter, float* losses_out_device, void* stream) {
    if (!a || niter <= 0 || niter > a->loss_cap || !losses_out_device) return D3R_ERR_INVALID;
    HIPCHK(hipMemcpyAsync(losses_out_device, a->loss_hist, niter * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return D3R_OK;
}

// one forward/backward WITHOUT a step: loss (device float[1]) and gradients (device, any may be null)
extern "C" int d3r_aligner_loss_grad(d3r_aligner* a, float* loss_device, float* g_pw_poses, float* g_im_poses, float* g_im_depth,
                                     float* g_im_focals, float* g_im_pp, float* g_pw_adaptors, void* stream) {
    if (!a) return D3R_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    aligner_wait_ready(a, st);
    float* gpw = g_pw_poses ? g_pw_poses : a->g_scratch;  // forces the gradient branch of the small kernel
    const int rc = aligner_pass(a, false, 0.0, 0, gpw, g_im_poses, g_im_depth, g_im_focals, true, st, g_im_pp, g_pw_adaptors);
    if (rc != D3R_OK) return rc;
    if (loss_device) HIPCHK(hipMemcpyAsync(loss_device, a->loss_hist, sizeof(float), hipMemcpyDeviceToDevice, st));
    return D3R_OK;
}

// ---- clean_pointcloud (reference dust3r/cloud_opt/base_opt.py:369-405) ----------------------------------------------
// A point of image i that projects IN FRONT of image j's depthmap while being less confident than the pixel it lands on
// gets its confidence clipped to bad_conf. The reference is a host-driven double loop (i outer, j inner) that updates the
// confidences in place, so image i sees the already-cleaned confidences of images j < i: one launch per i keeps exactly
// that order; inside a launch a thread owns one pixel of image i and walks the cameras j in order.
namespace d3r {
__global__ __launch_bounds__(256) void clean_pointcloud_kernel(int i, int n, float* __restrict__ conf, const float* __restrict__ depth,
                                                                const float* __restrict__ pts, const float* __restrict__ K,
                                                                const float* __restrict__ w2c, const int* __restrict__ Hs,
                                                                const int* __restrict__ Ws, int maxA, float tol, float bad_conf) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Hs[i] * Ws[i]) return;
    const float* X = pts + ((size_t)i * maxA + p) * 3;
    const float x = X[0], y = X[1], z = X[2];
    float c = conf[(size_t)i * maxA + p];
    for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        const float* M = w2c + j * 16;
        const float px = M[0] * x + M[1] * y + M[2] * z + M[3];
        const float py = M[4] * x + M[5] * y + M[6] * z + M[7];
        const float pz = M[8] * x + M[9] * y + M[10] * z + M[11];
        if (!(pz > 0.f)) continue;
        const float* Kj = K + j * 9;
        const float ku = Kj[0] * px + Kj[1] * py + Kj[2] * pz, kv = Kj[3] * px + Kj[4] * py + Kj[5] * pz, kw = Kj[6] * px + Kj[7] * py + Kj[8] * pz;
        const float u = rintf(ku / kw), v = rintf(kv / kw);   // torch.round: half to even
        const int Wj = Ws[j], Hj = Hs[j];
        if (!(u >= 0.f && u < (float)Wj && v >= 0.f && v < (float)Hj)) continue;
        const size_t q = (size_t)j * maxA + (size_t)((int)v * Wj + (int)u);
        if (pz < (1.f - tol) * depth[q] && c < conf[q]) c = fminf(c, bad_conf);
    }
    conf[(size_t)i * maxA + p] = c;
}
}  // namespace d3r

extern "C" int d3r_clean_pointcloud(int n_imgs, float* conf, const float* depth, const float* pts3d, const float* intrinsics,
                                    const float* world2cam, const int* img_h_dev, const int* img_w_dev, int max_area, float tol, float bad_conf,
                                    void* stream) {
    if (n_imgs <= 0 || !conf || !depth || !pts3d || !intrinsics || !world2cam || !img_h_dev || !img_w_dev || max_area <= 0 || tol < 0.f || tol >= 1.f)
        return D3R_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    // one launch per image, in order (image i sees the cleaned confidences of the images before it); every launch covers max_area
    // pixels and bounds itself by the image's own size, so nothing is allocated, copied or synchronised here
    for (int i = 0; i < n_imgs; ++i)
        hipLaunchKernelGGL(d3r::clean_pointcloud_kernel, dim3((max_area + 255) / 256), dim3(256), 0, st, i, n_imgs, conf, depth, pts3d, intrinsics,
                           world2cam, img_h_dev, img_w_dev, max_area, tol, bad_conf);
    return hipGetLastError() == hipSuccess ? D3R_OK : D3R_ERR_LAUNCH;
}

// ---- brute-force 3-D nearest neighbour (find_reciprocal_matches, reference dust3r/utils/geometry.py:345-361) ---------------
// The reference builds two SciPy KD-trees on the host; on the GPU an exhaustive scan is simpler and faster at pointmap
// sizes (196 608^2 distance evaluations = 2 x 10^11 flops): one thread per query, reference points streamed through LDS
// as (x, y, z, -) float4 tiles. Ties resolve to the lowest index.
namespace d3r {
__global__ __launch_bounds__(256) void nearest_neighbor_kernel(const float* __restrict__ Q, int nq, const float* __restrict__ R, int nr,
                                                                int* __restrict__ out) {
    __shared__ float4 tile[1024];
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool live = q < nq;
    const float qx = live ? Q[(size_t)q * 3] : 0.f, qy = live ? Q[(size_t)q * 3 + 1] : 0.f, qz = live ? Q[(size_t)q * 3 + 2] : 0.f;
    float best = 3.4e38f;
    int bi = 0;
    for (int r0 = 0; r0 < nr; r0 += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024; t += 256) {
            const int r = r0 + t;
            tile[t] = r < nr ? make_float4(R[(size_t)r * 3], R[(size_t)r * 3 + 1], R[(size_t)r * 3 + 2], 0.f) : make_float4(3e18f, 3e18f, 3e18f, 0.f);
        }
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < 1024; ++t) {
            const float4 p = tile[t];
            const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; bi = r0 + t; }
        }
    }
    if (live) out[q] = bi;
}
}  // namespace d3r

extern "C" int d3r_nearest_neighbors(const float* query, int n_query, const float* ref, int n_ref, int* idx_out, void* stream) {
    if (!query || !ref || !idx_out || n_query <= 0 || n_ref <= 0) return D3R_ERR_INVALID;
    hipLaunchKernelGGL(d3r::nearest_neighbor_kernel, dim3((n_query + 255) / 256), dim3(256), 0, (hipStream_t)stream, query, n_query, ref, n_ref, idx_out);
    return hipGetLastError() == hipSuccess ? D3R_OK : D3R_ERR_LAUNCH;
}

// ---- host-only self test of the analytic gradients (no GPU): used by the CPU test-suite to check
// aligner_math.hpp against autograd before any kernel runs. NOT a compute path of the product.
extern "C" int d3r_selftest_aligner_math_host(int n_imgs, int n_edges, const int* ei, const int* ej, int H, int W,
                                              const float* pred_i, const float* pred_j, const float* w_i, const float* w_j,
                                              const float* pw_poses, const float* im_poses, const float* im_depth,
                                              const float* im_focals, float base_scale, float focal_break, double* loss_out,
                                              double* g_pw, double* g_imp, double* g_depth, double* g_foc) {
    const int A = H * W;
    const double inv_area = 1.0 / ((double)n_edges * A);
    double mean7 = 0.0;
    for (int e = 0; e < n_edges; ++e) mean7 += pw_poses[e * 8 + 7];
    mean7 /= n_edges;
    const float nf = expf(logf(base_scale) - (float)mean7);
    std::vector<double> GM((size_t)n_edges * 12, 0.0), GI((size_t)n_imgs * 9, 0.0), GT((size_t)n_imgs * 3, 0.0);
    std::vector<float> gx((size_t)n_imgs * A * 3, 0.f);
    std::vector<float> Rimg((size_t)n_imgs * 9);
    double loss = 0.0;
    for (int i = 0; i < n_imgs; ++i) quat_to_rotmat(im_poses + i * 7, &Rimg[i * 9]);
    auto point = [&](int i, int p, float X[3], float cam[3]) {
        const float* R = &Rimg[i * 9];
        const float F = expf(im_focals[i] / focal_break), ed = expf(im_depth[(size_t)i * A + p]);
        const int v = p / W, u = p - v * W;
        cam[0] = ed * ((float)u - 0.5f * W) / F; cam[1] = ed * ((float)v - 0.5f * H) / F; cam[2] = ed;
        for (int r = 0; r < 3; ++r)
            X[r] = R[r * 3] * cam[0] + R[r * 3 + 1] * cam[1] + R[r * 3 + 2] * cam[2] + signed_expm1f(im_poses[i * 7 + 4 + r]);
    };
    for (int e = 0; e < n_edges; ++e) {
        const float* P = pw_poses + e * 8;
        float R[9], M[12];
        quat_to_rotmat(P, R);
        const float st = expf(P[7]) * nf;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) M[r * 4 + c] = st * R[r * 3 + c];
            M[r * 4 + 3] = st * signed_expm1f(P[4 + r]);
        }
        for (int side = 0; side < 2; ++side) {
            const int img = side ? ej[e] : ei[e];
            const float* pred = side ? pred_j : pred_i;
            const float* wg = side ? w_j : w_i;
            for (int p = 0; p < A; ++p) {
                float X[3], cam[3], g[3] = {0, 0, 0}, gm[12] = {0}, l = 0.f;
                point(img, p, X, cam);
                residual_accumulate(X, M, pred + ((size_t)e * A + p) * 3, (float)(wg[(size_t)e * A + p] * inv_area), false, l, g, gm);
                loss += l;
                for (int k = 0; k < 12; ++k) GM[(size_t)e * 12 + k] += gm[k];
                for (int r = 0; r < 3; ++r) gx[((size_t)img * A + p) * 3 + r] += g[r];
            }
        }
    }
    for (int i = 0; i < n_imgs; ++i)
        for (int p = 0; p < A; ++p) {
            float X[3], cam[3];
            point(i, p, X, cam);
            const float* g = &gx[((size_t)i * A + p) * 3];
            double gd = 0.0;
            for (int r = 0; r < 3; ++r) {
                gd += (double)g[r] * (X[r] - signed_expm1f(im_poses[i * 7 + 4 + r]));
                for (int c = 0; c < 3; ++c) GI[(size_t)i * 9 + r * 3 + c] += (double)g[r] * cam[c];
                GT[(size_t)i * 3 + r] += g[r];
            }
            g_depth[(size_t)i * A + p] = gd;
        }
    std::vector<double> gss(n_edges);
    double gsum = 0.0;
    for (int e = 0; e < n_edges; ++e) {
        const float* P = pw_poses + e * 8;
        float R[9];
        quat_to_rotmat(P, R);
        const float st = expf(P[7]) * nf, adapt[3] = {1.f, 1.f, 1.f};
        double gP[7], gs;
        edge_chain(P, R, st, adapt, &GM[(size_t)e * 12], gP, gs);
        for (int k = 0; k < 7; ++k) g_pw[e * 8 + k] = gP[k];
        gss[e] = gs * st;
        gsum += gss[e];
    }
    for (int e = 0; e < n_edges; ++e) g_pw[e * 8 + 7] = gss[e] - gsum / n_edges;
    for (int i = 0; i < n_imgs; ++i) {
        double gP[7], gf;
        image_chain(im_poses + i * 7, &Rimg[i * 9], focal_break, &GI[(size_t)i * 9], &GT[(size_t)i * 3], gP, gf);
        for (int k = 0; k < 7; ++k) g_imp[i * 7 + k] = gP[k];
        g_foc[i] = gf;
    }
    *loss_out = loss;
    return D3R_OK;
}
