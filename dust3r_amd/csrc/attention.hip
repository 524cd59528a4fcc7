// dust3r_amd -- fused softmax(Q K^T * scale) V for the croco Attention / CrossAttention blocks
// (reference call sites: dust3r/model.py:136-137 encoder blocks, :180-186 decoder blocks; the
// croco modules themselves are restated in oracle/croco_ref/models/blocks.py).
//
// q, k arrive head-major [B][H][N][64] with 2-D RoPE already applied by the projection GEMM's
// epilogue; v arrives transposed [B][H][64][ldv]. Output is token-major [B][Nq][H*64], the
// operand layout of the following proj GEMM.
//
// Structure (one workgroup = 4 waves = 128 queries of one (b, h); each wave owns 32 queries):
//   S^T = K Q^T with 32x32 MFMAs, so one lane holds 16 of the 32 keys of ONE query -> the
//   softmax row reduction is in-register plus a single lane<->lane+32 exchange;
//   O^T += V^T P^T consumes P straight from the S registers: the MFMA contraction index is
//   permuted identically on both operands (keys {0-3,8-11}+4*half), so no cross-lane shuffle
//   and no transposing LDS read is needed -- V^T rows are read as two 4-key groups.
//   K / V^T tiles (64 keys) are staged through LDS (padded rows: conflict-free b128 / b64 reads),
//   double buffered with the global loads of tile t+1 in flight during the math of tile t.
#include <atomic>

#include "kernels.hpp"

namespace d3r {

template <int DT> struct AttnCfg {
    static constexpr int EB = Traits<DT>::EB;
    static constexpr int ROWB = 64 * EB;                      // bytes of one 64-element row
    static constexpr int KROW = ROWB + 16;                    // padded K row stride in LDS
    static constexpr int VROW = (DT == D3R_F32) ? ROWB + 16 : ROWB + 8;   // F16X3: 264-byte rows, b64 reads
    static constexpr int CPR = ROWB / 16;                     // 16-byte chunks per row
    static constexpr int NLD = 64 * CPR / 256;                // chunks per thread per tile (2 or 4)
    static constexpr int NKS = ROWB / 32;                     // QK^T k-steps (two chunks each)
    static constexpr int STAGE = 64 * KROW + 64 * VROW;
    static constexpr int LDS = 2 * STAGE;
};

// ODT: layout of the output rows -- DT, or the fp16 + fp8 activation rows when the following proj GEMM runs in that mode
template <int DT, int ODT = DT>
__global__ __launch_bounds__(256, 2) void attention_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using C = AttnCfg<DT>;
    using TR = Traits<DT>;
    constexpr int EB = C::EB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;

    const int nqb = (p.Nq + 127) / 128;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = lid / nqb, qb = lid - bh * nqb;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qb * 128 + wave * 32;

    const char* qptr = reinterpret_cast<const char*>(p.q) + (size_t)bh * p.Nq * C::ROWB;
    const char* kptr = reinterpret_cast<const char*>(p.k) + (size_t)bh * p.Nk * C::ROWB;
    const char* vptr = reinterpret_cast<const char*>(p.vt) + (size_t)bh * 64 * p.ldv * EB;

    // ---- Q fragments stay in registers for the whole kernel -------------------------------------
    // F16X3: a 64-element row is 8 groups [hi x8][lo x8]; k-step ks (16 k) gives lane half hh the group 2*ks+hh,
    // kept as qf[2*ks] (hi chunk) and qf[2*ks+1] (lo chunk).
    uint4 qf[C::NKS];
    {
        int qrow = q0 + l31;
        qrow = qrow < p.Nq ? qrow : p.Nq - 1;
        if constexpr (DT == D3R_F16X3) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qf[2 * ks] = *reinterpret_cast<const uint4*>(qptr + (size_t)qrow * C::ROWB + (2 * ks + hh) * 32);        // hi
                qf[2 * ks + 1] = *reinterpret_cast<const uint4*>(qptr + (size_t)qrow * C::ROWB + (2 * ks + hh) * 32 + 16);  // lo
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks)
                qf[ks] = *reinterpret_cast<const uint4*>(qptr + (size_t)qrow * C::ROWB + (ks * 2 + hh) * 16);
        }
    }

    // ---- tile staging (registers carry tile t+1 across the math of tile t). Named scalars, not
    // arrays: hipcc sends a loop-carried register array to scratch (rule 20 of the CDNA guide).
    uint4 kst0, kst1, kst2, kst3, vst0, vst1, vst2, vst3;
    kst0 = kst1 = kst2 = kst3 = vst0 = vst1 = vst2 = vst3 = make_uint4(0, 0, 0, 0);
#define D3R_LOAD1(i_, key0_, kdst_, vdst_)                                                               \
    if constexpr ((i_) < C::NLD) {                                                                          \
        const int idx_ = tid + 256 * (i_);                                                                  \
        const int row_ = idx_ / C::CPR, ch_ = idx_ - row_ * C::CPR;                                         \
        int krow_ = (key0_) + row_;                                                                         \
        krow_ = krow_ < p.Nk ? krow_ : p.Nk - 1;                                                            \
        kdst_ = *reinterpret_cast<const uint4*>(kptr + (size_t)krow_ * C::ROWB + ch_ * 16);                 \
        vdst_ = *reinterpret_cast<const uint4*>(vptr + ((size_t)row_ * p.ldv + (key0_)) * EB + ch_ * 16);   \
    }
#define D3R_ISSUE_LOADS(key0_)                                                                              \
    D3R_LOAD1(0, key0_, kst0, vst0) D3R_LOAD1(1, key0_, kst1, vst1) D3R_LOAD1(2, key0_, kst2, vst2)         \
    D3R_LOAD1(3, key0_, kst3, vst3)
#define D3R_WRITE1(i_, kb_, vb_, ksrc_, vsrc_)                                                              \
    if constexpr ((i_) < C::NLD) {                                                                          \
        const int idx_ = tid + 256 * (i_);                                                                  \
        const int row_ = idx_ / C::CPR, ch_ = idx_ - row_ * C::CPR;                                         \
        *reinterpret_cast<uint4*>(kb_ + row_ * C::KROW + ch_ * 16) = ksrc_;                                 \
        if constexpr (DT == D3R_F32) {                                                                      \
            *reinterpret_cast<uint4*>(vb_ + row_ * C::VROW + ch_ * 16) = vsrc_;                             \
        } else { /* 136-byte rows are only 8-byte aligned */                                                \
            *reinterpret_cast<uint2*>(vb_ + row_ * C::VROW + ch_ * 16) = make_uint2(vsrc_.x, vsrc_.y);      \
            *reinterpret_cast<uint2*>(vb_ + row_ * C::VROW + ch_ * 16 + 8) = make_uint2(vsrc_.z, vsrc_.w);  \
        }                                                                                                   \
    }
#define D3R_WRITE_LDS(buf_)                                                                                 \
    {                                                                                                       \
        char* kb_ = smem + (buf_) * C::STAGE;                                                               \
        char* vb_ = kb_ + 64 * C::KROW;                                                                     \
        D3R_WRITE1(0, kb_, vb_, kst0, vst0) D3R_WRITE1(1, kb_, vb_, kst1, vst1)                             \
        D3R_WRITE1(2, kb_, vb_, kst2, vst2) D3R_WRITE1(3, kb_, vb_, kst3, vst3)                             \
    }

    f32x16_t o[2];
    o[0] = o[1] = (f32x16_t){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float m_run = -1e30f, l_run = 0.f;
    const float c = p.scale * 1.44269504088896340736f;  // fold log2(e): p = exp2(s*c - m*c)

    const int ntiles = (p.Nk + 63) / 64;
    D3R_ISSUE_LOADS(0)
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        D3R_WRITE_LDS(buf)
        __syncthreads();
        if (t + 1 < ntiles) { D3R_ISSUE_LOADS((t + 1) * 64) }
        const char* kb = smem + buf * C::STAGE;
        const char* vb = kb + 64 * C::KROW;

        // S^T[key][query] = K Q^T
        f32x16_t s[2];
        s[0] = s[1] = (f32x16_t){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (DT == D3R_F16X3) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const char* kr = kb + (rb * 32 + l31) * C::KROW + (2 * ks + hh) * 32;
                    const uint4 kh = *reinterpret_cast<const uint4*>(kr), kl = *reinterpret_cast<const uint4*>(kr + 16);
                    TR::mma32x3(s[rb], kh, kl, qf[2 * ks], qf[2 * ks + 1]);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const uint4 kf = *reinterpret_cast<const uint4*>(kb + (rb * 32 + l31) * C::KROW + (ks * 2 + hh) * 16);
                    TR::mma32(s[rb], kf, qf[ks]);
                }
            }
        }
        // mask keys beyond Nk (only the last tile of a ragged sequence)
        if ((t + 1) * 64 > p.Nk) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= p.Nk) s[rb][r] = -1e30f;
                }
        }
        // online softmax; lanes l and l^32 share one query
        float mt = s[0][0];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[rb][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
        float psum = 0.f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[rb][r] * c - mc);
                s[rb][r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;

        // O^T[d][query] += V^T P^T   (contraction over keys, permuted identically on both sides)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                const int kbase = rb * 32 + 16 * sh + 4 * hh;
                if constexpr (DT == D3R_F32) {
                    const uint4 plo = make_uint4(__float_as_uint(s[rb][8 * sh + 0]), __float_as_uint(s[rb][8 * sh + 1]),
                                                 __float_as_uint(s[rb][8 * sh + 2]), __float_as_uint(s[rb][8 * sh + 3]));
                    const uint4 phi = make_uint4(__float_as_uint(s[rb][8 * sh + 4]), __float_as_uint(s[rb][8 * sh + 5]),
                                                 __float_as_uint(s[rb][8 * sh + 6]), __float_as_uint(s[rb][8 * sh + 7]));
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const char* vrow = vb + (db * 32 + l31) * C::VROW + kbase * EB;
                        const uint4 vlo = *reinterpret_cast<const uint4*>(vrow);
                        const uint4 vhi = *reinterpret_cast<const uint4*>(vrow + 8 * EB);
                        TR::mma32(o[db], vlo, plo);
                        TR::mma32(o[db], vhi, phi);
                    }
                } else if constexpr (DT == D3R_F16X3) {
                    // keys kbase..+3 sit in 8-group kbase/8 at element offset 4*hh, keys kbase+8..+11 in the next group
                    uint4 ph, pl;
                    // p = exp2(s c - m c) with m the running maximum: 0 <= p <= 1, no range clamp in front of the fp16 conversions
                    TR::split2_inrange(s[rb][8 * sh + 0], s[rb][8 * sh + 1], ph.x, pl.x);
                    TR::split2_inrange(s[rb][8 * sh + 2], s[rb][8 * sh + 3], ph.y, pl.y);
                    TR::split2_inrange(s[rb][8 * sh + 4], s[rb][8 * sh + 5], ph.z, pl.z);
                    TR::split2_inrange(s[rb][8 * sh + 6], s[rb][8 * sh + 7], ph.w, pl.w);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const char* vrow = vb + (db * 32 + l31) * C::VROW + (kbase >> 3) * 32 + hh * 8;
                        const uint2 h0 = *reinterpret_cast<const uint2*>(vrow), l0 = *reinterpret_cast<const uint2*>(vrow + 16);
                        const uint2 h1 = *reinterpret_cast<const uint2*>(vrow + 32), l1 = *reinterpret_cast<const uint2*>(vrow + 48);
                        TR::mma32x3(o[db], make_uint4(h0.x, h0.y, h1.x, h1.y), make_uint4(l0.x, l0.y, l1.x, l1.y), ph, pl);
                    }
                } else {
                    uint4 pf;
                    pf.x = TR::pack2(s[rb][8 * sh + 0], s[rb][8 * sh + 1]);
                    pf.y = TR::pack2(s[rb][8 * sh + 2], s[rb][8 * sh + 3]);
                    pf.z = TR::pack2(s[rb][8 * sh + 4], s[rb][8 * sh + 5]);
                    pf.w = TR::pack2(s[rb][8 * sh + 6], s[rb][8 * sh + 7]);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const char* vrow = vb + (db * 32 + l31) * C::VROW + kbase * EB;
                        const uint2 vlo = *reinterpret_cast<const uint2*>(vrow);
                        const uint2 vhi = *reinterpret_cast<const uint2*>(vrow + 8 * EB);
                        TR::mma32(o[db], make_uint4(vlo.x, vlo.y, vhi.x, vhi.y), pf);
                    }
                }
            }
        }
    }

    // ---- normalise and store (4 consecutive d per register group) -------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
    if (q < p.Nq) {
        const size_t obase = ((size_t)b * p.Nq + q) * (size_t)(p.H * 64) + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = db * 32 + 8 * g + 4 * hh;
                store4<ODT>(p.out, obase + d, o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv,
                            o[db][4 * g + 3] * inv);
            }
    }
}

template <int DT, int ODT = DT> static hipError_t launch_t(const AttnParams& p, hipStream_t s) {
    // the dynamic-LDS limit is a per-device function attribute: raise it once on every device this process launches on
    static std::atomic<unsigned long long> attr_done{0};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    const unsigned long long dev_bit = 1ull << (dev_id & 63);
    if (!(attr_done.load(std::memory_order_relaxed) & dev_bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<DT, ODT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  AttnCfg<DT>::LDS);
        attr_done.fetch_or(dev_bit, std::memory_order_relaxed);
    }
    const int grid = p.B * p.H * ((p.Nq + 127) / 128);
    hipLaunchKernelGGL((attention_kernel<DT, ODT>), dim3(grid), dim3(256), AttnCfg<DT>::LDS, s, p);
    return hipGetLastError();
}

hipError_t launch_attention(int dt, const AttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0 || p.ldv % 64 != 0 || p.ldv < ((p.Nk + 63) / 64) * 64)
        return hipErrorInvalidValue;
    if (p.out_dt >= 0 && p.out_dt != dt && !(dt == D3R_F16X3 && p.out_dt == D3R_F16F8)) return hipErrorInvalidValue;
    switch (dt) {
        case D3R_BF16: return launch_t<D3R_BF16>(p, s);
        case D3R_F16: return launch_t<D3R_F16>(p, s);
        case D3R_F32: return launch_t<D3R_F32>(p, s);
        case D3R_F16X3:
            if (p.out_dt == D3R_F16F8) return launch_t<D3R_F16X3, D3R_F16F8>(p, s);   // q, k, v^T split-fp16; output rows for an fp16 + fp8 proj GEMM
            return launch_t<D3R_F16X3>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace d3r
