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
#include <stdlib.h>

#include <atomic>
#include <type_traits>

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

// =========================================================================================================================
// split-fp16 attention, software pipelined (round 3). Same data layouts, MFMA shapes and lane maps as attention_kernel above
// (one workgroup = 4 waves = 128 queries of one (b, h), 32 queries per wave, 64-key tiles), but a wave's instruction stream is
// laid out so that its VALU work runs UNDER its own MFMAs instead of between them:
//   * S of tile t + 1 (24 MFMAs) is issued while the softmax of tile t runs (max, exp2, row sum, rescale of O): two S
//     register sets, loop unrolled by two. The old kernel ran QK^T -> softmax -> PV back to back: 1536 MFMA cycles and ~1700
//     VALU cycles per wave and tile with nothing to overlap them but the SIMD's other wave (MFMA busy 39 %).
//   * PV of tile t (24 MFMAs) carries the fp16 hi / lo split of P: group g + 1 is split under the MFMAs of group g.
//   * program order alternates one MFMA with a handful of VALU instructions (sched_group_barrier): an in-order wave cannot
//     issue past an MFMA that waits for the matrix pipe, so a burst of MFMAs followed by a burst of VALU serialises the two.
//   * the lane <-> lane + 32 exchange of the row maximum is v_permlane32_swap (VALU, inline asm) instead of ds_bpermute (an LDS round trip
//     in the middle of the softmax); V^T tiles sit in LDS as a hi plane and a lo plane per row, so that one ds_read2_b64 returns a
//     whole MFMA operand (the interleaved [hi x8][lo x8] image cost 52 v_mov per tile to regroup); packed fp32 math for the
//     exponent argument, the row sum and the rescale.
//   * K tiles run two tiles ahead of the PV product, V^T tiles one: separate rings, ONE barrier per tile.
// D3R_ATTN_V1=1 selects the previous kernel (parity tests compare the two).
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4b_t __attribute__((ext_vector_type(4)));

// v_permlane32_swap_b32 a, b exchanges lanes 32-63 of a with lanes 0-31 of b: with a = b = v on entry, a holds v[lane & 31] and b holds
// v[32 + (lane & 31)] in every lane afterwards, i.e. {a, b} = {v[l], v[l ^ 32]} in some order. Issued from inline asm: hipcc 7.2's
// __builtin_amdgcn_permlane32_swap returns its FIRST result in both elements (measured: fmaxf(r[0], r[1]) compiles to a move of r[0]),
// which made the two lane halves of a query disagree on the running maximum. s_nop 1: a VALU-written VGPR needs wait states before a
// permlane reads it, and the assembler does not see inside the asm.
D3R_DEV void xor32_pair(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
D3R_DEV float xor32_max(float v) {
    float a, b;
    xor32_pair(v, a, b);
    return fmaxf(a, b);
}
D3R_DEV float xor32_sum(float v) {
    float a, b;
    xor32_pair(v, a, b);
    return a + b;
}

// PROBE (measurement aid, results invalid when != 0; tools/gpu_probe.py attnparts): bit 0 drops the VALU slices (softmax, P split), bit 1 the
// MFMAs, bit 2 the per-tile barrier, bit 3 the staging of the next tiles (global loads + LDS writes) -- what each part costs next to the others
// NW = waves per workgroup (4: 128 queries, two workgroups per CU; 8: 256 queries, one per CU -- every K / V^T tile staged once per 256
// queries instead of 128: half the L2 -> LDS traffic per query, but nothing covers a workgroup's prologue).
// DMA (round 4): the K and V^T tiles go L2 -> LDS with global_load_lds_dwordx4 (1 KiB per wave instruction, no VGPR round trip, no ds_write:
// the ablation of round 3 priced the register staging -- 8 buffer loads + 12 LDS writes per thread and tile -- at 30 % of the kernel). The LDS
// image of a DMA is lane-linear (lane l's 16 bytes land at base + 16 l), so rows cannot be padded: they are 256 bytes and the bank spread comes
// from an XOR swizzle applied on the per-lane SOURCE address and again on the fragment reads -- physical 16-byte slot = logical slot ^ (row & 15):
//   K rows   (logical slot = memory chunk: 8 groups [hi x8][lo x8]): the 16 rows of a ds_read_b128 service group hit 16 distinct slots;
//   V^T rows (logical slot = (hi / lo plane) * 8 + 8-key group, as in the padded image): the 8-byte operand reads of 16 consecutive rows
//            spread over all 16 slots of the 256-byte row, i.e. two rows per 128-byte bank window (2-way; the reads are ds_read_b64).
// SC (round 5): the softmax / split slices on SCALAR fp32 VALU (v_fma_f32 / v_add_f32 / v_mul_f32 / v_sub_f32) instead of the packed forms
// (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32): MI355X_MICROARCH.md prices a packed fp32 op beside MFMAs at +11..13 cycles over its issue slot
// (it holds the SIMD's issue port twice as long, and the matrix pipe's next instruction waits behind it) -- 64 of them per tile and wave here.
// Same values bit for bit (the packed forms are two independent IEEE operations). attention.hip is compiled with -fno-slp-vectorize so that
// hipcc does not re-pack the scalar pairs (dust3r_amd/build.py).
// LZ (round 5): LAZY running maximum. The maximum a query's exponents are taken against moves only when the tile's maximum exceeds it by more than 6 octaves
// (p = exp2(s c - m c) then stays <= 64: far inside fp16's range for the hi / lo split, fp32 for the row sum), so that on most tiles NO query of the wave moves it,
// alpha = 1 everywhere, and the 32 multiplications per lane that rescale the O accumulators are skipped behind one wave-uniform branch (with an eager maximum at
// least one of a wave's 32 queries moves it on nearly every tile of a 768-key row). The result is the same softmax(QK^T)V up to fp32 rounding of differently scaled partial sums.
template <int ODT, int PROBE = 0, int NW = 4, bool DMA = false, bool SC = false, bool LZ = false>
__global__ __launch_bounds__(NW * 64, 2) void attention_x3_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TR = Traits<D3R_F16X3>;
    constexpr int ROWB = 256, KROW = DMA ? ROWB : ROWB + 16, VROW = DMA ? ROWB : ROWB + 8;
    constexpr int KT = 64 * KROW, VT = 64 * VROW;   // K ring: smem[0, 2 KT), V^T ring: smem[2 KT, 2 KT + 2 VT)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;

    constexpr int QPB = NW * 32;
    const int nqb = (p.Nq + QPB - 1) / QPB;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = lid / nqb, qb = lid - bh * nqb;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qb * QPB + wave * 32;

    const char* qptr = reinterpret_cast<const char*>(p.q) + (size_t)bh * p.Nq * ROWB;
    const char* kptr = reinterpret_cast<const char*>(p.k) + (size_t)bh * p.Nk * ROWB;
    const char* vptr = reinterpret_cast<const char*>(p.vt) + (size_t)bh * 64 * p.ldv * 4;

    // Q fragments (B operand of S^T = K Q^T): k-step ks covers d = 16 ks .. 16 ks + 15; lane half hh holds the 8-group 2 ks + hh
    uint4 qf[8];
    {
        int qrow = q0 + l31;
        qrow = qrow < p.Nq ? qrow : p.Nq - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[2 * ks] = *reinterpret_cast<const uint4*>(qptr + (size_t)qrow * ROWB + (2 * ks + hh) * 32);
            qf[2 * ks + 1] = *reinterpret_cast<const uint4*>(qptr + (size_t)qrow * ROWB + (2 * ks + hh) * 32 + 16);
        }
    }

    // ---- tile staging: thread t moves chunks t, t + NT, ... of the 64 x 16 chunks of a K tile and of a V^T tile (NP = 1024 / NT per operand)
    constexpr int RPP = NW * 4, NP = 64 / RPP;          // rows per pass, passes
    const int srow = tid >> 4, sch = tid & 15;          // chunk i: row srow + RPP i, 16-byte chunk sch
    uint4 kst[NP], vst[NP];
    // Raw buffer loads: a wave-uniform resource per operand, a per-thread 32-bit offset, the tile offset in a scalar register -- no 64-bit
    // per-thread addresses (16 VGPRs and ~50 address instructions per tile in the first version of this kernel). K rows beyond Nk
    // (ragged last tile) are out of the resource's range and read as zero (their scores are masked anyway); the row part of a K address
    // therefore sits in the VGPR offset, the only part gfx9 range-checks. V^T tiles are always inside the zero-padded ldv.
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kptr), 0, p.Nk * ROWB, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vptr), 0, 64 * p.ldv * 4, 0x00020000);
    const int kvo = srow * ROWB + sch * 16;              // chunk i: + i * RPP rows
    const int vvo = srow * p.ldv * 4 + sch * 16;
    const int vstep = RPP * p.ldv * 4;
    auto as_uint4 = [](const u32x4b_t& v) __attribute__((always_inline)) { return make_uint4(v[0], v[1], v[2], v[3]); };
    auto gload_k = [&](int key0) __attribute__((always_inline)) {
        const int t0 = key0 * ROWB + kvo;
#pragma unroll
        for (int i = 0; i < NP; ++i) kst[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(krs, t0 + i * RPP * ROWB, 0, 0));
    };
    auto gload_v = [&](int key0) __attribute__((always_inline)) {
        const int s0 = key0 * 4;
#pragma unroll
        for (int i = 0; i < NP; ++i) vst[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(vrs, vvo, s0 + i * vstep, 0));
    };
    auto lds_put_k = [&](int buf) __attribute__((always_inline)) {
        char* kb = smem + buf * KT + srow * KROW + sch * 16;
#pragma unroll
        for (int i = 0; i < NP; ++i) *reinterpret_cast<uint4*>(kb + i * RPP * KROW) = kst[i];
    };
    // V^T row image: [hi of the 64 keys (128 B) | lo of the 64 keys (128 B)]; memory chunk c = (8-group c >> 1, hi / lo = c & 1)
    auto lds_put_v = [&](int buf) __attribute__((always_inline)) {
        char* vb = smem + 2 * KT + buf * VT + srow * VROW + (sch & 1) * 128 + (sch >> 1) * 16;
#pragma unroll
        for (int i = 0; i < NP; ++i) {   // 264-byte rows are only 8-byte aligned
            char* d = vb + i * RPP * VROW;
            *reinterpret_cast<uint2*>(d) = make_uint2(vst[i].x, vst[i].y);
            *reinterpret_cast<uint2*>(d + 8) = make_uint2(vst[i].z, vst[i].w);
        }
    };

    // ---- DMA staging: piece i of wave w fills rows 4 (w + NW i) .. + 3 of a tile; lane l -> row + (l >> 4), physical slot l & 15
    constexpr int DPW = 16 / NW;                                  // 1 KiB pieces per wave, tile and operand
    const int drow = wave * 4 + (lane >> 4);                      // + 4 NW i: the low four bits of the row (the swizzle key) do not depend on i
    const int dlog = (lane & 15) ^ (drow & 15);                   // logical slot held by this lane's physical slot
    const uint32_t dk_off = (uint32_t)(dlog * 16);                                        // K: logical slot = memory chunk
    const uint32_t dv_off = (uint32_t)(drow * p.ldv * 4 + ((((dlog & 7) << 1) | (dlog >> 3)) * 16));   // V^T: plane-major logical slot -> memory chunk (group, hi / lo)
    const uint32_t lds_w = lds_addr(smem) + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    auto dma_piece = [&](const void* sbase, uint32_t vo, uint32_t dst) __attribute__((always_inline)) {
        // M0 = LDS destination of the wave's 1 KiB; wave-uniform 64-bit base in an SGPR pair, 32-bit byte offset per lane. s_nop 4: an SGPR
        // freshly written by v_readfirstlane must not be read by a VMEM instruction within five wait states (hipcc does not see inside the asm)
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo), "s"(sbase), "s"(dst) : "memory", "m0");
    };
    auto dma_k = [&](int key0, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int key = min(key0 + drow + 4 * NW * i, p.Nk - 1);          // ragged last tile: rows beyond Nk repeat the last key (their scores are masked)
            dma_piece(kptr, (uint32_t)(key * ROWB) + dk_off, lds_w + (uint32_t)(buf * KT + i * NW * 1024));
        }
    };
    auto dma_v = [&](int key0, int buf) __attribute__((always_inline)) {   // V^T tiles are always inside the zero-padded ldv
#pragma unroll
        for (int i = 0; i < DPW; ++i)
            dma_piece(vptr + (size_t)key0 * 4 + (size_t)(4 * NW * i) * p.ldv * 4, dv_off, lds_w + (uint32_t)(2 * KT + buf * VT + i * NW * 1024));
    };

    const int ntiles = (p.Nk + 63) / 64;
    const float c = p.scale * 1.44269504088896340736f;  // fold log2(e): p = exp2(s c - m c)
    const float lz_thr = 6.0f / c;                        // LZ: six octaves, in score units
    const int koff = DMA ? l31 * KROW : l31 * KROW + hh * 32;   // this lane's K row (/ group inside a 32-key block: padded image)
    const int voff = DMA ? l31 * VROW : l31 * VROW + hh * 8;    // this lane's V^T row (/ 4-key slot: padded image)
    // swizzled images: byte offsets inside the 256-byte row. K: chunk 4 ks + 2 hh + lo; V^T: logical slot 2 g (+ 1: second 8-key group, + 8: lo plane)
    const int sx = (l31 & 15) * 16;
    int kxo[8], vxo[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) kxo[i] = (((i >> 1) * 4 + 2 * hh + (i & 1)) * 16) ^ sx;
#pragma unroll
    for (int g = 0; g < 4; ++g) vxo[g] = ((2 * g * 16) ^ sx) + hh * 8;
    // K fragment (step st = (ks, rb), half lo) and its address in both images
    auto kfrag_ptr = [&](const char* kb, int st, int lo) __attribute__((always_inline)) -> const char* {
        if constexpr (DMA) return kb + (st & 1) * 32 * KROW + kxo[(st >> 1) * 2 + lo];
        else return kb + (st & 1) * 32 * KROW + (st >> 1) * 64 + lo * 16;
    };

    // S^T[key][query] of one tile: 24 MFMAs; fragment reads one (ks, rb) step ahead
    auto qk_tile = [&](int kbuf, f32x16_t (&s)[2]) __attribute__((always_inline)) {
        const char* kb = smem + kbuf * KT + koff;
        s[0] = s[1] = (f32x16_t){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint4 kh = *reinterpret_cast<const uint4*>(kfrag_ptr(kb, 0, 0)), kl = *reinterpret_cast<const uint4*>(kfrag_ptr(kb, 0, 1));
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int ks = st >> 1, rb = st & 1;
            uint4 nh = kh, nl = kl;
            if (st + 1 < 8) {
                nh = *reinterpret_cast<const uint4*>(kfrag_ptr(kb, st + 1, 0));
                nl = *reinterpret_cast<const uint4*>(kfrag_ptr(kb, st + 1, 1));
            }
            TR::mma32x3(s[rb], kh, kl, qf[2 * ks], qf[2 * ks + 1]);
            kh = nh; kl = nl;
        }
    };

    f32x16_t o[2];
    o[0] = o[1] = (f32x16_t){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float m_run = -1e30f, l_run = 0.f;

    const int last = ntiles - 1;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    auto h8 = [](const u32x4_t& v) __attribute__((always_inline)) { return __builtin_bit_cast(f16x8_t, v); };
    auto q8 = [](const uint4& v) __attribute__((always_inline)) { return __builtin_bit_cast(f16x8_t, v); };
    // One tile: softmax of s_cur (tile t) under the QK^T of tile t + 1 into s_nxt, then PV of tile t with the split of P under it.
    // Interleaving is pinned at the IR level: an empty `asm volatile` that takes every live accumulator as a read-write operand
    // (PIN_A / PIN_B) is an ordering point no pure instruction can cross, so the program order between two pins is exactly
    // {one MFMA, one slice of VALU work} -- what an in-order wave needs to run its VALU under its own MFMAs. (Left to itself hipcc sinks
    // the 24 QK^T MFMAs, whose results are only needed one tile later, below the whole softmax; sched_group_barrier cannot pull them back.)
    // What depends on t + 1 < ntiles is a template parameter (the last tile is a peeled instance without QK^T); tail loads are clamped.
    auto tile_step = [&](auto has_next_c, int t, f32x16_t (&s_cur)[2], f32x16_t (&s_nxt)[2]) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next_c)::value;
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of K_{t+1} / V_t (issued one tile ago) have landed
        if constexpr (!(PROBE & 4)) __syncthreads();   // K_{t+1}, V_t are visible; every wave is done with K_t (ring slot t & 1) and V_{t-1} (slot (t + 1) & 1)
        const char* kb = smem + ((t + 1) & 1) * KT + koff;
        uint4 kh = make_uint4(0, 0, 0, 0), kl = kh;
        if constexpr (HAS_NEXT) {       // the first K fragment of S_{t+1}: requested right behind the barrier, used ~60 instructions later
            kh = *reinterpret_cast<const uint4*>(kfrag_ptr(kb, 0, 0));
            kl = *reinterpret_cast<const uint4*>(kfrag_ptr(kb, 0, 1));
        }
        if constexpr (HAS_NEXT && DMA) {
            dma_k(min(t + 2, last) * 64, t & 1);       // K_{t+2} (at t = ntiles - 2 a second copy of the last tile: that slot is not read again)
            dma_v((t + 1) * 64, (t + 1) & 1);          // V_{t+1}
        } else if constexpr (HAS_NEXT && !(PROBE & 8)) {
            if constexpr (!(PROBE & 16)) {
                lds_put_k(t & 1);            // K_{t+2} (at t = ntiles - 2 a second copy of the last tile: that slot is not read again)
                lds_put_v((t + 1) & 1);      // V_{t+1}
            } else {                         // probe: loads without the LDS writes (kept alive)
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const u32x4b_t kk = {kst[i].x, kst[i].y, kst[i].z, kst[i].w}, vv = {vst[i].x, vst[i].y, vst[i].z, vst[i].w};
                    asm volatile("" :: "v"(kk), "v"(vv));
                }
            }
            if constexpr (!(PROBE & 32)) {
                gload_k(min(t + 3, last) * 64);
                gload_v(min(t + 2, last) * 64);
            }
        } else if constexpr (!HAS_NEXT) {
            // keys beyond Nk: only the last tile of a ragged sequence has them (selects, no branch)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    s_cur[rb][r] = key < p.Nk ? s_cur[rb][r] : -1e30f;
                }
        }
        float mt = -1e30f, m_new = 0.f, alpha = 1.f, mcn = 0.f;
        bool resc = true;                             // LZ: does any query of this wave move its maximum on this tile (wave-uniform)
        v2f_t ps2 = {0.f, 0.f}, aa = {0.f, 0.f};      // row sum; the pair of exponents in flight between two slices
        float ps0 = 0.f, ps1 = 0.f, aa0 = 0.f, aa1 = 0.f;   // SC: the same as four scalars (the pins below carry whichever set is live)
        u32x4_t pH[2], pL[2];                         // P operands of PV group g in set g & 1 (hi and lo halves)
        // "defined" without an instruction: the sets are filled element by element under the MFMAs and pinned from the start
        asm volatile("" : "=v"(pH[0]), "=v"(pL[0]), "=v"(pH[1]), "=v"(pL[1]));
        // ordering points: every live accumulator is a read-write operand, "memory" keeps the LDS reads where they are written
#define PIN_A1() do { if constexpr (SC) asm volatile("" : "+v"(s_nxt[0]), "+v"(s_cur[0]), "+v"(s_cur[1]), "+v"(o[0]), "+v"(o[1]), "+v"(mt), "+v"(m_new), \
                              "+v"(alpha), "+v"(mcn), "+v"(ps0), "+v"(ps1), "+v"(aa0), "+v"(aa1) : : "memory"); \
                      else asm volatile("" : "+v"(s_nxt[0]), "+v"(s_cur[0]), "+v"(s_cur[1]), "+v"(o[0]), "+v"(o[1]), "+v"(mt), "+v"(m_new), \
                              "+v"(alpha), "+v"(mcn), "+v"(ps2), "+v"(aa) : : "memory"); } while (0)
#define PIN_A() do { if constexpr (SC) asm volatile("" : "+v"(s_nxt[0]), "+v"(s_nxt[1]), "+v"(s_cur[0]), "+v"(s_cur[1]), "+v"(o[0]), "+v"(o[1]), "+v"(mt), "+v"(m_new), \
                             "+v"(alpha), "+v"(mcn), "+v"(ps0), "+v"(ps1), "+v"(aa0), "+v"(aa1), "+v"(pH[0]), "+v"(pL[0]) : : "memory"); \
                     else asm volatile("" : "+v"(s_nxt[0]), "+v"(s_nxt[1]), "+v"(s_cur[0]), "+v"(s_cur[1]), "+v"(o[0]), "+v"(o[1]), "+v"(mt), "+v"(m_new), \
                             "+v"(alpha), "+v"(mcn), "+v"(ps2), "+v"(aa), "+v"(pH[0]), "+v"(pL[0]) : : "memory"); } while (0)
    // (hi, lo) fp16 split of two probabilities as packed operations: v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_add_f32 (negated), v_cvt_pk_f16_f32.
    // p = exp2(s c - m c) with m the running maximum: 0 <= p <= 1, no range clamp in front of the fp16 conversions
#define SPLIT2(x_, y_, hv_, lv_, idx_)                                                                    \
    {                                                                                                      \
        typedef _Float16 h2v_t __attribute__((ext_vector_type(2)));                                        \
        const v2f_t xv_ = {x_, y_};                                                                        \
        const h2v_t hh_ = __builtin_convertvector(xv_, h2v_t);                                             \
        v2f_t dv_;                                                                                         \
        if constexpr (SC) {   /* x - float(hi) as ONE v_fma_mix_f32 per element: the f16 -> f32 conversion rides in the instruction (hipcc emits v_cvt_f32_f16 + v_sub_f32 for the C form) */ \
            float d0_, d1_;                                                                                \
            const unsigned hu_ = __builtin_bit_cast(unsigned, hh_);                                        \
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0_) : "v"(hu_), "v"(x_));        \
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1_) : "v"(hu_), "v"(y_)); \
            dv_ = (v2f_t){d0_, d1_};                                                                       \
        }                                                                                                  \
        else dv_ = xv_ - __builtin_convertvector(hh_, v2f_t);                                              \
        const h2v_t ll_ = __builtin_convertvector(dv_, h2v_t);                                             \
        hv_[idx_] = __builtin_bit_cast(unsigned, hh_); lv_[idx_] = __builtin_bit_cast(unsigned, ll_);      \
    }
        // The softmax of tile t in 24 slices. The exponent / exp2 / row-sum chain of a pair of scores is spread over three consecutive
        // slices (a transcendental result cannot be consumed by the next instruction without wait states: fill them with the neighbours).
        auto pair_of = [&](int i, int& rb, int& r) __attribute__((always_inline)) { rb = (2 * i) >> 4; r = (2 * i) & 15; };
        auto valu_slice = [&](int v) __attribute__((always_inline)) {
            if (v < 4) {                       // running maximum of the tile: 8 scores per slice
                const f32x16_t& sv = s_cur[v >> 1];
                const int r0 = (v & 1) * 8;
                mt = fmaxf(fmaxf(mt, sv[r0]), sv[r0 + 1]);
                mt = fmaxf(fmaxf(mt, sv[r0 + 2]), sv[r0 + 3]);
                mt = fmaxf(fmaxf(mt, sv[r0 + 4]), sv[r0 + 5]);
                mt = fmaxf(fmaxf(mt, sv[r0 + 6]), sv[r0 + 7]);
            } else if (v == 4) {               // lanes l and l ^ 32 share a query
                mt = xor32_max(mt);
                if constexpr (LZ) {
                    m_new = mt > m_run + lz_thr ? mt : m_run;
                    resc = __builtin_amdgcn_ballot_w64(m_new != m_run) != 0ull;
                } else {
                m_new = fmaxf(m_run, mt);
                }
                alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
                mcn = -m_new * c;
                if constexpr (SC) {
                    aa0 = __builtin_fmaf(s_cur[0][0], c, mcn); aa1 = __builtin_fmaf(s_cur[0][1], c, mcn);
                } else {
                const v2f_t sv = {s_cur[0][0], s_cur[0][1]}, c2 = {c, c}, m2 = {mcn, mcn};
                aa = __builtin_elementwise_fma(sv, c2, m2);            // exponents of pair 0
                }
            } else if (v < 22) {               // slice 5 + i: exp2 of pair i (in place), exponents of pair i + 1, row sum += pair i - 1
                const int i = v - 5;
                if (i >= 1 && i <= 16) {
                    int rb, r;
                    pair_of(i - 1, rb, r);
                    if constexpr (SC) { ps0 += s_cur[rb][r]; ps1 += s_cur[rb][r + 1]; }
                    else {
                    const v2f_t pv = {s_cur[rb][r], s_cur[rb][r + 1]};
                    ps2 += pv;
                    }
                }
                if (i < 16) {
                    int rb, r;
                    pair_of(i, rb, r);
                    if constexpr (SC) { s_cur[rb][r] = __builtin_amdgcn_exp2f(aa0); s_cur[rb][r + 1] = __builtin_amdgcn_exp2f(aa1); }
                    else { s_cur[rb][r] = __builtin_amdgcn_exp2f(aa[0]); s_cur[rb][r + 1] = __builtin_amdgcn_exp2f(aa[1]); }
                }
                if (i + 1 < 16) {
                    int rb, r;
                    pair_of(i + 1, rb, r);
                    if constexpr (SC) {
                        aa0 = __builtin_fmaf(s_cur[rb][r], c, mcn); aa1 = __builtin_fmaf(s_cur[rb][r + 1], c, mcn);
                    } else {
                    const v2f_t sv = {s_cur[rb][r], s_cur[rb][r + 1]}, c2 = {c, c}, m2 = {mcn, mcn};
                    aa = __builtin_elementwise_fma(sv, c2, m2);
                    }
                }
                if (i == 16 && resc) {         // rescale O: first d-block (the PV MFMAs of this tile come after phase A)
                    if constexpr (SC) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[0][r] *= alpha;
                    } else {
                    const v2f_t al2 = {alpha, alpha};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const v2f_t ov = {o[0][r], o[0][r + 1]};
                        const v2f_t t2 = ov * al2;
                        o[0][r] = t2[0]; o[0][r + 1] = t2[1];
                    }
                    }
                }
            } else if (v == 22) {              // second d-block
                if (!resc) {
                } else
                if constexpr (SC) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[1][r] *= alpha;
                } else {
                const v2f_t al2 = {alpha, alpha};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const v2f_t ov = {o[1][r], o[1][r + 1]};
                    const v2f_t t2 = ov * al2;
                    o[1][r] = t2[0]; o[1][r + 1] = t2[1];
                }
                }
            } else {                           // bookkeeping + the first P operand of phase B
                l_run = l_run * alpha + (SC ? ps0 + ps1 : ps2[0] + ps2[1]);
                m_run = m_new;
                SPLIT2(s_cur[0][0], s_cur[0][1], pH[0], pL[0], 0)
                SPLIT2(s_cur[0][2], s_cur[0][3], pH[0], pL[0], 1)
                SPLIT2(s_cur[0][4], s_cur[0][5], pH[0], pL[0], 2)
                SPLIT2(s_cur[0][6], s_cur[0][7], pH[0], pL[0], 3)
            }
        };
        // ---- phase A: S_{t+1} = K_{t+1} Q^T  ||  softmax(S_t) -----------------------------------------------------------------
        if constexpr (HAS_NEXT) {
            const f32x16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int ks = st >> 1, rb = st & 1;
                uint4 nh = kh, nl = kl;
                if (st + 1 < 8) {          // the next step's fragments, one step (3 MFMAs) ahead of their use
                    nh = *reinterpret_cast<const uint4*>(kfrag_ptr(kb, st + 1, 0));
                    nl = *reinterpret_cast<const uint4*>(kfrag_ptr(kb, st + 1, 1));
                }
                if constexpr (!(PROBE & 2)) s_nxt[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q8(kl), q8(qf[2 * ks]), st < 2 ? zero16 : s_nxt[rb], 0, 0, 0);
                else if (st < 2) s_nxt[rb] = zero16;
                if (st == 0) { PIN_A1(); } else { PIN_A(); }
                if constexpr (!(PROBE & 1)) valu_slice(3 * st);
                if constexpr (!(PROBE & 2)) s_nxt[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q8(kh), q8(qf[2 * ks + 1]), s_nxt[rb], 0, 0, 0);
                if (st == 0) { PIN_A1(); } else { PIN_A(); }
                if constexpr (!(PROBE & 1)) valu_slice(3 * st + 1);
                if constexpr (!(PROBE & 2)) s_nxt[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q8(kh), q8(qf[2 * ks]), s_nxt[rb], 0, 0, 0);
                if (st == 0) { PIN_A1(); } else { PIN_A(); }
                if constexpr (!(PROBE & 1)) valu_slice(3 * st + 2);
                kh = nh; kl = nl;
            }
            PIN_A();
        } else {
#pragma unroll
            for (int v = 0; v < 24; ++v) if constexpr (!(PROBE & 1)) valu_slice(v);
        }
        // ---- phase B: O^T += V_t^T P_t^T, contraction over keys permuted identically on both operands ---------------------------
        // unit u = (group g = (rb, sh): 16 keys, d-block db): 3 MFMAs; the next unit's V^T fragments are read, and (over a group's two
        // units) the next group's P operand is split, under them
        const char* vb = smem + 2 * KT + (t & 1) * VT + voff;
#define PIN_B() asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(s_cur[0]), "+v"(s_cur[1]), "+v"(pH[0]), "+v"(pL[0]), "+v"(pH[1]), "+v"(pL[1]) : : "memory")
        auto vfrag = [&](int u, u32x4_t& vh, u32x4_t& vl) __attribute__((always_inline)) {
            const int g = u >> 1, db = u & 1, rb = g >> 1, sh = g & 1;
            // keys kbase .. + 3 and kbase + 8 .. + 11 of this lane half (the 4 hh part sits in voff): hi plane byte 2 kbase, lo plane + 128
            uint2 h0, h1, l0, l1;
            if constexpr (DMA) {            // swizzled image: logical slots 2 g, 2 g + 1 (hi plane) and + 8 (lo plane), XORed with the row key
                const char* vr = vb + db * 32 * VROW;
                const int a0 = vxo[g];
                h0 = *reinterpret_cast<const uint2*>(vr + a0); h1 = *reinterpret_cast<const uint2*>(vr + (a0 ^ 16));
                l0 = *reinterpret_cast<const uint2*>(vr + (a0 ^ 128)); l1 = *reinterpret_cast<const uint2*>(vr + (a0 ^ 144));
            } else {
                const char* vd = vb + (rb * 32 + 16 * sh) * 2 + db * 32 * VROW;
                h0 = *reinterpret_cast<const uint2*>(vd); h1 = *reinterpret_cast<const uint2*>(vd + 16);
                l0 = *reinterpret_cast<const uint2*>(vd + 128); l1 = *reinterpret_cast<const uint2*>(vd + 144);
            }
            vh = (u32x4_t){h0.x, h0.y, h1.x, h1.y};
            vl = (u32x4_t){l0.x, l0.y, l1.x, l1.y};
        };
        u32x4_t vh, vl;
        vfrag(0, vh, vl);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int g = u >> 1, db = u & 1, cs = g & 1, ns = cs ^ 1;
            const int nrb = (g + 1) >> 1, nsh = (g + 1) & 1;
            u32x4_t nvh = vh, nvl = vl;
            if (u + 1 < 8) vfrag(u + 1, nvh, nvl);
            if constexpr (!(PROBE & 2)) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(vl), h8(pH[cs]), o[db], 0, 0, 0);
            PIN_B();
            if constexpr (!(PROBE & 1)) if (g + 1 < 4) SPLIT2(s_cur[nrb][8 * nsh + 4 * db], s_cur[nrb][8 * nsh + 4 * db + 1], pH[ns], pL[ns], 2 * db)
            if constexpr (!(PROBE & 2)) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(vh), h8(pL[cs]), o[db], 0, 0, 0);
            PIN_B();
            if constexpr (!(PROBE & 1)) if (g + 1 < 4) SPLIT2(s_cur[nrb][8 * nsh + 4 * db + 2], s_cur[nrb][8 * nsh + 4 * db + 3], pH[ns], pL[ns], 2 * db + 1)
            if constexpr (!(PROBE & 2)) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(vh), h8(pH[cs]), o[db], 0, 0, 0);
            PIN_B();
            vh = nvh; vl = nvl;
        }
#undef PIN_A
#undef PIN_A1
#undef PIN_B
#undef SPLIT2
    };
    const std::true_type has_next{};
    const std::false_type is_last{};

    // ---- prologue: K_0, V_0 -> LDS; K_1 -> LDS; K_2 / V_1 in flight; S_0 ---------------------------------------------------
    f32x16_t sa[2], sb[2];
    if constexpr (DMA) {
        dma_k(0, 0);
        dma_v(0, 0);
        dma_k(min(1, last) * 64, 1);     // K_1 (ntiles == 1: a second copy of K_0, never read)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        qk_tile(0, sa);
    } else {
    gload_k(0);
    gload_v(0);
    lds_put_k(0);
    lds_put_v(0);
    gload_k(min(1, last) * 64);
    __syncthreads();
    qk_tile(0, sa);
    lds_put_k(1);                        // K_1 (ntiles == 1: a second copy of K_0, never read)
    gload_k(min(2, last) * 64);
    gload_v(min(1, last) * 64);
    }
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {     // both steps have a next tile
        tile_step(has_next, t, sa, sb);
        tile_step(has_next, t + 1, sb, sa);
    }
    if (t + 1 < ntiles) {                // two tiles left
        tile_step(has_next, t, sa, sb);
        tile_step(is_last, t + 1, sb, sa);
    } else {                             // one tile left
        tile_step(is_last, t, sa, sb);
    }

    // ---- normalise and store (4 consecutive d per register group) -------------------------------
    const float inv = 1.0f / xor32_sum(l_run);
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

template <int ODT, int PROBE, int NW = 4, bool DMA = false, bool SC = false, bool LZ = false> static hipError_t launch_x3_v2p(const AttnParams& p, hipStream_t s) {
    constexpr int LDS = DMA ? 4 * 64 * 256 : 2 * 64 * (256 + 16) + 2 * 64 * (256 + 8);
    static std::atomic<unsigned long long> attr_done{0};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    const unsigned long long dev_bit = 1ull << (dev_id & 63);
    if (!(attr_done.load(std::memory_order_relaxed) & dev_bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_x3_kernel<ODT, PROBE, NW, DMA, SC, LZ>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done.fetch_or(dev_bit, std::memory_order_relaxed);
    }
    const int grid = p.B * p.H * ((p.Nq + NW * 32 - 1) / (NW * 32));
    hipLaunchKernelGGL((attention_x3_kernel<ODT, PROBE, NW, DMA, SC, LZ>), dim3(grid), dim3(NW * 64), LDS, s, p);
    return hipGetLastError();
}
template <int ODT> static hipError_t launch_x3_v2(const AttnParams& p, hipStream_t s) {
    if constexpr (kProbes && ODT == D3R_F16X3) {
        if (const char* e = probe_env("D3R_ATTN_PROBE")) {      // ablation instances (results invalid), see the kernel
            switch (atoi(e)) {
                case 1: return launch_x3_v2p<ODT, 1>(p, s);
                case 2: return launch_x3_v2p<ODT, 2>(p, s);
                case 4: return launch_x3_v2p<ODT, 4>(p, s);
                case 8: return launch_x3_v2p<ODT, 8>(p, s);
                case 12: return launch_x3_v2p<ODT, 12>(p, s);
                case 13: return launch_x3_v2p<ODT, 13>(p, s);
                case 14: return launch_x3_v2p<ODT, 14>(p, s);
                case 16: return launch_x3_v2p<ODT, 16>(p, s);
                case 32: return launch_x3_v2p<ODT, 32>(p, s);
                default: break;
            }
        }
    }
    if constexpr (kProbes) if (const char* e = probe_env("D3R_ATTN_NW")) {
        if (e[0] == '8' && e[1] == 0) return launch_x3_v2p<ODT, 0, 8>(p, s);   // probe: 256 queries per workgroup (register staging, packed softmax: round 3's instance)
        // probe '8d' / '8e': 256 queries per workgroup with DMA staging and the scalar softmax slices, everywhere / for launches of >= 2048 workgroups of 128 queries only
        if (e[0] == '8' && (e[1] == 'd' || (e[1] == 'e' && (long)p.B * p.H * ((p.Nq + 127) / 128) >= 4096))) return launch_x3_v2p<ODT, 0, 8, true, true>(p, s);
    }
    // K / V^T tiles by global_load_lds DMA into swizzled 256-byte rows (default since round 4; D3R_ATTN_DMA=0: staged through registers into padded
    // rows; read per launch). Measured (profiles/r04_b/attndma.log, ab_attn_dma.txt): 64 x 16 heads 512 -> 492 us, 32 x 12 heads equal, forward
    // 193.15 -> 193.5 pairs/s; bit-identical outputs (tests/test_kernels_gpu.py::test_attention_split_fp16_dma_staging_is_bit_identical).
    const char* e_dma = getenv("D3R_ATTN_DMA");
    // The softmax / split slices on scalar fp32 VALU (default since round 5; D3R_ATTN_SC=0: the packed v_pk_* form; bit-identical; read per launch).
    // Measured in one process on one box (tools/ab_probe.py, profiles/r05_b/ab_probe.log, three alternating repetitions of the 32-pair forward):
    // attention 21.20 -> 20.72 ms per step, forward 171.29 -> 170.83 ms.
    const char* e_sc = probe_env("D3R_ATTN_SC");
    const char* e_lz = probe_env("D3R_ATTN_LAZY");          // 1: lazy running maximum (round 5 probe; read per launch)
    const bool dma = e_dma ? e_dma[0] != '0' : true;
    if constexpr (kProbes) {        // the lazy-maximum and DMA + packed-softmax instances: probe builds only (-DD3R_PROBES)
        if (dma && !(e_sc && e_sc[0] == '0') && e_lz && e_lz[0] == '1') return launch_x3_v2p<ODT, 0, 4, true, true, true>(p, s);
        if (dma && e_sc && e_sc[0] == '0') return launch_x3_v2p<ODT, 0, 4, true>(p, s);
    }
    (void)e_sc; (void)e_lz;
    return dma ? launch_x3_v2p<ODT, 0, 4, true, true>(p, s) : launch_x3_v2p<ODT, 0>(p, s);
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

hipError_t launch_attention(int dt, const AttnParams& p_in, hipStream_t s) {
    AttnParams p = p_in;
    if (p.out_dt >= 0) p.out_dt = d3r_act_dt(p.out_dt);     // rows for a 2.5-unit proj GEMM are fp16 + fp8 activation rows
    if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0 || p.ldv % 64 != 0 || p.ldv < ((p.Nk + 63) / 64) * 64)
        return hipErrorInvalidValue;
    if (p.out_dt >= 0 && p.out_dt != dt && !(dt == D3R_F16X3 && p.out_dt == D3R_F16F8)) return hipErrorInvalidValue;
    switch (dt) {
        case D3R_BF16: return launch_t<D3R_BF16>(p, s);
        case D3R_F16: return launch_t<D3R_F16>(p, s);
        case D3R_F32: return launch_t<D3R_F32>(p, s);
        case D3R_F16X3: {
            const char* e_v1 = getenv("D3R_ATTN_V1");          // 1: the round-2 kernel (A/B runs, parity tests); read on every launch
            const bool v1 = e_v1 && e_v1[0] == '1';
            if (!v1) return p.out_dt == D3R_F16F8 ? launch_x3_v2<D3R_F16F8>(p, s) : launch_x3_v2<D3R_F16X3>(p, s);
            if (p.out_dt == D3R_F16F8) return launch_t<D3R_F16X3, D3R_F16F8>(p, s);   // q, k, v^T split-fp16; output rows for an fp16 + fp8 proj GEMM
            return launch_t<D3R_F16X3>(p, s);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace d3r
