// dust3r_amd -- persistent split-fp16 GEMM whose epilogue runs UNDER the next tile's K loop (gfx950; round 6).
//
// What it is for (reference call sites): the nn.Linear layers of the 24 encoder / 2 x 12 decoder blocks at batch sizes whose GEMMs fill the
// chip (dust3r/model.py:136-137,180-186 -> croco Block / DecoderBlock: qkv, proj, fc1, fc2, projq / projk|projv). In gemm.hip one block
// computes one tile and then stores it: for K <= 1024 a tile spends 15-25 % of its life in the epilogue with the CU's matrix pipes idle
// (tools/tile_probe.py: the same launches without their epilogue run 505-514 TFLOP/s against 371-438 with it).
//
// Shape. One block per CU, FOUR waves, one per SIMD, each with the whole 512-entry register file: 256 accumulator registers = TWO sets of
// the 128 (n) x 64 (m) wave tile (8 x 4 fragments of v_mfma_f32_16x16x32_f16, the wave tile of gemm.hip's 256 x 256 shape, so the LDS read
// traffic per MFMA is the same). Block tile M 256 x N 128. A block walks its tiles (the XCD-contiguous panel order of gemm.hip, strided by
// the grid); set C accumulates tile t while set D -- tile t - 1 -- is drained in 16 micro-slices, one per K step, placed between the MFMAs:
// accumulator -> bias / folded LayerNorm / GELU / split -> wave-private LDS rows -> 16-byte global stores. Operands arrive by
// global_load_lds DMA into a THREE-slot ring that never drains between tiles (the loads run two K steps ahead of the math, across tile
// boundaries); one s_barrier per K step, in the MIDDLE of the step's MFMA stream (the slot it publishes is the NEXT step's, the slot it
// frees is filled behind it), the 12 DMA pieces of a step and the head fragments of the next are spread between the MFMA rows.
// Every VMEM operation of the kernel is issued from inline asm and counted by hand (hipcc's own counter cannot see the DMA: any load it
// knows about would wait for everything in flight): vmcnt(n) below always names how many YOUNGER operations may stay outstanding.
//
// Arithmetic: per output element the same MFMA sequence as every gemm.hip tile (per 32 k: lo.hi, hi.lo, hi.hi; K ascending) and the same
// epilogue expressions, so results are bit-identical to them (tests/test_kernels_gpu.py::test_persistent_gemm_*).
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "kernels.hpp"

namespace d3r {
namespace p4 {

constexpr int BM = 256, BN = 128, NW = 4, NT = 256, FI = 8, FJ = 4, KTB = 128, NST = 3;
constexpr int A_BYTES = BM * KTB, W_BYTES = BN * KTB, STAGE = A_BYTES + W_BYTES;   // 32 KiB + 16 KiB per ring slot
constexpr int STG0 = NST * STAGE, STG_W = 16 * 128;          // drain staging: 16 rows x 128 bytes per wave (XOR-swizzled 16-byte slots)
constexpr int SIDE0 = STG0 + NW * STG_W, SIDE_W = 1536;      // per wave: bias[128] | colsum[128] | rstd[64] | nmr[64]  (fp32)
constexpr int DUMMY0 = SIDE0 + NW * SIDE_W;                  // 256 bytes per wave: where the DMA of an operand the launch does not have lands
constexpr int LDS = DUMMY0 + NW * 256;                       // 162 816 of the CU's 163 840 bytes
constexpr int DS = 16;                                       // drain micro-slices = K steps that carry one
static_assert(LDS <= 160 * 1024, "one block owns the CU's LDS");

enum { EPK_TYPED = 0, EPK_GELU = 1, EPK_X3RES = 2, EPK_X3RES_LN = 3, EPK_NONE = 4 };   // _LN: the launch also writes the row partial sums of a folded LayerNorm; NONE: probe builds (GF_NOSTORE), the K loops alone

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// 4-byte LDS-DMA: lane l's dword lands at lds_dst + 4 l
D3R_DEV void glds4(const void* gsrc, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
// 16-byte store, wave-uniform 64-bit base + per-lane 32-bit byte offset; NT: non-temporal policy
template <bool NTP> D3R_DEV void gst16(void* sbase, uint32_t voff, const u32x4_t v) {
    if constexpr (NTP) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
D3R_DEV void gst8(void* sbase, uint32_t voff, const float2 v) {
    asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
// 16-byte load into registers, invisible to hipcc's vmcnt bookkeeping: the caller waits (counted) before the first use and passes the
// value through use_after_wait() so that no consumer is scheduled above the wait
D3R_DEV u32x4_t gld16(const void* sbase, uint32_t voff) {
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
// LDS-DMA piece: source = wave-uniform 64-bit base + this lane's 32-bit offset; destination = slot base (SGPR) + a compile-time offset, formed in M0
template <int IMM> D3R_DEV void glds16_imm(const void* sbase, uint32_t voff, uint32_t lds_slot) {
    asm volatile("s_add_i32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_slot), "n"(IMM) : "memory", "m0", "scc");     // s_add writes SCC
}
template <int N> D3R_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
D3R_DEV void pin(u32x4_t& v) { asm volatile("" : "+v"(v)); }

D3R_DEV void tile_origin(int v, int ntiles, int tiles_m, int tiles_n, int panel_w, int& m0, int& n0) {
    const int lid = xcd_remap(v, ntiles);
    const int per_panel = panel_w * tiles_m;
    const int panel = lid / per_panel, rem = lid - panel * per_panel;
    const int width = min(panel_w, tiles_n - panel * panel_w);
    const int tm = rem / width, tn = panel * panel_w + (rem - tm * width);
    m0 = tm * BM;
    n0 = tn * BN;
}

D3R_DEV int key2(int row) { return (row ^ (row >> 3)) & 7; }      // staging rows 0..15: rows r, r + 8 and the 8 rows of a pass all differ

template <int EPK>
__global__ __launch_bounds__(NT, 1) void gemm_p4_kernel(GemmParams p, int tiles_m, int tiles_n, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TX = Traits<D3R_F16X3>;
    typedef std::integral_constant<bool, true> T_;
    typedef std::integral_constant<bool, false> F_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K >> 5;
    const int G = gridDim.x;
    constexpr bool X3R = EPK == EPK_X3RES || EPK == EPK_X3RES_LN;     // typed residual stream epilogue (GF_X3RES)
    constexpr bool LNP = EPK == EPK_X3RES_LN;

    // ---- DMA: one 32-bit offset per operand, the 32-row pass stride added to the wave-uniform tile base (SGPRs) --------------------------
    const int lrow = wave * 8 + (lane >> 3);
    const int lslot = (lane & 7) ^ ((lrow >> 1) & 7);
    const int lchunk = (lslot & 3) * 2 + (lslot >> 2);            // LDS image [hi0..hi3 | lo0..lo3] of the memory row's [hi x8][lo x8] groups
    // per-lane byte offsets of the 8 activation and 4 weight passes of a K step (32 rows apart): every piece then takes the SAME wave-uniform base
    uint32_t a_off[8], w_off[4];
#pragma unroll
    for (int q = 0; q < 8; ++q) a_off[q] = (uint32_t)(((size_t)(q * 32 + lrow) * p.lda) * 4 + lchunk * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) w_off[q] = (uint32_t)(((size_t)(q * 32 + lrow) * p.K) * 4 + lchunk * 16);
    const uint32_t lds0 = lds_addr(smem) + (uint32_t)wave * 1024;
    // The load cursor runs two K steps ahead of the math and never stops: behind this block's last tile it wraps to its first one (two steps of
    // loads nobody reads) so that every step issues exactly 12 pieces -- no branch in the MFMA stream, and the vmcnt arithmetic below is exact.
    int ld_v = blockIdx.x, ld_kt = 0;
    uint32_t ld_sb = lds0;                       // LDS base of the cursor's ring slot (+ this wave's 1 KiB)
    const char* ld_a = nullptr;
    const char* ld_w = nullptr;
    auto ld_set_tile = [&](int v) __attribute__((always_inline)) {
        int m0, n0;
        tile_origin(v, ntiles, tiles_m, tiles_n, p.panel, m0, n0);
        ld_a = reinterpret_cast<const char*>(p.act) + (size_t)m0 * p.lda * 4;
        ld_w = reinterpret_cast<const char*>(p.wgt) + (size_t)n0 * p.K * 4;
    };
    // piece IDX (0..11) of the load cursor's K step: 8 activation passes, 4 weight passes
    auto dma_piece = [&](auto idx_tag) __attribute__((always_inline)) {
        constexpr int IDX = decltype(idx_tag)::value;
        if constexpr (IDX < 8) glds16_imm<IDX * 4096>(ld_a, a_off[IDX], ld_sb);
        else glds16_imm<A_BYTES + (IDX - 8) * 4096>(ld_w, w_off[IDX - 8], ld_sb);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {
        ld_a += KTB;
        ld_w += KTB;
        ld_sb = ld_sb == lds0 + (NST - 1) * STAGE ? lds0 : ld_sb + STAGE;
        if (++ld_kt == nk) {
            ld_kt = 0;
            ld_v += G;
            if (ld_v >= ntiles) ld_v = blockIdx.x;
            ld_set_tile(ld_v);
        }
    };

    // ---- fragment reads ---------------------------------------------------------------------------------------------------------------------
    const int frow = lane & 15, fgrp = lane >> 4, fsw = (frow >> 1) & 7;
    const int chi = (fgrp ^ fsw) * 16, clo = ((4 + fgrp) ^ fsw) * 16;
    const int q_base = (wave * 64 + frow) * KTB;                  // activation rows of this wave (j side)
    const int p_base = A_BYTES + frow * KTB;                      // weight rows (i side: 4 consecutive n per lane)
    auto frag = [&](int slot, int off) __attribute__((always_inline)) { return *reinterpret_cast<const uint4*>(smem + slot * STAGE + off); };

    // ---- drain: constants and state -----------------------------------------------------------------------------------------------------------
    const int i4 = (lane >> 4) * 4, jl = lane & 15;
    const int rrow = lane >> 3, rch = lane & 7;
    char* const stg = smem + STG0 + wave * STG_W;
    float* const side = reinterpret_cast<float*>(smem + SIDE0 + wave * SIDE_W);
    const bool has_ln = p.ln_rstd != nullptr, has_bias = p.bias != nullptr;
    const float* const bsrc = p.bias ? p.bias : reinterpret_cast<const float*>(p.wgt);
    const float* const lnr = has_ln ? p.ln_rstd : bsrc;
    const float* const lnn = has_ln ? p.ln_nmr : bsrc;
    const float* const lns = has_ln ? p.ln_colsum : bsrc;
    const bool has_res = p.res1 != nullptr;
    int dm0 = 0, dn0 = 0;                                          // origin of the tile held by the drain set
    // Side buffer of this wave: bias[128] | colsum[128] | rstd[64] | nmr[64]. An operand the launch does not have keeps the NEUTRAL value written
    // here once (bias 0; no folded LayerNorm: colsum 0, rstd 1, nmr 0 -- fma(acc, 1, fma(0, 0, b)) = acc + b bit for bit, as in gemm.hip) and its
    // DMA is pointed at a dummy area instead: the same six instructions every tile, no branch, no select in the drain.
    const uint32_t sd = lds_addr(side), dummy = lds_addr(smem) + DUMMY0 + (uint32_t)wave * 256;
    side[lane] = 0.f; side[64 + lane] = 0.f; side[128 + lane] = 0.f; side[192 + lane] = 0.f; side[256 + lane] = 1.f; side[320 + lane] = 0.f;
    auto side_loads = [&]() __attribute__((always_inline)) {
        glds4(bsrc + dn0 + lane, has_bias ? sd : dummy);
        glds4(bsrc + dn0 + 64 + lane, has_bias ? sd + 256 : dummy);
        glds4(lns + (has_ln ? dn0 : 0) + lane, has_ln ? sd + 512 : dummy);
        glds4(lns + (has_ln ? dn0 + 64 : 0) + lane, has_ln ? sd + 768 : dummy);
        glds4(lnr + (has_ln ? dm0 + wave * 64 : 0) + lane, has_ln ? sd + 1024 : dummy);
        glds4(lnn + (has_ln ? dm0 + wave * 64 : 0) + lane, has_ln ? sd + 1280 : dummy);
    };

    f32x4_t acc[FI][FJ], dacc[FI][FJ];
    uint4 qh[FJ], ql[FJ];            // activation fragments of the current K step (reloaded in place behind their last MFMA)
    uint4 ph, pl;                    // weight fragment of the next MFMA row

    // Drain step D (0..15) = micro-slice D of the drain set: fragment columns fi = 2 g, 2 g + 1 (32 columns), fragment row fj (16 rows), g = D >> 2,
    // fj = D & 3. Its work is cut into the 24 SLOTS of a K step (one behind each group of four MFMAs), a handful of instructions each, so that the
    // matrix pipe never waits for a block of VALU work (hipcc left alone emits one 70-instruction block per fragment):
    //   typed / GELU:  fragment fl at slots 11 fl + 0..10: operands | folded-LayerNorm fmas | GELU in 8 pieces (scalar fp32: packed VALU beside MFMAs
    //                  is an anti-lever, MI355X_MICROARCH.md) | split + staging writes;   slot 22: read the two row passes back;  slot 23: two 16-byte stores
    //   typed residual stream: slot 0 / 1: fragment + bias -> staging (fp32);  row pass ps at slots 2 + 10 ps + 0..9: read back | residual halves swapped
    //                  | joins | add | splits | swap back | sums | DPP tree;  stores at slots 18, 19 (pass 0) and 22, 23 (pass 1): behind the last DMA piece
    constexpr int NSTORE = X3R ? (LNP ? 4 : 2) : 2;                // VMEM stores of a drain step, all younger than its last DMA piece
    constexpr int NHEAD = X3R ? 2 : 0;                             // residual-row requests at the top of a drain step (for the NEXT step)
    u32x4_t rr[2][2];                                              // X3R: residual rows [step parity][pass]
    auto rr_request = [&](auto d_tag) __attribute__((always_inline)) {      // the residual rows drain step D will add (issued one step ahead)
        constexpr int D = decltype(d_tag)::value;
        if constexpr (X3R && D >= 0 && D < DS) {
            constexpr int g = D >> 2, fj = D & 3;
            const char* rsrc = reinterpret_cast<const char*>(has_res ? p.res1 : (const void*)p.out2);
            const int rld = has_res ? p.ldr : p.ldo2;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const char* base = rsrc + TX::boff((size_t)(dm0 + wave * 64 + fj * 16 + ps * 8) * rld + dn0 + g * 32);
                rr[D & 1][ps] = gld16(base, (uint32_t)(TX::boff((size_t)rrow * rld + (rch >> 1) * 8) + (rch & 1) * 16));
            }
        }
    };
    // state that lives across the slots of a step
    float fv0, fv1, fv2, fv3;                                      // fragment values in flight
    float gz0, gz1, gz2, gz3, gt0, gt1, gt2, gt3, gp0, gp1, gp2, gp3, ge0, ge1, ge2, ge3;     // GELU: |z|, t, polynomial, exponent / half
    float4 sB, sX; float sR, sN;                                   // side operands of the fragment
    uint4 rv0, rv1;                                                // typed: the two row passes read back
    float4 xv; uint32_t xhx, xhy, xlx, xly; float xj0, xj1, xj2, xj3; uint2 xh, xl; u32x4_t xsv, xsv0; float xsm, xsq, xsm0, xsq0;
    auto swap_pair = [](uint32_t x) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true); };   // lane ^ 1
    auto dpp_add = [](float x, auto ctl) __attribute__((always_inline)) { return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctl)::value, 0xF, 0xF, false)); };
    auto drain_slot = [&](auto d_tag, auto slot_tag, auto tail_tag) __attribute__((always_inline)) {
        constexpr int D = decltype(d_tag)::value, SL = decltype(slot_tag)::value;
        constexpr bool TAIL = decltype(tail_tag)::value;             // the block's last tile: no K loop (no DMA pieces) around the slots
        constexpr int g = D >> 2, fj = D & 3;
        if constexpr (!X3R) {
            constexpr int fl = SL >= 11 ? 1 : 0, k = SL - 11 * fl;      // fragment and stage (SL 22, 23: the row passes)
            constexpr int fi = 2 * g + fl;
            if constexpr (SL < 22) {
                if constexpr (k == 0) {
                    const f32x4_t a = dacc[fi][fj];
                    fv0 = a[0]; fv1 = a[1]; fv2 = a[2]; fv3 = a[3];
                    sB = *reinterpret_cast<const float4*>(side + g * 32 + fl * 16 + i4);
                    sX = *reinterpret_cast<const float4*>(side + 128 + g * 32 + fl * 16 + i4);
                    sR = side[256 + fj * 16 + jl];
                    sN = side[320 + fj * 16 + jl];
                } else if constexpr (k == 1) {      // fma(acc, R_j, fma(S_i, Nm_j, bias_i)): the expression of gemm.hip's wide split-fp16 epilogue
                    fv0 = __builtin_fmaf(fv0, sR, __builtin_fmaf(sX.x, sN, sB.x)); fv1 = __builtin_fmaf(fv1, sR, __builtin_fmaf(sX.y, sN, sB.y));
                    fv2 = __builtin_fmaf(fv2, sR, __builtin_fmaf(sX.z, sN, sB.z)); fv3 = __builtin_fmaf(fv3, sR, __builtin_fmaf(sX.w, sN, sB.w));
                } else if constexpr (k < 10) {
                    if constexpr (EPK == EPK_GELU) {
                        // gelu_pk (common.hpp) element by element, same operations in the same order: x/2 + |x|/2 erf(|z|), z = x / sqrt 2,
                        // erf(|z|) = 1 - poly(t) exp(-z^2), t = 1 / (1 + p |z|)   (Abramowitz-Stegun 7.1.26)
                        constexpr float K0 = 0.70710678118654752440f, KL = -1.44269504088896340736f;
                        if constexpr (k == 2) {
                            gz0 = fabsf(fv0 * K0); gz1 = fabsf(fv1 * K0); gz2 = fabsf(fv2 * K0); gz3 = fabsf(fv3 * K0);
                            gt0 = __builtin_fmaf(gz0, 0.3275911f, 1.0f); gt1 = __builtin_fmaf(gz1, 0.3275911f, 1.0f);
                            gt2 = __builtin_fmaf(gz2, 0.3275911f, 1.0f); gt3 = __builtin_fmaf(gz3, 0.3275911f, 1.0f);
                        } else if constexpr (k == 3) {
                            gt0 = __builtin_amdgcn_rcpf(gt0); gt1 = __builtin_amdgcn_rcpf(gt1); gt2 = __builtin_amdgcn_rcpf(gt2); gt3 = __builtin_amdgcn_rcpf(gt3);
                        } else if constexpr (k == 4) {
                            gp0 = __builtin_fmaf(gt0, 1.061405429f, -1.453152027f); gp1 = __builtin_fmaf(gt1, 1.061405429f, -1.453152027f);
                            gp2 = __builtin_fmaf(gt2, 1.061405429f, -1.453152027f); gp3 = __builtin_fmaf(gt3, 1.061405429f, -1.453152027f);
                            gp0 = __builtin_fmaf(gp0, gt0, 1.421413741f); gp1 = __builtin_fmaf(gp1, gt1, 1.421413741f);
                            gp2 = __builtin_fmaf(gp2, gt2, 1.421413741f); gp3 = __builtin_fmaf(gp3, gt3, 1.421413741f);
                        } else if constexpr (k == 5) {
                            gp0 = __builtin_fmaf(gp0, gt0, -0.284496736f); gp1 = __builtin_fmaf(gp1, gt1, -0.284496736f);
                            gp2 = __builtin_fmaf(gp2, gt2, -0.284496736f); gp3 = __builtin_fmaf(gp3, gt3, -0.284496736f);
                            gp0 = __builtin_fmaf(gp0, gt0, 0.254829592f); gp1 = __builtin_fmaf(gp1, gt1, 0.254829592f);
                            gp2 = __builtin_fmaf(gp2, gt2, 0.254829592f); gp3 = __builtin_fmaf(gp3, gt3, 0.254829592f);
                        } else if constexpr (k == 6) {
                            gp0 = gp0 * gt0; gp1 = gp1 * gt1; gp2 = gp2 * gt2; gp3 = gp3 * gt3;
                            ge0 = (gz0 * KL) * gz0; ge1 = (gz1 * KL) * gz1; ge2 = (gz2 * KL) * gz2; ge3 = (gz3 * KL) * gz3;
                        } else if constexpr (k == 7) {
                            ge0 = __builtin_amdgcn_exp2f(ge0); ge1 = __builtin_amdgcn_exp2f(ge1); ge2 = __builtin_amdgcn_exp2f(ge2); ge3 = __builtin_amdgcn_exp2f(ge3);
                        } else if constexpr (k == 8) {
                            gp0 = __builtin_fmaf(-gp0, ge0, 1.0f); gp1 = __builtin_fmaf(-gp1, ge1, 1.0f); gp2 = __builtin_fmaf(-gp2, ge2, 1.0f); gp3 = __builtin_fmaf(-gp3, ge3, 1.0f);
                            ge0 = fv0 * 0.5f; ge1 = fv1 * 0.5f; ge2 = fv2 * 0.5f; ge3 = fv3 * 0.5f;
                        } else {
                            fv0 = __builtin_fmaf(gz0 * K0, gp0, ge0); fv1 = __builtin_fmaf(gz1 * K0, gp1, ge1);
                            fv2 = __builtin_fmaf(gz2 * K0, gp2, ge2); fv3 = __builtin_fmaf(gz3 * K0, gp3, ge3);
                        }
                    }
                } else {
                    uint2 hh, ll;
                    TX::split2(fv0, fv1, hh.x, ll.x);
                    TX::split2(fv2, fv3, hh.y, ll.y);
                    constexpr int c0b = fl * 16;                       // + i4: 4 consecutive logical columns inside one 8-group
                    const int c0 = c0b + i4, shi = (c0 >> 3) * 2;
                    char* w = stg + jl * 128 + (c0 & 7) * 2;
                    *reinterpret_cast<uint2*>(w + ((shi ^ key2(jl)) * 16)) = hh;
                    *reinterpret_cast<uint2*>(w + (((shi + 1) ^ key2(jl)) * 16)) = ll;
                }
            } else if constexpr (SL == 22) {
                rv0 = *reinterpret_cast<const uint4*>(stg + rrow * 128 + ((rch ^ key2(rrow)) * 16));
                rv1 = *reinterpret_cast<const uint4*>(stg + (8 + rrow) * 128 + ((rch ^ key2(8 + rrow)) * 16));
            } else {
                const int mrow = dm0 + wave * 64 + fj * 16;
                char* ob0 = reinterpret_cast<char*>(p.out) + ((size_t)mrow * p.ldo + dn0 + g * 32) * 4;
                char* ob1 = reinterpret_cast<char*>(p.out) + ((size_t)(mrow + 8) * p.ldo + dn0 + g * 32) * 4;
                const uint32_t vo = (uint32_t)(((size_t)rrow * p.ldo + (rch >> 1) * 8) * 4 + (rch & 1) * 16);
                gst16<true>(ob0, vo, (u32x4_t){rv0.x, rv0.y, rv0.z, rv0.w});
                gst16<true>(ob1, vo, (u32x4_t){rv1.x, rv1.y, rv1.z, rv1.w});
            }
        } else {
            if constexpr (SL < 2) {
                constexpr int fl = SL, fi = 2 * g + fl;
                const f32x4_t a = dacc[fi][fj];
                const float4 q4 = *reinterpret_cast<const float4*>(side + g * 32 + fl * 16 + i4);
                const int sl16 = fl * 4 + (lane >> 4);
                *reinterpret_cast<float4*>(stg + jl * 128 + ((sl16 ^ key2(jl)) * 16)) = make_float4(a[0] + q4.x, a[1] + q4.y, a[2] + q4.z, a[3] + q4.w);
            } else {
                constexpr int ps = SL >= 12 ? 1 : 0, r = SL - 2 - 10 * ps;     // row pass and piece (pieces 0..9 at slots 2 + 10 ps + r)
                const bool odd = rch & 1;
                if constexpr (SL <= 21) {
                    if constexpr (r == 0) {
                        const int row = ps * 8 + rrow;
                        xv = *reinterpret_cast<const float4*>(stg + row * 128 + ((rch ^ key2(row)) * 16));
                        if constexpr (ps == 0) {
                            // the residual rows of this step, requested one drain step ago: younger = the rest of that step (12 DMA pieces, its stores) and
                            // this step's own requests
                            wait_vm<(TAIL && D >= 1 ? 0 : 12) + (D >= 1 ? NSTORE : 0) + (D + 1 < DS ? NHEAD : 0)>();
                            pin(rr[D & 1][0]); pin(rr[D & 1][1]);
                        }
                    } else if constexpr (r == 1) {
                        // even lane holds hi0..7 (keeps hi0..3, hands over hi4..7), odd lane lo0..7 (keeps lo4..7, hands over lo0..3)
                        const u32x4_t raw = rr[D & 1][ps];
                        const uint32_t t0 = swap_pair(odd ? raw[0] : raw[2]), t1 = swap_pair(odd ? raw[1] : raw[3]);
                        xhx = odd ? t0 : raw[0]; xhy = odd ? t1 : raw[1]; xlx = odd ? raw[2] : t0; xly = odd ? raw[3] : t1;
                    } else if constexpr (r == 2) {
                        xj0 = TX::join_lo(xhx, xlx); xj1 = TX::join_hi(xhx, xlx);
                    } else if constexpr (r == 3) {
                        xj2 = TX::join_lo(xhy, xly); xj3 = TX::join_hi(xhy, xly);
                    } else if constexpr (r == 4) {
                        xv.x += has_res ? xj0 : 0.f; xv.y += has_res ? xj1 : 0.f; xv.z += has_res ? xj2 : 0.f; xv.w += has_res ? xj3 : 0.f;
                    } else if constexpr (r == 5) {
                        TX::split2(xv.x, xv.y, xh.x, xl.x);
                    } else if constexpr (r == 6) {
                        TX::split2(xv.z, xv.w, xh.y, xl.y);
                    } else if constexpr (r == 7) {
                        const uint32_t u0 = swap_pair(odd ? xh.x : xl.x), u1 = swap_pair(odd ? xh.y : xl.y);
                        xsv = odd ? (u32x4_t){u0, u1, xl.x, xl.y} : (u32x4_t){xh.x, xh.y, u0, u1};
                        if constexpr (ps == 0) xsv0 = xsv;
                    } else if constexpr (r == 8) {
                        if constexpr (LNP) ln_quad_sums(xv, xsm, xsq);
                    } else {
                        if constexpr (LNP) {
                            // the 8 lanes of a row: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror -- the one fixed tree of every tile shape
                            xsm = dpp_add(xsm, std::integral_constant<int, 0xB1>()); xsq = dpp_add(xsq, std::integral_constant<int, 0xB1>());
                            xsm = dpp_add(xsm, std::integral_constant<int, 0x4E>()); xsq = dpp_add(xsq, std::integral_constant<int, 0x4E>());
                            xsm = dpp_add(xsm, std::integral_constant<int, 0x141>()); xsq = dpp_add(xsq, std::integral_constant<int, 0x141>());
                            if constexpr (ps == 0) { xsm0 = xsm; xsq0 = xsq; }
                        }
                    }
                }
                // stores: behind the step's last DMA piece (slot 17)
                if constexpr (SL == 18 || SL == 22) {
                    constexpr int sp = SL == 18 ? 0 : 1;
                    const int mrow = dm0 + wave * 64 + fj * 16 + sp * 8;
                    char* obase = reinterpret_cast<char*>(p.out2) + TX::boff((size_t)mrow * p.ldo2 + dn0 + g * 32);
                    gst16<false>(obase, (uint32_t)(TX::boff((size_t)rrow * p.ldo2 + (rch >> 1) * 8) + (odd ? 16 : 0)), sp == 0 ? xsv0 : xsv);
                }
                if constexpr (LNP && (SL == 19 || SL == 23)) {
                    constexpr int sp = SL == 19 ? 0 : 1;
                    const int mrow = dm0 + wave * 64 + fj * 16 + sp * 8;
                    float* pbase = p.ln_part + ((size_t)mrow * (p.n_store >> 5) + ((dn0 >> 5) + g)) * 2;
                    if (rch == 0) gst8(pbase, (uint32_t)((size_t)rrow * (p.n_store >> 5) * 8), sp == 0 ? make_float2(xsm0, xsq0) : make_float2(xsm, xsq));
                }
            }
        }
    };

    // ---- one K step --------------------------------------------------------------------------------------------------------------------------
    int slot = 0;
    // One group of four MFMAs + its slot of other work (a DMA piece of the step two ahead in slots 6..17, a piece of the drain), closed by a
    // scheduling barrier: inside, hipcc may interleave; across, nothing moves.
    auto slot_work = [&](auto d_tag, auto slot_tag) __attribute__((always_inline)) {
        constexpr int D = decltype(d_tag)::value, SL = decltype(slot_tag)::value;
        if constexpr (SL >= 6 && SL <= 17) dma_piece(std::integral_constant<int, SL - 6>());
        if constexpr (D >= 0) drain_slot(d_tag, slot_tag, F_());
        __builtin_amdgcn_sched_barrier(0);
    };
    // MFMA row FIv (fragment column of the wave tile against its four row fragments). Rows 1..6 term-major (dependent MFMAs four apart: a single
    // wave per SIMD has no partner to fill a dependency stall); rows 0 and 7 fragment-row-major, so that the activation fragments of the NEXT step
    // can be requested in place behind their last use (row 7) and are used in request order (row 0).
    auto krow = [&](auto first, auto d_tag, auto waitn_tag, auto fi_tag, int nslot) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr int WAITN = decltype(waitn_tag)::value, FIv = decltype(fi_tag)::value;
        const uint4 ch = ph, cl = pl;
        if constexpr (FIv + 1 < FI) {
            ph = frag(slot, p_base + (FIv + 1) * 16 * KTB + chi);
            pl = frag(slot, p_base + (FIv + 1) * 16 * KTB + clo);
        }
        if constexpr (FIv == 2) {
            // this wave's pieces of the NEXT step's slot have landed (issued one step ago; WAITN younger operations may stay in flight); the barrier
            // publishes that slot to every wave and frees the slot the pieces below go into
            wait_vm<WAITN>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        const f16x8_t PH = TX::h8(ch), PL = TX::h8(cl);
        if constexpr (FIv == FI - 1) {      // the next step's first weight fragment: requested before the activation fragments below (LDS returns in order)
            ph = frag(nslot, p_base + chi);
            pl = frag(nslot, p_base + clo);
        }
        if constexpr (FIv == 0 || FIv == FI - 1) {
            constexpr int SL0 = FIv == 0 ? 0 : 21;
            auto one = [&](auto fj_tag) __attribute__((always_inline)) {
                constexpr int fj = decltype(fj_tag)::value;
                acc[FIv][fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PL, TX::h8(qh[fj]), FIRST ? z : acc[FIv][fj], 0, 0, 0);
                acc[FIv][fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PH, TX::h8(ql[fj]), acc[FIv][fj], 0, 0, 0);
                acc[FIv][fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PH, TX::h8(qh[fj]), acc[FIv][fj], 0, 0, 0);
                if constexpr (FIv == FI - 1) {
                    qh[fj] = frag(nslot, q_base + fj * 16 * KTB + chi);
                    ql[fj] = frag(nslot, q_base + fj * 16 * KTB + clo);
                }
            };
            one(std::integral_constant<int, 0>());
            slot_work(d_tag, std::integral_constant<int, SL0>());
            one(std::integral_constant<int, 1>());
            slot_work(d_tag, std::integral_constant<int, SL0 + 1>());
            one(std::integral_constant<int, 2>());
            one(std::integral_constant<int, 3>());
            slot_work(d_tag, std::integral_constant<int, SL0 + 2>());
        } else {
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) acc[FIv][fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PL, TX::h8(qh[fj]), FIRST ? z : acc[FIv][fj], 0, 0, 0);
            slot_work(d_tag, std::integral_constant<int, FIv * 3>());
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) acc[FIv][fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PH, TX::h8(ql[fj]), acc[FIv][fj], 0, 0, 0);
            slot_work(d_tag, std::integral_constant<int, FIv * 3 + 1>());
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) acc[FIv][fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PH, TX::h8(qh[fj]), acc[FIv][fj], 0, 0, 0);
            slot_work(d_tag, std::integral_constant<int, FIv * 3 + 2>());
        }
    };
    // FIRST: accumulators start from zero (the C operand of the first term).  D >= 0: the step carries drain step D.  PREVST: stores the PREVIOUS step issued
    // behind its last DMA piece.  LAST: last K step of a tile -- the vectors (and, typed residual stream, the first residual rows) its drain will read are
    // requested at the top of the step (the side buffer's previous tenant was drained 16 steps into this tile).
    auto kstep = [&](auto first, auto d_tag, auto prevst_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr int D = decltype(d_tag)::value, PREVST = decltype(prevst_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value && EPK != EPK_NONE;
        const int nslot = slot == NST - 1 ? 0 : slot + 1;
        if constexpr (LAST) { side_loads(); rr_request(std::integral_constant<int, 0>()); }
        if constexpr (D == 0) wait_vm<12 + NHEAD>();          // the side vectors (younger: the first residual request, the 12 pieces of the previous step)
        if constexpr (D >= 0) rr_request(std::integral_constant<int, D + 1>());
        constexpr int HEAD = (D >= 0 && D + 1 < DS ? NHEAD : 0) + (LAST ? 6 + NHEAD : 0);
        typedef std::integral_constant<int, PREVST + HEAD> WN;
        krow(first, d_tag, WN(), std::integral_constant<int, 0>(), nslot);
        krow(first, d_tag, WN(), std::integral_constant<int, 1>(), nslot);
        krow(first, d_tag, WN(), std::integral_constant<int, 2>(), nslot);
        krow(first, d_tag, WN(), std::integral_constant<int, 3>(), nslot);
        krow(first, d_tag, WN(), std::integral_constant<int, 4>(), nslot);
        krow(first, d_tag, WN(), std::integral_constant<int, 5>(), nslot);
        krow(first, d_tag, WN(), std::integral_constant<int, 6>(), nslot);
        krow(first, d_tag, WN(), std::integral_constant<int, 7>(), nslot);
        dma_advance();
        slot = nslot;
    };

    // ---- prologue: steps 0 and 1 of the first tile in flight, head fragments of step 0 ---------------------------------------------------------
    ld_set_tile(ld_v);
    dma_piece(std::integral_constant<int, 0>()); dma_piece(std::integral_constant<int, 1>()); dma_piece(std::integral_constant<int, 2>());
    dma_piece(std::integral_constant<int, 3>()); dma_piece(std::integral_constant<int, 4>()); dma_piece(std::integral_constant<int, 5>());
    dma_piece(std::integral_constant<int, 6>()); dma_piece(std::integral_constant<int, 7>()); dma_piece(std::integral_constant<int, 8>());
    dma_piece(std::integral_constant<int, 9>()); dma_piece(std::integral_constant<int, 10>()); dma_piece(std::integral_constant<int, 11>());
    dma_advance();
    dma_piece(std::integral_constant<int, 0>()); dma_piece(std::integral_constant<int, 1>()); dma_piece(std::integral_constant<int, 2>());
    dma_piece(std::integral_constant<int, 3>()); dma_piece(std::integral_constant<int, 4>()); dma_piece(std::integral_constant<int, 5>());
    dma_piece(std::integral_constant<int, 6>()); dma_piece(std::integral_constant<int, 7>()); dma_piece(std::integral_constant<int, 8>());
    dma_piece(std::integral_constant<int, 9>()); dma_piece(std::integral_constant<int, 10>()); dma_piece(std::integral_constant<int, 11>());
    dma_advance();
    wait_vm<12>();                       // step 0 landed (step 1's pieces may fly)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int fj = 0; fj < FJ; ++fj) {
        qh[fj] = frag(0, q_base + fj * 16 * KTB + chi);
        ql[fj] = frag(0, q_base + fj * 16 * KTB + clo);
    }
    ph = frag(0, p_base + chi);
    pl = frag(0, p_base + clo);

    typedef std::integral_constant<int, -1> NOMS;
    typedef std::integral_constant<int, 0> X0;
    typedef std::integral_constant<int, NSTORE> XT;
    bool have_d = false;
    for (int v = blockIdx.x; v < ntiles; v += G) {
        int m0, n0;
        tile_origin(v, ntiles, tiles_m, tiles_n, p.panel, m0, n0);
        int kt;
        if (EPK == EPK_NONE || !have_d) {
            kstep(T_(), NOMS(), X0(), F_());
            kt = 1;
        } else {
            // 16 steps, each with one micro-slice of the previous tile between its MFMAs
            kstep(T_(), std::integral_constant<int, 0>(), X0(), F_());
            kstep(F_(), std::integral_constant<int, 1>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 2>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 3>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 4>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 5>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 6>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 7>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 8>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 9>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 10>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 11>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 12>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 13>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 14>(), XT(), F_());
            kstep(F_(), std::integral_constant<int, 15>(), XT(), F_());
            kstep(F_(), NOMS(), XT(), F_());
            kt = DS + 1;
        }
#pragma unroll 1
        for (; kt < nk - 1; ++kt) kstep(F_(), NOMS(), X0(), F_());
        dm0 = m0;                        // the last step requests the drain's vectors of THIS tile
        dn0 = n0;
        kstep(F_(), NOMS(), X0(), T_());
        // the tile moves to the drain set
#pragma unroll
        for (int fi = 0; fi < FI; ++fi)
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) dacc[fi][fj] = acc[fi][fj];
        have_d = true;
    }
    // ---- the last tile of this block: drained with nothing to hide under ------------------------------------------------------------------------
    if constexpr (EPK == EPK_NONE) {     // probe: keep the math alive
        float t = 0.f;
#pragma unroll
        for (int fi = 0; fi < FI; ++fi)
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) t += dacc[fi][fj][0] + dacc[fi][fj][1] + dacc[fi][fj][2] + dacc[fi][fj][3];
        if (t == 123.456f) reinterpret_cast<float*>(p.out ? p.out : p.out2)[0] = t;
        wait_vm<0>();
        return;
    }
    if (have_d) {
        wait_vm<NHEAD>();                // the side vectors (the last K step requested them; younger: the first residual request)
        auto tail_step = [&](auto d_tag) __attribute__((always_inline)) {
            constexpr int D = decltype(d_tag)::value;
            rr_request(std::integral_constant<int, D + 1>());
            drain_slot(d_tag, std::integral_constant<int, 0>(), T_()); drain_slot(d_tag, std::integral_constant<int, 1>(), T_()); drain_slot(d_tag, std::integral_constant<int, 2>(), T_());
            drain_slot(d_tag, std::integral_constant<int, 3>(), T_()); drain_slot(d_tag, std::integral_constant<int, 4>(), T_()); drain_slot(d_tag, std::integral_constant<int, 5>(), T_());
            drain_slot(d_tag, std::integral_constant<int, 6>(), T_()); drain_slot(d_tag, std::integral_constant<int, 7>(), T_()); drain_slot(d_tag, std::integral_constant<int, 8>(), T_());
            drain_slot(d_tag, std::integral_constant<int, 9>(), T_()); drain_slot(d_tag, std::integral_constant<int, 10>(), T_()); drain_slot(d_tag, std::integral_constant<int, 11>(), T_());
            drain_slot(d_tag, std::integral_constant<int, 12>(), T_()); drain_slot(d_tag, std::integral_constant<int, 13>(), T_()); drain_slot(d_tag, std::integral_constant<int, 14>(), T_());
            drain_slot(d_tag, std::integral_constant<int, 15>(), T_()); drain_slot(d_tag, std::integral_constant<int, 16>(), T_()); drain_slot(d_tag, std::integral_constant<int, 17>(), T_());
            drain_slot(d_tag, std::integral_constant<int, 18>(), T_()); drain_slot(d_tag, std::integral_constant<int, 19>(), T_()); drain_slot(d_tag, std::integral_constant<int, 20>(), T_());
            drain_slot(d_tag, std::integral_constant<int, 21>(), T_()); drain_slot(d_tag, std::integral_constant<int, 22>(), T_()); drain_slot(d_tag, std::integral_constant<int, 23>(), T_());
        };
        tail_step(std::integral_constant<int, 0>()); tail_step(std::integral_constant<int, 1>()); tail_step(std::integral_constant<int, 2>()); tail_step(std::integral_constant<int, 3>());
        tail_step(std::integral_constant<int, 4>()); tail_step(std::integral_constant<int, 5>()); tail_step(std::integral_constant<int, 6>()); tail_step(std::integral_constant<int, 7>());
        tail_step(std::integral_constant<int, 8>()); tail_step(std::integral_constant<int, 9>()); tail_step(std::integral_constant<int, 10>()); tail_step(std::integral_constant<int, 11>());
        tail_step(std::integral_constant<int, 12>()); tail_step(std::integral_constant<int, 13>()); tail_step(std::integral_constant<int, 14>()); tail_step(std::integral_constant<int, 15>());
    }
    wait_vm<0>();                        // nothing of this block is in flight when its LDS is handed on (the load cursor ran two steps past the last tile)
}

}  // namespace p4

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------
// Which launches run here: split-fp16 nn.Linear operands, whole 256 x 128 tiles, at least DS + 2 K steps, one of the epilogues above.
bool gemm_p4_eligible(const GemmParams& p, int dt) {
    if (dt != D3R_F16X3 || p.amode != AMODE_LINEAR) return false;
    if (p.M % p4::BM != 0 || p.n_store % p4::BN != 0 || p.K % 32 != 0 || (p.K >> 5) < p4::DS + 2) return false;
    if (p.n_store > p.n_pad || p.ln_part_in || p.trace || (p.flags & (GF_RELU | GF_NOWIDE))) return false;
    if ((size_t)32 * p.lda * 4 >= (1ull << 31) || (size_t)32 * p.K * 4 >= (1ull << 31)) return false;
    if (p.epi == EPI_F32) {
        if (!(p.flags & GF_X3RES) || !p.out2 || (p.ldo2 & 7) || (p.res1 && (p.ldr & 7)) || p.res2 || (p.ln_part && p.n_store % 32 != 0)) return false;
        if ((size_t)16 * p.ldo2 * 4 >= (1ull << 31)) return false;
        return true;
    }
    if (p.epi == EPI_GELU || p.epi == EPI_T) {
        if (p.res1 || p.res2 || p.out2 || (p.ldo & 7) || !p.out) return false;
        if (p.ln_rstd && (!p.ln_nmr || !p.ln_colsum)) return false;
        if ((size_t)16 * p.ldo * 4 >= (1ull << 31)) return false;
        return true;
    }
    return false;
}

template <int EPK> static hipError_t launch_p4(const GemmParams& p, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    const unsigned long long dev_bit = 1ull << (dev_id & 63);
    if (!(attr_done.load(std::memory_order_relaxed) & dev_bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(p4::gemm_p4_kernel<EPK>), hipFuncAttributeMaxDynamicSharedMemorySize, p4::LDS);
        attr_done.fetch_or(dev_bit, std::memory_order_relaxed);
    }
    static std::atomic<int> cus_cache[64];
    int cus = cus_cache[dev_id & 63].load(std::memory_order_relaxed);
    if (cus <= 0) {
        int q = 0;
        cus = (hipDeviceGetAttribute(&q, hipDeviceAttributeMultiprocessorCount, dev_id) == hipSuccess && q > 0) ? q : 256;
        cus_cache[dev_id & 63].store(cus, std::memory_order_relaxed);
    }
    const int tiles_m = p.M / p4::BM, tiles_n = p.n_store / p4::BN, ntiles = tiles_m * tiles_n;
    int grid = cus < ntiles ? cus : ntiles;
    if (const char* e = getenv("D3R_P4_GRID")) { const int g = atoi(e); if (g >= 8 && g < grid) grid = g; }      // probe: fewer resident blocks (more tiles per block)
    grid &= ~7;                          // XCD-contiguous tile ranges need the grid stride to keep a block on its XCD (v & 7 == blockIdx & 7)
    if (grid < 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL((p4::gemm_p4_kernel<EPK>), dim3(grid), dim3(p4::NT), p4::LDS, s, p, tiles_m, tiles_n, ntiles);
    return hipGetLastError();
}

hipError_t launch_gemm_p4(const GemmParams& p, hipStream_t s) {
    if (p.flags & GF_NOSTORE) return launch_p4<p4::EPK_NONE>(p, s);       // probe: the K loops alone
    if (p.epi == EPI_F32) return p.ln_part ? launch_p4<p4::EPK_X3RES_LN>(p, s) : launch_p4<p4::EPK_X3RES>(p, s);
    if (p.epi == EPI_GELU) return launch_p4<p4::EPK_GELU>(p, s);
    return launch_p4<p4::EPK_TYPED>(p, s);
}

}  // namespace d3r
