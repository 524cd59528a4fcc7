// dust3r_amd -- persistent split-fp16 GEMM whose epilogue runs UNDER the next tile's K loop (gfx950; round 6).
//
// What it is for (reference call sites): the nn.Linear layers of the 24 encoder / 2 x 12 decoder blocks at batch sizes whose GEMMs fill the
// chip (dust3r/model.py:136-137,180-186 -> croco Block / DecoderBlock: proj, fc1, fc2). In gemm.hip one block computes one tile and then
// stores it: for K <= 1024 a tile spends 15-25 % of its life in the epilogue with the CU's matrix pipes idle (tools/tile_probe.py: the same
// launches without their epilogue run 505-514 TFLOP/s against 371-438 with it).
//
// Shape. One block per CU, FOUR waves, one per SIMD, each with the whole 512-entry register file: 256 accumulator registers = TWO sets of
// the 128 (n) x 64 (m) wave tile (8 x 4 fragments of v_mfma_f32_16x16x32_f16, the wave tile of gemm.hip's 256 x 256 shape, so the LDS read
// traffic per MFMA is the same). Block tile M 256 x N 128. A block walks its tiles (the XCD-contiguous panel order of gemm.hip, strided by
// the grid); set C accumulates tile t while set D -- tile t - 1 -- is drained ONE FRAGMENT PER K STEP (two when the tile has fewer than 32
// steps), a few instructions behind each group of four MFMAs: accumulator -> bias / folded LayerNorm / GELU / residual -> split -> the lanes
// that hold the two halves of an 8-element group swap them (v_permlane16_swap) -> one 16-byte global store per lane. No LDS staging.
// Operands arrive by global_load_lds DMA into a THREE-slot ring that never drains between tiles (the loads run two K steps ahead of the
// math, across tile boundaries); one s_barrier per K step, in the MIDDLE of the step's MFMA stream (the slot it publishes is the NEXT
// step's, the slot it frees is filled behind it), the 12 DMA pieces of a step and the head fragments of the next are spread between the
// MFMA rows. Every VMEM operation of the kernel is issued from inline asm and counted by hand (hipcc's own counter cannot see the DMA: any
// load it knows about would wait for everything in flight): vmcnt(n) below always names how many YOUNGER operations may stay outstanding.
//
// Arithmetic: per output element the same MFMA sequence as every gemm.hip tile (per 32 k: lo.hi, hi.lo, hi.hi; K ascending) and the same
// epilogue expressions and summation trees, so results are bit-identical to them (tests/test_kernels_gpu.py::test_persistent_gemm_*).
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "kernels.hpp"

namespace d3r {
namespace p4 {

constexpr int BM = 256, BN = 128, NW = 4, NT = 256, FI = 8, FJ = 4, KTB = 128, NST = 3;
constexpr int A_BYTES = BM * KTB, W_BYTES = BN * KTB, STAGE = A_BYTES + W_BYTES;   // 32 KiB + 16 KiB per ring slot
constexpr int SIDE0 = NST * STAGE, SIDE_W = 1536;            // per wave: bias[128] | colsum[128] | rstd[64] | nmr[64]  (fp32)
constexpr int DUMMY0 = SIDE0 + NW * SIDE_W;                  // 256 bytes per wave: where the DMA of an operand the launch does not have lands
constexpr int RBUF0 = DUMMY0 + NW * 256;                     // residual rows: 1 KiB (64 lanes x 16 bytes) per wave and drain lane, filled by DMA
constexpr int LDS = RBUF0 + NW * 2 * 1024;                   // 162 816 of the CU's 163 840 bytes
constexpr int NFRAG = FI * FJ;                               // 32 fragments per wave tile
static_assert(LDS <= 160 * 1024, "one block owns the CU's LDS");

// epilogue kinds; _LN: the launch also writes the row partial sums of a folded LayerNorm; NONE: probe builds (GF_NOSTORE), the K loops alone
enum { EPK_TYPED = 0, EPK_GELU = 1, EPK_X3RES = 2, EPK_X3RES_LN = 3, EPK_NONE = 4 };

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int V> using IC = std::integral_constant<int, V>;
template <bool V> using BC = std::integral_constant<bool, V>;

// 4-byte LDS-DMA: lane l's dword lands at lds_dst + 4 l
D3R_DEV void glds4(const void* gsrc, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
// LDS-DMA piece: source = wave-uniform 64-bit base + this lane's 32-bit offset; destination = slot base (SGPR) + a compile-time offset, formed in M0
template <int IMM> D3R_DEV void glds16_imm(const void* sbase, uint32_t voff, uint32_t lds_slot) {
    asm volatile("s_add_i32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_slot), "n"(IMM) : "memory", "m0", "scc");     // s_add writes SCC
}
// 16-byte store, wave-uniform 64-bit base + per-lane 32-bit byte offset; NTP: non-temporal policy
template <bool NTP> D3R_DEV void gst16(void* sbase, uint32_t voff, const u32x4_t v) {
    if constexpr (NTP) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
D3R_DEV void gst8(void* sbase, uint32_t voff, const float2 v) {
    asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(sbase) : "memory");
}
// 16-byte load into registers, invisible to hipcc's vmcnt bookkeeping: the caller waits (counted) before the first use and pins the value behind the wait.
// (A request whose value is never used must not exist: its destination is dead to the compiler while the data is still in flight.)
D3R_DEV u32x4_t gld16(const void* sbase, uint32_t voff) {
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
template <int N> D3R_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
D3R_DEV void pin(u32x4_t& v) { asm volatile("" : "+v"(v)); }
// rows {1, 3} (16-lane groups) of a  <->  rows {0, 2} of b   (wait states around the swap inside the statement: hipcc does not pad asm)
D3R_DEV void swap16(uint32_t& a, uint32_t& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
D3R_DEV void swap32(uint32_t& a, uint32_t& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
// x[l] + x[l ^ 16], x[l] + x[l ^ 32]: both lanes of a pair get the same sum
D3R_DEV float add_xor16(float x) { uint32_t a = __float_as_uint(x), b = a; swap16(a, b); return __uint_as_float(a) + __uint_as_float(b); }
D3R_DEV float add_xor32(float x) { uint32_t a = __float_as_uint(x), b = a; swap32(a, b); return __uint_as_float(a) + __uint_as_float(b); }

D3R_DEV void tile_origin(int v, int ntiles, int tiles_m, int tiles_n, int panel_w, int& m0, int& n0) {
    const int lid = xcd_remap(v, ntiles);
    const int per_panel = panel_w * tiles_m;
    const int panel = lid / per_panel, rem = lid - panel * per_panel;
    const int width = min(panel_w, tiles_n - panel * panel_w);
    const int tm = rem / width, tn = panel * panel_w + (rem - tm * width);
    m0 = tm * BM;
    n0 = tn * BN;
}

// compile-time loop: fn(IC<A>), ..., fn(IC<B>)
template <int A, int B, class Fn> D3R_DEV void for_pairs(Fn&& fn) {
    fn(IC<A>());
    if constexpr (A < B) for_pairs<A + 1, B>(fn);
}

// stores / residual requests of a K step that drains DR (0 nothing, 1 one fragment: A half, 2 one fragment: B half, 3 two fragments: A, B)
constexpr int nst_of(int dr, bool lnp) { return dr == 0 ? 0 : dr == 3 ? 2 + (lnp ? 1 : 0) : 1 + (lnp && dr == 2 ? 1 : 0); }

// NF: fragments of the drain set per K step (1: K >= 1024, every one of a tile's first 32 steps carries one; 2: 16 steps carry two)
// L32 (NF == 1 only): a tile has exactly 32 K steps, so its last step also carries the previous tile's last fragment
template <int EPK, int NF, bool L32, int DBG = 0>      // DBG: probe instances (results invalid): 1 no stores, 2 no lane swaps, 4 stores early in the step, 5 plain instead of non-temporal stores, 6 the row-2 wait does not wait
__global__ __launch_bounds__(NT, 1) void gemm_p4_kernel(GemmParams p, int tiles_m, int tiles_n, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TX = Traits<D3R_F16X3>;
    typedef BC<true> T_;
    typedef BC<false> F_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K >> 5;
    const int G = gridDim.x;
    constexpr bool X3R = EPK == EPK_X3RES || EPK == EPK_X3RES_LN;     // typed residual stream epilogue (GF_X3RES)
    constexpr bool LNP = EPK == EPK_X3RES_LN;
    constexpr bool DRAINS = EPK != EPK_NONE;
    constexpr int NREQ = X3R ? NF : 0;                                 // residual-row requests of a draining K step

    // ---- DMA: per-lane byte offsets of the 8 activation and 4 weight passes of a K step (32 rows apart); every piece takes the SAME wave-uniform base ----
    // Weight rows (read by all four waves): pass q covers rows 32 q + 8 wave + lane / 8. Activation rows are read by ONE wave each (rows 64 wave .. + 63 of
    // the tile): that wave stages them itself, pass q = rows 64 wave + 8 q + lane / 8 -- nobody else has to wait for them (see the K step).
    uint32_t a_off[8], w_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = q * 32 + wave * 8 + (lane >> 3);
        const int ls = (lane & 7) ^ ((row >> 1) & 7);             // LDS slot = chunk ^ key(row), key = (row >> 1) & 7: the fragment reads' swizzle
        w_off[q] = (uint32_t)(((size_t)row * p.K) * 4 + ((ls & 3) * 2 + (ls >> 2)) * 16);      // LDS image [hi0..hi3 | lo0..lo3] of the memory row's [hi x8][lo x8] groups
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int row = wave * 64 + q * 8 + (lane >> 3);
        const int ls = (lane & 7) ^ ((row >> 1) & 7);
        a_off[q] = (uint32_t)(((size_t)row * p.lda) * 4 + ((ls & 3) * 2 + (ls >> 2)) * 16);
    }
    const uint32_t lds0 = lds_addr(smem);
    const uint32_t a_dst = (uint32_t)wave * 8192, w_dst = A_BYTES + (uint32_t)wave * 1024;        // of piece 0, inside a ring slot
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
    // piece 0..11 of the cursor's K step: the 4 weight passes FIRST (vmcnt completes in order: the barrier's wait for them leaves the 8 activation passes in flight)
    auto dma_piece = [&](auto idx_tag) __attribute__((always_inline)) {
        constexpr int IDX = decltype(idx_tag)::value;
        if constexpr (IDX < 4) glds16_imm<IDX * 4096>(ld_w, w_off[IDX], ld_sb + w_dst);
        else glds16_imm<(IDX - 4) * 1024>(ld_a, a_off[IDX - 4], ld_sb + a_dst);
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
    const int i4 = (lane >> 4) * 4, jl = lane & 15;               // accumulator fragment: this lane holds columns i4..i4+3 of row jl
    // a lane's 16 bytes of a stored row: 8-group (lane >> 5) of the fragment's 16 columns, hi half (lanes with (lane >> 4) even) or lo half (odd)
    const uint32_t half_off = (uint32_t)((lane >> 5) * 32 + ((lane >> 4) & 1) * 16);
    float* const side = reinterpret_cast<float*>(smem + SIDE0 + wave * SIDE_W);
    const bool has_ln = p.ln_rstd != nullptr, has_bias = p.bias != nullptr;
    const float* const bsrc = p.bias ? p.bias : reinterpret_cast<const float*>(p.wgt);
    const float* const lnr = has_ln ? p.ln_rstd : bsrc;
    const float* const lnn = has_ln ? p.ln_nmr : bsrc;
    const float* const lns = has_ln ? p.ln_colsum : bsrc;
    const bool has_res = p.res1 != nullptr;
    const char* const rsrc = reinterpret_cast<const char*>(has_res ? p.res1 : (const void*)p.out2);
    const int rld = has_res ? p.ldr : p.ldo2;
    const int old_ = X3R ? p.ldo2 : p.ldo;                        // row stride (elements) of the stored rows
    char* const obase0 = reinterpret_cast<char*>(X3R ? p.out2 : p.out);
    const uint32_t o_voff = (uint32_t)((size_t)jl * old_ * 4) + half_off, r_voff = (uint32_t)((size_t)jl * rld * 4) + half_off;
    const uint32_t part_voff = (uint32_t)((size_t)jl * (p.n_store >> 5) * 8);
    int dm0 = 0, dn0 = 0;                                          // origin of the tile held by the drain set
    int cm0 = 0, cn0 = 0;                                          // origin of the tile being accumulated
    // Side buffer of this wave: bias[128] | colsum[128] | rstd[64] | nmr[64]. An operand the launch does not have keeps the NEUTRAL value written
    // here once (bias 0; no folded LayerNorm: colsum 0, rstd 1, nmr 0 -- fma(acc, 1, fma(0, 0, b)) = acc + b bit for bit, as in gemm.hip) and its
    // DMA is pointed at a dummy area instead: the same six instructions every tile, no branch, no select in the drain.
    const uint32_t sd = lds_addr(side), dummy = lds_addr(smem) + DUMMY0 + (uint32_t)wave * 256;
    side[lane] = 0.f; side[64 + lane] = 0.f; side[128 + lane] = 0.f; side[192 + lane] = 0.f; side[256 + lane] = 1.f; side[320 + lane] = 0.f;
    auto side_loads = [&](int tm0, int tn0) __attribute__((always_inline)) {
        glds4(bsrc + tn0 + lane, has_bias ? sd : dummy);
        glds4(bsrc + tn0 + 64 + lane, has_bias ? sd + 256 : dummy);
        glds4(lns + (has_ln ? tn0 : 0) + lane, has_ln ? sd + 512 : dummy);
        glds4(lns + (has_ln ? tn0 + 64 : 0) + lane, has_ln ? sd + 768 : dummy);
        glds4(lnr + (has_ln ? tm0 + wave * 64 : 0) + lane, has_ln ? sd + 1024 : dummy);
        glds4(lnn + (has_ln ? tm0 + wave * 64 : 0) + lane, has_ln ? sd + 1280 : dummy);
    };

    f32x4_t acc[FI][FJ], dacc[FI][FJ];
    uint4 qh[FJ], ql[FJ];            // activation fragments of the current K step (reloaded in place behind their last MFMA)
    uint4 ph, pl;                    // weight fragment of the next MFMA row

    // Fragment f (0..31) of the drain set: fragment column fi = 2 (f >> 3) + (f & 1), fragment row fj = (f >> 1) & 3 -- the two 16-column halves
    // of a 32-column group of a row are consecutive (their LayerNorm partial sums are added: A half, then B half, the last level of gemm.hip's tree).
    // The accumulator array is indexed statically only (a dynamic index moves it to scratch; a run-time switch over the 32 fragments compiles to a tree of ten
    // scalar branches per pick, measured -10 %): the drain steps are unrolled, the fragment index is a compile-time constant.
    // The work on one fragment is cut into 12 PIECES, a handful of instructions each; piece k of drain lane L (0: the step's first fragment, 1: its
    // second, NF == 2) sits in slot 2 k + L of the K step's 24 slots (one behind each group of four MFMAs), so the matrix pipe never waits for a block
    // of VALU work (hipcc left alone emits one 70-instruction block per fragment). Stores sit in pieces >= 9 = slots >= 18: behind the step's last
    // DMA piece (slot 17), so that the next step's wait for that DMA never waits for a store.
    //   typed / GELU:            0 operands | 1 folded-LayerNorm fmas | 2..9 GELU (scalar fp32: packed VALU beside MFMAs is an anti-lever, MI355X_MICROARCH.md)
    //                            | 10 split | 11 halves swapped, 16-byte store
    //   typed residual stream:   0 acc + bias, residual row arrives | 1 its halves swapped, next request | 2, 3 joins | 4 add | 5, 6 splits | 7 halves swapped
    //                            | 8 quad sums | 9 store, 16-column sums | 10 (B half) A + B | 11 (B half) 8-byte store of the pair
    // X3R: the residual row of each drain lane's next fragment arrives by DMA in this wave's LDS (a load into registers issued from asm is written, for
    // the compiler, when it is issued: any copy it makes of the destination before the data lands -- a loop-carried move is enough -- reads garbage)
    const uint32_t rbuf_lds = lds_addr(smem) + RBUF0 + (uint32_t)wave * 2048;
    const char* const rbuf = smem + RBUF0 + wave * 2048 + lane * 16;
    float fv[2][4], gz[2][4], gt[2][4], gp[2][4], ge[2][4];        // fragment values; GELU: |z|, t, polynomial, exponent / half
    float4 sB[2], sX[2]; float sR[2], sN[2];                       // side operands
    uint32_t hx[2], hy[2], lx[2], ly[2];                           // hi / lo halves (residual in, result out)
    float xj[2][4], xsm[2], xsq[2], xsmA = 0.f, xsqA = 0.f;
    u32x4_t xsv[2];
    auto rr_request = [&](auto f_tag, int tm0, int tn0, auto l_tag) __attribute__((always_inline)) {      // the residual row of fragment f of the tile at (tm0, tn0)
        constexpr int L = decltype(l_tag)::value, f = decltype(f_tag)::value;
        if constexpr (X3R && f < NFRAG) {
            constexpr int fi = 2 * (f >> 3) + (f & 1), fj = (f >> 1) & 3;
            glds16_so(rsrc + ((size_t)(tm0 + wave * 64 + fj * 16) * rld + tn0 + fi * 16) * 4, r_voff, rbuf_lds + L * 1024);
        }
    };
    // ISB: the fragment is the B half of its 32-column group (f odd).  RWAIT: VMEM operations issued since this fragment's residual request.
    // REQ: piece 1 requests the residual row of this drain lane's next fragment.
    auto drain_piece = [&](auto f_tag, auto l_tag, auto k_tag, auto isb_tag, auto rwait_tag, auto req_tag) __attribute__((always_inline)) {
        constexpr int L = decltype(l_tag)::value, k = decltype(k_tag)::value, RWAIT = decltype(rwait_tag)::value, f = decltype(f_tag)::value;
        constexpr bool ISB = decltype(isb_tag)::value, REQ = decltype(req_tag)::value;
        constexpr int fi = 2 * (f >> 3) + (f & 1), fj = (f >> 1) & 3;
        if constexpr (!X3R) {
            if constexpr (k == 0) {
                const f32x4_t a = dacc[fi][fj];
                fv[L][0] = a[0]; fv[L][1] = a[1]; fv[L][2] = a[2]; fv[L][3] = a[3];
                sB[L] = *reinterpret_cast<const float4*>(side + fi * 16 + i4);
                sX[L] = *reinterpret_cast<const float4*>(side + 128 + fi * 16 + i4);
                sR[L] = side[256 + fj * 16 + jl];
                sN[L] = side[320 + fj * 16 + jl];
            } else if constexpr (k == 1) {      // fma(acc, R_j, fma(S_i, Nm_j, bias_i)): the expression of gemm.hip's wide split-fp16 epilogue
                fv[L][0] = __builtin_fmaf(fv[L][0], sR[L], __builtin_fmaf(sX[L].x, sN[L], sB[L].x)); fv[L][1] = __builtin_fmaf(fv[L][1], sR[L], __builtin_fmaf(sX[L].y, sN[L], sB[L].y));
                fv[L][2] = __builtin_fmaf(fv[L][2], sR[L], __builtin_fmaf(sX[L].z, sN[L], sB[L].z)); fv[L][3] = __builtin_fmaf(fv[L][3], sR[L], __builtin_fmaf(sX[L].w, sN[L], sB[L].w));
            } else if constexpr (k < 10) {
                if constexpr (EPK == EPK_GELU) {
                    // gelu_pk (common.hpp) element by element, the same operations: x/2 + |x|/2 erf(|z|), z = x / sqrt 2,
                    // erf(|z|) = 1 - poly(t) exp(-z^2), t = 1 / (1 + p |z|)   (Abramowitz-Stegun 7.1.26)
                    constexpr float K0 = 0.70710678118654752440f, KL = -1.44269504088896340736f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (k == 2) { gz[L][e] = fabsf(fv[L][e] * K0); gt[L][e] = __builtin_fmaf(gz[L][e], 0.3275911f, 1.0f); }
                        else if constexpr (k == 3) gt[L][e] = __builtin_amdgcn_rcpf(gt[L][e]);
                        else if constexpr (k == 4) { gp[L][e] = __builtin_fmaf(gt[L][e], 1.061405429f, -1.453152027f); gp[L][e] = __builtin_fmaf(gp[L][e], gt[L][e], 1.421413741f); }
                        else if constexpr (k == 5) { gp[L][e] = __builtin_fmaf(gp[L][e], gt[L][e], -0.284496736f); gp[L][e] = __builtin_fmaf(gp[L][e], gt[L][e], 0.254829592f); }
                        else if constexpr (k == 6) { gp[L][e] = gp[L][e] * gt[L][e]; ge[L][e] = (gz[L][e] * KL) * gz[L][e]; }
                        else if constexpr (k == 7) ge[L][e] = __builtin_amdgcn_exp2f(ge[L][e]);
                        else if constexpr (k == 8) { gp[L][e] = __builtin_fmaf(-gp[L][e], ge[L][e], 1.0f); ge[L][e] = fv[L][e] * 0.5f; }
                        else fv[L][e] = __builtin_fmaf(gz[L][e] * K0, gp[L][e], ge[L][e]);
                    }
                }
            } else if constexpr (k == 10) {
                TX::split2(fv[L][0], fv[L][1], hx[L], lx[L]);
                TX::split2(fv[L][2], fv[L][3], hy[L], ly[L]);
            } else {
                // lanes l and l ^ 16 hold columns 0-3 / 4-7 of one 8-group of a row: the first keeps its hi words and takes the partner's hi words (16 hi bytes),
                // the second takes the first's lo words and keeps its own (16 lo bytes): [hi x8][lo x8] = the row's 32 bytes, one 16-byte store per lane
                if constexpr (DBG != 2) { swap16(hx[L], lx[L]); swap16(hy[L], ly[L]); }
                char* ob = obase0 + ((size_t)(dm0 + wave * 64 + fj * 16) * old_ + dn0 + fi * 16) * 4;
                if constexpr (DBG == 5) gst16<false>(ob, o_voff, (u32x4_t){hx[L], hy[L], lx[L], ly[L]});
                else if constexpr (DBG != 1) gst16<true>(ob, o_voff, (u32x4_t){hx[L], hy[L], lx[L], ly[L]});
                else asm volatile("" :: "v"(hx[L]), "v"(hy[L]), "v"(lx[L]), "v"(ly[L]), "s"(ob));
            }
        } else {
            if constexpr (k == 0) {
                const f32x4_t a = dacc[fi][fj];
                const float4 q4 = *reinterpret_cast<const float4*>(side + fi * 16 + i4);
                fv[L][0] = a[0] + q4.x; fv[L][1] = a[1] + q4.y; fv[L][2] = a[2] + q4.z; fv[L][3] = a[3] + q4.w;
                wait_vm<RWAIT>();               // this fragment's residual row has landed
                const uint4 rv = *reinterpret_cast<const uint4*>(rbuf + L * 1024);
                hx[L] = rv.x; hy[L] = rv.y; lx[L] = rv.z; ly[L] = rv.w;
            } else if constexpr (k == 1) {
                // the row's 16 bytes: the first lane of a pair holds hi0..7 (keeps hi0..3, hands over hi4..7), the second lo0..7 (keeps lo4..7, hands over lo0..3)
                swap16(hx[L], lx[L]);
                swap16(hy[L], ly[L]);
                if constexpr (REQ) rr_request(IC<f + NF>(), dm0, dn0, l_tag);     // (the read of this buffer has returned: its words were just swapped)
            } else if constexpr (k == 2) {
                xj[L][0] = TX::join_lo(hx[L], lx[L]); xj[L][1] = TX::join_hi(hx[L], lx[L]);
            } else if constexpr (k == 3) {
                xj[L][2] = TX::join_lo(hy[L], ly[L]); xj[L][3] = TX::join_hi(hy[L], ly[L]);
            } else if constexpr (k == 4) {
                fv[L][0] += has_res ? xj[L][0] : 0.f; fv[L][1] += has_res ? xj[L][1] : 0.f; fv[L][2] += has_res ? xj[L][2] : 0.f; fv[L][3] += has_res ? xj[L][3] : 0.f;
            } else if constexpr (k == 5) {
                TX::split2(fv[L][0], fv[L][1], hx[L], lx[L]);
            } else if constexpr (k == 6) {
                TX::split2(fv[L][2], fv[L][3], hy[L], ly[L]);
            } else if constexpr (k == 7) {
                swap16(hx[L], lx[L]);
                swap16(hy[L], ly[L]);
                xsv[L] = (u32x4_t){hx[L], hy[L], lx[L], ly[L]};
            } else if constexpr (k == 8) {
                if constexpr (LNP) ln_quad_sums(make_float4(fv[L][0], fv[L][1], fv[L][2], fv[L][3]), xsm[L], xsq[L]);
            } else if constexpr (k == 9) {
                char* ob = obase0 + ((size_t)(dm0 + wave * 64 + fj * 16) * old_ + dn0 + fi * 16) * 4;
                gst16<false>(ob, o_voff, xsv[L]);
                if constexpr (LNP) {    // gemm.hip's fixed tree over the 8 column quads of a row's 32-column group: (q0 + q1) + (q2 + q3) here, A + B below
                    xsm[L] = add_xor16(xsm[L]); xsq[L] = add_xor16(xsq[L]);
                    xsm[L] = add_xor32(xsm[L]); xsq[L] = add_xor32(xsq[L]);
                }
            } else if constexpr (k == 10) {
                if constexpr (LNP) {
                    if constexpr (!ISB) { xsmA = xsm[L]; xsqA = xsq[L]; }
                    else { xsm[L] = xsmA + xsm[L]; xsq[L] = xsqA + xsq[L]; }
                }
            } else {
                if constexpr (LNP && ISB) {
                    float* pb = p.ln_part + ((size_t)(dm0 + wave * 64 + fj * 16) * (p.n_store >> 5) + ((dn0 >> 5) + (fi >> 1))) * 2;
                    if (lane < 16) gst8(pb, part_voff, make_float2(xsm[L], xsq[L]));
                }
            }
        }
    };

    // ---- one K step --------------------------------------------------------------------------------------------------------------------------
    int slot = 0;
    // One group of four MFMAs + its slot of other work (a DMA piece of the step two ahead in slots 6..17, a piece of the drain; in a tile's last step the
    // requests for the next drain behind slot 4), closed by a scheduling barrier: inside, hipcc may interleave; across, nothing moves.
    // DR: what the step drains (nst_of).  PREVNST: stores of the previous K step.  REQ: the drain lanes request their next fragments' residual rows.
    auto slot_work = [&](auto f_tag, auto dr_tag, auto prevnst_tag, auto req_tag, auto last_tag, auto slot_tag) __attribute__((always_inline)) {
        constexpr int DR = decltype(dr_tag)::value, SL = decltype(slot_tag)::value, PREVNST = decltype(prevnst_tag)::value, F = decltype(f_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        if constexpr (SL >= 6 && SL <= 17) dma_piece(IC<SL - 6>());
        if constexpr (DR != 0) {
            constexpr int L = SL & 1, k = DBG == 4 ? ((SL >> 1) + 8) % 12 : SL >> 1;      // DBG 4: the fragment's pieces rotated so that its store sits in slot 6
            // operations issued since this lane's residual request (behind piece 1 of the previous step, or slot 4 of the previous tile's last step):
            // the other lane's request (lane 0 only), that step's 12 DMA pieces and its stores
            typedef IC<(NF == 2 && L == 0 ? 1 : 0) + 12 + PREVNST> RW;
            if constexpr (DR == 3) drain_piece(IC<F + L>(), IC<L>(), IC<k>(), BC<L == 1>(), RW(), req_tag);
            else if constexpr (L == 0) drain_piece(IC<F>(), IC<0>(), IC<k>(), BC<DR == 2>(), RW(), req_tag);
        }
        if constexpr (LAST && SL == 4) {
            // the vectors the drain of the tile that ends here will read -- the side buffer's operands of the previous drain's last fragment were consumed
            // in slots <= 3 -- and the residual rows of its first fragment(s)
            side_loads(cm0, cn0);
            rr_request(IC<0>(), cm0, cn0, IC<0>());
            if constexpr (NF == 2) rr_request(IC<1>(), cm0, cn0, IC<1>());
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // MFMA row FIv (fragment column of the wave tile against its four row fragments). Rows 1..6 term-major (dependent MFMAs four apart: a single
    // wave per SIMD has no partner to fill a dependency stall); rows 0 and 7 fragment-row-major, so that the activation fragments of the NEXT step
    // can be requested in place behind their last use (row 7) and are used in request order (row 0).
    auto krow = [&](auto f, auto first, auto dr_tag, auto prevnst_tag, auto req_tag, auto last_tag, auto waitn_tag, auto waita_tag, auto fi_tag, int nslot) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr int WAITN = decltype(waitn_tag)::value, WAITA = decltype(waita_tag)::value, FIv = decltype(fi_tag)::value;
        const uint4 ch = ph, cl = pl;
        if constexpr (FIv + 1 < FI) {
            ph = frag(slot, p_base + (FIv + 1) * 16 * KTB + chi);
            pl = frag(slot, p_base + (FIv + 1) * 16 * KTB + clo);
        }
        if constexpr (FIv == 2) {
            // this wave's WEIGHT pieces of the next step's slot have landed (issued one step ago, ahead of the activation pieces: WAITN younger operations may
            // stay in flight); the barrier publishes them to every wave and frees the weight slot the pieces below go into
            wait_vm<DBG == 6 ? WAITN + 14 : WAITN>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        const f16x8_t PH = TX::h8(ch), PL = TX::h8(cl);
        if constexpr (FIv == FI - 1) {
            // the next step's ACTIVATION rows -- staged by this wave for itself, nobody else reads them -- have landed: ~1.5 K steps after their issue, where the
            // barrier above gives the shared weight rows ~0.5 (they come from L2; the activation rows from HBM, slower still beside the drain's stores)
            wait_vm<WAITA>();
            // the next step's first weight fragment: requested before the activation fragments below (LDS returns in order)
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
            one(IC<0>());
            slot_work(f, dr_tag, prevnst_tag, req_tag, last_tag, IC<SL0>());
            one(IC<1>());
            slot_work(f, dr_tag, prevnst_tag, req_tag, last_tag, IC<SL0 + 1>());
            one(IC<2>());
            one(IC<3>());
            slot_work(f, dr_tag, prevnst_tag, req_tag, last_tag, IC<SL0 + 2>());
        } else {
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) acc[FIv][fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PL, TX::h8(qh[fj]), FIRST ? z : acc[FIv][fj], 0, 0, 0);
            slot_work(f, dr_tag, prevnst_tag, req_tag, last_tag, IC<FIv * 3>());
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) acc[FIv][fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PH, TX::h8(ql[fj]), acc[FIv][fj], 0, 0, 0);
            slot_work(f, dr_tag, prevnst_tag, req_tag, last_tag, IC<FIv * 3 + 1>());
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) acc[FIv][fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(PH, TX::h8(qh[fj]), acc[FIv][fj], 0, 0, 0);
            slot_work(f, dr_tag, prevnst_tag, req_tag, last_tag, IC<FIv * 3 + 2>());
        }
    };
    // f: first fragment of the drain set this step works on.  FIRST: accumulators start from zero (the C operand of the first term).  PDR: what the PREVIOUS
    // step drained (its stores are younger than its DMA).  REQ: the drain lanes request their next fragments (false on the drain's last step).  LAST: last K
    // step of a tile.
    auto kstep = [&](auto f, auto first, auto dr_tag, auto pdr_tag, auto req_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr int DR = decltype(dr_tag)::value, PDR = decltype(pdr_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value && DRAINS, REQ = decltype(req_tag)::value && X3R && DR != 0;
        constexpr bool TSTART = decltype(first)::value && DR != 0;
        constexpr int PREVNST = nst_of(PDR, LNP);
        const int nslot = slot == NST - 1 ? 0 : slot + 1;
        // first drain step of a tile: the side vectors (requested in slot 4 of the previous tile's last step; younger: the first residual requests, that step's
        // 12 DMA pieces and stores)
        if constexpr (TSTART) wait_vm<NREQ + 12 + PREVNST>();
        // row-2 wait (weight pieces of the previous step): younger = that step's 8 activation pieces and stores + what this step issues in slots 0..5 (residual
        // requests behind piece 1; last step: 6 + NREQ).  Row-7 wait (its activation pieces): younger = the same without the 8, + this step's 12 pieces and the stores
        // it has issued by then (typed residual stream: piece 9 = slots 18, 19; the typed / GELU store sits in piece 11, behind the wait)
        constexpr int HEADN = (REQ ? NREQ : 0) + (LAST ? 6 + NREQ : 0);
        typedef IC<8 + PREVNST + HEADN> WN;
        typedef IC<PREVNST + HEADN + 12 + (X3R ? (DR == 3 ? 2 : DR != 0 ? 1 : 0) : 0)> WA;
        typedef IC<PREVNST> PN;
        typedef BC<REQ> RQ;
        typedef BC<LAST> LT;
        krow(f, first, dr_tag, PN(), RQ(), LT(), WN(), WA(), IC<0>(), nslot);
        krow(f, first, dr_tag, PN(), RQ(), LT(), WN(), WA(), IC<1>(), nslot);
        krow(f, first, dr_tag, PN(), RQ(), LT(), WN(), WA(), IC<2>(), nslot);
        krow(f, first, dr_tag, PN(), RQ(), LT(), WN(), WA(), IC<3>(), nslot);
        krow(f, first, dr_tag, PN(), RQ(), LT(), WN(), WA(), IC<4>(), nslot);
        krow(f, first, dr_tag, PN(), RQ(), LT(), WN(), WA(), IC<5>(), nslot);
        krow(f, first, dr_tag, PN(), RQ(), LT(), WN(), WA(), IC<6>(), nslot);
        krow(f, first, dr_tag, PN(), RQ(), LT(), WN(), WA(), IC<7>(), nslot);
        dma_advance();
        slot = nslot;
    };

    // ---- prologue: steps 0 and 1 of the first tile in flight, head fragments of step 0 ---------------------------------------------------------
    ld_set_tile(ld_v);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        dma_piece(IC<0>()); dma_piece(IC<1>()); dma_piece(IC<2>()); dma_piece(IC<3>()); dma_piece(IC<4>()); dma_piece(IC<5>());
        dma_piece(IC<6>()); dma_piece(IC<7>()); dma_piece(IC<8>()); dma_piece(IC<9>()); dma_piece(IC<10>()); dma_piece(IC<11>());
        dma_advance();
    }
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

    // ---- tiles ---------------------------------------------------------------------------------------------------------------------------------
    // NF == 1 (nk >= 32): steps 0..31 of a tile carry fragments 0..31 of the previous one (A, B, A, B, ...); when nk == 32 the last of them is also the
    // tile's last step. NF == 2 (17 <= nk): steps 0..15 carry two fragments each. The loops are rolled: the fragment index is a run-time scalar.
    constexpr bool l32 = NF == 1 && L32;
    bool have_d = false;
    for (int v = blockIdx.x; v < ntiles; v += G) {
        tile_origin(v, ntiles, tiles_m, tiles_n, p.panel, cm0, cn0);
        int kt = 0;
        bool last_done = false;
        if (!DRAINS || !have_d) {
            kstep(IC<NFRAG>(), T_(), IC<0>(), IC<0>(), F_(), F_());
            kt = 1;
        } else if constexpr (NF == 1) {
            // (the previous tile's last step drained its fragment 31 -- and stored -- only when a tile has exactly 32 steps)
            kstep(IC<0>(), T_(), IC<1>(), IC<l32 ? 2 : 0>(), T_(), F_());
            kstep(IC<1>(), F_(), IC<2>(), IC<1>(), T_(), F_());
            for_pairs<1, 15>([&](auto i_tag) __attribute__((always_inline)) {
                constexpr int f = 2 * decltype(i_tag)::value;
                kstep(IC<f>(), F_(), IC<1>(), IC<2>(), T_(), F_());
                if constexpr (f + 1 < NFRAG - 1) kstep(IC<f + 1>(), F_(), IC<2>(), IC<1>(), T_(), F_());
            });
            if constexpr (l32) {
                kstep(IC<31>(), F_(), IC<2>(), IC<1>(), F_(), T_());       // fragment 31 of the previous tile + the requests for this one
                last_done = true;
            } else {
                kstep(IC<31>(), F_(), IC<2>(), IC<1>(), F_(), F_());
                kstep(IC<NFRAG>(), F_(), IC<0>(), IC<2>(), F_(), F_());
                kt = 33;
            }
        } else {
            kstep(IC<0>(), T_(), IC<3>(), IC<0>(), T_(), F_());
            for_pairs<1, 14>([&](auto i_tag) __attribute__((always_inline)) { kstep(IC<2 * decltype(i_tag)::value>(), F_(), IC<3>(), IC<3>(), T_(), F_()); });
            kstep(IC<NFRAG - 2>(), F_(), IC<3>(), IC<3>(), F_(), F_());
            kstep(IC<NFRAG>(), F_(), IC<0>(), IC<3>(), F_(), F_());
            kt = 17;
        }
        if (!last_done) {
#pragma unroll 1
            for (; kt < nk - 1; ++kt) kstep(IC<NFRAG>(), F_(), IC<0>(), IC<0>(), F_(), F_());
            kstep(IC<NFRAG>(), F_(), IC<0>(), IC<0>(), F_(), T_());
        }
        // the tile moves to the drain set
        dm0 = cm0;
        dn0 = cn0;
#pragma unroll
        for (int fi = 0; fi < FI; ++fi)
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) dacc[fi][fj] = acc[fi][fj];
        have_d = true;
    }
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
    // ---- the last tile of this block: drained with nothing to hide under ------------------------------------------------------------------------
    if (have_d) {
        wait_vm<0>();                    // side vectors and first residual rows (requested by the last K step), and the two steps of loads past the last tile
        // through the drain lanes of the K loop (lane L takes fragments f = L mod NF), one fragment after the other; RWAIT = operations since the fragment's
        // request: NF == 1: the previous fragment's stores; NF == 2: the stores of both lanes' previous fragments and the other lane's request
        auto tail_frag = [&](auto f, auto l_tag, auto isb_tag, auto rwait_tag, auto req_tag) __attribute__((always_inline)) {
            drain_piece(f, l_tag, IC<0>(), isb_tag, rwait_tag, req_tag); drain_piece(f, l_tag, IC<1>(), isb_tag, rwait_tag, req_tag);
            drain_piece(f, l_tag, IC<2>(), isb_tag, rwait_tag, req_tag); drain_piece(f, l_tag, IC<3>(), isb_tag, rwait_tag, req_tag);
            drain_piece(f, l_tag, IC<4>(), isb_tag, rwait_tag, req_tag); drain_piece(f, l_tag, IC<5>(), isb_tag, rwait_tag, req_tag);
            drain_piece(f, l_tag, IC<6>(), isb_tag, rwait_tag, req_tag); drain_piece(f, l_tag, IC<7>(), isb_tag, rwait_tag, req_tag);
            drain_piece(f, l_tag, IC<8>(), isb_tag, rwait_tag, req_tag); drain_piece(f, l_tag, IC<9>(), isb_tag, rwait_tag, req_tag);
            drain_piece(f, l_tag, IC<10>(), isb_tag, rwait_tag, req_tag); drain_piece(f, l_tag, IC<11>(), isb_tag, rwait_tag, req_tag);
        };
        constexpr int NSA = nst_of(1, LNP), NSB = nst_of(2, LNP);
        typedef IC<NF == 1 ? NSB : NSA + NSB + 1> RWA;      // A half: behind the previous B half's stores (NF == 2: + lane 0's own previous stores + lane 1's request)
        typedef IC<NF == 1 ? NSA : NSA + NSB + 1> RWB;
        for_pairs<0, 14>([&](auto i_tag) __attribute__((always_inline)) {
            constexpr int f = 2 * decltype(i_tag)::value;
            tail_frag(IC<f>(), IC<0>(), F_(), RWA(), T_());
            tail_frag(IC<f + 1>(), IC<NF - 1>(), T_(), RWB(), T_());
        });
        tail_frag(IC<NFRAG - 2>(), IC<0>(), F_(), RWA(), BC<NF == 1>());       // NF == 1: fragment 30 requests 31; NF == 2: nothing left to request
        tail_frag(IC<NFRAG - 1>(), IC<NF - 1>(), T_(), IC<NF == 1 ? NSA : NSA + NSB>(), F_());      // (NF == 2: fragment 30 requested nothing)
    }
    wait_vm<0>();                        // nothing of this block is in flight when its LDS is handed on (the load cursor ran two steps past the last tile)
}

}  // namespace p4

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------
// Which launches can run here: split-fp16 nn.Linear operands, whole 256 x 128 tiles, at least 18 K steps, one of the epilogues above.
bool gemm_p4_eligible(const GemmParams& p, int dt) {
    if (dt != D3R_F16X3 || p.amode != AMODE_LINEAR) return false;
    if (p.M % p4::BM != 0 || p.n_store % p4::BN != 0 || p.K % 32 != 0 || (p.K >> 5) < 18) return false;
    if (p.n_store > p.n_pad || p.ln_part_in || p.trace || (p.flags & (GF_RELU | GF_NOWIDE))) return false;
    if ((size_t)256 * p.lda * 4 >= (1ull << 31) || (size_t)128 * p.K * 4 >= (1ull << 31)) return false;       // 32-bit row offsets inside a tile
    if (p.epi == EPI_F32) {
        if (!(p.flags & GF_X3RES) || !p.out2 || (p.ldo2 & 7) || (p.res1 && (p.ldr & 7)) || p.res2 || (p.ln_part && p.n_store % 32 != 0)) return false;
        if ((size_t)16 * p.ldo2 * 4 >= (1ull << 31) || (p.res1 && (size_t)16 * p.ldr * 4 >= (1ull << 31))) return false;
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

template <int EPK, int NF, bool L32 = false, int DBG = 0> static hipError_t launch_p4(const GemmParams& p, hipStream_t s) {
    static std::atomic<unsigned long long> attr_done{0};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    const unsigned long long dev_bit = 1ull << (dev_id & 63);
    if (!(attr_done.load(std::memory_order_relaxed) & dev_bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(p4::gemm_p4_kernel<EPK, NF, L32, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, p4::LDS);
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
    if (const char* e = probe_env("D3R_P4_GRID")) { const int g = atoi(e); if (g >= 8 && g < grid) grid = g; }      // probe: fewer resident blocks (more tiles per block)
    grid &= ~7;                          // XCD-contiguous tile ranges need the grid stride to keep a block on its XCD (v & 7 == blockIdx & 7)
    if (grid < 8) return hipErrorInvalidValue;
    GemmParams q = p;
    if (const char* e = probe_env("D3R_P4_PANEL")) { const int v = atoi(e); if (v >= 1 && v <= 64) q.panel = v; }      // probe: width in tiles of the column panels of the tile walk
    hipLaunchKernelGGL((p4::gemm_p4_kernel<EPK, NF, L32, DBG>), dim3(grid), dim3(p4::NT), p4::LDS, s, q, tiles_m, tiles_n, ntiles);
    return hipGetLastError();
}

hipError_t launch_gemm_p4(const GemmParams& p, hipStream_t s) {
    // Two fragments per K step (16 draining steps) everywhere. The one-fragment form (NF = 1: 32 unrolled step bodies) measured SLOWER on MI355X
    // (typed store 408 vs 436 TFLOP/s, GELU 338 vs ~400 at K = 1024): ~100 KB of straight-line code per tile against a 64 KB instruction cache.
    // It stays in the template (and in the probe below) but is not instantiated by default.
    if (p.flags & GF_NOSTORE) return launch_p4<p4::EPK_NONE, 2>(p, s);       // probe: the K loops alone
    if (p.epi == EPI_F32) return p.ln_part ? launch_p4<p4::EPK_X3RES_LN, 2>(p, s) : launch_p4<p4::EPK_X3RES, 2>(p, s);
    if (p.epi == EPI_GELU) return launch_p4<p4::EPK_GELU, 2>(p, s);
#ifdef D3R_PROBES
    if (const char* e = probe_env("D3R_P4_DBG")) {       // probe instances (typed store, 32 K steps): results INVALID
        if ((p.K >> 5) == 32 && e[0] == '0') return launch_p4<p4::EPK_TYPED, 1, true, 0>(p, s);
        if ((p.K >> 5) == 32 && e[0] == '1') return launch_p4<p4::EPK_TYPED, 1, true, 1>(p, s);
        if ((p.K >> 5) == 32 && e[0] == '6') return launch_p4<p4::EPK_TYPED, 1, true, 6>(p, s);
    }
#endif
    return launch_p4<p4::EPK_TYPED, 2>(p, s);
}

}  // namespace d3r
