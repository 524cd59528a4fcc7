// dust3r_amd -- MFMA GEMM / implicit-GEMM convolution with fused epilogues (gfx950).
//
// One kernel template serves every dense contraction of the DUSt3R forward (reference call sites):
//   Linear layers of croco Block / DecoderBlock (qkv, proj, fc1, fc2, projq/k/v) and
//   decoder_embed (dust3r/model.py:136-137,176-186), PatchEmbed's k16s16 conv as a GEMM
//   over pre-gathered patches (dust3r/patch_embed.py:19-29), the DPT head's 1x1 / 3x3 /
//   stride-2 convolutions and ConvTranspose k=s (dust3r/heads/dpt_head.py:34-65) as implicit
//   GEMMs over NHWC activations, and LinearPts3d.proj (dust3r/heads/linear_head.py:30-41).
//
// Shape: out[m][n] = sum_k act[m][k] * wgt[n][k]   ("NT": both operands K-contiguous).
// Tile configurations (template Cfg; 128 bytes of K per step = 64 bf16; 16x16x32 MFMA fragments):
//   256 x 256, 8 waves (2 x 4), each wave 128 (n) x 64 (m) = 8 x 4 fragments   -- the large-GEMM shape
//   256 x 128, 8 waves, each wave 64 x 64                                       -- mid-size problems
//   512 x 128, 8 waves (1 x 8), each wave 128 (n) x 64 (m)                      -- N <= 128 convolutions (DPT head)
//   128 x 128, 4 waves (2 x 2), each wave 64 x 64, 2 blocks per CU              -- small problems / tails
// Operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered; the LDS image is
// lane-linear, the bank swizzle (chunk ^= (row>>1)&7, conflict-free for ds_read_b128 on 128-byte rows) is applied
// on the per-lane SOURCE address and again on the fragment read.
// Block -> tile map: XCD-aware (each XCD owns a contiguous range of tile ids) and panel-rasterised (8 n-tiles wide)
// so the tiles an XCD works on concurrently share A rows and W rows inside its 4 MiB L2.
// Precision modes share the tiles, the LDS image and the epilogues (common.hpp, Traits<DT>): one 16-bit or exact-f32 MFMA per product;
// split-fp16 (hi + lo rows, three f16 MFMAs per product: the default engine); fp16 + fp8 rows (opt-in engines' transformer-block linears: 128-byte K
// steps alternate between the fp16 half and the e4m3 half of a 256-byte super-group -- two f16 MFMAs, then ONE 16x16x128 fp8 MFMA that
// adds both cross terms, its E8M0 scale undoing the 2^17 of the encodings; DMA row addresses as one 32-bit offset per operand plus
// scalar strides, DMA pieces interleaved with the MFMA rows on the 256-wide tiles).
// MFMA operand roles: D[i][j] with 4 consecutive i per lane. Normally i = n (weights) so each
// lane owns 4 consecutive output columns of one row -> 8/16-byte stores; for V^T tiles of the
// attention projections the roles are swapped (i = m) so 4 consecutive TOKENS land together.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include <atomic>

#include "kernels.hpp"

namespace d3r {

// NWI x NWJ waves; a wave owns FI x FJ 16x16 fragments. The "i" side (4 consecutive per lane) is n (weights) unless
// the block works on a V^T region (roles swapped; square configurations only). KTB: bytes of K per LDS row and K step
// (128 = 64 bf16: two MFMA k-steps per barrier; 64 = 32 bf16: one k-step, half the LDS, so two blocks fit on a CU).
template <int NWI_, int NWJ_, int FI_, int FJ_, int MINW_, int KTB_ = 128, int NSTAGE_ = 2, int PP_ = 0> struct GemmCfg {
    static constexpr int NWI = NWI_, NWJ = NWJ_, FI = FI_, FJ = FJ_, MINW = MINW_, KTB = KTB_, NSTAGE = NSTAGE_, PP = PP_;
    static constexpr int NW = NWI * NWJ, NT = NW * 64;
    static constexpr int BN = NWI * FI * 16, BM = NWJ * FJ * 16;
    static constexpr int CPR = KTB / 16;                     // 16-byte chunks per row
    static constexpr int RPI = 64 / CPR;                     // rows covered by one 1 KiB wave DMA instruction
    static constexpr int PASS_ROWS = NW * RPI;               // rows staged by one global_load_lds per wave
    static constexpr int APASS = BM / PASS_ROWS, WPASS = BN / PASS_ROWS;
    static constexpr int STAGE_BYTES = (BM + BN) * KTB;
    // PP == 3: asymmetric ring -- THREE slots for the activation rows (streamed from HBM: two K steps of lookahead) and TWO for the
    // weight rows (L2 / MALL resident: one step), 3 x 32 + 2 x 32 KiB = all 160 KiB of a CU for the 256 x 256 tile
    // PP == 6: TWO slots for the activation rows and ONE for the weight rows (read into registers at the top of every K step): 80 KiB
    // for the 256 x 128 tile, so that two blocks share a CU
    static constexpr int LDS = PP_ == 3 ? (3 * NWJ_ * FJ_ * 16 + 2 * NWI_ * FI_ * 16) * KTB_ : PP_ == 6 ? (2 * NWJ_ * FJ_ * 16 + NWI_ * FI_ * 16) * KTB_ : NSTAGE * STAGE_BYTES;
    static constexpr int LPS = APASS + WPASS;                // DMA instructions per lane per K step
    // bank swizzle of the lane-linear LDS image: slot = chunk ^ key(row). Checked against the ds_read_b128 service groups
    // of gfx950 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32): a fragment read (lane -> row lane & 15, chunk group lane >> 4)
    // touches 16 distinct 16-byte slots of the 256-byte bank row in every group, for both row widths.
    __host__ __device__ static constexpr int key(int row) { return KTB == 128 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3); }
};
typedef GemmCfg<2, 4, 8, 4, 2> Cfg256;      // 256 x 256, 512 threads, 128 KiB LDS, 1 block / CU
typedef GemmCfg<2, 4, 4, 4, 2> Cfg256x128;  // M 256 x N 128, 512 threads, 96 KiB LDS, 1 block / CU
typedef GemmCfg<2, 2, 4, 4, 2> Cfg128;      // 128 x 128, 256 threads, 64 KiB LDS, 2 blocks / CU
typedef GemmCfg<1, 8, 8, 4, 2> Cfg512x128;  // M 512 x N 128, 512 threads, 160 KiB LDS (all of it): N <= 128 convolutions with the
                                            // 128 x 64 per-wave tile of Cfg256 (12 ds_read_b128 per 32 MFMAs instead of 8 per 16)
typedef GemmCfg<2, 4, 8, 4, 2, 64, 4> Cfg256s4;   // 256 x 256, 8 waves, 64-byte K rows, FOUR LDS stages (128 KiB): three K steps of DMA in flight
typedef GemmCfg<2, 4, 8, 4, 2, 64, 4, 1> Cfg256pp;   // 256 x 256, 64-byte K rows, 4-stage ring, PING-PONG schedule: waves 0-3 and 4-7 (one of each per
                                                     // SIMD) run half a phase apart, so one group's 16-MFMA burst covers the other's ds_reads + DMA issue
// split-fp16 (x3) software-pipelined K loop (PP = 2): same tiles, barrier moved into the tail of the MFMA stream (see the K loop)
typedef GemmCfg<2, 4, 8, 4, 2, 128, 2, 2> Cfg256sw;
typedef GemmCfg<2, 4, 4, 4, 2, 128, 2, 2> Cfg256x128sw;
typedef GemmCfg<2, 2, 4, 4, 2, 128, 2, 2> Cfg128sw;
typedef GemmCfg<1, 8, 8, 4, 2, 128, 2, 2> Cfg512x128sw;
// fp16 + fp8 rows, DMA pieces interleaved with the MFMA rows (PP = 4; see the K loop)
typedef GemmCfg<2, 4, 8, 4, 2, 128, 2, 4> Cfg256il;
typedef GemmCfg<2, 4, 4, 4, 2, 128, 2, 4> Cfg256x128il;
typedef GemmCfg<2, 2, 4, 4, 2, 128, 2, 4> Cfg128il;
typedef GemmCfg<1, 8, 8, 4, 2, 128, 2, 4> Cfg512x128il;
typedef GemmCfg<1, 4, 8, 4, 2, 64, 3, 5> Cfg256x128f8;   // fp16 + fp8 rows: M 256 x N 128 by four waves of 128 (n) x 64 (m), 64-byte K steps, 3 x 24 KiB: TWO blocks per CU
typedef GemmCfg<2, 4, 8, 4, 2, 128, 2, 3> Cfg256a3;     // 256 x 256, asymmetric ring (A x 3, W x 2), 160 KiB LDS
// split-fp16, TWO blocks per CU (round 3): M 256 x N 128 by four waves of 128 (n) x 64 (m) -- the per-wave tile of Cfg256, so the LDS read
// traffic per MFMA is unchanged --, 128-byte K steps, activation rows double buffered, weight rows single buffered and held in REGISTERS
// for the step (80 KiB). One block's epilogue and barrier bubbles run under the other block's MFMAs (see the K loop).
typedef GemmCfg<1, 4, 8, 4, 2, 128, 2, 6> Cfg256x128r;
// (256 x 256 by FOUR waves of 128 x 128 -- 256 accumulator registers per lane, one wave per SIMD, a third less LDS read traffic per
// MFMA -- compiles to 256 VGPR + 256 AGPR with the accumulator array in scratch: 90-105 TF/s algorithmic against 390-480, round 2.
// With hipcc as the register allocator the 128 x 64 wave tile at two waves per SIMD is the largest that stays in registers.)
// split-fp16, SMALL problems (round 3): 64 x 64 by four waves of 32 x 32 -- four times the waves of the 128 x 128 tile for the same
// problem, so that the one-pair forward (M = 1536 token rows: 96 tiles of 128 x 128 on 256 CUs) fills the chip. Twice the LDS read
// traffic per MFMA (8 ds_read_b128 per 12 MFMAs); 32 KiB of LDS, up to four blocks per CU. Same K order per output element as every
// other shape: bit-identical results.
typedef GemmCfg<2, 2, 2, 2, 4> Cfg64;
// the same tile on a three- / four-slot ring (48 / 64 KiB): two / three K steps of DMA in flight -- a small problem has too few waves per CU
// to hide the load latency of a one-step lookahead (measured: 1 us per K step of 0.1 us of MFMA work, profiles/r03_f). Default: three slots.
typedef GemmCfg<2, 2, 2, 2, 4, 128, 3> Cfg64s3;
typedef GemmCfg<2, 2, 2, 2, 2, 128, 4> Cfg64s4;
// M 96 x N 64 by four waves of 32 (n) x 48 (m) on the three-slot ring (60 KiB; round 6): the small tile is bound by what it pulls out of L2 per flop and by the blocks a
// CU has to run one after the other -- 1536 x 1024 is 384 tiles of 64 x 64 (a CU in two runs two: 256 operand rows per K step) or exactly 256 of
// 96 x 64 (one per CU: 160 rows). Tile configuration 11; same K order per output element: bit-identical.
typedef GemmCfg<2, 2, 2, 3, 4, 128, 3> Cfg96x64;
// 128 x 128 by EIGHT waves of 64 (n) x 32 (m): twice the waves per tile (16 per CU with two resident blocks) for the mid-size problems of the
// small-batch forwards, where a K step is bound by latency rather than by the matrix pipe. Bit-identical to the four-wave shape.
typedef GemmCfg<2, 4, 4, 2, 4> Cfg128w8;
// M 384 x N 192 by eight waves stacked along m, each 192 (n) x 48 (m) = 12 x 3 fragments (144 accumulators; 147 KiB of LDS): the decoder's 24576-row GEMMs with
// N = 768 / 1536 / 2304 / 3072 are exactly 1 / 2 / 3 / 4 rounds of 256 CUs on it (128 x 128: 1152 tiles = 2.25 rounds of 512 slots). Round 4; tile configuration 9,
// chosen by gemm_pick_config (D3R_GEMM_T384=0: never). Same K order per output element as every other shape: bit-identical results.
typedef GemmCfg<1, 8, 12, 3, 2> Cfg384x192;
typedef GemmCfg<1, 4, 8, 4, 2, 64, 3> Cfg256x128w4;  // M 256 x N 128, 4 waves of 128 (n) x 64 (m), 64-byte K rows: 48 KiB LDS, TWO
                                                  // blocks per CU, three stages (72 KiB)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Region lookups written as selects (a dynamically indexed member array of the by-value kernel argument can make hipcc
// copy the whole struct to scratch).
D3R_DEV int head_kind_of(const GemmParams& p, int region) { return region == 0 ? p.head_kind[0] : (region == 1 ? p.head_kind[1] : p.head_kind[2]); }
D3R_DEV void* head_dst_of(const GemmParams& p, int region) { return region == 0 ? p.head_dst[0] : (region == 1 ? p.head_dst[1] : p.head_dst[2]); }

// RoPE table rows of token row jj (clamped to M - 1): (cos0,sin0,cos1,sin1), (cos2,sin2,cos3,sin3) of the y position, then of x
D3R_DEV void load_rope_rows(const float* table, int ntok, int tok_w, int M, int jj, int i4, float4 (&d)[4]) {
    const int mm = jj < M ? jj : M - 1;
    const int b = mm / ntok, t = mm - b * ntok;
    const int ty = t / tok_w, tx = t - ty * tok_w;
    const float4* cy = reinterpret_cast<const float4*>(table + ((size_t)ty * 16 + i4) * 2);
    const float4* cx = reinterpret_cast<const float4*>(table + ((size_t)tx * 16 + i4) * 2);
    d[0] = cy[0]; d[1] = cy[1]; d[2] = cx[0]; d[3] = cx[1];
}

// K step kt of an implicit-GEMM operand -> filter tap and first input channel. Tap-major (k = tap * Cin + c) re-reads a tile's
// whole input neighbourhood once per tap: with 32 tiles in flight per XCD that is 8-10 MiB between two uses of a line, past the
// 4 MiB L2 (measured: 6.6x the algorithmic fetch on the 3x3 128->128 head convolution). Slice-major walks the taps of ONE K step's
// channel slice before moving on: the lines of a slice (128 bytes per pixel) are touched by all k x k taps back to back.
D3R_DEV void conv_k_step(const GemmParams& p, int kel, int S, int& tap, int& c0) {   // kel: first K element of the step; S: elements per 128-byte slice
    if (p.kslice_major) {
        const int per = p.ksize * p.ksize * S;
        const int sl = kel / per, rem = kel - sl * per;
        tap = rem / S;
        c0 = sl * S + (rem - tap * S);
    } else {
        tap = kel / p.Cin;
        c0 = kel - tap * p.Cin;
    }
}

// the same for ONE 32-column half of a head (xhalf 0: the y position's rows, 1: x): what a 32-wide epilogue group needs
D3R_DEV void load_rope_half(const float* table, int ntok, int tok_w, int M, int jj, int i4, int xhalf, float4 (&d)[2]) {
    const int mm = jj < M ? jj : M - 1;
    const int b = mm / ntok, t = mm - b * ntok;
    const int ty = t / tok_w, tx = t - ty * tok_w;
    const float4* c = reinterpret_cast<const float4*>(table + ((size_t)(xhalf ? tx : ty) * 16 + i4) * 2);
    d[0] = c[0]; d[1] = c[1];
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
D3R_DEV void store16(void* dst, uint4 v, bool nt) {
    const u32x4_t w = {v.x, v.y, v.z, v.w};
    // inline asm: behind a uniform branch hipcc merges a __builtin_nontemporal_store with the plain store and drops the policy
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
    else *reinterpret_cast<u32x4_t*>(dst) = w;
}

template <int DT, class CF>
__global__ __launch_bounds__(CF::NT, CF::MINW) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TR = Traits<DT>;
    constexpr int EB = TR::EB;
    constexpr int KTB = CF::KTB;
    constexpr int KT = KTB / EB;  // elements of K per tile
    constexpr int BM = CF::BM, BN = CF::BN, FI = CF::FI, FJ = CF::FJ, STAGE_BYTES = CF::STAGE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.trace && tid == 0) {
        unsigned long long* tr = p.trace + (size_t)blockIdx.x * 8;
        tr[0] = (unsigned long long)wall_clock64();
        tr[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        tr[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        tr[7] = blockIdx.x;
    }
    // ---- start stagger of the first round of resident blocks -----------------------------------------------------------
    // All tiles of a launch take the same time, so the resident blocks run in lockstep: every CU is in its K loop (HBM idle), then
    // every CU is in its epilogue (MFMA idle, 256 x 128-512 KiB hitting HBM at once: measured 22-28 us per round at 3-4.7 TB/s
    // where one CU alone needs a few us). Delaying the FIRST-round blocks by up to one such burst spreads the epilogues of all later
    // rounds over time (a block's successor on the same CU inherits its phase); the price is half a burst once per launch.
    if (p.stagger_ticks > 0 && (int)blockIdx.x < p.first_round) {
        const unsigned slot = p.stagger_mode ? ((blockIdx.x & 7u) * 4u) : ((blockIdx.x >> 3) & 31u);   // mode 1: by XCD; mode 0: across each XCD's blocks
        const long long delay = ((long long)p.stagger_ticks * slot) >> 5;
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < delay) __builtin_amdgcn_s_sleep(32);
    }
    // ---- block -> tile: XCD-contiguous ids, then 8-wide column panels walked row by row ------------------
    const int tiles_n = (p.n_store + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    int lid = xcd_remap(blockIdx.x, gridDim.x);
    int sk_slice = 0;
    if (p.splitk > 1) { sk_slice = lid % p.splitk; lid /= p.splitk; }      // a tile's K slices: consecutive ids = one XCD (the combine reads same-XCD slabs)
    const int sk_tile = lid;
    const int PANEL = p.panel;
    const int per_panel = PANEL * tiles_m;
    const int panel = lid / per_panel, rem_p = lid - panel * per_panel;
    const int width = min(PANEL, tiles_n - panel * PANEL);
    const int tm = rem_p / width, tn = panel * PANEL + (rem_p - tm * width);
    const int m0 = tm * BM, n0 = tn * BN;
    // V^T regions of the attention projections: with the wide epilogue (16-bit types, token count a multiple of 64) the
    // transposition happens in the LDS staging tile and any tile shape works; otherwise the MFMA operand roles are
    // swapped for that block (square tiles only) so that a lane owns 4 consecutive tokens.
    const bool vt_wide = (DT == D3R_BF16 || DT == D3R_F16) && p.epi == EPI_HEADS && (p.ntok & 63) == 0 && !(p.flags & GF_NOWIDE);
    const bool swap = !vt_wide && (BM == BN) && (p.epi == EPI_HEADS) && (head_kind_of(p, n0 / p.head_c) == HEAD_VT);
    // Folded LayerNorm, SMALL problems (GemmParams::ln_part_in, the one- to eight-pair forwards): the rstd / -mean rstd of this tile's BM rows are formed HERE from the
    // producer's partial sums instead of by a launch of ln_finalize_kernel between the two GEMMs (a launch + gap in a dependent chain of ~25 us kernels). The arithmetic IS
    // that kernel's (ln_row_stats, kernels.hpp: its 32-lane butterfly as a binary tree, lower levels in registers, upper levels by lane exchange), so a row's statistics are
    // bit-identical whichever route formed them and however many threads shared the row (the batch-vs-one-pair tests compare the routes). Every column tile of a row panel writes the same values
    // to the same addresses and reads back what it wrote itself; the stores are complete before the epilogue through the K loop's barriers (each waits vmcnt(0)).
    if constexpr (DT == D3R_F16X3) {
        if (p.ln_part_in) {
            constexpr int TPR = (CF::NT / BM) >= 4 ? 4 : ((CF::NT / BM) >= 2 ? 2 : 1);      // threads per row (adjacent lanes)
            const int G = p.K >> 5;
            // uniform trip count, no branch around the lane exchanges: threads past the tile's rows work on a clamped row and only the store is predicated
#pragma unroll 1
            for (int r0 = 0; r0 < BM; r0 += CF::NT / TPR) {
                const int r = r0 + tid / TPR, m = m0 + r;
                float rs, nm;
                ln_row_stats<TPR>(reinterpret_cast<const float2*>(p.ln_part_in) + (size_t)min(m, p.M - 1) * G, G, tid % TPR, p.ln_inv_c, p.ln_eps, rs, nm);
                if (tid % TPR == 0 && r < BM && m < p.M) {
                    const_cast<float*>(p.ln_rstd)[m] = rs;
                    const_cast<float*>(p.ln_nmr)[m] = nm;
                }
            }
        }
    }

    // ---- staging addresses (per lane: one 16-byte chunk of APASS activation rows and WPASS weight rows) ---
    const int lrow = wave * CF::RPI + lane / CF::CPR;                // row inside a PASS_ROWS slab
    const int lslot = (lane % CF::CPR) ^ CF::key(lrow);              // logical chunk of the LDS row image held by this lane's slot (slot = chunk ^ key(row))
    // split-fp16 rows are [hi x8][lo x8] groups in memory; their LDS image is [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] so that the four
    // lane groups of a fragment read touch four CONSECUTIVE chunks, like the 16-bit types (with the memory order in LDS lane group g
    // reads chunk 2g: rows r and r + 4 of a ds_read_b128 service group land on the same banks -- a 2-way conflict on every read)
    const int lchunk = (DT == D3R_F16X3) ? ((lslot & 3) * 2 + (lslot >> 2)) : lslot;   // chunk of the MEMORY row fetched by this lane
    const char* wsrc[CF::WPASS];
    // per activation row of a pass: linear operand -> the row's source address; implicit-GEMM operand -> the packed
    // (image base pixel | top-left input y << 16 | top-left input x) of the output pixel. One 64-bit slot either way.
    unsigned long long arow[CF::APASS];
#pragma unroll
    for (int q = 0; q < CF::WPASS; ++q) {
        const int r = q * CF::PASS_ROWS + lrow;
        wsrc[q] = reinterpret_cast<const char*>(p.wgt) + ((size_t)(n0 + r) * p.K) * EB + lchunk * 16;
    }
#pragma unroll
    for (int q = 0; q < CF::APASS; ++q) {
        const int r = q * CF::PASS_ROWS + lrow;
        int m = m0 + r;
        m = m < p.M ? m : p.M - 1;
        if (p.amode == AMODE_LINEAR) {
            arow[q] = (unsigned long long)(size_t)(reinterpret_cast<const char*>(p.act) + ((size_t)m * p.lda) * EB + lchunk * 16);
        } else {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;   // |.| < 32768
            const unsigned pk = ((unsigned)iy0 << 16) | ((unsigned)ix0 & 0xFFFFu);
            arow[q] = ((unsigned long long)pk << 32) | (unsigned)(b * p.Hin * p.Win);
        }
    }
    const char* zsrc = reinterpret_cast<const char*>(p.zero_page) + lchunk * 16;

    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    const uint32_t lds0 = lds_addr(smem) + wave_u * 1024;   // wave-uniform: one 1 KiB DMA piece per wave and pass
    constexpr bool A3 = CF::PP == 3;
    auto stage_a = [&](int kt, int buf) __attribute__((always_inline)) {   // activation rows of K step kt -> stage buf
        const uint32_t sb = lds0 + (A3 ? buf * (BM * KTB) : buf * STAGE_BYTES);
        const size_t koff = (size_t)kt * KTB;
        if (p.amode == AMODE_LINEAR) {
#pragma unroll
            for (int q = 0; q < CF::APASS; ++q) glds16(reinterpret_cast<const char*>((size_t)arow[q]) + koff, sb + q * (CF::NW * 1024));
        } else {
            int tap, c0;
            conv_k_step(p, kt * KT, 128 / EB, tap, c0);
            const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
#pragma unroll
            for (int q = 0; q < CF::APASS; ++q) {
                const int pk = (int)(arow[q] >> 32), ibase = (int)(unsigned)arow[q];
                const int iy = (pk >> 16) + ky, ix = (int)(short)(pk & 0xFFFF) + kx;
                const bool ok = (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
                const char* src = reinterpret_cast<const char*>(p.act) +
                                  ((size_t)(ibase + iy * p.Win + ix) * p.cstride + c0) * EB + lchunk * 16;
                src = ok ? src : zsrc;
                glds16(src, sb + q * (CF::NW * 1024));
            }
        }
    };
    auto stage_w = [&](int kt, int buf) __attribute__((always_inline)) {   // weight rows of K step kt -> stage buf
        const uint32_t sb = lds0 + (A3 ? 3 * (BM * KTB) + buf * (BN * KTB) : buf * STAGE_BYTES + BM * KTB);
        const size_t koff = (size_t)kt * KTB;
#pragma unroll
        for (int q = 0; q < CF::WPASS; ++q) glds16(wsrc[q] + koff, sb + q * (CF::NW * 1024));
    };
    auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
        stage_a(kt, buf);
        stage_w(kt, buf);
    };

    // ---- fragment read addresses ---------------------------------------------------------------
    const int wi = wave / CF::NWJ, wj = wave - wi * CF::NWJ;
    const int frow = lane & 15, fsw = CF::key(frow), fgrp = lane >> 4;   // fragment rows start at multiples of 16: key(row) == key(frow)
    // P tile supplies i (4 consecutive per lane), Q tile supplies j
    const int p_off = swap ? 0 : BM * KTB;  // activations live at 0, weights at BM*KTB
    const int q_off = swap ? BM * KTB : 0;
    const int p_row0 = wi * (FI * 16) + frow, q_row0 = wj * (FJ * 16) + frow;

    if (p.trace && tid == 0) p.trace[(size_t)blockIdx.x * 8 + 1] = (unsigned long long)wall_clock64();
    __builtin_amdgcn_s_setprio(2);   // K loop: ahead of a co-resident block's epilogue in the SIMD's issue arbitration
    static_assert(CF::NSTAGE >= 2 && CF::NSTAGE <= 4, "vmcnt ladder below covers up to 2 younger steps in flight");
    f32x4_t acc[FI][FJ];
#pragma unroll
    for (int a = 0; a < FI; ++a)
#pragma unroll
        for (int b = 0; b < FJ; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // K loop over an NSTAGE-deep LDS ring. Step kt: [wait until this wave's DMA of step kt has landed, leaving the younger
    // steps' loads in flight: a counted vmcnt -- the asm-issued loads are the only VMEM ops of the loop] -> barrier (step
    // kt is visible to every wave; every wave is done reading the stage of step kt-1) -> issue the DMA of step
    // kt+NSTAGE-1 into that freed stage -> math on step kt. NSTAGE-1 steps of HBM/L2 latency are covered.
    constexpr int NS = CF::NSTAGE, LPS = CF::LPS;
    int nk = p.K / KT;
    if (p.splitk > 1) {               // this block's share of the K steps (launch_gemm: nn.Linear operands, the plain loop, nk % splitk == 0)
        nk /= p.splitk;
        const size_t skip = (size_t)sk_slice * nk * KTB;
#pragma unroll
        for (int q = 0; q < CF::WPASS; ++q) wsrc[q] += skip;
#pragma unroll
        for (int q = 0; q < CF::APASS; ++q) arow[q] += skip;
    }
    if constexpr (CF::PP == 3) {
        // ---- asymmetric ring: activations two K steps ahead, weights one ------------------------------------------------------
        // Measured with the per-block trace (tools/gpu_probe.py gemmtrace): next to other blocks' epilogue traffic a K step of the
        // two-stage loop takes 2.2-2.7 us against 1.8-1.9 us alone -- the activation rows come from HBM and one step of lookahead
        // does not cover their latency under load. Issue order per step: W(kt+1), then A(kt+2); loads complete in order, so the
        // wait at the top of step kt+1 leaves the APASS pieces of A(kt+2) in flight.
        static_assert(KTB == 128 && CF::NSTAGE == 2, "asymmetric ring: 128-byte K rows");
        constexpr int ASLOT = BM * KTB, WSLOT = BN * KTB, WBASE = 3 * ASLOT;
        stage_a(0, 0);
        stage_w(0, 0);
        if (nk > 1) stage_a(1, 1);
        int ab = 0, wb = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CF::APASS) : "memory");
            else d3r_wait_vm0();
            __syncthreads();
            if (kt + 1 < nk) stage_w(kt + 1, wb ^ 1);
            if (kt + 2 < nk) stage_a(kt + 2, ab >= 1 ? ab - 1 : 2);     // (ab + 2) % 3
            const char* abase = smem + ab * ASLOT;
            const char* wbase = smem + WBASE + wb * WSLOT;
            const char* pb = swap ? abase : wbase;
            const char* qb = swap ? wbase : abase;
            if constexpr (DT == D3R_F16X3) {
                const int chi = (fgrp ^ fsw) * 16, clo = ((4 + fgrp) ^ fsw) * 16;
                uint4 qf[FJ], ql[FJ];
#pragma unroll
                for (int f = 0; f < FJ; ++f) {
                    const char* qr = qb + (q_row0 + f * 16) * KTB;
                    qf[f] = *reinterpret_cast<const uint4*>(qr + chi);
                    ql[f] = *reinterpret_cast<const uint4*>(qr + clo);
                }
#pragma unroll
                for (int fi = 0; fi < FI; ++fi) {
                    const char* pr = pb + (p_row0 + fi * 16) * KTB;
                    const uint4 pf = *reinterpret_cast<const uint4*>(pr + chi), pl = *reinterpret_cast<const uint4*>(pr + clo);
#pragma unroll
                    for (int fj = 0; fj < FJ; ++fj) TR::mma16x3(acc[fi][fj], pf, pl, qf[fj], ql[fj]);
                }
            } else if constexpr (DT != D3R_F16F8 && DT != D3R_F16X2F8) {
#pragma unroll
                for (int ks = 0; ks < KTB / 64; ++ks) {
                    const int coff = ((ks * 4 + fgrp) ^ fsw) * 16;
                    uint4 pf[FI], qf[FJ];
#pragma unroll
                    for (int f = 0; f < FJ; ++f) qf[f] = *reinterpret_cast<const uint4*>(qb + (q_row0 + f * 16) * KTB + coff);
#pragma unroll
                    for (int f = 0; f < FI; ++f) pf[f] = *reinterpret_cast<const uint4*>(pb + (p_row0 + f * 16) * KTB + coff);
#pragma unroll
                    for (int fi = 0; fi < FI; ++fi)
#pragma unroll
                        for (int fj = 0; fj < FJ; ++fj) TR::mma16(acc[fi][fj], pf[fi], qf[fj]);
                }
            }
            ab = ab == 2 ? 0 : ab + 1;
            wb ^= 1;
        }
    } else if constexpr (DT == D3R_F16X3 && CF::PP == 4) {
        // ---- split-fp16, nn.Linear operands, DMA pieces interleaved with the MFMA rows (round 4) ---------------------------------------------
        // The plain loop opens every K step with [vmcnt(0) | s_barrier | LPS DMA issues | first fragment reads]: a burst during which neither wave of
        // a SIMD has an MFMA to issue. Here the pieces of step kt + 1 go out one fragment row at a time behind the MFMAs of step kt (fragment reads
        // one row ahead, rows pinned with sched_barrier), as in the fp16 + fp8 and 2.5-unit loops. DMA addressing: a wave-uniform 64-bit tile base
        // (SGPRs) + ONE 32-bit offset per operand (the plain loop keeps one 64-bit address per staged row: 16 VGPRs). Same MFMA order per accumulator
        // as the plain loop: bit-identical results.
        static_assert(KTB == 128 && NS == 2, "split-fp16 rows: 128-byte K steps, two stages");
        const bool edge = __builtin_amdgcn_readfirstlane(m0 + BM > p.M ? 1 : 0) != 0;
        const char* const tile_a = reinterpret_cast<const char*>(p.act) + (size_t)__builtin_amdgcn_readfirstlane(m0) * p.lda * EB;
        const char* const tile_w = reinterpret_cast<const char*>(p.wgt) + (size_t)__builtin_amdgcn_readfirstlane(n0) * p.K * EB;
        const uint32_t a0 = (uint32_t)(((size_t)(min(m0 + lrow, p.M - 1) - m0) * p.lda) * EB + lchunk * 16);
        const uint32_t w0 = (uint32_t)(((size_t)lrow * p.K) * EB + lchunk * 16);
        const size_t stride_a = (size_t)CF::PASS_ROWS * p.lda * EB, stride_w = (size_t)CF::PASS_ROWS * p.K * EB;
        auto piece3 = [&](int idx, int kt, int buf) __attribute__((always_inline)) {
            const uint32_t sb = lds0 + buf * STAGE_BYTES;
            if (idx < CF::APASS) {
                const int q = idx;
                const char* ab = tile_a + (size_t)kt * KTB;
                if (!edge) {
                    glds16_so(ab + q * stride_a, a0, sb + q * (CF::NW * 1024));
                } else {
                    const int m = min(m0 + q * CF::PASS_ROWS + lrow, p.M - 1);
                    glds16_so(ab, (uint32_t)(((size_t)(m - m0) * p.lda) * EB + lchunk * 16), sb + q * (CF::NW * 1024));
                }
            } else {
                const int q = idx - CF::APASS;
                glds16_so(tile_w + (size_t)kt * KTB + q * stride_w, w0, sb + BM * KTB + q * (CF::NW * 1024));
            }
        };
        constexpr int PPR = (LPS + FI - 1) / FI;
#pragma unroll
        for (int i = 0; i < LPS; ++i) piece3(i, 0, 0);
        const int chi = (fgrp ^ fsw) * 16, clo = ((4 + fgrp) ^ fsw) * 16;   // LDS image: [hi0..hi3 | lo0..lo3]
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            d3r_wait_vm0();
            __syncthreads();
            const bool more = kt + 1 < nk;
            const char* sb = smem + buf * STAGE_BYTES;
            uint4 qf[FJ], ql[FJ];
#pragma unroll
            for (int f = 0; f < FJ; ++f) {
                const char* qr = sb + q_off + (q_row0 + f * 16) * KTB;
                qf[f] = *reinterpret_cast<const uint4*>(qr + chi);
                ql[f] = *reinterpret_cast<const uint4*>(qr + clo);
            }
            const char* pr0 = sb + p_off + p_row0 * KTB;
            uint4 cf = *reinterpret_cast<const uint4*>(pr0 + chi), cl = *reinterpret_cast<const uint4*>(pr0 + clo);
#pragma unroll
            for (int fi = 0; fi < FI; ++fi) {
                uint4 nf = cf, nl = cl;
                if (fi + 1 < FI) {
                    const char* pr = sb + p_off + (p_row0 + (fi + 1) * 16) * KTB;
                    nf = *reinterpret_cast<const uint4*>(pr + chi);
                    nl = *reinterpret_cast<const uint4*>(pr + clo);
                }
#pragma unroll
                for (int fj = 0; fj < FJ; ++fj) TR::mma16x3(acc[fi][fj], cf, cl, qf[fj], ql[fj]);
                if (more) {
#pragma unroll
                    for (int i = fi * PPR; i < (fi + 1) * PPR && i < LPS; ++i) piece3(i, kt + 1, buf ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                cf = nf; cl = nl;
            }
            buf ^= 1;
        }
    } else if constexpr (CF::PP == 2) {
        // ---- split-fp16, software pipelined (2 stages, ONE barrier per K step, no bubble at the step boundary) ---------------
        // The plain loop below opens every K step with [vmcnt(0) | s_barrier | 8 DMA issues | first ds_reads] during which none of
        // the CU's 8 waves has an MFMA to issue: ~25 % of the step at 96 MFMAs per wave. Here the barrier of step kt sits in
        // front of the LAST fragment row (TAIL = 1) of step kt (every ds_read of stage kt & 1 has been issued and waited for by
        // then, so the stage is dead), and behind it, interleaved with those TAIL x FJ x 3 MFMAs: the DMA of step kt + 2 into
        // the stage just freed, and the ds_reads of step kt + 1's q fragments and first p fragment (published by the same
        // barrier: every wave waited for ITS DMA pieces of step kt + 1, issued a whole step earlier, before arriving). The next
        // step's MFMAs then start from registers. Hazards: RAW on stage (kt+1)&1 = vmcnt(0) + barrier; WAR on stage kt&1 =
        // lgkmcnt(0) + the same barrier.
        static_assert(DT == D3R_F16X3 && KTB == 128 && NS == 2, "software-pipelined loop: split-fp16 rows, two stages");
        constexpr int TAIL = 1, NSLOT = TAIL * FJ;
        const int chi = (fgrp ^ fsw) * 16, clo = ((4 + fgrp) ^ fsw) * 16;   // LDS image: [hi0..hi3 | lo0..lo3]
        int cky = 0, ckx = 0, cc0 = 0;           // implicit-GEMM operand: filter tap / first channel of the K step being staged
        auto conv_step = [&](int kt) __attribute__((always_inline)) {
            if (p.amode != AMODE_LINEAR) {
                int tap;
                conv_k_step(p, kt * KT, 128 / EB, tap, cc0);
                cky = tap / p.ksize;
                ckx = tap - cky * p.ksize;
            }
        };
        auto dma_piece = [&](int idx, int kt, int buf) __attribute__((always_inline)) {   // idx: compile-time constant after unrolling
            const uint32_t sb = lds0 + buf * STAGE_BYTES;
            const size_t koff = (size_t)kt * KTB;
            if (idx < CF::APASS) {
                const int q = idx;
                if (p.amode == AMODE_LINEAR) {
                    glds16(reinterpret_cast<const char*>((size_t)arow[q]) + koff, sb + q * (CF::NW * 1024));
                } else {
                    const int pk = (int)(arow[q] >> 32), ibase = (int)(unsigned)arow[q];
                    const int iy = (pk >> 16) + cky, ix = (int)(short)(pk & 0xFFFF) + ckx;
                    const bool ok = (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
                    const char* src = reinterpret_cast<const char*>(p.act) + ((size_t)(ibase + iy * p.Win + ix) * p.cstride + cc0) * EB + lchunk * 16;
                    glds16(ok ? src : zsrc, sb + q * (CF::NW * 1024));
                }
            } else {
                const int q = idx - CF::APASS;
                glds16(wsrc[q] + koff, sb + BM * KTB + q * (CF::NW * 1024));
            }
        };
        auto frag = [&](const char* sbase, int off, int row, int coff) __attribute__((always_inline)) {
            return *reinterpret_cast<const uint4*>(sbase + off + row * KTB + coff);
        };
        // DMA pieces of one K step (LPS per wave) are spread over the MFMA stream instead of being issued in one burst (8 waves x
        // 8 pieces of 1 KiB keep the CU's vector-memory issue path busy for ~1000 cycles during which no wave issues MFMAs):
        // pieces [0, PT) of step kt + 2 go out in the tail of step kt (behind the barrier that frees their stage), pieces
        // [PT, LPS) in the first MROWS - GUARD main rows of step kt + 1; the last GUARD rows carry none, so that the youngest
        // piece has ~GUARD x 12 MFMAs of time to land before the barrier that publishes it.
        constexpr int MROWS = FI - TAIL, GUARD = MROWS >= 5 ? 2 : 1, DROWS = MROWS - GUARD;
        constexpr int PT = (LPS + DROWS) / (DROWS + 1);         // tail share: about one row's worth
        stage(0, 0);
        conv_step(1);
        if (nk > 1) {
#pragma unroll
            for (int pc = 0; pc < PT; ++pc) dma_piece(pc, 1, 1);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PT) : "memory");   // step 0 has landed (the PT younger pieces may still fly)
        if (nk <= 1) d3r_wait_vm0();
        __syncthreads();
        // Fragments that cross the step boundary: the first q fragment and the first p fragment of the next step, read behind the
        // barrier under the tail MFMAs. Two register sets with swapped roles in a loop unrolled by two: no moves on the back edge.
        struct Head { uint4 qh, ql, ph, pl; };
        Head ha, hb;
        ha.qh = frag(smem, q_off, q_row0, chi); ha.ql = frag(smem, q_off, q_row0, clo);
        ha.ph = frag(smem, p_off, p_row0, chi); ha.pl = frag(smem, p_off, p_row0, clo);
        auto kstep = [&](int kt, const Head& cur, Head& nxt) __attribute__((always_inline)) {
            const char* sb = smem + (kt & 1) * STAGE_BYTES;
            const char* sn = smem + ((kt + 1) & 1) * STAGE_BYTES;
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
            uint4 qh[FJ], ql[FJ];
            qh[0] = cur.qh; ql[0] = cur.ql;
#pragma unroll
            for (int f = 1; f < FJ; ++f) { qh[f] = frag(sb, q_off, q_row0 + f * 16, chi); ql[f] = frag(sb, q_off, q_row0 + f * 16, clo); }
            uint4 ch = cur.ph, cl = cur.pl;
#pragma unroll
            for (int fi = 0; fi < MROWS; ++fi) {
                const uint4 nh = frag(sb, p_off, p_row0 + (fi + 1) * 16, chi), nl = frag(sb, p_off, p_row0 + (fi + 1) * 16, clo);
                if (more1 && fi < DROWS) {     // the rest of step kt + 1's DMA (conv_step(kt + 1) was evaluated in the previous tail / the prologue)
#pragma unroll
                    for (int pc = PT; pc < LPS; ++pc)
                        if ((pc - PT) * DROWS / (LPS - PT) == fi) dma_piece(pc, kt + 1, (kt + 1) & 1);
                }
#pragma unroll
                for (int fj = 0; fj < FJ; ++fj) TR::mma16x3(acc[fi][fj], ch, cl, qh[fj], ql[fj]);
                ch = nh; cl = nl;
            }
            if (more2) conv_step(kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // the next step's first fragments (stale but in-bounds bytes after the last step: never used)
            nxt.qh = frag(sn, q_off, q_row0, chi); nxt.ql = frag(sn, q_off, q_row0, clo);
            nxt.ph = frag(sn, p_off, p_row0, chi); nxt.pl = frag(sn, p_off, p_row0, clo);
            __builtin_amdgcn_sched_barrier(0);
            // tail row, term-major over its FJ accumulators (dependent MFMAs FJ apart), with the first PT pieces of step kt + 2
            constexpr int NMF = 3 * NSLOT;
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
                for (int fj = 0; fj < FJ; ++fj) {
                    const int mf = term * FJ + fj;
                    if (more2) {
#pragma unroll
                        for (int pc = 0; pc < PT; ++pc)
                            if (pc * NMF / PT == mf) dma_piece(pc, kt + 2, kt & 1);
                    }
                    TR::mma16_term(term, acc[MROWS][fj], ch, cl, qh[fj], ql[fj]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            kstep(kt, ha, hb);
            kstep(kt + 1, hb, ha);
        }
        if (kt < nk) kstep(kt, ha, hb);
        d3r_wait_vm0();   // nothing of this block's DMA is in flight when the epilogue reuses the stages
    } else if constexpr (CF::PP == 1) {
        // ---- ping-pong schedule (non-swapped 16-bit / fp32 operands; KTB = 64: one MFMA k-step per K step) -----------------
        // A K step is two phases, each [L: issue half of the DMA of step kt+2, ds_read this phase's fragments] | s_barrier |
        // [C: 16 MFMAs] | s_barrier. Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in C while its
        // partner is in L. Data flow (ring of 4 stages, loads two steps ahead): every wave waits for ITS loads of step kt+1
        // (counted vmcnt: the 4 loads of step kt+2 stay in flight) in the segment that ends at the barrier in front of the
        // first read of step kt+1 -- C_b for the leading group, L_b for the trailing one; the stage written by step kt+2's
        // DMA was last read in step kt-2, four phases back.
        static_assert(CF::KTB == 64 && CF::NSTAGE == 4 && CF::NW == 8 && FI == 8 && FJ == 4, "ping-pong schedule is written for the 256x256 / 64-byte-row tile");
        const bool trailing = wave_u >= 4;
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else d3r_wait_vm0();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (trailing) __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const char* sb = smem + (kt & 3) * STAGE_BYTES;
            const int coff = (fgrp ^ fsw) * 16;
            const bool more = kt + 2 < nk;
            auto wait_next = [&]() __attribute__((always_inline)) {   // this wave's loads of step kt+1 have landed
                if (kt + 1 < nk) {
                    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                    else d3r_wait_vm0();
                }
            };
            uint4 qf[FJ], pf[4];
            // ---- phase a: P fragments 0-3 x all Q fragments
            if (more) stage_a(kt + 2, (kt + 2) & 3);
#pragma unroll
            for (int f = 0; f < FJ; ++f) qf[f] = *reinterpret_cast<const uint4*>(sb + q_off + (q_row0 + f * 16) * KTB + coff);
#pragma unroll
            for (int f = 0; f < 4; ++f) pf[f] = *reinterpret_cast<const uint4*>(sb + p_off + (p_row0 + f * 16) * KTB + coff);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                for (int fj = 0; fj < FJ; ++fj) TR::mma16(acc[fi][fj], pf[fi], qf[fj]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase b: P fragments 4-7 x all Q fragments
            if (more) stage_w(kt + 2, (kt + 2) & 3);
#pragma unroll
            for (int f = 0; f < 4; ++f) pf[f] = *reinterpret_cast<const uint4*>(sb + p_off + (p_row0 + (4 + f) * 16) * KTB + coff);
            if (trailing) wait_next();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                for (int fj = 0; fj < FJ; ++fj) TR::mma16(acc[4 + fi][fj], pf[fi], qf[fj]);
            __builtin_amdgcn_s_setprio(0);
            if (!trailing) wait_next();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!trailing) __builtin_amdgcn_s_barrier();
    } else if constexpr (DT == D3R_F16F8 && KTB == 64) {
        // ---- fp16 + fp8 rows, 64-byte K steps, three LDS slots of 24 KiB: a 256 (m) x 128 (n) tile by FOUR waves of 128 (n) x 64 (m) --------
        // Why: one 8-wave 256 x 256 block per CU spends a quarter of a K = 1024 tile's life in its epilogue with the MFMA pipes idle (the
        // same forward with the epilogues removed runs 282 instead of 213 pairs/s). This shape keeps the per-wave tile (the LDS read
        // traffic per MFMA) and halves the block, so that TWO blocks are resident per CU (2 x 72 KiB of LDS, 256 VGPRs per wave at two
        // waves per SIMD) and one block's K loop runs under the other's epilogue.
        // A 256-byte super-group [hi fp16 x64 | a8 x64 | b8 x64] is four steps h0, h1, fa, fb; step s lives in slot s % 3 = (j + i) % 3
        // for step i of super-group j. Per super-group three phases, each [wait | barrier | issue loads | math]:
        //   B0: math on h0 (one 16x16x32 k-step, lane group g on chunk g);   loads issued: h1(j), fa(j)
        //   B1: math on h1;                                                  loads issued: fb(j)      -> the slot h0(j) just left
        //   B2: math on (fa, fb): ONE fp8 MFMA per fragment pair, a8 from fa's slot and b8 from fb's;   loads issued: h0(j + 1)
        // A load goes out behind the barrier that follows the last read of its slot's previous tenant (three steps earlier); loads
        // complete in order, so B1 waits with vmcnt(LPS) (fa's pieces may still fly) and B0 / B2 with vmcnt(0).
        static_assert(NS == 3 && CF::NWI == 1, "fp16 + fp8 rows on 64-byte K steps: three slots, waves stacked along m");
        const bool edge = __builtin_amdgcn_readfirstlane(m0 + BM > p.M ? 1 : 0) != 0;
        // offsets are relative to the TILE's first row (its 64-bit address is wave-uniform: SGPRs), so they fit 32 bits whatever the operand size
        const char* const tile_a = reinterpret_cast<const char*>(p.act) + (size_t)__builtin_amdgcn_readfirstlane(m0) * p.lda * EB;
        const char* const tile_w = reinterpret_cast<const char*>(p.wgt) + (size_t)__builtin_amdgcn_readfirstlane(n0) * p.K * EB;
        const uint32_t a0 = (uint32_t)(((size_t)(min(m0 + lrow, p.M - 1) - m0) * p.lda) * EB + lchunk * 16);
        const uint32_t w0 = (uint32_t)(((size_t)lrow * p.K) * EB + lchunk * 16);
        const size_t stride_a = (size_t)CF::PASS_ROWS * p.lda * EB, stride_w = (size_t)CF::PASS_ROWS * p.K * EB;
        auto load_step = [&](int step, int slot) __attribute__((always_inline)) {   // 64-byte K step `step` of every row -> LDS slot
            const uint32_t sb = lds0 + slot * STAGE_BYTES;
            const char* ab = tile_a + (size_t)step * KTB;
            const char* wb = tile_w + (size_t)step * KTB;
#pragma unroll
            for (int q = 0; q < CF::APASS; ++q) {
                if (!edge) {
                    glds16_so(ab + q * stride_a, a0, sb + q * (CF::NW * 1024));
                } else {
                    const int m = min(m0 + q * CF::PASS_ROWS + lrow, p.M - 1);
                    glds16_so(ab, (uint32_t)(((size_t)(m - m0) * p.lda) * EB + lchunk * 16), sb + q * (CF::NW * 1024));
                }
            }
#pragma unroll
            for (int q = 0; q < CF::WPASS; ++q) glds16_so(wb + q * stride_w, w0, sb + BM * KTB + q * (CF::NW * 1024));
        };
        const int coff = (fgrp ^ fsw) * 16;
        auto math_hi = [&](int slot) __attribute__((always_inline)) {
            const char* sb = smem + slot * STAGE_BYTES;
            uint4 qf[FJ];
#pragma unroll
            for (int f = 0; f < FJ; ++f) qf[f] = *reinterpret_cast<const uint4*>(sb + q_off + (q_row0 + f * 16) * KTB + coff);
#pragma unroll
            for (int fi = 0; fi < FI; ++fi) {
                const uint4 pf = *reinterpret_cast<const uint4*>(sb + p_off + (p_row0 + fi * 16) * KTB + coff);
#pragma unroll
                for (int fj = 0; fj < FJ; ++fj) TR::mma16_hi(acc[fi][fj], pf, qf[fj]);
            }
        };
        const int nsg = nk / 4;
        load_step(0, 0);
        int b = 0;
        for (int j = 0; j < nsg; ++j) {
            const int b1 = b == 2 ? 0 : b + 1, b2 = b1 == 2 ? 0 : b1 + 1;
            d3r_wait_vm0();
            __syncthreads();
            load_step(4 * j + 1, b1);
            load_step(4 * j + 2, b2);
            math_hi(b);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            __syncthreads();
            load_step(4 * j + 3, b);
            math_hi(b1);
            d3r_wait_vm0();
            __syncthreads();
            if (j + 1 < nsg) load_step(4 * j + 4, b1);
            {
                const char* sa = smem + b2 * STAGE_BYTES;
                const char* sbb = smem + b * STAGE_BYTES;
                uint4 qa[FJ], qb[FJ];
#pragma unroll
                for (int f = 0; f < FJ; ++f) {
                    const int ro = q_off + (q_row0 + f * 16) * KTB + coff;
                    qa[f] = *reinterpret_cast<const uint4*>(sa + ro);
                    qb[f] = *reinterpret_cast<const uint4*>(sbb + ro);
                }
#pragma unroll
                for (int fi = 0; fi < FI; ++fi) {
                    const int ro = p_off + (p_row0 + fi * 16) * KTB + coff;
                    const uint4 pa = *reinterpret_cast<const uint4*>(sa + ro), pb = *reinterpret_cast<const uint4*>(sbb + ro);
#pragma unroll
                    for (int fj = 0; fj < FJ; ++fj) TR::mma16_f8(acc[fi][fj], pa, pb, qa[fj], qb[fj]);
                }
            }
            b = b1;
        }
    } else if constexpr (DT == D3R_F16X2F8) {
        // ---- 2.5-unit rows (common.hpp, Traits<D3R_F16X2F8>): five K steps per 128 logical k ------------------------------------------------
        //   step s     activation slot (BM x 128 B)                                   weight slot (BN x 128 B)          MFMAs per fragment pair
        //   0          hi of k 0..63      (chunk 0 of super-group 2b)                 w_hi k 0..63     (chunk 5b)       2 x f16   a_hi . w_hi
        //   1          the same chunk again                                           w_lo k 0..63     (chunk 5b + 1)   2 x f16   a_hi . w_lo
        //   2          hi of k 64..127    (chunk 0 of super-group 2b + 1)             w_hi k 64..127   (chunk 5b + 2)   2 x f16
        //   3          the same chunk again                                           w_lo k 64..127   (chunk 5b + 3)   2 x f16
        //   4          b8 of k 0..63 | b8 of k 64..127 (gathered by the DMA sources)  h8 k 0..127      (chunk 5b + 4)   1 x e4m3  a_lo . w_hi  (K = 128)
        // The hi chunk of the activations is fetched twice (an L2 hit one step later) rather than held in registers across a step: both
        // operands keep the plain [slot | slot] stage of every other loop, so the operand-role swap of the V^T regions and every epilogue
        // work unchanged. nn.Linear operands only; same one-offset-per-operand DMA addressing as the fp16 + fp8 loop.
        static_assert(KTB == 128 && NS == 2, "2.5-unit rows: 128-byte K steps, two stages");
        const bool edge = __builtin_amdgcn_readfirstlane(m0 + BM > p.M ? 1 : 0) != 0;
        const size_t wrow = (size_t)p.K * 5;
        const char* const tile_a = reinterpret_cast<const char*>(p.act) + (size_t)__builtin_amdgcn_readfirstlane(m0) * p.lda * EB;
        const char* const tile_w = reinterpret_cast<const char*>(p.wgt) + (size_t)__builtin_amdgcn_readfirstlane(n0) * wrow;
        // source byte offset of this lane's 16-byte slot inside a row: a plain chunk, or (step 4) slots 0-3 <- the b8 quarter of the first
        // super-group's second chunk, slots 4-7 <- the b8 quarter of the second super-group's (relative to the 512-byte block of the row)
        const uint32_t coff_plain = (uint32_t)(lchunk * 16);
        const uint32_t coff_b8 = (uint32_t)(lchunk < 4 ? 128 + 64 + lchunk * 16 : 384 + 64 + (lchunk - 4) * 16);
        const uint32_t arow0 = (uint32_t)(((size_t)(min(m0 + lrow, p.M - 1) - m0) * p.lda) * EB);
        const uint32_t w0 = (uint32_t)((size_t)lrow * wrow) + coff_plain;
        const size_t stride_a = (size_t)CF::PASS_ROWS * p.lda * EB, stride_w = (size_t)CF::PASS_ROWS * wrow;
        // LDS: two ACTIVATION slots and two WEIGHT slots that advance independently -- a weight chunk lives for one step, an activation chunk for
        // the two steps that pair it with w_hi and w_lo, so it is fetched ONCE (8 chunk-rows of DMA per 128 k, like split-fp16, instead of 10) and its
        // successor has two steps to land. Slot s of the activations at s * BM * KTB, of the weights at 2 * BM * KTB + s * BN * KTB (the fragment
        // addresses below do not use p_off / q_off: the operand-role swap of the V^T regions exchanges the two bases).
        // The four f16 steps of a block are ONE loop body (not unrolled: five unrolled bodies with their own address arithmetic spilled), the
        // fp8 step a second one. A pieces: indices WPASS .. LPS - 1, issued by the steps that open a pair (st even) only.
        constexpr int ASLOT = BM * KTB, WSLOT = BN * KTB, WBASE = 2 * ASLOT;
        static_assert(2 * ASLOT + 2 * WSLOT == 2 * STAGE_BYTES, "same LDS footprint as the two-stage loops");
        const uint32_t a_plain = arow0 + coff_plain, a_b8 = arow0 + coff_b8;
        // W piece q of step st of block kb -> weight slot wbuf; A piece q of chunk ac (0: hi k 0..63, 1: hi k 64..127, 2: the b8 gather) -> slot abuf
        auto piece_w = [&](int q, int kb, int st, int wbuf) __attribute__((always_inline)) {
            const char* wb = tile_w + (size_t)kb * 640 + st * 128;
            glds16_so(wb + q * stride_w, w0, lds0 + WBASE + wbuf * WSLOT + q * (CF::NW * 1024));
        };
        auto piece_a = [&](int q, int kb, int ac, int abuf) __attribute__((always_inline)) {
            const char* ab = tile_a + (size_t)kb * 512 + (ac == 1 ? 256 : 0);
            const bool gather = ac == 2;
            const uint32_t dst = lds0 + abuf * ASLOT + q * (CF::NW * 1024);
            if (!edge) {
                glds16_so(ab + q * stride_a, gather ? a_b8 : a_plain, dst);
            } else {
                const int m = min(m0 + q * CF::PASS_ROWS + lrow, p.M - 1);
                glds16_so(ab, (uint32_t)(((size_t)(m - m0) * p.lda) * EB) + (gather ? coff_b8 : coff_plain), dst);
            }
        };
        // IL (PP == 4), as in the fp16 + fp8 loop: the DMA of the coming steps is issued a piece at a time behind the MFMAs of the first fragment rows
        // (fragment reads one row ahead, rows pinned with sched_barrier) instead of in one burst behind the barrier, where neither wave of a SIMD
        // has an MFMA to issue. A step here is 64 MFMAs per wave (1024 cycles) against 96 in the split-fp16 loop: the burst weighs more
        // (measured on the forward: 198.2 -> 202.4 pairs/s, profiles/r04_e).
        constexpr bool IL = CF::PP == 4;
        constexpr int PPR = (LPS + FI - 1) / FI;                  // pieces per fragment row, over the FI rows of the step's first half
        const int nblk = p.K >> 7;
        // what step (kb, st) issues behind its barrier: the W chunk of the NEXT step into the other weight slot, and -- when it opens a pair or is
        // the fp8 step -- the NEXT activation chunk into the other activation slot (chunk 1 at st 0, the gather at st 2, chunk 0 of kb + 1 at st 4)
        auto issue = [&](int i, int kb, int st, int abuf, int wbuf) __attribute__((always_inline)) {     // i: piece index 0 .. LPS - 1
            if (i < CF::WPASS) {
                if (st < 4) piece_w(i, kb, st + 1, wbuf ^ 1);
                else if (kb + 1 < nblk) piece_w(i, kb + 1, 0, wbuf ^ 1);
            } else {
                const int q = i - CF::WPASS;
                if (st == 0) piece_a(q, kb, 1, abuf ^ 1);
                else if (st == 2) piece_a(q, kb, 2, abuf ^ 1);
                else if (st == 4 && kb + 1 < nblk) piece_a(q, kb + 1, 0, abuf ^ 1);
            }
        };
#pragma unroll
        for (int q = 0; q < CF::APASS; ++q) piece_a(q, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < CF::WPASS; ++q) piece_w(q, 0, 0, 0);
        int abuf = 0, wbuf = 0;
        for (int kb = 0; kb < nblk; ++kb) {
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {          // f16 steps: (a_hi chunk, w_hi chunk), (the same a_hi, w_lo chunk), twice
                d3r_wait_vm0();
                __syncthreads();
                if (!IL) {
#pragma unroll
                    for (int i = 0; i < LPS; ++i) issue(i, kb, st, abuf, wbuf);
                }
                const char* ab_ = smem + abuf * ASLOT;
                const char* wb_ = smem + WBASE + wbuf * WSLOT;
                const char* pb = swap ? ab_ : wb_;       // P tile supplies i (weights unless swapped), Q tile supplies j
                const char* qb = swap ? wb_ : ab_;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int coff = ((ks * 4 + fgrp) ^ fsw) * 16;
                    uint4 qf[FJ];
#pragma unroll
                    for (int f = 0; f < FJ; ++f) qf[f] = *reinterpret_cast<const uint4*>(qb + (q_row0 + f * 16) * KTB + coff);
                    if constexpr (!IL) {
#pragma unroll
                        for (int fi = 0; fi < FI; ++fi) {
                            const uint4 pf = *reinterpret_cast<const uint4*>(pb + (p_row0 + fi * 16) * KTB + coff);
#pragma unroll
                            for (int fj = 0; fj < FJ; ++fj) TR::mma16_hi(acc[fi][fj], pf, qf[fj]);
                        }
                    } else {
                        uint4 cur = *reinterpret_cast<const uint4*>(pb + p_row0 * KTB + coff);
#pragma unroll
                        for (int fi = 0; fi < FI; ++fi) {
                            uint4 nxt = cur;
                            if (fi + 1 < FI) nxt = *reinterpret_cast<const uint4*>(pb + (p_row0 + (fi + 1) * 16) * KTB + coff);
#pragma unroll
                            for (int fj = 0; fj < FJ; ++fj) TR::mma16_hi(acc[fi][fj], cur, qf[fj]);
                            if (ks == 0) {
#pragma unroll
                                for (int i = fi * PPR; i < (fi + 1) * PPR && i < LPS; ++i) issue(i, kb, st, abuf, wbuf);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            cur = nxt;
                        }
                    }
                }
                wbuf ^= 1;
                abuf ^= (st & 1);                     // the activation chunk changes after the second step of a pair
            }
            {                                         // fp8 step: b8 of the block's 128 k (gathered) against the h8 chunk
                d3r_wait_vm0();
                __syncthreads();
                if (!IL) {
#pragma unroll
                    for (int i = 0; i < LPS; ++i) issue(i, kb, 4, abuf, wbuf);
                }
                const char* ab_ = smem + abuf * ASLOT;
                const char* wb_ = smem + WBASE + wbuf * WSLOT;
                const char* pb = swap ? ab_ : wb_;
                const char* qb = swap ? wb_ : ab_;
                const int ca = (fgrp ^ fsw) * 16, cb = ((4 + fgrp) ^ fsw) * 16;
                uint4 qa[FJ], qbb[FJ];
#pragma unroll
                for (int f = 0; f < FJ; ++f) {
                    const char* qr = qb + (q_row0 + f * 16) * KTB;
                    qa[f] = *reinterpret_cast<const uint4*>(qr + ca);
                    qbb[f] = *reinterpret_cast<const uint4*>(qr + cb);
                }
                if constexpr (!IL) {
#pragma unroll
                    for (int fi = 0; fi < FI; ++fi) {
                        const char* pr = pb + (p_row0 + fi * 16) * KTB;
                        const uint4 pa = *reinterpret_cast<const uint4*>(pr + ca), pbv = *reinterpret_cast<const uint4*>(pr + cb);
#pragma unroll
                        for (int fj = 0; fj < FJ; ++fj) TR::mma16_f8(acc[fi][fj], pa, pbv, qa[fj], qbb[fj]);
                    }
                } else {
                    const char* pr0 = pb + p_row0 * KTB;
                    uint4 ca0 = *reinterpret_cast<const uint4*>(pr0 + ca), cb0 = *reinterpret_cast<const uint4*>(pr0 + cb);
#pragma unroll
                    for (int fi = 0; fi < FI; ++fi) {
                        uint4 na = ca0, nb = cb0;
                        if (fi + 1 < FI) {
                            const char* pr = pb + (p_row0 + (fi + 1) * 16) * KTB;
                            na = *reinterpret_cast<const uint4*>(pr + ca);
                            nb = *reinterpret_cast<const uint4*>(pr + cb);
                        }
#pragma unroll
                        for (int fj = 0; fj < FJ; ++fj) TR::mma16_f8(acc[fi][fj], ca0, cb0, qa[fj], qbb[fj]);
#pragma unroll
                        for (int i = fi * PPR; i < (fi + 1) * PPR && i < LPS; ++i) issue(i, kb, 4, abuf, wbuf);
                        __builtin_amdgcn_sched_barrier(0);
                        ca0 = na; cb0 = nb;
                    }
                }
                wbuf ^= 1;
                abuf ^= 1;
            }
        }
    } else if constexpr (DT == D3R_F16F8) {
        // ---- fp16 + fp8 rows: the K loop walks 256-byte super-groups [hi fp16 x64 | a8 x64 | b8 x64] (64 logical k) in two K steps --------
        // Step 2t stages the fp16 half into stage 0: two f16 MFMA k-steps, lane group g on chunks g and 4 + g. Step 2t + 1 stages the fp8
        // half into stage 1: lane group g reads chunk g (a8 of its 16 logical k) and chunk 4 + g (b8 of the same k) and ONE 16x16x128
        // fp8 MFMA adds both cross terms (its E8M0 scale undoes the 2^17 of the encodings). nk is even (K % 64 == 0).
        static_assert(KTB == 128 && NS == 2, "fp16 + fp8 rows: 128-byte K steps, two stages");
        // nn.Linear operands only. The rows a lane stages are PASS_ROWS apart: ONE 32-bit byte offset per operand in a VGPR, the row
        // stride added to the wave-uniform 64-bit address of the tile's first row in SGPRs. With one 64-bit address
        // per staged row (16 per lane) this loop spilled, and a scratch reload's vmcnt wait also drains the DMA in flight: load and
        // math serialised. Only a tile that hangs over the last row (rows clamped to M - 1) computes its offsets per piece.
        const bool edge = __builtin_amdgcn_readfirstlane(m0 + BM > p.M ? 1 : 0) != 0;   // wave-uniform, and told so (a scalar branch, not an exec mask)
        // offsets are relative to the TILE's first row (its 64-bit address is wave-uniform: SGPRs), so they fit 32 bits whatever the operand size
        const char* const tile_a = reinterpret_cast<const char*>(p.act) + (size_t)__builtin_amdgcn_readfirstlane(m0) * p.lda * EB;
        const char* const tile_w = reinterpret_cast<const char*>(p.wgt) + (size_t)__builtin_amdgcn_readfirstlane(n0) * p.K * EB;
        const uint32_t a0 = (uint32_t)(((size_t)(min(m0 + lrow, p.M - 1) - m0) * p.lda) * EB + lchunk * 16);
        const uint32_t w0 = (uint32_t)(((size_t)lrow * p.K) * EB + lchunk * 16);
        const size_t stride_a = (size_t)CF::PASS_ROWS * p.lda * EB, stride_w = (size_t)CF::PASS_ROWS * p.K * EB;
        // one of the LPS DMA pieces of K step kt (activation rows first, then weight rows); idx is a compile-time constant after unrolling
        auto piece8 = [&](int idx, int kt, int buf) __attribute__((always_inline)) {
            const uint32_t sb = lds0 + buf * STAGE_BYTES;
            if (idx < CF::APASS) {
                const int q = idx;
                const char* ab = tile_a + (size_t)kt * KTB;
                if (!edge) {
                    glds16_so(ab + q * stride_a, a0, sb + q * (CF::NW * 1024));
                } else {
                    const int m = min(m0 + q * CF::PASS_ROWS + lrow, p.M - 1);
                    glds16_so(ab, (uint32_t)(((size_t)(m - m0) * p.lda) * EB + lchunk * 16), sb + q * (CF::NW * 1024));
                }
            } else {
                const int q = idx - CF::APASS;
                const char* wb = tile_w + (size_t)kt * KTB;
                glds16_so(wb + q * stride_w, w0, sb + BM * KTB + q * (CF::NW * 1024));
            }
        };
        auto stage8 = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < LPS; ++i) piece8(i, kt, buf);
        };
        // IL (PP == 4): the DMA of the next K step is not issued in one burst behind the barrier -- where neither wave of a SIMD has an
        // MFMA to issue yet -- but a few pieces at a time behind the MFMAs of the first fragment rows of the step (fragment reads one row
        // ahead of their use, rows pinned with sched_barrier), so that the matrix pipe always has work from one of the two waves.
        constexpr bool IL = CF::PP == 4;
        constexpr int PPR_HI = (LPS + FI - 1) / FI;               // pieces per fragment row, fp16 step: over the FI rows of its first k-step
        constexpr int PPR_F8 = (LPS + FI / 2 - 1) / (FI / 2);     // fp8 step: over the first half of its rows
        stage8(0, 0);
        for (int kt = 0; kt < nk; kt += 2) {
            d3r_wait_vm0();
            __syncthreads();
            if (!IL) stage8(kt + 1, 1);
            {
                const char* sb = smem;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int coff = ((ks * 4 + fgrp) ^ fsw) * 16;
                    uint4 qf[FJ];
#pragma unroll
                    for (int f = 0; f < FJ; ++f) qf[f] = *reinterpret_cast<const uint4*>(sb + q_off + (q_row0 + f * 16) * KTB + coff);
                    if constexpr (!IL) {
#pragma unroll
                        for (int fi = 0; fi < FI; ++fi) {
                            const uint4 pf = *reinterpret_cast<const uint4*>(sb + p_off + (p_row0 + fi * 16) * KTB + coff);
#pragma unroll
                            for (int fj = 0; fj < FJ; ++fj) TR::mma16_hi(acc[fi][fj], pf, qf[fj]);
                            if (p.f8_proxy) {
#pragma unroll
                                for (int fj = 0; fj < FJ; ++fj) TR::mma16_hi(acc[fi][fj], pf, qf[fj]);
                            }
                        }
                    } else {
                        uint4 cur = *reinterpret_cast<const uint4*>(sb + p_off + p_row0 * KTB + coff);
#pragma unroll
                        for (int fi = 0; fi < FI; ++fi) {
                            uint4 nxt = cur;
                            if (fi + 1 < FI) nxt = *reinterpret_cast<const uint4*>(sb + p_off + (p_row0 + (fi + 1) * 16) * KTB + coff);
#pragma unroll
                            for (int fj = 0; fj < FJ; ++fj) TR::mma16_hi(acc[fi][fj], cur, qf[fj]);
                            if (p.f8_proxy) {     // MEASUREMENT AID (results invalid): the MFMA mix of a 2.5-unit scheme, see launch_gemm
#pragma unroll
                                for (int fj = 0; fj < FJ; ++fj) TR::mma16_hi(acc[fi][fj], cur, qf[fj]);
                            }
                            if (ks == 0) {
#pragma unroll
                                for (int i = fi * PPR_HI; i < (fi + 1) * PPR_HI && i < LPS; ++i) piece8(i, kt + 1, 1);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            cur = nxt;
                        }
                    }
                }
            }
            d3r_wait_vm0();
            __syncthreads();
            const bool more = kt + 2 < nk;
            if (!IL && more) stage8(kt + 2, 0);
            {
                const char* sb = smem + STAGE_BYTES;
                const int ca = (fgrp ^ fsw) * 16, cb = ((4 + fgrp) ^ fsw) * 16;
                uint4 qa[FJ], qb[FJ];
#pragma unroll
                for (int f = 0; f < FJ; ++f) {
                    const char* qr = sb + q_off + (q_row0 + f * 16) * KTB;
                    qa[f] = *reinterpret_cast<const uint4*>(qr + ca);
                    qb[f] = *reinterpret_cast<const uint4*>(qr + cb);
                }
                if constexpr (!IL) {
#pragma unroll
                    for (int fi = 0; fi < FI; ++fi) {
                        const char* pr = sb + p_off + (p_row0 + fi * 16) * KTB;
                        const uint4 pa = *reinterpret_cast<const uint4*>(pr + ca), pb = *reinterpret_cast<const uint4*>(pr + cb);
                        if (!p.f8_proxy || (kt & 2)) {
#pragma unroll
                        for (int fj = 0; fj < FJ; ++fj) TR::mma16_f8(acc[fi][fj], pa, pb, qa[fj], qb[fj]);
                        }
                    }
                } else {
                    const char* pr0 = sb + p_off + p_row0 * KTB;
                    uint4 ca0 = *reinterpret_cast<const uint4*>(pr0 + ca), cb0 = *reinterpret_cast<const uint4*>(pr0 + cb);
#pragma unroll
                    for (int fi = 0; fi < FI; ++fi) {
                        uint4 na = ca0, nb = cb0;
                        if (fi + 1 < FI) {
                            const char* pr = sb + p_off + (p_row0 + (fi + 1) * 16) * KTB;
                            na = *reinterpret_cast<const uint4*>(pr + ca);
                            nb = *reinterpret_cast<const uint4*>(pr + cb);
                        }
                        if (!p.f8_proxy || (kt & 2)) {
#pragma unroll
                        for (int fj = 0; fj < FJ; ++fj) TR::mma16_f8(acc[fi][fj], ca0, cb0, qa[fj], qb[fj]);
                        }
                        if (more && fi < FI / 2) {
#pragma unroll
                            for (int i = fi * PPR_F8; i < (fi + 1) * PPR_F8 && i < LPS; ++i) piece8(i, kt + 2, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        ca0 = na; cb0 = nb;
                    }
                }
            }
        }
    } else if constexpr (CF::PP == 6) {
        // ---- split-fp16, two blocks per CU: activation ring of 2, ONE weight slot, weights of the step in registers -------------------------
        // Why: with one 8-wave block per CU a tile's epilogue (10-19 us of an 85 us K = 1024 tile) and the [wait | barrier | DMA issue |
        // first fragment reads] bubble of every K step run with the CU's MFMA pipes idle: whole-kernel MFMA duty 55 % (round 3 traces: the K
        // loop itself is not slowed by the stores, the epilogue simply adds to it). The shapes that fit two blocks in 160 KiB so far halved
        // the per-wave tile (128 x 128 by 64 x 64 waves: twice the LDS reads per MFMA) or the K step (64-byte steps: a barrier every 680
        // MFMA cycles and one phase of lookahead for the DMA). This one keeps both: the 128 (n) x 64 (m) wave tile and 128-byte K steps;
        // what it gives up is the second weight slot -- a wave reads its 8 weight fragments (hi, lo: 64 VGPRs) at the top of the step,
        // a second barrier frees the slot, and the DMA of the NEXT step's weight rows goes into it while the MFMAs run from registers.
        //   step kt:  vmcnt(0) | barrier A (rows of step kt landed; every wave done with activation slot (kt+1)&1)
        //             -> 16 ds_read_b128: weight fragments | lgkmcnt(0) | barrier B (weight slot free)
        //             -> DMA of step kt+1: activation rows -> slot (kt+1)&1, weight rows -> THE slot
        //             -> 4 x [2 ds_read_b128 activation fragment (one ahead), 24 MFMAs]
        // Same accumulation order per output element as every other configuration (three terms per k-step, K ascending): bit-identical.
        static_assert(DT == D3R_F16X3 && KTB == 128 && CF::NWI == 1 && FI == 8, "weights-in-registers loop: split-fp16, waves stacked along m");
        constexpr int ASLOT = BM * KTB, WBASE = 2 * ASLOT;
        const int chi = (fgrp ^ fsw) * 16, clo = ((4 + fgrp) ^ fsw) * 16;   // LDS image: [hi0..hi3 | lo0..lo3]
        auto stage_a6 = [&](int kt, int slot) __attribute__((always_inline)) {
            const uint32_t sb = lds0 + slot * ASLOT;
            const size_t koff = (size_t)kt * KTB;
            if (p.amode == AMODE_LINEAR) {
#pragma unroll
                for (int q = 0; q < CF::APASS; ++q) glds16(reinterpret_cast<const char*>((size_t)arow[q]) + koff, sb + q * (CF::NW * 1024));
            } else {
                int tap, c0;
                conv_k_step(p, kt * KT, 128 / EB, tap, c0);
                const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
#pragma unroll
                for (int q = 0; q < CF::APASS; ++q) {
                    const int pk = (int)(arow[q] >> 32), ibase = (int)(unsigned)arow[q];
                    const int iy = (pk >> 16) + ky, ix = (int)(short)(pk & 0xFFFF) + kx;
                    const bool ok = (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
                    const char* src = reinterpret_cast<const char*>(p.act) + ((size_t)(ibase + iy * p.Win + ix) * p.cstride + c0) * EB + lchunk * 16;
                    glds16(ok ? src : zsrc, sb + q * (CF::NW * 1024));
                }
            }
        };
        auto stage_w6 = [&](int kt) __attribute__((always_inline)) {
            const size_t koff = (size_t)kt * KTB;
#pragma unroll
            for (int q = 0; q < CF::WPASS; ++q) glds16(wsrc[q] + koff, lds0 + WBASE + q * (CF::NW * 1024));
        };
        stage_a6(0, 0);
        stage_w6(0);
        for (int kt = 0; kt < nk; ++kt) {
            d3r_wait_vm0();
            __syncthreads();                                  // barrier A
            uint4 pf[FI], pl[FI];
            {
                const char* wb = smem + WBASE;
#pragma unroll
                for (int f = 0; f < FI; ++f) {
                    const char* pr = wb + (p_row0 + f * 16) * KTB;
                    pf[f] = *reinterpret_cast<const uint4*>(pr + chi);
                    pl[f] = *reinterpret_cast<const uint4*>(pr + clo);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();                                  // barrier B: the weight slot is free
            if (kt + 1 < nk) {
                stage_a6(kt + 1, (kt + 1) & 1);
                stage_w6(kt + 1);
            }
            const char* ab = smem + (kt & 1) * ASLOT;
            uint4 qh = *reinterpret_cast<const uint4*>(ab + q_row0 * KTB + chi), ql = *reinterpret_cast<const uint4*>(ab + q_row0 * KTB + clo);
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) {
                uint4 nh = qh, nl = ql;
                if (fj + 1 < FJ) {
                    const char* qr = ab + (q_row0 + (fj + 1) * 16) * KTB;
                    nh = *reinterpret_cast<const uint4*>(qr + chi);
                    nl = *reinterpret_cast<const uint4*>(qr + clo);
                }
#pragma unroll
                for (int fi = 0; fi < FI; ++fi) TR::mma16x3(acc[fi][fj], pf[fi], pl[fi], qh, ql);
                qh = nh; ql = nl;
            }
        }
    } else {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) stage(s, s);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(NS - 2, nk - 1 - kt);   // younger steps already issued
        if (NS == 2 || ahead <= 0) d3r_wait_vm0();
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
        __syncthreads();
        if (kt + NS - 1 < nk) {
            int nb = buf + NS - 1;
            nb = nb >= NS ? nb - NS : nb;
            stage(kt + NS - 1, nb);
        }
        const char* sb = smem + buf * STAGE_BYTES;
        if constexpr (DT == D3R_F16X3) {
            static_assert(DT != D3R_F16X3 || KTB == 128, "split-fp16 rows are [hi x8][lo x8] groups: 128-byte K rows only");
            // 128 bytes of a row = 32 logical k = 4 groups [hi x8][lo x8]; lane group fgrp owns group fgrp (hi chunk fgrp, lo chunk 4 + fgrp of the LDS image)
            const int chi = (fgrp ^ fsw) * 16, clo = ((4 + fgrp) ^ fsw) * 16;   // LDS image: [hi0..hi3 | lo0..lo3]
            uint4 qf[FJ], ql[FJ];
#pragma unroll
            for (int f = 0; f < FJ; ++f) {
                const char* qr = sb + q_off + (q_row0 + f * 16) * KTB;
                qf[f] = *reinterpret_cast<const uint4*>(qr + chi);
                ql[f] = *reinterpret_cast<const uint4*>(qr + clo);
            }
#pragma unroll
            for (int fi = 0; fi < FI; ++fi) {
                const char* pr = sb + p_off + (p_row0 + fi * 16) * KTB;
                const uint4 pf = *reinterpret_cast<const uint4*>(pr + chi), pl = *reinterpret_cast<const uint4*>(pr + clo);
#pragma unroll
                for (int fj = 0; fj < FJ; ++fj) TR::mma16x3(acc[fi][fj], pf, pl, qf[fj], ql[fj]);
            }
        } else if constexpr (DT != D3R_F16F8 && DT != D3R_F16X2F8) {
#pragma unroll
            for (int ks = 0; ks < KTB / 64; ++ks) {
                const int coff = ((ks * 4 + fgrp) ^ fsw) * 16;
                uint4 pf[FI], qf[FJ];
#pragma unroll
                for (int f = 0; f < FJ; ++f) qf[f] = *reinterpret_cast<const uint4*>(sb + q_off + (q_row0 + f * 16) * KTB + coff);
#pragma unroll
                for (int f = 0; f < FI; ++f) pf[f] = *reinterpret_cast<const uint4*>(sb + p_off + (p_row0 + f * 16) * KTB + coff);
#pragma unroll
                for (int fi = 0; fi < FI; ++fi)
#pragma unroll
                    for (int fj = 0; fj < FJ; ++fj) TR::mma16(acc[fi][fj], pf[fi], qf[fj]);
            }
        }
        buf = buf + 1 >= NS ? 0 : buf + 1;
    }
    }  // !PP

    // ---- split-K: partial tile -> slab, ticket, the last arriver adds the slices up in slice order ------------------------------------------------
    if constexpr (DT == D3R_F16X3 && CF::PP == 0) {
        if (p.splitk > 1) {
            __builtin_amdgcn_s_setprio(0);
            // Partial tiles travel through memory operations of AGENT scope (global_store / global_load ... sc1: written through to, and read from, the
            // level every XCD sees) instead of ordinary stores bracketed by release / acquire fences: on gfx950 an agent-scope release is a write-back of
            // the XCD's whole L2 (buffer_wbl2) and an acquire invalidates it (buffer_inv) -- per BLOCK, under the K loops of the other blocks of the launch.
            // Measured with the fences (profiles/r06_c/splitk_probe_fences.log): one pair 9.99 -> 11.36 ms, i.e. split-K LOST 9 us per launch.
            float* slab = p.sk_slab + ((size_t)sk_tile * p.splitk) * (BM * BN) + (size_t)wave * (FI * FJ * 256) + lane;
            float* mine = slab + (size_t)sk_slice * (BM * BN);
#pragma unroll
            for (int a = 0; a < FI; ++a)
#pragma unroll
                for (int b = 0; b < FJ; ++b)
#pragma unroll
                    for (int c = 0; c < 4; ++c) __hip_atomic_store(mine + ((a * FJ + b) * 4 + c) * 64, acc[a][b][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // publish (cdna_hip_programming.md guideline 16, counter form): every wave's stores acknowledged -> block barrier -> ticket
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* flag = reinterpret_cast<unsigned*>(smem);      // the K loop's LDS stages are free (every wave is past its last fragment read)
            if (tid == 0) *flag = __hip_atomic_fetch_add(p.sk_cnt + sk_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned ticket = *flag;
            if (ticket != (unsigned)(p.splitk - 1)) return;          // not the last slice of this tile: done
            if (tid == 0) __hip_atomic_store(p.sk_cnt + sk_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
            // ((s0 + s1) + s2) + ...: every slice read back from its slab (its own too), so the sum does not depend on which block arrived last
#pragma unroll
            for (int a = 0; a < FI; ++a)
#pragma unroll
                for (int b = 0; b < FJ; ++b)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float t = __hip_atomic_load(slab + ((a * FJ + b) * 4 + c) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (int sl = 1; sl < p.splitk; ++sl)
                            t += __hip_atomic_load(slab + (size_t)sl * (BM * BN) + ((a * FJ + b) * 4 + c) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        acc[a][b][c] = t;
                    }
        }
    }
    // ---- epilogue ------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(0);
    if (p.trace && tid == 0) p.trace[(size_t)blockIdx.x * 8 + 2] = (unsigned long long)wall_clock64();
    struct TraceEnd {   // stamps "epilogue issued" and "stores drained" on every return path
        const GemmParams& p; int tid;
        __device__ ~TraceEnd() {
            if (p.trace) {
                const unsigned long long t3 = (unsigned long long)wall_clock64();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) { p.trace[(size_t)blockIdx.x * 8 + 3] = t3; p.trace[(size_t)blockIdx.x * 8 + 4] = (unsigned long long)wall_clock64(); }
            }
        }
    } trace_end{p, tid};
    if (p.flags & GF_NOSTORE) {   // measurement aid (D3R_GEMM_NOSTORE=1): keep the math, skip the epilogue's memory traffic
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < FI; ++a)
#pragma unroll
            for (int b = 0; b < FJ; ++b) t += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
        if (t == 123.456f) reinterpret_cast<float*>(p.out)[0] = t;
        return;
    }
    const int i4 = (lane >> 4) * 4;  // first of this lane's 4 consecutive i inside a fragment
    const int jl = lane & 15;

    // ---- fused DPT head tail (EPI_HEAD4, kernels.hpp): this wave holds every output channel of its FJ x 16 rows --------------
    if constexpr (DT == D3R_F16X3 && CF::NWI == 1 && FI == 8) {
        if (p.epi == EPI_HEAD4) {
            const float* hw = reinterpret_cast<const float*>(p.res1);
            const float* hb = reinterpret_cast<const float*>(p.res2);
            const float* bsrc = p.bias ? p.bias : hw;
            const int C = p.n_store;
            const bool has_bias = p.bias != nullptr;
            const float floor_v = (p.flags & GF_RELU) ? 0.f : -3.0e38f;
            float part[FJ][4];
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) part[fj][0] = part[fj][1] = part[fj][2] = part[fj][3] = 0.f;
            // 1x1 weights and bias of fragment column fi: requested one fragment ahead (double buffered); the fence keeps hipcc from hoisting
            // all 40 loads (160 registers next to the 128 accumulators) to the top
            float4 wb[2][5];
            auto request = [&](int fi, float4 (&d)[5]) __attribute__((always_inline)) {
                const int n = fi * 16 + i4;                     // n0 == 0: one tile spans the channels
                const int nc = min(n, C - 4);                   // channels >= C: zero accumulators (padded weight rows), zero 1x1 weights
                d[4] = *reinterpret_cast<const float4*>(bsrc + nc);
#pragma unroll
                for (int o = 0; o < 4; ++o) d[o] = *reinterpret_cast<const float4*>(hw + (size_t)o * C + nc);
            };
            request(0, wb[0]);
#pragma unroll
            for (int fi = 0; fi < FI; ++fi) {
                if (fi + 1 < FI) request(fi + 1, wb[(fi + 1) & 1]);
                const float live = fi * 16 + i4 < C ? 1.f : 0.f;
                const float4 bt = wb[fi & 1][4];
                const float4 bi = make_float4(has_bias ? bt.x : 0.f, has_bias ? bt.y : 0.f, has_bias ? bt.z : 0.f, has_bias ? bt.w : 0.f);
                float4 w4[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float4 t = wb[fi & 1][o];
                    w4[o] = make_float4(t.x * live, t.y * live, t.z * live, t.w * live);
                }
#pragma unroll
                for (int fj = 0; fj < FJ; ++fj) {
                    const f32x4_t a = acc[fi][fj];
                    // ReLU as a select-free maximum (a branch here splits the block and the register allocator spills around it)
                    const float v0 = fmaxf(a[0] + bi.x, floor_v), v1 = fmaxf(a[1] + bi.y, floor_v), v2 = fmaxf(a[2] + bi.z, floor_v), v3 = fmaxf(a[3] + bi.w, floor_v);
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        part[fj][o] = fmaf(v3, w4[o].w, fmaf(v2, w4[o].z, fmaf(v1, w4[o].y, fmaf(v0, w4[o].x, part[fj][o]))));
                }
                asm volatile("" ::: "memory");
            }
            // the four lane groups hold four quarters of a row's channels: add them up; lane group g then owns the rows of fragment g
            const int g = lane >> 4;
            float r4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj)
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float t = rows_sum4(part[fj][o]);
                    r4[o] = (fj == g) ? t : r4[o];
                }
            static_assert(FJ == 4, "one row fragment per lane group");
            const int m = m0 + wj * (FJ * 16) + g * 16 + jl;
            if (m < p.M)
                postprocess_store(r4[0] + hb[0], r4[1] + hb[1], r4[2] + hb[2], r4[3] + hb[3], reinterpret_cast<float*>(p.out),
                                  reinterpret_cast<float*>(p.out2), (size_t)m, p.ldo, p.ldo2, p.post);
            return;
        }
    }

    // ---- wide epilogues: accumulators -> wave-private LDS tile -> whole 128-byte rows ---------------------------------
    // An MFMA accumulator fragment gives a lane 4 consecutive columns of ONE row and 16 different rows per wave
    // instruction: stored directly that is 32 B (16-bit types) or 64 B (fp32) per row per instruction, and the store
    // tail costs as much as the whole K loop at K = 1024 (measured with D3R_GEMM_NOSTORE: 750 -> 1270 TF/s,
    // profiles/r01_call5). Staging a [64 rows j][64 x 16-bit or 32 x fp32 columns i] tile per wave in the (now idle) LDS
    // stages turns every global access into 8 lanes x 16 B = one full 128-byte line per row, 8 rows per instruction.
    constexpr int WROW = 144;   // 128 payload bytes + 16: keeps ds_write_b64/b128 and ds_read_b128 (nearly) conflict free
    constexpr bool DT16 = (DT == D3R_BF16 || DT == D3R_F16);
    const bool heads_wide = p.epi == EPI_HEADS && (!swap || (p.ntok & 63) == 0);
    const bool wide16 = DT16 && !(p.flags & GF_NOWIDE) &&
                        (((p.epi == EPI_GELU || (p.epi == EPI_T && !p.res1 && !p.res2 && !p.out2)) && (p.ldo & 7) == 0) || heads_wide);
    const bool wide32 = p.epi == EPI_F32 && !(p.flags & GF_NOWIDE);
    // split-fp16 outputs: the same [64 rows j][32 columns i] staging tile as the fp32 epilogue (a 32-column piece of an x3 row
    // is 128 bytes: 4 groups [hi x8][lo x8]), written in the final byte layout so that the read phase stores whole lines
    constexpr bool DTX3 = (DT == D3R_F16X3);
    // fp16 + fp8 GEMMs: the attention projections hand q / k / v^T to the attention kernel in the split-fp16 layout (HDT), the typed
    // copy of an fp32-residual launch (out2: the DPT hooks) is split-fp16 too (O2DT); GELU / plain outputs are fp16 + fp8 activation rows
    constexpr bool DTF8 = (DT == D3R_F16F8 || DT == D3R_F16X2F8);   // outputs of both are fp16 + fp8 ACTIVATION rows (or split-fp16 heads / fp32)
    constexpr int HDT = DTF8 ? D3R_F16X3 : DT, O2DT = DTF8 ? D3R_F16X3 : DT;
    const bool widex3 = (DTX3 || DTF8) && !(p.flags & GF_NOWIDE) &&
                        ((DTX3 && (p.epi == EPI_GELU || (p.epi == EPI_T && !p.res1 && !p.res2 && !p.out2)) && (p.ldo & 7) == 0) ||
                         (p.epi == EPI_HEADS && (p.ntok & 31) == 0 && (p.ldv & 7) == 0));
    const bool widef8 = DTF8 && !(p.flags & GF_NOWIDE) && (p.epi == EPI_GELU || (p.epi == EPI_T && !p.res1 && !p.res2 && !p.out2)) &&
                        (p.ldo & 63) == 0 && (p.n_store & 63) == 0;
    if (wide16 || wide32 || widex3 || widef8) {
        __syncthreads();   // every wave is done with the K loop's LDS stages
        constexpr int WR = FJ * 16, RP = FJ * 2;   // staging rows per wave (its j range) / read-phase passes of 8 rows
        char* wreg = smem + wave * (WR * WROW);
        // P side (i, 4 consecutive per lane) base / Q side (j) base in global coordinates
        const int ib = (swap ? m0 : n0) + wi * (FI * 16), jb = (swap ? n0 : m0) + wj * (FJ * 16);
        const int rrow = lane >> 3, rch = lane & 7;   // read phase: 8 lanes cover one 128-byte row
        const bool nt = (p.flags & GF_NTSTORE) != 0;
        // No global read of the wide epilogues sits behind a run-time condition. A load inside `if (p.bias)` / `if (rope)` /
        // `if (m < M)` whose value is used after the join is a PHI: hipcc copies it at the end of the block and puts
        // `s_waitcnt vmcnt(0)` there -- load -> wait -> use once per fragment, and vmcnt also counts the stores issued in
        // between. Measured on the network: residual-stream projections ran 380-580 TF/s against 620-950 for the same shapes
        // without a residual (32 serial HBM round trips per tile). So every read is issued unconditionally, ahead of its use,
        // from a clamped index (columns >= n_store / rows >= M are never stored) or, when the operand is absent, from another
        // readable buffer of the launch, and what is conditional is a register select on the loaded value.
        if constexpr (DT16) {
            if (wide16) {
                const bool bias_i = !swap && p.bias != nullptr, bias_j = swap && p.bias != nullptr;
                const float* bsrc = p.bias ? p.bias : reinterpret_cast<const float*>(p.wgt);          // >= n_pad floats either way
                const bool has_tab = p.epi == EPI_HEADS && p.rope_table != nullptr;
                const float* rtab = has_tab ? p.rope_table : reinterpret_cast<const float*>(p.wgt);   // dummy reads stay inside row 0 of the weights
                const int r_ntok = has_tab ? p.ntok : 1, r_tokw = has_tab ? p.tok_w : 1, r_M = has_tab ? p.M : 1;
#pragma unroll
                for (int g = 0; g < FI / 4; ++g) {
                    // ---- registers -> LDS (bias, activation / RoPE applied here, rounded once to the 16-bit type)
                    const int ig = ib + g * 64;                 // first i (column n, or token m when swapped) of this group
                    int region = 0, h = 0;
                    bool rope = false, vt = false;
                    char* hdst = nullptr;
                    if (p.epi == EPI_HEADS) {
                        // wave-uniform, and told so (readfirstlane): left in a VGPR the destination-pointer select became a vector
                        // load from the kernel-argument segment inside the store loop, followed by vmcnt(0) -- which also waits
                        // for the previous row's store, i.e. one HBM round trip per stored row
                        const int nh = __builtin_amdgcn_readfirstlane(swap ? jb : ig);   // 64-aligned n of this (wave, group): one head
                        region = nh / p.head_c;
                        h = (nh - region * p.head_c) >> 6;
                        rope = !swap && head_kind_of(p, region) == HEAD_ROPE;
                        vt = !swap && head_kind_of(p, region) == HEAD_VT;   // transposed through the staging tile
                        hdst = reinterpret_cast<char*>(head_dst_of(p, region));
                    }
                    float4 bq[4];
#pragma unroll
                    for (int fl = 0; fl < 4; ++fl) {
                        const float4 t = *reinterpret_cast<const float4*>(bsrc + max(min(ig + fl * 16 + i4, p.n_store - 4), 0));
                        bq[fl] = make_float4(bias_i ? t.x : 0.f, bias_i ? t.y : 0.f, bias_i ? t.z : 0.f, bias_i ? t.w : 0.f);
                    }
                    float bjv[FJ];     // per-row bias of the operand-swapped (V^T) tiles, requested with the rest
#pragma unroll
                    for (int fj = 0; fj < FJ; ++fj) bjv[fj] = bsrc[max(min(jb + fj * 16 + jl, p.n_store - 1), 0)];
                    float4 rt[2][4];   // RoPE table rows, double buffered: fragment fj + 1 is requested before fragment fj is rotated
                    load_rope_rows(rtab, r_ntok, r_tokw, r_M, jb + jl, i4, rt[0]);
#pragma unroll
                    for (int fj = 0; fj < FJ; ++fj) {
                        const int j = jb + fj * 16 + jl;        // row m (or feature n when swapped)
                        const float bj = bias_j ? bjv[fj] : 0.f;
                        if (fj + 1 < FJ) load_rope_rows(rtab, r_ntok, r_tokw, r_M, jb + (fj + 1) * 16 + jl, i4, rt[(fj + 1) & 1]);
                        const float4 a0 = rt[fj & 1][0], a1 = rt[fj & 1][1], b0 = rt[fj & 1][2], b1 = rt[fj & 1][3];
                        const float cc[4] = {a0.x, a0.z, a1.x, a1.z}, ss[4] = {a0.y, a0.w, a1.y, a1.w};      // y position: pairs (fl 0, fl 1)
                        const float cc2[4] = {b0.x, b0.z, b1.x, b1.z}, ss2[4] = {b0.y, b0.w, b1.y, b1.w};    // x position: pairs (fl 2, fl 3)
                        float vals[4][4];
#pragma unroll
                        for (int fl = 0; fl < 4; ++fl) {
                            const float4 bi = swap ? make_float4(bj, bj, bj, bj) : bq[fl];
                            const f32x4_t a = acc[g * 4 + fl][fj];
                            vals[fl][0] = a[0] + bi.x; vals[fl][1] = a[1] + bi.y; vals[fl][2] = a[2] + bi.z; vals[fl][3] = a[3] + bi.w;
                        }
                        if (rope) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float u = vals[0][r], v = vals[1][r], u2 = vals[2][r], v2 = vals[3][r];
                                vals[0][r] = u * cc[r] - v * ss[r];
                                vals[1][r] = v * cc[r] + u * ss[r];
                                vals[2][r] = u2 * cc2[r] - v2 * ss2[r];
                                vals[3][r] = v2 * cc2[r] + u2 * ss2[r];
                            }
                        }
#pragma unroll
                        for (int fl = 0; fl < 4; ++fl) {
                            float v0 = vals[fl][0], v1 = vals[fl][1], v2 = vals[fl][2], v3 = vals[fl][3];
                            if (p.epi == EPI_GELU) gelu4<DT>(v0, v1, v2, v3);
                            if (p.flags & GF_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                            uint2 pk;
                            pk.x = TR::pack2(v0, v1);
                            pk.y = TR::pack2(v2, v3);
                            if (!vt) {
                                *reinterpret_cast<uint2*>(wreg + (fj * 16 + jl) * WROW + (fl * 16 + i4) * 2) = pk;
                            } else {   // tile[feature][token]: this lane's 4 features of token fj*16+jl
                                char* tcol = wreg + (fl * 16 + i4) * WROW + (fj * 16 + jl) * 2;
                                *reinterpret_cast<uint16_t*>(tcol) = (uint16_t)(pk.x & 0xFFFFu);
                                *reinterpret_cast<uint16_t*>(tcol + WROW) = (uint16_t)(pk.x >> 16);
                                *reinterpret_cast<uint16_t*>(tcol + 2 * WROW) = (uint16_t)(pk.y & 0xFFFFu);
                                *reinterpret_cast<uint16_t*>(tcol + 3 * WROW) = (uint16_t)(pk.y >> 16);
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS ops are in order; this only fences the compiler
                    // ---- LDS -> global: row = j, 8 lanes x 16 B = 64 consecutive i
#pragma unroll
                    for (int pass = 0; pass < 8; ++pass) {
                        const int row = pass * 8 + rrow;
                        const uint4 v = *reinterpret_cast<const uint4*>(wreg + row * WROW + rch * 16);
                        const int j = jb + row, i = ig + rch * 8;
                        if (p.epi == EPI_HEADS) {
                            char* dst = hdst;
                            if (vt) {               // v^T from the transposed tile: row = feature, 8 consecutive tokens per lane
                                const int tok = jb + rch * 8;
                                if (tok < p.M) {
                                    const int b = tok / p.ntok, t = tok - b * p.ntok;
                                    store16(dst + ((((size_t)(b * p.heads + h) * 64 + row) * p.ldv) + t) * 2, v, nt);
                                }
                            } else if (!swap) {     // q / k: [b][h][token][64]
                                if (j < p.M) {
                                    const int b = j / p.ntok, t = j - b * p.ntok;
                                    store16(dst + ((((size_t)(b * p.heads + h) * p.ntok + t) * 64) + rch * 8) * 2, v, nt);
                                }
                            } else if (i < p.M && j < p.n_store) {   // v^T: [b][h][feature][ldv], 8 consecutive tokens (ntok % 64 == 0)
                                const int b = i / p.ntok, t = i - b * p.ntok;
                                const int dd = j - (j / 64) * 64;
                                *reinterpret_cast<uint4*>(dst + ((((size_t)(b * p.heads + h) * 64 + dd) * p.ldv) + t) * 2) = v;
                            }
                        } else if (j < p.M && i < p.n_store) {
                            char* o = reinterpret_cast<char*>(p.out) + ((size_t)j * p.ldo + i) * 2;
                            if (i + 8 <= p.n_store) store16(o, v, nt);
                            else *reinterpret_cast<uint2*>(o) = make_uint2(v.x, v.y);   // n_store % 4 == 0
                        }
                    }
                    asm volatile("" ::: "memory");
                }
                return;
            }
        }
        if constexpr (DTF8) {
            if (widef8) {
                // fp16 + fp8 activation rows: a wave's 64-column group of a row is one 256-byte super-group [hi x64 | a8 x64 | b8 x64].
                // Staged 32 rows at a time ([32][256 + 16] bytes per wave), read back 16 lanes per row: whole 256-byte lines to HBM.
                using TF = Traits<D3R_F16F8>;
                constexpr int RW = 272;
                char* freg = smem + wave * (32 * RW);
                const float* bsrc = p.bias ? p.bias : reinterpret_cast<const float*>(p.wgt);
                const bool has_bias = p.bias != nullptr;
                const int r16 = lane >> 4, c16 = lane & 15;
#pragma unroll
                for (int g = 0; g < FI / 4; ++g) {
                    const int ig = ib + g * 64;                 // first column n of this 64-wide group (64-aligned)
                    float4 bq[4];
#pragma unroll
                    for (int fl = 0; fl < 4; ++fl) {
                        const float4 t = *reinterpret_cast<const float4*>(bsrc + max(min(ig + fl * 16 + i4, p.n_store - 4), 0));
                        bq[fl] = make_float4(has_bias ? t.x : 0.f, has_bias ? t.y : 0.f, has_bias ? t.z : 0.f, has_bias ? t.w : 0.f);
                    }
#pragma unroll
                    for (int half = 0; half < FJ / 2; ++half) {
#pragma unroll
                        for (int fjj = 0; fjj < 2; ++fjj) {
                            const int fj = half * 2 + fjj;
                            char* w = freg + (fjj * 16 + jl) * RW;
#pragma unroll
                            for (int fl = 0; fl < 4; ++fl) {
                                const f32x4_t a = acc[g * 4 + fl][fj];
                                float v0 = a[0] + bq[fl].x, v1 = a[1] + bq[fl].y, v2 = a[2] + bq[fl].z, v3 = a[3] + bq[fl].w;
                                if (p.epi == EPI_GELU) gelu4<DT>(v0, v1, v2, v3);
                                if (p.flags & GF_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                                uint2 hh; uint32_t a8, b8;
                                TF::enc4<false>(v0, v1, v2, v3, hh, a8, b8);
                                const int c0 = fl * 16 + i4;       // 4 consecutive logical columns of the group
                                *reinterpret_cast<uint2*>(w + c0 * 2) = hh;
                                *reinterpret_cast<uint32_t*>(w + 128 + c0) = a8;
                                *reinterpret_cast<uint32_t*>(w + 192 + c0) = b8;
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int pass = 0; pass < 8; ++pass) {
                            const int row = pass * 4 + r16;
                            const uint4 v = *reinterpret_cast<const uint4*>(freg + row * RW + c16 * 16);
                            const int j = jb + half * 32 + row;
                            if (j < p.M && ig < p.n_store) store16(reinterpret_cast<char*>(p.out) + ((size_t)j * p.ldo + ig) * 4 + c16 * 16, v, nt);
                        }
                        asm volatile("" ::: "memory");
                    }
                }
                return;
            }
        }
        if constexpr (DTX3 || DTF8) {
            if (widex3) {
                using TX = Traits<D3R_F16X3>;
                const bool bias_i = !swap && p.bias != nullptr, bias_j = swap && p.bias != nullptr;
                const float* bsrc = p.bias ? p.bias : reinterpret_cast<const float*>(p.wgt);
                const bool heads = p.epi == EPI_HEADS;
                const bool has_tab = heads && p.rope_table != nullptr;
                const float* rtab = has_tab ? p.rope_table : reinterpret_cast<const float*>(p.wgt);
                const int r_ntok = has_tab ? p.ntok : 1, r_tokw = has_tab ? p.tok_w : 1, r_M = has_tab ? p.M : 1;
                // Folded LayerNorm (kernels.hpp GemmParams::ln_*): rstd_m (acc - mean_m s_n) + b'_n = fma(acc, R, fma(S, Nm, bias)) with, per element,
                // R = rstd of its ROW m, Nm = -mean rstd of its row, S = column sum of its COLUMN n. The four operands live in the registers the
                // bias alone used to take, plus 12:
                //   not swapped (m = j, n = i):  R = lnC[fj] (scalar), Nm = bjv[fj] (scalar), S = lnX[fl] (float4), bias = bq[fl] (float4)
                //   swapped, V^T (m = i, n = j): R = lnX[fl] (float4), Nm = bq[fl] (float4), S = lnC[fj] (scalar),  bias = bjv[fj] (scalar)
                // Without a fold the same two fmas run on (R, Nm, S) = (1, 0, 0) = acc + bias bit for bit; every load then reads the bias vector.
                const bool has_ln = DTX3 && p.ln_rstd != nullptr;
                const float* lnr = has_ln ? p.ln_rstd : bsrc;
                const float* lnn = has_ln ? p.ln_nmr : bsrc;
                const float* lns = has_ln ? p.ln_colsum : bsrc;
                const int ln_rmax = has_ln ? p.M - 1 : 0, ln_cmax = has_ln ? p.n_store - 1 : 0;
                const bool use_bj = swap ? bias_j : has_ln;     // bjv: the per-feature bias of a V^T tile, or the per-row -mean rstd
                float bjv[FJ], lnC[FJ];
#pragma unroll
                for (int fj = 0; fj < FJ; ++fj) {
                    const int j = jb + fj * 16 + jl;
                    const float* pb = swap ? bsrc + max(min(j, p.n_store - 1), 0) : lnn + max(min(j, ln_rmax), 0);
                    bjv[fj] = *pb;
                    if constexpr (DTX3) {
                        const float* pc = swap ? lns + max(min(j, ln_cmax), 0) : lnr + max(min(j, ln_rmax), 0);
                        lnC[fj] = *pc;
                    }
                }
#pragma unroll
                for (int g = 0; g < FI / 2; ++g) {
                    const int ig = ib + g * 32;                 // first i (column n, or token m when swapped) of this 32-wide group
                    int h = 0;
                    bool rope = false;
                    char* hdst = nullptr;
                    if (heads) {
                        const int nh = __builtin_amdgcn_readfirstlane(swap ? jb : ig);   // n of this (wave, group): inside one 64-wide head
                        const int region = nh / p.head_c;
                        h = (nh - region * p.head_c) >> 6;
                        rope = !swap && head_kind_of(p, region) == HEAD_ROPE;
                        hdst = reinterpret_cast<char*>(head_dst_of(p, region));
                    }
                    const int xhalf = (ig >> 5) & 1;            // RoPE: columns 0-31 of a head rotate with the y position, 32-63 with x
                    const bool use_bq = swap ? has_ln : bias_i; // bq: the per-column bias, or (V^T tiles) the -mean rstd of 4 consecutive tokens
                    float4 bq[2], lnX[2];
#pragma unroll
                    for (int fl = 0; fl < 2; ++fl) {
                        const int i0 = ig + fl * 16 + i4;
                        const float* pq = swap ? lnn + max(min(i0, ln_rmax - 3), 0) : bsrc + max(min(i0, p.n_store - 4), 0);
                        const float4 t = *reinterpret_cast<const float4*>(pq);
                        bq[fl] = make_float4(use_bq ? t.x : 0.f, use_bq ? t.y : 0.f, use_bq ? t.z : 0.f, use_bq ? t.w : 0.f);
                        if constexpr (DTX3) {
                            const float* px = swap ? lnr + max(min(i0, ln_rmax - 3), 0) : lns + max(min(i0, ln_cmax - 3), 0);
                            lnX[fl] = *reinterpret_cast<const float4*>(px);
                        }
                    }
                    float4 rt[2][2];   // RoPE table rows of this half of the head, double buffered across fragments
                    load_rope_half(rtab, r_ntok, r_tokw, r_M, jb + jl, i4, xhalf, rt[0]);
#pragma unroll
                    for (int fj = 0; fj < FJ; ++fj) {
                        const float bj = use_bj ? bjv[fj] : 0.f;
                        if (fj + 1 < FJ) load_rope_half(rtab, r_ntok, r_tokw, r_M, jb + (fj + 1) * 16 + jl, i4, xhalf, rt[(fj + 1) & 1]);
                        const float4 a0 = rt[fj & 1][0], a1 = rt[fj & 1][1];
                        const float cc[4] = {a0.x, a0.z, a1.x, a1.z}, ss[4] = {a0.y, a0.w, a1.y, a1.w};
                        float vals[2][4];
#pragma unroll
                        for (int fl = 0; fl < 2; ++fl) {
                            const f32x4_t a = acc[g * 2 + fl][fj];
                            if constexpr (DTX3) {
                                const float cj = has_ln ? lnC[fj] : (swap ? 0.f : 1.f);                 // S (swapped) or R
                                const float4 X = lnX[fl];
                                const float xd = swap ? 1.f : 0.f;                                      // no fold: R = 1 (swapped), S = 0
                                const float x0 = has_ln ? X.x : xd, x1 = has_ln ? X.y : xd, x2 = has_ln ? X.z : xd, x3 = has_ln ? X.w : xd;
                                const float4 q4 = bq[fl];
                                if (swap) {      // fma(acc, R_i, fma(S_j, Nm_i, bias_j))
                                    vals[fl][0] = __builtin_fmaf(a[0], x0, __builtin_fmaf(cj, q4.x, bj)); vals[fl][1] = __builtin_fmaf(a[1], x1, __builtin_fmaf(cj, q4.y, bj));
                                    vals[fl][2] = __builtin_fmaf(a[2], x2, __builtin_fmaf(cj, q4.z, bj)); vals[fl][3] = __builtin_fmaf(a[3], x3, __builtin_fmaf(cj, q4.w, bj));
                                } else {         // fma(acc, R_j, fma(S_i, Nm_j, bias_i))
                                    vals[fl][0] = __builtin_fmaf(a[0], cj, __builtin_fmaf(x0, bj, q4.x)); vals[fl][1] = __builtin_fmaf(a[1], cj, __builtin_fmaf(x1, bj, q4.y));
                                    vals[fl][2] = __builtin_fmaf(a[2], cj, __builtin_fmaf(x2, bj, q4.z)); vals[fl][3] = __builtin_fmaf(a[3], cj, __builtin_fmaf(x3, bj, q4.w));
                                }
                            } else {
                            const float4 bi = swap ? make_float4(bj, bj, bj, bj) : bq[fl];
                            vals[fl][0] = a[0] + bi.x; vals[fl][1] = a[1] + bi.y; vals[fl][2] = a[2] + bi.z; vals[fl][3] = a[3] + bi.w;
                            }
                        }
                        if (rope) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float u = vals[0][r], v = vals[1][r];
                                vals[0][r] = u * cc[r] - v * ss[r];
                                vals[1][r] = v * cc[r] + u * ss[r];
                            }
                        }
#pragma unroll
                        for (int fl = 0; fl < 2; ++fl) {
                            float v0 = vals[fl][0], v1 = vals[fl][1], v2 = vals[fl][2], v3 = vals[fl][3];
                            if (p.epi == EPI_GELU) gelu4<DT>(v0, v1, v2, v3);
                            if (p.flags & GF_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                            uint2 hh, ll;
                            TX::split2(v0, v1, hh.x, ll.x);
                            TX::split2(v2, v3, hh.y, ll.y);
                            const int c0 = fl * 16 + i4;       // 4 consecutive logical columns inside one 8-group
                            char* w = wreg + (fj * 16 + jl) * WROW + (c0 >> 3) * 32 + (c0 & 7) * 2;
                            *reinterpret_cast<uint2*>(w) = hh;
                            *reinterpret_cast<uint2*>(w + 16) = ll;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // ---- LDS -> global: 8 lanes x 16 B = the 128 bytes of 32 consecutive logical i of row j
#pragma unroll
                    for (int pass = 0; pass < RP; ++pass) {
                        const int row = pass * 8 + rrow;
                        const uint4 v = *reinterpret_cast<const uint4*>(wreg + row * WROW + rch * 16);
                        const int j = jb + row;
                        const int i8 = ig + (rch >> 1) * 8;    // first logical i of this 16-byte chunk (hi or lo half of an 8-group)
                        if (heads) {
                            if (!swap) {                        // q / k: [b][h][token][64]
                                if (j < p.M) {
                                    const int b = j / p.ntok, t = j - b * p.ntok;
                                    store16(hdst + (((size_t)(b * p.heads + h) * p.ntok + t) * 64 + xhalf * 32) * 4 + rch * 16, v, nt);
                                }
                            } else if (i8 < p.M && j < p.n_store) {   // v^T: [b][h][feature][ldv], 8 consecutive tokens per chunk
                                const int b = i8 / p.ntok, t = i8 - b * p.ntok;
                                const int dd = j & 63;
                                store16(hdst + (((size_t)(b * p.heads + h) * 64 + dd) * p.ldv + t) * 4 + (rch & 1) * 16, v, nt);
                            }
                        } else if (j < p.M && i8 < p.n_store) {
                            char* o = reinterpret_cast<char*>(p.out) + ((size_t)j * p.ldo + i8) * 4 + (rch & 1) * 16;
                            if (i8 + 8 <= p.n_store) store16(o, v, nt);
                            else *reinterpret_cast<uint2*>(o) = make_uint2(v.x, v.y);   // n_store % 8 == 4: the group's first 4 elements
                        }
                    }
                    asm volatile("" ::: "memory");
                }
                return;
            }
        }
        if (wide32) {
            const bool has_res = p.res1 != nullptr, has_bias = p.bias != nullptr;
            const float* bsrc = has_bias ? p.bias : reinterpret_cast<const float*>(p.wgt);
            // GF_X3RES (split-fp16 launches; the folded-LayerNorm engine, kernels.hpp): the residual stream lives in split-fp16 rows ONLY -- res1 is read
            // as typed rows (8 + 8 bytes per 4 elements), out2 receives the typed sum, no fp32 row is stored. The two forms are two instances of
            // one body so that no load feeds a PHI (see the note on conditional reads above).
            if constexpr (DTX3) {
                if (p.flags & GF_X3RES) {
                    using TXR = Traits<D3R_F16X3>;
                    // without a residual the same addresses of the (same shape) output are read and discarded: no branch, no PHI
                    const char* rsrc = reinterpret_cast<const char*>(has_res ? p.res1 : (const void*)p.out2);
                    const int rld = has_res ? p.ldr : p.ldo2;
                    // Lanes 2k / 2k + 1 of the read phase hold the two halves of one 8-element group [hi x8 (16 B)][lo x8 (16 B)]: the even lane reads /
                    // writes the group's 16 hi bytes, the odd lane its 16 lo bytes -- ONE 16-byte access per lane and direction (8-byte accesses are
                    // issue-bound, MI355X_MICROARCH.md store tail) -- and the halves are swapped with a DPP move. n_store % 8 == 0.
                    // ONE residual buffer: the row of group g + 1 is requested into the register its group-g value has just left (a whole group
                    // ahead of its use, like the double buffer of the fp32 form, at half the registers: this kernel has none to spare).
                    const bool odd = rch & 1;
                    const bool x3nt = p.x3res_nt != 0;       // probe D3R_GEMM_X3NT: the typed stream stored with the non-temporal policy (it is re-read by the next two launches)
                    auto swap_pair = [](uint32_t x) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true); };   // quad_perm [1,0,3,2]: lane ^ 1
                    auto request_row = [&](int g, int pass) __attribute__((always_inline)) -> uint4 {
                        const int m = min(jb + pass * 8 + rrow, p.M - 1);
                        const int n8 = max(min((ib + g * 32 + rch * 4) & ~7, p.n_store - 8), 0);
                        return *reinterpret_cast<const uint4*>(rsrc + TXR::boff((size_t)m * rld + n8) + (odd ? 16 : 0));
                    };
                    uint4 rr1[RP];
#pragma unroll
                    for (int pass = 0; pass < RP; ++pass) rr1[pass] = request_row(0, pass);
#pragma unroll
                    for (int g = 0; g < FI / 2; ++g) {
                        const int ig = ib + g * 32;
                        float4 bi2[2];
#pragma unroll
                        for (int fl = 0; fl < 2; ++fl) {
                            const float4 t = *reinterpret_cast<const float4*>(bsrc + max(min(ig + fl * 16 + i4, p.n_store - 4), 0));
                            bi2[fl] = make_float4(has_bias ? t.x : 0.f, has_bias ? t.y : 0.f, has_bias ? t.z : 0.f, has_bias ? t.w : 0.f);
                        }
#pragma unroll
                        for (int fl = 0; fl < 2; ++fl) {
                            const float4 bi = bi2[fl];
#pragma unroll
                            for (int fj = 0; fj < FJ; ++fj) {
                                const f32x4_t a = acc[g * 2 + fl][fj];
                                *reinterpret_cast<float4*>(wreg + (fj * 16 + jl) * WROW + (fl * 16 + i4) * 4) = make_float4(a[0] + bi.x, a[1] + bi.y, a[2] + bi.z, a[3] + bi.w);
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int pass = 0; pass < RP; ++pass) {
                            const int row = pass * 8 + rrow;
                            float4 v = *reinterpret_cast<const float4*>(wreg + row * WROW + rch * 16);
                            const int m = jb + row, n = ig + rch * 4;
                            const uint4 raw = rr1[pass];
                            if (g + 1 < FI / 2) rr1[pass] = request_row(g + 1, pass);
                            // even lane holds hi0..7 (keeps hi0..3, hands over hi4..7), odd lane lo0..7 (keeps lo4..7, hands over lo0..3)
                            const uint32_t t0 = swap_pair(odd ? raw.x : raw.z), t1 = swap_pair(odd ? raw.y : raw.w);
                            const uint32_t hx = odd ? t0 : raw.x, hy = odd ? t1 : raw.y, lx = odd ? raw.z : t0, ly = odd ? raw.w : t1;
                            v.x += has_res ? TXR::join_lo(hx, lx) : 0.f; v.y += has_res ? TXR::join_hi(hx, lx) : 0.f;
                            v.z += has_res ? TXR::join_lo(hy, ly) : 0.f; v.w += has_res ? TXR::join_hi(hy, ly) : 0.f;
                            uint2 h, l;
                            TXR::split2(v.x, v.y, h.x, l.x);
                            TXR::split2(v.z, v.w, h.y, l.y);
                            const uint32_t u0 = swap_pair(odd ? h.x : l.x), u1 = swap_pair(odd ? h.y : l.y);
                            if (m < p.M && n < p.n_store) {      // the typed sum: 16 hi bytes from the even lane, 16 lo bytes from the odd lane
                                char* o2 = reinterpret_cast<char*>(p.out2) + TXR::boff((size_t)m * p.ldo2 + (n & ~7)) + (odd ? 16 : 0);
                                store16(o2, odd ? make_uint4(u0, u1, l.x, l.y) : make_uint4(h.x, h.y, u0, u1), nt && x3nt);
                            }
                            if (p.ln_part) {    // folded LayerNorm: (sum, sum of squares) of the 32 values of row m in this column group (8 lanes x 4), one fixed tree
                                float sm, sq;
                                ln_quad_sums(v, sm, sq);
                                // the 8 lanes of a row (rch = lane & 7): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror -- VALU-side DPP moves, not the LDS crossbar
                                sm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm), 0xB1, 0xF, 0xF, false));
                                sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0xB1, 0xF, 0xF, false));
                                sm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm), 0x4E, 0xF, 0xF, false));
                                sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x4E, 0xF, 0xF, false));
                                sm += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sm), 0x141, 0xF, 0xF, false));
                                sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x141, 0xF, 0xF, false));
                                if (rch == 0 && m < p.M && n < p.n_store)
                                    *reinterpret_cast<float2*>(p.ln_part + ((size_t)m * (p.n_store >> 5) + (ig >> 5)) * 2) = make_float2(sm, sq);
                            }
                        }
                        asm volatile("" ::: "memory");
                    }
                    return;
                }
            }
            {
                // without a residual the same addresses of the (fp32, same shape) output are read and discarded: no branch, no PHI
                const float* rsrc = has_res ? reinterpret_cast<const float*>(p.res1) : reinterpret_cast<const float*>(p.out);
                const int rld = has_res ? p.ldr : p.ldo;
                // The residual rows a lane adds in the read phase of group g are requested one group AHEAD (double buffered): the per-block
                // trace (tools/gpu_probe.py gemmtrace) showed this epilogue as four serial HBM round trips, ~19 us per 256 x 256 tile.
                // In place (res1 == out) is fine: a lane reads exactly the elements it stores later, and group g + 1's columns are
                // disjoint from the columns group g is storing.
                float4 rr[2][RP];
                auto request_rows = [&](int g, float4 (&dst)[RP]) __attribute__((always_inline)) {
                    const int ig = ib + g * 32;
#pragma unroll
                    for (int pass = 0; pass < RP; ++pass) {
                        const int m = min(jb + pass * 8 + rrow, p.M - 1), n = max(min(ig + rch * 4, p.n_store - 4), 0);
                        dst[pass] = *reinterpret_cast<const float4*>(rsrc + (size_t)m * rld + n);
                    }
                };
                request_rows(0, rr[0]);
#pragma unroll
                for (int g = 0; g < FI / 2; ++g) {
                    const int ig = ib + g * 32;
                    if (g + 1 < FI / 2) request_rows(g + 1, rr[(g + 1) & 1]);
                    float4 bi2[2];
#pragma unroll
                    for (int fl = 0; fl < 2; ++fl) {
                        const float4 t = *reinterpret_cast<const float4*>(bsrc + max(min(ig + fl * 16 + i4, p.n_store - 4), 0));
                        bi2[fl] = make_float4(has_bias ? t.x : 0.f, has_bias ? t.y : 0.f, has_bias ? t.z : 0.f, has_bias ? t.w : 0.f);
                    }
#pragma unroll
                    for (int fl = 0; fl < 2; ++fl) {
                        const float4 bi = bi2[fl];
#pragma unroll
                        for (int fj = 0; fj < FJ; ++fj) {
                            const f32x4_t a = acc[g * 2 + fl][fj];
                            *reinterpret_cast<float4*>(wreg + (fj * 16 + jl) * WROW + (fl * 16 + i4) * 4) = make_float4(a[0] + bi.x, a[1] + bi.y, a[2] + bi.z, a[3] + bi.w);
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int pass = 0; pass < RP; ++pass) {
                        const int row = pass * 8 + rrow;
                        float4 v = *reinterpret_cast<const float4*>(wreg + row * WROW + rch * 16);
                        const int m = jb + row, n = ig + rch * 4;
                        const float4 rv = rr[g & 1][pass];
                        v.x += has_res ? rv.x : 0.f; v.y += has_res ? rv.y : 0.f;
                        v.z += has_res ? rv.z : 0.f; v.w += has_res ? rv.w : 0.f;
                        if (m < p.M && n < p.n_store) {
                            store16(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)), nt);
                            if (p.out2) store4<O2DT>(p.out2, (size_t)m * p.ldo2 + n, v.x, v.y, v.z, v.w);
                        }
                    }
                    asm volatile("" ::: "memory");
                }
            }
            return;
        }
    }

    if (!swap) {
        const int nb = n0 + wi * (FI * 16), mb = m0 + wj * (FJ * 16);
        if (p.epi == EPI_HEADS) {
            // q / k projections: bias, 2-D RoPE on the fp32 accumulator, head-major store. A wave's n range is walked in 32-column halves
            // of a head (columns 0-31 rotate with the token's y position, 32-63 with x; pairs are (c, c + 16) inside a half).
#pragma unroll
            for (int hb = 0; hb < FI / 2; ++hb) {
                const int nh = nb + hb * 32;
                if (nh >= p.n_store) continue;
                const int region = nh / p.head_c;
                const int h = (nh - region * p.head_c) >> 6, half = (nh >> 5) & 1;
                void* dst = p.head_dst[region];
                const bool rope = p.head_kind[region] == HEAD_ROPE;
                const float4 bu = p.bias ? *reinterpret_cast<const float4*>(p.bias + nh + i4) : make_float4(0, 0, 0, 0);
                const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + nh + 16 + i4) : make_float4(0, 0, 0, 0);
#pragma unroll
                for (int fj = 0; fj < FJ; ++fj) {
                    const int m = mb + fj * 16 + jl;
                    if (m >= p.M) continue;
                    const int b = m / p.ntok, t = m - b * p.ntok;
                    const int ty = t / p.tok_w, tx = t - ty * p.tok_w;
                    const size_t obase = ((size_t)(b * p.heads + h) * p.ntok + t) * 64;
                    const f32x4_t u = acc[hb * 2][fj], v = acc[hb * 2 + 1][fj];
                    float uu[4] = {u[0] + bu.x, u[1] + bu.y, u[2] + bu.z, u[3] + bu.w};
                    float vv[4] = {v[0] + bv.x, v[1] + bv.y, v[2] + bv.z, v[3] + bv.w};
                    if constexpr (DTX3) {
                        if (p.ln_rstd) {     // folded LayerNorm (kernels.hpp): rstd_m (acc - mean_m s_n) + b'_n
                            const float rs = p.ln_rstd[m], nm = p.ln_nmr[m];
                            const float4 su = *reinterpret_cast<const float4*>(p.ln_colsum + nh + i4), sv = *reinterpret_cast<const float4*>(p.ln_colsum + nh + 16 + i4);
                            uu[0] = __builtin_fmaf(u[0], rs, __builtin_fmaf(su.x, nm, bu.x)); uu[1] = __builtin_fmaf(u[1], rs, __builtin_fmaf(su.y, nm, bu.y));
                            uu[2] = __builtin_fmaf(u[2], rs, __builtin_fmaf(su.z, nm, bu.z)); uu[3] = __builtin_fmaf(u[3], rs, __builtin_fmaf(su.w, nm, bu.w));
                            vv[0] = __builtin_fmaf(v[0], rs, __builtin_fmaf(sv.x, nm, bv.x)); vv[1] = __builtin_fmaf(v[1], rs, __builtin_fmaf(sv.y, nm, bv.y));
                            vv[2] = __builtin_fmaf(v[2], rs, __builtin_fmaf(sv.z, nm, bv.z)); vv[3] = __builtin_fmaf(v[3], rs, __builtin_fmaf(sv.w, nm, bv.w));
                        }
                    }
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
                    store4<HDT>(dst, obase + half * 32 + i4, uu[0], uu[1], uu[2], uu[3]);
                    store4<HDT>(dst, obase + half * 32 + 16 + i4, vv[0], vv[1], vv[2], vv[3]);
                }
            }
            return;
        }
#pragma unroll
        for (int fi = 0; fi < FI; ++fi) {
            const int n = nb + fi * 16 + i4;
            if (n >= p.n_store) continue;
            const float4 bias = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int fj = 0; fj < FJ; ++fj) {
                const int m = mb + fj * 16 + jl;
                if (m >= p.M) continue;
                const f32x4_t a = acc[fi][fj];
                float v0 = a[0] + bias.x, v1 = a[1] + bias.y, v2 = a[2] + bias.z, v3 = a[3] + bias.w;
                if constexpr (DTX3) {
                    if (p.ln_rstd) {         // folded LayerNorm (kernels.hpp): rstd_m (acc - mean_m s_n) + b'_n
                        const float rs = p.ln_rstd[m], nm = p.ln_nmr[m];
                        const float4 s4 = *reinterpret_cast<const float4*>(p.ln_colsum + n);
                        v0 = __builtin_fmaf(a[0], rs, __builtin_fmaf(s4.x, nm, bias.x)); v1 = __builtin_fmaf(a[1], rs, __builtin_fmaf(s4.y, nm, bias.y));
                        v2 = __builtin_fmaf(a[2], rs, __builtin_fmaf(s4.z, nm, bias.z)); v3 = __builtin_fmaf(a[3], rs, __builtin_fmaf(s4.w, nm, bias.w));
                    }
                }
                if constexpr (DTF8) {
                    // fp16 + fp8 launches come with EPI_F32 (+ residual, + split-fp16 typed copy) or GELU / plain activation rows only
                    // (launch_gemm checks); a lean body keeps these fully unrolled loops under the unroller's size cap -- past it the
                    // loops stay rolled, acc is indexed dynamically and the whole accumulator array moves to scratch
                    if (p.epi == EPI_F32) {
                        float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n;
                        if (p.res1) {
                            const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res1) + (size_t)m * p.ldr + n);
                            v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                        }
                        *reinterpret_cast<float4*>(o) = make_float4(v0, v1, v2, v3);
                        if (p.out2) store4<O2DT>(p.out2, (size_t)m * p.ldo2 + n, v0, v1, v2, v3);
                    } else {
                        if (p.epi == EPI_GELU) gelu4<DT>(v0, v1, v2, v3);
                        if (p.flags & GF_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        store4<DT>(p.out, (size_t)m * p.ldo + n, v0, v1, v2, v3);
                    }
                    continue;
                }
                switch (p.epi) {
                    case EPI_F32: {
                        float* o = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n;
                        if (p.res1) {
                            const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res1) + (size_t)m * p.ldr + n);
                            v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                        }
                        *reinterpret_cast<float4*>(o) = make_float4(v0, v1, v2, v3);
                        if (p.out2) store4<O2DT>(p.out2, (size_t)m * p.ldo2 + n, v0, v1, v2, v3);
                    } break;
                    case EPI_GELU:
                        gelu4<DT>(v0, v1, v2, v3);
                        store4<DT>(p.out, (size_t)m * p.ldo + n, v0, v1, v2, v3);
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
        // V^T tiles: lane owns 4 consecutive tokens (i = m) of one feature (j = n); a wave's n range (FJ*16 = 64 or 32) lies inside one head
        const int mb = m0 + wi * (FI * 16), nb = n0 + wj * (FJ * 16);
        if (nb >= p.n_store) return;
        const int region = nb / p.head_c;
        const int h = (nb - region * p.head_c) >> 6;
        void* dst = p.head_dst[region];
#pragma unroll
        for (int fj = 0; fj < FJ; ++fj) {
            const int dd = (nb & 63) + fj * 16 + jl;
            const float bias = p.bias ? p.bias[nb + fj * 16 + jl] : 0.f;
            float lns_j = 0.f;
            if constexpr (DTX3) lns_j = p.ln_rstd ? p.ln_colsum[nb + fj * 16 + jl] : 0.f;
#pragma unroll
            for (int fi = 0; fi < FI; ++fi) {
                const int m = mb + fi * 16 + i4;
                if (m >= p.M) continue;
                f32x4_t a = acc[fi][fj];
                float badd = bias;
                if constexpr (DTX3) {
                    if (p.ln_rstd) {         // folded LayerNorm (kernels.hpp), the 4 consecutive tokens of this lane: fma(acc, rstd_m, fma(s_n, nmr_m, b'_n)), the wide route's expression
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int mm = min(m + r, p.M - 1);
                            a[r] = __builtin_fmaf(a[r], p.ln_rstd[mm], __builtin_fmaf(lns_j, p.ln_nmr[mm], bias));
                        }
                        badd = 0.f;
                    }
                }
                const int b = m / p.ntok, t = m - b * p.ntok;
                const size_t rowbase = ((size_t)(b * p.heads + h) * 64 + dd) * p.ldv;
                if (t + 3 < p.ntok && ((p.ldv | t) & 3) == 0) {
                    store4<HDT>(dst, rowbase + t, a[0] + badd, a[1] + badd, a[2] + badd, a[3] + badd);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int mm = m + r;
                        if (mm >= p.M) break;
                        const int bb = mm / p.ntok, tt = mm - bb * p.ntok;
                        store1<HDT>(dst, ((size_t)(bb * p.heads + h) * 64 + dd) * p.ldv + tt, a[r] + badd);
                    }
                }
            }
        }
    }
}

struct DevInfo { int cus; int wall_khz; };
static DevInfo dev_info() {
    DevInfo d{256, 100000};
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) d.cus = v;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, dev) == hipSuccess && v > 0) d.wall_khz = v;
    }
    return d;
}

// ---- host side: configuration choice + launch ------------------------------------------------------------------
template <int DT, class CF> static hipError_t launch_cfg(const GemmParams& p, hipStream_t s) {
    // the dynamic-LDS limit is a per-device function attribute: raise it once on every device this process launches on
    static std::atomic<unsigned long long> attr_done{0};
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    const unsigned long long dev_bit = 1ull << (dev_id & 63);
    if (!(attr_done.load(std::memory_order_relaxed) & dev_bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<DT, CF>), hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS);
        attr_done.fetch_or(dev_bit, std::memory_order_relaxed);
    }
    const int grid = cdiv(p.M, CF::BM) * cdiv(p.n_store, CF::BN) * (p.splitk > 1 ? p.splitk : 1);
    GemmParams q = p;
    {   // first-round stagger (see the kernel): spread = factor x (epilogue bytes of the resident tiles / ~4.5 TB/s)
        const char* e_st = probe_env("D3R_GEMM_STAGGER");       // default off: measured no gain (profiles/README.md), costs half a burst per launch; read per launch (probes toggle it)
        const float factor = e_st ? (float)atof(e_st) : 0.0f;
        const char* e_sm = probe_env("D3R_GEMM_STAGGER_MODE");
        const int mode = e_sm ? atoi(e_sm) : 0;
        static const DevInfo dev = dev_info();
        const int resident = dev.cus * (CF::LDS * 2 <= 160 * 1024 ? 2 : 1);
        if (factor > 0.f && grid > resident && !(p.flags & GF_NOSTORE)) {
            const double eb_out = p.epi == EPI_F32 ? 4.0 : (double)Traits<DT>::EB;
            double bytes = (double)CF::BM * CF::BN * eb_out;
            if (p.res1) bytes += (double)CF::BM * CF::BN * eb_out;
            if (p.res2) bytes += (double)CF::BM * CF::BN * eb_out;
            if (p.out2) bytes += (double)CF::BM * CF::BN * Traits<DT>::EB;
            const double burst_s = bytes * resident / 4.5e12;
            q.stagger_ticks = (int)(burst_s * factor * dev.wall_khz * 1e3);
            q.first_round = resident;
            q.stagger_mode = mode;
        }
    }
    hipLaunchKernelGGL((gemm_kernel<DT, CF>), dim3(grid), dim3(CF::NT), CF::LDS, s, q);
    return hipGetLastError();
}

// (A persistent variant -- resident blocks walking several tiles and prefetching the next tile's first K step across the
// epilogue -- was measured 5-10 % slower on MI355X, profiles/r01_call4: on gfx9 vmcnt also counts the epilogue's stores, so
// the next tile's first wait drains them, which a fresh block never does. One block per tile it is.)
// Tile configuration choice, from measurements on MI355X (tools/gpu_probe.py gemm, profiles/): the 256x256 tile wins
// once there are at least ~3 full rounds of 256 resident blocks (M = 49152: 760-1070 vs 640-830 TF/s), the 128x128 tile
// (2 blocks per CU: one block's prologue / epilogue hides behind the other's K loop) wins below that; N <= 128 problems
// (DPT head convolutions) take the 512x128 / 256x128 tiles so that no half-empty 256-wide tile is computed.
static int pick_config_raw(const GemmParams& p, int dt);
// compute units of the current device, cached per device id (the whole-rounds rules below are about THIS chip's CU count, not a constant)
static int device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int slot = dev & 63;
    int v = cache[slot].load(std::memory_order_relaxed);
    if (v <= 0) {
        int q = 0;
        v = (hipDeviceGetAttribute(&q, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && q > 0) ? q : 256;
        cache[slot].store(v, std::memory_order_relaxed);
    }
    return v;
}
// the tile configuration a launch of (p, dt) RUNS on -- what the engine's profile records and d3r_gemm_tile_config reports: the heuristic's
// choice, then the remaps launch_t applies for operand types that do not have every shape
static int device_cus();
// D3R_GEMM_CFG=0..9: the tile configuration pinned from the environment (parity tests, A/B runs); read per call, -1 = not set
static int env_forced_cfg() {
    const char* e = getenv("D3R_GEMM_CFG");
    if (!e || e[0] < '0' || e[0] > '9') return -1;
    if (e[1] == 0) return e[0] - '0';
    return (e[0] == '1' && e[1] == '1' && e[2] == 0) ? GEMM_CFG_96x64 : -1;
}
int gemm_p4_mode() {
    const char* e = getenv("D3R_GEMM_PERSIST");
    return e ? (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : -1)) : -1;
}
// persistent kernel (gemm_p4.hip) for this launch? Measured on MI355X inside the 32-pair forward (tools/launch_table.py, profiles/r06_*): it wins where the
// epilogue is a large share of a tile's life and the tiles fill whole rounds of the CUs -- fc1 + GELU 395 -> 427 (encoder) / 363 -> 398 TFLOP/s (decoder),
// the typed-residual projections at K <= 1024 348 -> 362, typed stores +4..12 % -- ties at K = 4096 (fc2: the K loop dominates; the one-tile-per-block
// kernels keep it) and loses where its 256 x 128 tiles leave the last round of CUs mostly empty (the decoder's N = 768 GEMMs: 576 tiles = 2.25 rounds).
static bool use_p4(const GemmParams& p, int dt) {
    if (dt != D3R_F16X3 || p.force_cfg >= 0 || env_forced_cfg() >= 0) return false;
    const int mode = gemm_p4_mode();
    if (mode == 0 || !gemm_p4_eligible(p, dt)) return false;
    if (mode == 1) return true;
    const long tiles = (long)(p.M / 256) * (p.n_store / 128);
    const int cus = device_cus();
    const long rounds = (tiles + cus - 1) / cus;
    if (tiles < cus || tiles * 100 < rounds * cus * 88) return false;      // the last round at least ~half full on average: >= 88 % of the tile slots used
    if (p.epi == EPI_F32 && p.K > 1024) return false;
    return true;
}
int gemm_pick_config(const GemmParams& p, int dt) {
    if (use_p4(p, dt)) return GEMM_CFG_P4;
    int cfg = pick_config_raw(p, dt);
    const bool split = dt == D3R_F16X3 || dt == D3R_F16F8 || dt == D3R_F16X2F8;
    if (cfg == GEMM_CFG_256x128W4 && split) cfg = GEMM_CFG_256x128;
    if ((cfg == GEMM_CFG_256x128R || cfg == GEMM_CFG_64 || cfg == GEMM_CFG_96x64 || cfg == GEMM_CFG_384x192) && dt != D3R_F16X3) cfg = cfg == GEMM_CFG_256x128R ? GEMM_CFG_256x128 : GEMM_CFG_128;
    if ((cfg == GEMM_CFG_256S4 || cfg == GEMM_CFG_256PP) && split) cfg = GEMM_CFG_256;
    if (dt == D3R_F16X2F8 && cfg != GEMM_CFG_256) cfg = GEMM_CFG_128;      // the 2.5-unit K loop exists on the two square tiles
    return cfg;
}
static int pick_config_raw(const GemmParams& p, int dt) {
    const int n_rows = p.n_rows > 0 ? p.n_rows : p.n_pad;
    const bool wide_vt = (dt == D3R_BF16 || dt == D3R_F16) && (p.ntok & 63) == 0 && !(p.flags & GF_NOWIDE);
    const bool heads = p.epi == EPI_HEADS && !wide_vt;   // "heads" = needs a square tile (operand-role swap for V^T)
    const bool ok256 = cdiv(p.n_store, 256) * 256 <= n_rows && (p.epi != EPI_HEADS || p.head_c % 256 == 0);
    int forced = p.force_cfg;
    if (forced < 0) {   // D3R_GEMM_CFG=0|1|2|3 pins the tile configuration (parity tests, probes); infeasible choices are ignored
        forced = env_forced_cfg();
    }
    if (forced == GEMM_CFG_384x192 && dt == D3R_F16X3 && !heads && p.epi != EPI_HEADS && p.epi != EPI_HEAD4 && p.amode == AMODE_LINEAR && cdiv(p.n_store, 192) * 192 <= n_rows) return forced;
    if (forced == GEMM_CFG_96x64 && dt == D3R_F16X3 && !heads && p.epi != EPI_HEADS && p.epi != EPI_HEAD4) return forced;
    if (forced == GEMM_CFG_128 || forced == GEMM_CFG_64 || (forced == GEMM_CFG_256 && ok256) ||
        ((forced == GEMM_CFG_256x128 || forced == GEMM_CFG_512x128 || forced == GEMM_CFG_256x128W4 || forced == GEMM_CFG_256x128R) && !heads) || ((forced == GEMM_CFG_256S4 || forced == GEMM_CFG_256PP) && ok256))
        return forced;
    if (p.epi == EPI_F32 && p.K <= 1024) {   // probe: tile of the HBM-heavy residual-stream epilogues at short K (D3R_GEMM_F32CFG=0|2|4)
        if (const char* e = probe_env("D3R_GEMM_F32CFG"))
            if ((e[0] == '0' || e[0] == '2' || e[0] == '4') && e[1] == 0) return e[0] - '0';
    }
    if (!heads && p.n_store <= 128) {
        if (cdiv(p.M, 512) >= 512) return GEMM_CFG_512x128;
        if (cdiv(p.M, 256) >= 512) return GEMM_CFG_256x128;
        return GEMM_CFG_128;
    }
    // split-fp16, nn.Linear launches without attention heads whose (M / 384) x (N / 192) tiles fill whole rounds of the 256 CUs: the decoder's 24576-row
    // GEMMs at 32 pairs per step (N = 768 / 3072 -> 256 / 1024 tiles; on the 128 x 128 tile 1152 tiles = 2.25 rounds of the 512 slots). Measured
    // (profiles/r04_i, r04_k). In isolation, us per launch, default tile -> this one: fc2 24576 x 768 x 3072 + residual 389 -> 280, fc1 24576 x 3072 x 768
    // + GELU 384 -> 364, plain stores 108 -> 88 / 222 -> 161 / 286 -> 224 (N = 768 / 1536 / 2304), the fp32-residual projections at K = 768 90 -> 94.
    // On the forward the isolated gains mostly vanish (in the network the default tiles run faster than back to back on one shape, and the two decoder
    // sides overlap their tails on the two streams): every eligible launch on it 194.55 -> 196.75 and 191.7 -> 192.7 pairs/s (two boxes, every run above every
    // baseline run); WITHOUT the K <= 1024 projections 191.7 -> 190.9 -- so the rule is "every eligible launch". D3R_GEMM_T384=0: never; =1: not the
    // fp32-residual projections at K <= 1024 (the probe of that A/B).
    if (dt == D3R_F16X3 && !heads && p.epi != EPI_HEADS && p.epi != EPI_HEAD4 && p.amode == AMODE_LINEAR && p.n_store % 192 == 0 && cdiv(p.n_store, 192) * 192 <= n_rows) {
        const char* e384 = probe_env("D3R_GEMM_T384");
        const long t384 = (long)cdiv(p.M, 384) * cdiv(p.n_store, 192);
        const bool short_res = p.epi == EPI_F32 && p.K <= 1024;
        const int cus = device_cus();       // whole rounds of THIS device's compute units (256 on MI355X; a partitioned device has fewer)
        if (!(e384 && e384[0] == '0') && !(short_res && e384 && e384[0] == '1') && t384 >= cus * 9 / 10 && (t384 % cus == 0 || t384 % cus >= cus * 9 / 10)) return GEMM_CFG_384x192;
    }
    // split-fp16, launches without attention heads (their V^T regions need a square tile): the two-blocks-per-CU 256 x 128 shape with the
    // weights of a K step in registers. Measured on MI355X (profiles/r03_c/gemmtrace_cfg7.log, bench_r*.log): it wins where the epilogue
    // is an HBM burst that the other block's MFMAs can run under -- the fp32-residual projections at K <= 1024 (proj 49152 x 1024 x
    // 1024: 337 vs 296 TFLOP/s) -- and loses 1-6 % where the epilogue is VALU work (GELU, typed stores: on gfx950 VALU and MFMA of a
    // SIMD overlap only by half, tools/issue_probe.hip) or the K loop dominates (fc2): default = those projections only.
    // D3R_GEMM_R=0: never; =1: every eligible launch (the A/B of the round).
    if (dt == D3R_F16X3 && !heads && p.epi != EPI_HEADS && p.n_store > 128) {
        const char* e_r = probe_env("D3R_GEMM_R");      // read per call, like the other probes (tests move it with monkeypatch)
        const int r_on = e_r ? atoi(e_r) : -1;
        const long tiles = (long)cdiv(p.M, 256) * cdiv(p.n_store, 128);
        if (r_on == 1 && tiles >= 512) return GEMM_CFG_256x128R;
        if (r_on < 0 && p.epi == EPI_F32 && p.res1 != nullptr && p.K <= 1024 && p.amode == AMODE_LINEAR && tiles >= 1024) return GEMM_CFG_256x128R;
    }
    // split-fp16, small problems (the one- and two-pair forwards of dust3r/demo.py:156 / visloc.py:88, batch_size = 1): below ~1.5
    // 128 x 128 tiles per CU the chip is not full -- the 64 x 64 tile by four waves of 32 x 32 runs the same problem on four times the
    // waves. D3R_GEMM_T64 moves the crossover (0 = never); measured at 200 / 400 / 600 / 1000 / 2000 for 1-8 pairs per call: 200 is the
    // best or within noise of it everywhere (profiles/r03_f/latency_small_tiles.log).
    if (dt == D3R_F16X3 && p.n_store > 128) {
        long t64 = 200;
        if (const char* e = probe_env("D3R_GEMM_T64")) t64 = atol(e);
        if ((long)cdiv(p.M, 128) * cdiv(p.n_store, 128) < t64) {
            // round 6: M 96 x N 64 instead, where its tiles come to 1.5 ... 2 per CU. The small tiles run at the rate their operand rows arrive: the 96 x 64
            // shape moves 17 % fewer bytes per flop, but a CU needs two or three resident blocks' worth of loads in flight to keep that rate -- measured on MI355X
            // (tools/tile_probe.py, profiles/r06_f; 64 x 64 -> 96 x 64, us per launch): 3072 x 1024 x 4096 81 -> 71, 2304 x 1024 x 4096 75 -> 61, 3072 x 768 x 3072 66 -> 53,
            // 1536 x 1536 x 768 20 -> 18, 768 x 3072 x 768 20 -> 18 (384 or 512 tiles); 1536 x 1024 x 4096 56 -> 61 (256 tiles: one block per CU), 1152 x 768 x 3072 33 -> 40 (144),
            // 1536 x 2304 x 768 28 -> 30 (576). nn.Linear operands without attention heads only. INSIDE the forward (tools/env_latency_ab.py, same process, rule on | off) the
            // K = 768 / 1024 cases do not carry over -- one pair 10.04 vs 9.96 ms (the decoder's two sides run side by side: twice the tiles in flight), two pairs 14.49 vs
            // 14.68 -- so the rule keeps the long K loops only (K >= 2048: fc2 of the encoder at two / three pairs, of the decoder at four).
            const long t96 = (long)cdiv(p.M, 96) * cdiv(p.n_store, 64);
            const int cus = device_cus();
            const char* e96 = probe_env("D3R_GEMM_T96");        // probe builds: 0 = never
            if (!(e96 && e96[0] == '0') && !heads && p.epi != EPI_HEADS && p.epi != EPI_HEAD4 && p.amode == AMODE_LINEAR && p.K >= 2048 && t96 * 2 >= (long)cus * 3 && t96 <= (long)cus * 2) return GEMM_CFG_96x64;
            return GEMM_CFG_64;
        }
    }
    const long tiles256 = (long)cdiv(p.M, 256) * cdiv(p.n_store, 256);
    // fp16 + fp8 rows: the 256-wide tile is 1.3-1.6x ahead of the 128 x 128 one per tile (its K loop lost a third of its MFMA work, the small
    // tile's LDS read traffic per MFMA is twice as high), so it pays from a single round of resident blocks on: measured +1.5 % on the
    // forward with the decoder's 24576 x 768 GEMMs (288 tiles, two such launches side by side on the two streams) on it
    long t256 = (dt == D3R_F16F8 || dt == D3R_F16X2F8) ? 250 : 700;     // probes: D3R_GEMM_T256 moves the 256x256 / 128x128 crossover, D3R_GEMM_MID=2 sends the shapes below it to 256x128
    if (const char* e = probe_env("D3R_GEMM_T256")) t256 = atol(e);
    if (ok256 && tiles256 >= t256) {
        // nn.Linear operands: the ping-pong schedule measured 1-8 % ahead of the plain 2-stage loop (profiles/r01_call13);
        // implicit-GEMM operands: behind it (the per-tap address arithmetic sits in the load segment) -> plain loop
        const char* e_pp = probe_env("D3R_GEMM_PP");
        const bool pp = e_pp ? e_pp[0] == '1' : false;
        return (pp && p.amode == AMODE_LINEAR) ? GEMM_CFG_256PP : GEMM_CFG_256;
    }
    if (const char* e = probe_env("D3R_GEMM_MID")) if (e[0] == '2' && !heads) return GEMM_CFG_256x128;
    return GEMM_CFG_128;
}

template <int DT> static hipError_t launch_t(const GemmParams& p, hipStream_t s) {
    int cfg = gemm_pick_config(p, DT);
    constexpr bool SPLIT = (DT == D3R_F16X3 || DT == D3R_F16F8 || DT == D3R_F16X2F8);   // split-fp16 and fp16 + fp8 rows need 128-byte K rows, two stages
    if (cfg == GEMM_CFG_256x128W4 && SPLIT) cfg = GEMM_CFG_256x128;
    if (cfg == GEMM_CFG_256x128R && DT != D3R_F16X3) cfg = GEMM_CFG_256x128;      // the weights-in-registers shape exists for split-fp16 only
    if ((cfg == GEMM_CFG_64 || cfg == GEMM_CFG_96x64) && DT != D3R_F16X3) cfg = GEMM_CFG_128;                // the 64 x 64 / 96 x 64 shapes too
    if (cfg == GEMM_CFG_384x192 && DT != D3R_F16X3) cfg = GEMM_CFG_128;           // and the 384 x 192 one
    // the fused head tail needs a wave to hold every output channel of its rows: waves stacked along m, 128 columns per wave
    if (p.epi == EPI_HEAD4 && (DT != D3R_F16X3 || !(cfg == GEMM_CFG_512x128 || cfg == GEMM_CFG_256x128R))) return hipErrorInvalidValue;
    if ((cfg == GEMM_CFG_256S4 || cfg == GEMM_CFG_256PP) && (SPLIT || !kProbes)) cfg = GEMM_CFG_256;      // the four-wave 256 x 128, four-stage and ping-pong shapes never won a
    if (cfg == GEMM_CFG_256x128W4 && !kProbes) cfg = GEMM_CFG_256x128;                                       // default (profiles/r01_*): compiled in probe builds only (-DD3R_PROBES)
    // the ping-pong schedule has no operand-role swap: attention projections only through the wide V^T route
    if (cfg == GEMM_CFG_256PP && p.epi == EPI_HEADS && !((DT == D3R_BF16 || DT == D3R_F16) && (p.ntok & 63) == 0 && !(p.flags & GF_NOWIDE))) cfg = GEMM_CFG_256;
    if constexpr (!SPLIT && kProbes) {
        if (cfg == GEMM_CFG_256x128W4) return launch_cfg<DT, Cfg256x128w4>(p, s);
        if (cfg == GEMM_CFG_256S4) return launch_cfg<DT, Cfg256s4>(p, s);
        if (cfg == GEMM_CFG_256PP) return launch_cfg<DT, Cfg256pp>(p, s);
    }
    if constexpr (kProbes && DT != D3R_F16F8 && DT != D3R_F16X2F8) {
        const char* e_a3 = probe_env("D3R_GEMM_A3");
        const bool a3 = e_a3 ? e_a3[0] != '0' : false;
        if (a3 && cfg == GEMM_CFG_256) return launch_cfg<DT, Cfg256a3>(p, s);
    }
    if constexpr (DT == D3R_F16X3) {
        if (cfg == GEMM_CFG_384x192) return launch_cfg<DT, Cfg384x192>(p, s);
        if (cfg == GEMM_CFG_256x128R) return launch_cfg<DT, Cfg256x128r>(p, s);
        if (cfg == GEMM_CFG_96x64) return launch_cfg<DT, Cfg96x64>(p, s);
        if (cfg == GEMM_CFG_64) {
            const char* e_ns = probe_env("D3R_GEMM_64NS");     // probe: LDS ring depth of the 64 x 64 tile (2 | 3 | 4)
            const int ns = e_ns ? atoi(e_ns) : 3;             // measured (profiles/r03_f): one pair 14.56 ms on the 128 x 128 tile, 12.36 / 10.38 / 10.48 ms with 2 / 3 / 4 slots
            if constexpr (kProbes) {
                if (ns == 2) return launch_cfg<DT, Cfg64>(p, s);
                if (ns == 4) return launch_cfg<DT, Cfg64s4>(p, s);
            }
            (void)ns;
            return launch_cfg<DT, Cfg64s3>(p, s);
        }
        // split-fp16: D3R_GEMM_X3SW=1 selects the software-pipelined K loop. Measured on MI355X (profiles/r02_*): equal to the plain
        // two-stage loop on the 256-wide tiles, 10-15 % behind on the 128 x 128 tile -- the K loop is not where the time goes (the
        // same launches without their epilogue run 30 % faster in either form), so the plain loop stays the default.
        // D3R_GEMM_X3IL=1 / 0: nn.Linear launches on the 256 x 256 tile with the DMA pieces interleaved with the MFMA rows (bit-identical)
        if constexpr (kProbes) {
        const char* e_il = probe_env("D3R_GEMM_X3IL");
        const bool x3il = e_il ? e_il[0] == '1' : false;
        if (x3il && cfg == GEMM_CFG_256 && p.amode == AMODE_LINEAR && (size_t)512 * p.lda * 4 < (1ull << 32) && (size_t)512 * p.K * 4 < (1ull << 32)) return launch_cfg<DT, Cfg256il>(p, s);
        const char* e_sw = probe_env("D3R_GEMM_X3SW");
        const bool sw = e_sw ? e_sw[0] == '1' : false;
        if (sw) {
            switch (cfg) {
                case GEMM_CFG_256: return launch_cfg<DT, Cfg256sw>(p, s);
                case GEMM_CFG_256x128: return launch_cfg<DT, Cfg256x128sw>(p, s);
                case GEMM_CFG_512x128: return launch_cfg<DT, Cfg512x128sw>(p, s);
                default: return launch_cfg<DT, Cfg128sw>(p, s);
            }
        }
        }
    }
    if constexpr (DT == D3R_F16F8) {
        // two blocks per CU (256 x 128 tile, four waves, 64-byte K steps): one block's K loop under the other's epilogue. The attention
        // projections keep the square tile (their V^T regions swap the MFMA operand roles). D3R_GEMM_F8W4=1 / 0.
        const char* e_w4 = probe_env("D3R_GEMM_F8W4");
        const int w4 = e_w4 ? (e_w4[0] == '1' ? 1 : 0) : 0;
        if constexpr (kProbes) { if (w4 == 1 && cfg == GEMM_CFG_256 && p.epi != EPI_HEADS && p.n_store % 128 == 0) return launch_cfg<DT, Cfg256x128f8>(p, s); }
        (void)w4;
        // DMA pieces interleaved with the MFMA rows (PP = 4). Measured on MI355X (profiles/r02_f8/bench_f8_il.log): +3 % on the 256-wide
        // tiles (one block per CU: behind the barrier neither wave of a SIMD has MFMAs to issue), -2.5 % on the 128 x 128 tile (the
        // CU's second block already fills that gap). D3R_GEMM_F8IL=0 / 1 forces the burst / interleaved loop everywhere.
        static const int il = [] { const char* e = probe_env("D3R_GEMM_F8IL"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
        if (il == 1 || (il < 0 && cfg != GEMM_CFG_128)) {
            switch (cfg) {
                case GEMM_CFG_256: return launch_cfg<DT, Cfg256il>(p, s);
                case GEMM_CFG_256x128: return launch_cfg<DT, Cfg256x128il>(p, s);
                case GEMM_CFG_512x128: return launch_cfg<DT, Cfg512x128il>(p, s);
                default: if constexpr (kProbes) return launch_cfg<DT, Cfg128il>(p, s); break;
            }
        }
    }
    if constexpr (DT == D3R_F16X3) {
        // 128 x 128 launches of fewer than 1100 tiles (the one- to four-pair forwards) on the eight-wave shape. Measured (profiles/r03_k):
        // one pair 10.59 -> 10.26 ms, two 16.0 -> 15.3, four 26.3 -> 25.9; applied to the 1152-tile launches of the 32-pair step as well:
        // -0.1 %, hence the limit. D3R_GEMM_T128W8 moves it (0 = never).
        if (cfg == GEMM_CFG_128) {
            const char* e = getenv("D3R_GEMM_T128W8");
            const long t = e ? atol(e) : 1100;
            if ((long)cdiv(p.M, 128) * cdiv(p.n_store, 128) < t) return launch_cfg<DT, Cfg128w8>(p, s);
        }
    }
    if constexpr (DT == D3R_F16X2F8) {      // nn.Linear matrices of the transformer blocks only (N >= 768): the two square tiles; D3R_GEMM_X2IL=0 / 1:
        const char* e_il = probe_env("D3R_GEMM_X2IL");     // DMA pieces of the next step in one burst behind the barrier / interleaved with the MFMA rows (default: on the 256-wide tile)
        const int il = e_il ? (e_il[0] == '1' ? 1 : 0) : -1;
        if constexpr (kProbes) {
            if (cfg == GEMM_CFG_256 && il == 0) return launch_cfg<DT, Cfg256>(p, s);
            if (cfg != GEMM_CFG_256 && il == 1) return launch_cfg<DT, Cfg128il>(p, s);
        }
        (void)il;
        return cfg == GEMM_CFG_256 ? launch_cfg<DT, Cfg256il>(p, s) : launch_cfg<DT, Cfg128>(p, s);
    } else {
    if constexpr (DT == D3R_F16F8 && !kProbes) return launch_cfg<DT, Cfg128>(p, s);     // every other shape left through the interleaved loop above
    else
    switch (cfg) {
        case GEMM_CFG_256: return launch_cfg<DT, Cfg256>(p, s);
        case GEMM_CFG_256x128: return launch_cfg<DT, Cfg256x128>(p, s);
        case GEMM_CFG_512x128: return launch_cfg<DT, Cfg512x128>(p, s);
        default: return launch_cfg<DT, Cfg128>(p, s);
    }
    }
}

bool conv_k_slice_major() {
    static const bool v = [] { const char* e = probe_env("D3R_CONV_KORDER"); return !(e && e[0] == '0'); }();   // D3R_CONV_KORDER=0: tap-major (probe)
    return v;
}

static unsigned long long* g_trace_buf = nullptr;
static size_t g_trace_cap = 0;
void gemm_set_trace(unsigned long long* buf, size_t capacity_blocks) { g_trace_buf = buf; g_trace_cap = capacity_blocks; }

hipError_t launch_gemm(int dt, const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    if (g_trace_buf && (size_t)cdiv(p.M, 128) * cdiv(p.n_store, 128) <= g_trace_cap) p.trace = g_trace_buf;   // capacity for the smallest tile
    if (const char* e = probe_env("D3R_GEMM_NOSTORE")) if (e[0] == '1') p.flags |= GF_NOSTORE;
    // (the folded-LayerNorm producer launches keep their wide epilogue: it is where the row sums and the typed residual stream are written)
    if (const char* e = getenv("D3R_GEMM_NOWIDE")) if (e[0] == '1' && !p.ln_part && !(p.flags & GF_X3RES)) p.flags |= GF_NOWIDE;
    // measurement aid (results INVALID): the fp16 + fp8 K loop issues the MFMA mix of a 2.5-unit scheme -- per 64 k four f16 MFMAs (hi.hi and
    // hi.w_lo on the f16 pipe) and half an e4m3 MFMA (a_lo.w_hi, K = 128 spans two groups) = 80 MFMA cycles instead of 64 (fp16f8) / 96 (fp16x3)
    if (const char* e = probe_env("D3R_F8_PROXY")) if (e[0] == '1') {
        p.f8_proxy = 1;
        static std::atomic<bool> told{false};
        if (!told.exchange(true)) fprintf(stderr, "[dust3r_amd] D3R_F8_PROXY=1: MEASUREMENT AID -- the fp16 + fp8 K loops issue a different MFMA mix and every result of an fp16f8 engine is INVALID\n");
    }
    // wide epilogues store with the non-temporal policy (measured +3..10 % on isolated GEMMs, +1 % on the forward); D3R_GEMM_NT=0: plain stores
    { const char* e = probe_env("D3R_GEMM_NT"); if (!e || e[0] != '0') p.flags |= GF_NTSTORE; }
    const int kt = 128 / (int)dt_bytes(dt);
    if (p.M <= 0 || p.n_pad % 128 != 0 || p.n_store > p.n_pad || p.K % kt != 0 || p.K <= 0) return hipErrorInvalidValue;
    if (p.amode == AMODE_CONV && (p.Cin % kt != 0 || p.zero_page == nullptr)) return hipErrorInvalidValue;
    if (p.amode == AMODE_CONV) p.kslice_major = conv_k_slice_major() ? 1 : 0;
    if (const char* e = probe_env("D3R_GEMM_X3NT")) p.x3res_nt = e[0] == '1' ? 1 : 0;
    if (const char* e = probe_env("D3R_GEMM_PANEL")) { const int v = atoi(e); if (v >= 1 && v <= 64) p.panel = v; }
    if (p.epi == EPI_HEADS && p.head_c % 128 != 0 && p.head_c < (1 << 29)) return hipErrorInvalidValue;
    if (p.epi == EPI_HEAD4 && (p.n_store > 128 || p.n_store % 4 != 0 || !p.res1 || !p.res2 || !p.out || !p.out2)) return hipErrorInvalidValue;
    // folded LayerNorm (kernels.hpp): statistics come out of the wide fp32 epilogue only; the consumer side exists for split-fp16 operands, typed / GELU / head outputs
    if (p.ln_part && (p.epi != EPI_F32 || !(p.flags & GF_X3RES) || p.n_store % 32 != 0)) return hipErrorInvalidValue;   // the row sums come out of the typed-stream epilogue
    if ((p.flags & GF_X3RES) && (dt != D3R_F16X3 || p.epi != EPI_F32 || (p.flags & GF_NOWIDE) || !p.out2 || (p.ldo2 & 7) || (p.n_store & 7) || (p.res1 && (p.ldr & 7)))) return hipErrorInvalidValue;
    if (p.ln_part_in && (!p.ln_rstd || p.K % 32 != 0 || p.K > 2048)) return hipErrorInvalidValue;
    if (p.ln_rstd && (dt != D3R_F16X3 || !p.ln_nmr || !p.ln_colsum || p.amode != AMODE_LINEAR || !(p.epi == EPI_T || p.epi == EPI_GELU || p.epi == EPI_HEADS) || p.res1 || p.res2 || p.out2)) return hipErrorInvalidValue;
    // fp16 + fp8 rows: nn.Linear operands only, whole 64-element super-groups; outputs: fp32 (+ residual), GELU / plain activation rows, heads
    const bool f8rows = dt == D3R_F16F8 || dt == D3R_F16X2F8;
    if (f8rows && ((size_t)512 * p.lda * 4 >= (1ull << 32) || (size_t)512 * p.K * 5 >= (1ull << 32))) return hipErrorInvalidValue;   // 32-bit offsets inside a tile
    if (dt == D3R_F16X2F8 && p.K % 128 != 0) return hipErrorInvalidValue;      // whole 128-k blocks of five chunks
    if (f8rows && (p.amode != AMODE_LINEAR || p.K % 64 != 0 || p.lda % 64 != 0 || p.epi == EPI_CONVT || p.res2 ||
                            (p.epi == EPI_T && (p.res1 || p.out2)) || ((p.epi == EPI_T || p.epi == EPI_GELU) && (p.ldo % 64 != 0 || p.n_store % 4 != 0))))
        return hipErrorInvalidValue;
    if (use_p4(p, dt)) return launch_gemm_p4(p, s);
    {   // split-K (kernels.hpp): only with the caller's buffers, split-fp16 nn.Linear launches of the small-batch forwards (the problems the heuristic sends to the
        // 64 x 64 tile: fewer than 200 tiles of 128 x 128), the plain K loop. Probe knobs (probe builds): D3R_SK_TILE=64|128 the tile the split launch runs on,
        // D3R_SK_BLOCKS the most blocks a launch may have after the split, D3R_SK_MINSTEPS the fewest K steps (of 32) per slice.
        int sk = 1;
        if (p.sk_slab && p.sk_cnt && p.splitk != 1 && dt == D3R_F16X3 && p.amode == AMODE_LINEAR && !p.trace && !(p.flags & GF_NOSTORE) && p.force_cfg < 0 &&
            gemm_pick_config(p, dt) == GEMM_CFG_64) {
            const int cus = device_cus();
            int tile = 64, minsteps = 16;
            long maxblocks = (long)cus * 3;
            if (const char* e = probe_env("D3R_SK_TILE")) tile = atoi(e) == 128 ? 128 : 64;
            if (const char* e = probe_env("D3R_SK_BLOCKS")) maxblocks = atol(e);
            if (const char* e = probe_env("D3R_SK_MINSTEPS")) minsteps = atoi(e);
            const long tiles = (long)cdiv(p.M, tile) * cdiv(p.n_store, tile);
            const int nk32 = p.K / 32;
            for (int c : {8, 6, 4, 3, 2}) {
                if (tiles * c <= maxblocks && nk32 % c == 0 && nk32 / c >= minsteps && (size_t)tiles * c * tile * tile <= p.sk_slab_floats && tiles <= p.sk_cnt_n) { sk = c; break; }
            }
            if (sk > 1 && tile == 128) p.force_cfg = GEMM_CFG_128;
        }
        p.splitk = sk;
    }
#ifdef D3R_GEMM_ONLY_DT      // development builds (-DD3R_GEMM_ONLY_DT=4): one precision mode only, a tenth of the compile time
    if (dt == D3R_GEMM_ONLY_DT) return launch_t<D3R_GEMM_ONLY_DT>(p, s);
#else
    switch (dt) {
        case D3R_BF16: return launch_t<D3R_BF16>(p, s);
        case D3R_F16: return launch_t<D3R_F16>(p, s);
        case D3R_F32: return launch_t<D3R_F32>(p, s);
        case D3R_F16X3: return launch_t<D3R_F16X3>(p, s);
        case D3R_F16F8: return launch_t<D3R_F16F8>(p, s);
        case D3R_F16X2F8: return launch_t<D3R_F16X2F8>(p, s);
    }
#endif
    return hipErrorInvalidValue;
}

}  // namespace d3r
