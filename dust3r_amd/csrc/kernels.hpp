// dust3r_amd -- internal kernel launch interface (C++ side, below the C-ABI of include/dust3r_hip.h).
#pragma once
#include "common.hpp"

namespace d3r {

// ------------------------------------------------------------------------------ GEMM / conv
enum { AMODE_LINEAR = 0, AMODE_CONV = 1 };
enum { EPI_T = 0, EPI_F32 = 1, EPI_GELU = 2, EPI_HEADS = 3, EPI_CONVT = 4, EPI_HEAD4 = 5 };
// EPI_HEAD4 (split-fp16, tiles whose waves hold all <= 128 output channels of their rows: the 512 x 128 / 256 x 128 R shapes): the DPT head's
// tail fused into its last 3x3 convolution -- bias, ReLU (GF_RELU), Conv2d(C, 4, 1) on the fp32 accumulators, postprocess, store of
// pts / conf (dpt_head.py:63 + postprocess.py:10-58). Field reuse: out = pts (float*), ldo = its pixel stride, out2 = conf (float*),
// ldo2 = its pixel stride, res1 = the 1x1 weights [4][n_store] fp32, res2 = its bias [4] fp32. Nothing of the C-channel map is stored.
enum { HEAD_ROPE = 1, HEAD_VT = 2, HEAD_PLAIN = 3 };
enum { GF_RELU = 1, GF_NOSTORE = 2, GF_NOWIDE = 4, GF_NTSTORE = 8, GF_X3RES = 16 };   // GF_X3RES (EPI_F32, split-fp16): res1 and the result live in split-fp16 rows only (out2); no fp32 row is stored   // GF_NTSTORE: wide epilogues store with the non-temporal policy (default; D3R_GEMM_NT=0 clears it)
enum { GEMM_CFG_128 = 0, GEMM_CFG_256 = 1, GEMM_CFG_256x128 = 2, GEMM_CFG_512x128 = 3, GEMM_CFG_256x128W4 = 4, GEMM_CFG_256S4 = 5, GEMM_CFG_256PP = 6, GEMM_CFG_256x128R = 7, GEMM_CFG_64 = 8, GEMM_CFG_384x192 = 9, GEMM_CFG_P4 = 10, GEMM_CFG_96x64 = 11 };

struct GemmParams {
    const void* act = nullptr;   // [M][lda] (linear) or NHWC image batch (conv), element type DT
    const void* wgt = nullptr;   // [n_pad][K], element type DT, rows >= N zero filled
    const float* bias = nullptr; // [n_pad] or null
    int M = 0, K = 0, n_pad = 0; // K in elements, multiple of 128/sizeof(DT); n_pad: multiple of 128, >= n_store
    int n_rows = 0;              // weight rows actually allocated (>= n_pad; 0 = n_pad): bounds the tile width choice
    int force_cfg = -1;          // GEMM_CFG_* to override the heuristic (tests / probes)
    int n_store = 0;             // columns written (multiple of 4, <= n_pad)
    int lda = 0;
    int amode = AMODE_LINEAR;
    // implicit-GEMM conv geometry (NHWC, K index = (ky, kx, cin))
    int Hin = 0, Win = 0, Cin = 0, cstride = 0, Hout = 0, Wout = 0, ksize = 1, stride = 1, pad = 0;
    const void* zero_page = nullptr;
    // epilogue
    int epi = EPI_T;
    int flags = 0;
    void* out = nullptr; int ldo = 0;
    const void* res1 = nullptr; const void* res2 = nullptr; int ldr = 0;
    void* out2 = nullptr; int ldo2 = 0;
    int ct_cout = 0;             // EPI_CONVT: (padded) output channels per tap
    // EPI_HEADS: column n belongs to region n / head_c; head (n % head_c) / 64
    int head_c = 1 << 30;
    int head_kind[3] = {0, 0, 0};
    void* head_dst[3] = {nullptr, nullptr, nullptr};
    int heads = 0, ntok = 1, tok_w = 1, ldv = 0;
    const float* rope_table = nullptr;  // [max_pos][16] (cos, sin) pairs
    // first-round start stagger (set by launch_gemm): blocks with blockIdx < first_round wait stagger_ticks * slot / 32 wall-clock
    // ticks before they start, so that the epilogue bursts of the resident tiles do not all hit HBM at the same time
    int stagger_ticks = 0, first_round = 0, stagger_mode = 0;
    int panel = 8;               // tile map: width of the column panels in tiles (launch_gemm: D3R_GEMM_PANEL probe)
    int kslice_major = 0;        // implicit-GEMM K order: 0 = (tap, channel), 1 = (channel slice of one K step, tap, channel in slice); set by launch_gemm
    // diagnostics (d3r_gemm_set_trace): 8 x uint64 per block -- wall-clock ticks at entry / K-loop start / K-loop end / epilogue
    // issued / stores drained, then HW_ID, XCC_ID, blockIdx
    unsigned long long* trace = nullptr;
    PostMode post;               // EPI_HEAD4: depth_mode / conf_mode of the head's postprocess
    // ---- LayerNorm folded into the GEMMs on both sides of it (round 5, split-fp16 engine; engine.hip `ln_fold`) ----------------------------
    // LN(x) W^T + b = rstd (x (W diag(gamma))^T - mean s) + b',  s_n = sum_k gamma_k W_nk,  b' = b + W beta.
    // PRODUCER (an EPI_F32 launch, wide epilogue): next to the fp32 rows it stores (and their typed copy out2 = the RAW x the consumer GEMM
    //   reads as its activation operand) it writes ln_part[m][n / 32] = (sum, sum of squares) of the 32 stored values of row m in that column
    //   group, summed in ONE fixed tree whatever the tile shape (quad sums, then xor 1 / 2 / 4 over the 8 quads) -- so the statistics, like every
    //   output, do not depend on the tile configuration. n_store % 32 == 0.
    // CONSUMER (weights = W diag(gamma) packed at load time, bias = b'): every epilogue forms rstd_m (acc - mean_m s_n) + b'_n as
    //   fma(acc, ln_rstd[m], fma(ln_colsum[n], ln_nmr[m], bias[n])) with ln_nmr = -mean rstd; with ln_rstd == nullptr the same expression
    //   runs on (1, 0, 0): fma(acc, 1, fma(0, 0, b)) = acc + b bit for bit.
    float* ln_part = nullptr;
    const float* ln_rstd = nullptr; const float* ln_nmr = nullptr; const float* ln_colsum = nullptr;
    // CONSUMER of a small problem: the partial sums [M][K / 32][2] of its input rows; the kernel's prologue then forms ln_rstd / ln_nmr of its tile's rows itself (and WRITES
    // them to those two arrays) with the arithmetic of ln_finalize_kernel, which is not launched
    const float* ln_part_in = nullptr; float ln_eps = 1e-6f, ln_inv_c = 0.f;
    int x3res_nt = 0;            // GF_X3RES: typed-stream stores with the non-temporal policy (probe D3R_GEMM_X3NT=1; default plain: the rows are re-read at once)
    // ---- split-K for SMALL problems (round 6; split-fp16, nn.Linear operands, the plain K loop): `splitk` blocks share a tile, each over 1 / splitk of the K steps;
    // every block stores its fp32 partial tile in sk_slab ([tile][slice][wave][fragment][lane] float4), the block that draws the last ticket of sk_cnt[tile] adds the
    // slices up in slice order (deterministic whichever block is last) and runs the epilogue. Chosen by launch_gemm when the caller lends it the two buffers.
    int splitk = 1; float* sk_slab = nullptr; unsigned* sk_cnt = nullptr; size_t sk_slab_floats = 0; int sk_cnt_n = 0;
    int f8_proxy = 0;            // MEASUREMENT AID (D3R_F8_PROXY=1, results INVALID): fp16 + fp8 K loop with the MFMA mix of a 2.5-unit scheme (4 f16 + 1/2 fp8 MFMA per 64 k)
};
void gemm_set_trace(unsigned long long* buf, size_t capacity_blocks);

hipError_t launch_gemm(int dt, const GemmParams& p, hipStream_t s);
int gemm_pick_config(const GemmParams& p, int dt);
// gemm_p4.hip: the persistent split-fp16 kernel whose epilogue runs under the next tile's K loop (tile configuration 10 in profiles)
bool gemm_p4_eligible(const GemmParams& p, int dt);
hipError_t launch_gemm_p4(const GemmParams& p, hipStream_t s);
int gemm_p4_mode();              // -1: the heuristic decides, 0: never, 1: every eligible launch (D3R_GEMM_PERSIST, read per call: probes and tests move it)

// ------------------------------------------------------------------------------ attention
struct AttnParams {
    const void* q = nullptr;   // [B][H][Nq][64]
    const void* k = nullptr;   // [B][H][Nk][64]
    const void* vt = nullptr;  // [B][H][64][ldv]   (ldv >= round_up(Nk, 64), pad zero filled)
    void* out = nullptr;       // [B][Nq][H*64]
    int B = 0, H = 0, Nq = 0, Nk = 0, ldv = 0;
    float scale = 0.125f;
    int out_dt = -1;           // layout of `out` when it differs from the operands' (D3R_F16F8 rows out of a split-fp16 attention); -1: same
};
hipError_t launch_attention(int dt, const AttnParams& p, hipStream_t s);

// ------------------------------------------------------------------------------ elementwise
// LayerNorm over the last dim of an fp32 [rows][C] tensor -> DT [rows][C]
hipError_t launch_layernorm(int dt, const float* x, const float* gamma, const float* beta, void* out, int rows, int C,
                            float eps, hipStream_t s);
// the same from split-fp16 input rows into split-fp16 output rows (enc_norm / dec_norm of a folded-LayerNorm engine: the residual stream is typed)
hipError_t launch_layernorm_x3in(const void* x3rows, const float* gamma, const float* beta, void* out, int rows, int C, float eps, hipStream_t s);
// fp32 -> DT copy (rows x C, contiguous)
hipError_t launch_convert(int dt, const float* x, void* out, size_t n, hipStream_t s);
// NCHW fp32 image -> [B*th*tw][3*ps*ps] patch rows of DT (k = (c, py, px))
hipError_t launch_patchify(int dt, const float* img, void* out, int B, int H, int W, int ps, hipStream_t s);
// standalone in-place 2-D RoPE on tokens[B][N][H][D] (fp32 / bf16 / f16): the reference's native op
hipError_t launch_rope2d(int dt, void* tokens, const int64_t* pos, int B, int N, int H, int D, float base, float F0,
                         hipStream_t s);
hipError_t launch_rope_table(float* table, int max_pos, float base, float F0, hipStream_t s);
// bilinear x2, align_corners=True, NHWC DT -> NHWC DT (optionally also relu copy), output cropped to (Ho, Wo)
hipError_t launch_upsample2x(int dt, const void* in, void* out, void* out_relu, int B, int Hi, int Wi, int C, int cstride,
                             int Ho, int Wo, hipStream_t s);
// final DPT stage: relu'd NHWC DT [pix][C] x W[4][C] + b -> postprocess -> pts3d [pix][3], conf [pix]
hipError_t launch_head_final(int dt, const void* feat, int C, const float* w, const float* b, float* pts, float* conf,
                             size_t npix, int pstride, int cstride, PostMode post, hipStream_t s);
// linear head: proj output fp32 [B*th*tw][(3+1)*ps*ps] -> pixel_shuffle -> postprocess
hipError_t launch_linear_head_post(const float* feat, float* pts, float* conf, int B, int th, int tw, int ps, int pstride, int cstride,
                                   PostMode post, hipStream_t s);
hipError_t launch_fill_zero(void* p, size_t bytes, hipStream_t s);
// folded LayerNorm (GemmParams::ln_part): per row, the G = C / 32 partial (sum, sum of squares) pairs of the producing GEMM -> rstd[m] and
// nmr[m] = -mean rstd (fixed summation order; variance = E[x^2] - mean^2 combined in fp64, eps inside the root like nn.LayerNorm)
hipError_t launch_ln_finalize(const float* part, int rows, int C, float eps, float* rstd, float* nmr, hipStream_t s);
// load-time fold of a LayerNorm into the nn.Linear that consumes it: for each of the N rows of W (fp32 [N][K]): colsum[n] = sum_k r(gamma_k W_nk)
// with r() the rounding of the packed operand type (split-fp16: hi + lo), bias_out[n] = bias_in[n] (or 0) + sum_k beta_k W_nk; fp64 sums
hipError_t launch_ln_fold_vectors(int dt, const float* W, const float* gamma, const float* beta, const float* bias_in, float* colsum, float* bias_out, int N, int K, hipStream_t s);

// load-time weight packing on the device (fp32 PyTorch-layout source -> engine layout in DT)
enum { PACK_MAT = 0, PACK_CONV = 1, PACK_CONVT = 2 };
struct PackParams {
    const float* src = nullptr; void* dst = nullptr;
    size_t numel = 0;
    int kind = PACK_MAT, cols = 0, row_off = 0, dst_cols = 0, cin = 0, cin_pad = 0, ksize = 1, cout_pad = 0;
    int kslice_major = 0;        // PACK_CONV: K order of the packed rows (conv_k_slice_major())
    const float* kscale = nullptr;   // PACK_MAT: multiply column k of the source by kscale[k] before the conversion (LayerNorm gamma folded into the weights)
};
bool conv_k_slice_major();       // process-wide K order of implicit-GEMM operands (kernel and weight packing agree on it)
hipError_t launch_pack_weight(int dt, const PackParams& p, hipStream_t s);
hipError_t launch_pack_convt_bias(const float* src, float* dst, int cout, int cout_pad, int taps, hipStream_t s);

// ------------------------------------------------------------------------------ aligner
struct AlignerDev;  // defined in aligner.hip

// Row statistics of a folded LayerNorm from the producer's G partial (sum x, sum x^2) pairs, formed by TPR = 1, 2 or 4 ADJACENT lanes per row in the consumer GEMM's prologue
// (GemmParams::ln_part_in) with EXACTLY the arithmetic of ln_finalize_kernel (elementwise.hip: 32 lanes per row, lane g adds pairs g and g + 32 in fp64, xor butterfly 1 .. 16):
// a butterfly over lanes is a balanced binary tree with adjacent pairing, and fp addition commutes, so a thread that owns W = 32 / TPR consecutive lane values reproduces the tree's
// lower levels in registers (a[i] += a[i + o], o = 1, 2, ...) and the upper levels by lane exchange (xor 1, xor 2 = the kernel's xor W, xor 2 W). rstd = 1 / sqrt(max(var, 0) + eps),
// nmr = -mean rstd; every lane of the row returns them.
#if defined(__HIPCC__)
// (sum, sum of squares) of the 4 values a lane holds in the read phase of a typed-residual-stream epilogue: the leaves of the fixed tree of GemmParams::ln_part.
// Contraction is switched OFF here: under -ffp-contract=fast hipcc fuses `x*x + y*y` in one kernel and not in another (gemm.hip's instances kept four rounded
// products, gemm_p4.hip's first version fused them: 6 % of the sums' last bits differed), and the statistics must not depend on which kernel stored the row.
// Four rounded products, pairwise sums -- the arithmetic rounds 5's engine ran and its oracle-parity figures were measured with.
D3R_DEV void ln_quad_sums(const float4& v, float& sm, float& sq) {
#pragma clang fp contract(off)
    const float xx = v.x * v.x, yy = v.y * v.y, zz = v.z * v.z, ww = v.w * v.w;
    sm = (v.x + v.y) + (v.z + v.w);
    sq = (xx + yy) + (zz + ww);
}
template <int TPR> D3R_DEV void ln_row_stats(const float2* __restrict__ pr, int G, int sub, float inv_c, float eps, float& rstd, float& nmr) {
    static_assert(TPR == 1 || TPR == 2 || TPR == 4, "threads per row");
    constexpr int W = 32 / TPR;
    double a[W], b[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int g = sub * W + i;
        double s = 0.0, t = 0.0;
        if (g < G) { const float2 v = pr[g]; s = (double)v.x; t = (double)v.y; }
        if (g + 32 < G) { const float2 v = pr[g + 32]; s += (double)v.x; t += (double)v.y; }
        a[i] = s; b[i] = t;
    }
#pragma unroll
    for (int o = 1; o < W; o <<= 1)
#pragma unroll
        for (int i = 0; i < W; i += 2 * o) { a[i] += a[i + o]; b[i] += b[i + o]; }
    double s = a[0], t = b[0];
    if constexpr (TPR >= 2) { s += __shfl_xor(s, 1); t += __shfl_xor(t, 1); }
    if constexpr (TPR == 4) { s += __shfl_xor(s, 2); t += __shfl_xor(t, 2); }
    const double mean = s * (double)inv_c;
    double var = t * (double)inv_c - mean * mean;
    var = var > 0.0 ? var : 0.0;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    nmr = (float)(-mean) * rstd;
}
#endif

}  // namespace d3r
