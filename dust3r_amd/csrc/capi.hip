// dust3r_amd -- C-ABI wrappers of the building-block kernels (include/dust3r_hip.h).
#include "../../include/dust3r_hip.h"
#include "kernels.hpp"

using namespace d3r;

static inline int rc_of(hipError_t e) { return e == hipSuccess ? D3R_OK : 1000 + (int)e; }
static inline int rup(int a, int b) { return (a + b - 1) / b * b; }

extern "C" int d3r_rope2d(void* tokens, const int64_t* positions, int B, int N, int H, int D, float base, float F0, int dtype, void* stream) {
    if (!tokens || !positions) return D3R_ERR_INVALID;
    return rc_of(launch_rope2d(dtype, tokens, positions, B, N, H, D, base, F0, (hipStream_t)stream));
}

extern "C" int d3r_layernorm(const float* x, const float* gamma, const float* beta, void* out, int rows, int C, float eps, int dtype, void* stream) {
    if (!x || !gamma || !beta || !out) return D3R_ERR_INVALID;
    return rc_of(launch_layernorm(dtype, x, gamma, beta, out, rows, C, eps, (hipStream_t)stream));
}

extern "C" int d3r_linear(const void* act, const void* wgt, const float* bias, void* out, const float* residual, int M, int N, int K, int epilogue,
                          int dtype, void* stream) {
    if (!act || !wgt || !out || N % 4 != 0) return D3R_ERR_INVALID;
    GemmParams p;
    p.act = act; p.lda = K; p.wgt = wgt; p.bias = bias; p.M = M; p.K = K; p.n_pad = rup(N, 128); p.n_rows = rup(N, 256); p.n_store = N;      // the engine's weight loader allocates rows the same way (engine.hip: Lin / ConvW)
    p.epi = epilogue == 1 ? EPI_F32 : (epilogue == 2 ? EPI_GELU : EPI_T);
    p.out = out; p.ldo = N; p.res1 = epilogue == 1 ? residual : nullptr; p.ldr = N;
    return rc_of(launch_gemm(dtype, p, (hipStream_t)stream));
}

// nn.Linear whose output joins the typed residual stream of a folded-LayerNorm engine (GemmParams GF_X3RES): out_rows = split-fp16 rows of
// (act . wgt^T + bias + residual_rows), ln_part (optional) = the (sum, sum of squares) of every 32-column group of every stored row.
extern "C" int d3r_linear_x3res(const void* act, const void* wgt, const float* bias, void* out_rows, const void* residual_rows, float* ln_part, int M, int N,
                                int K, void* stream) {
    if (!act || !wgt || !out_rows || N % 8 != 0) return D3R_ERR_INVALID;
    GemmParams p;
    p.act = act; p.lda = K; p.wgt = wgt; p.bias = bias; p.M = M; p.K = K; p.n_pad = rup(N, 128); p.n_rows = rup(N, 256); p.n_store = N;
    p.epi = EPI_F32; p.flags = GF_X3RES; p.res1 = residual_rows; p.ldr = N; p.out2 = out_rows; p.ldo2 = N; p.ln_part = ln_part;
    return rc_of(launch_gemm(D3R_F16X3, p, (hipStream_t)stream));
}

extern "C" int d3r_conv_k_slice_major(void) { return d3r::conv_k_slice_major() ? 1 : 0; }

extern "C" int d3r_conv2d_nhwc(const void* in, const void* wgt, const float* bias, void* out, const void* res1, const void* res2, void* out_relu_copy,
                               int B, int Hin, int Win, int Cin, int Cout, int ksize, int stride, int pad, int relu, const void* zero_page, int dtype,
                               void* stream) {
    if (!in || !wgt || !out || !zero_page || Cout % 4 != 0) return D3R_ERR_INVALID;
    GemmParams p;
    p.amode = AMODE_CONV; p.act = in; p.wgt = wgt; p.bias = bias;
    p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.cstride = Cin; p.ksize = ksize; p.stride = stride; p.pad = pad;
    p.Hout = (Hin + 2 * pad - ksize) / stride + 1; p.Wout = (Win + 2 * pad - ksize) / stride + 1;
    p.M = B * p.Hout * p.Wout; p.K = ksize * ksize * Cin; p.n_pad = rup(Cout, 128); p.n_rows = rup(Cout, 256); p.n_store = Cout; p.zero_page = zero_page;
    p.epi = EPI_T; p.flags = relu ? GF_RELU : 0; p.out = out; p.ldo = Cout; p.res1 = res1; p.res2 = res2; p.ldr = Cout;
    p.out2 = out_relu_copy; p.ldo2 = Cout;
    return rc_of(launch_gemm(dtype, p, (hipStream_t)stream));
}

extern "C" int d3r_attention(const void* q, const void* k, const void* vt, void* out, int B, int H, int Nq, int Nk, int ldv, float scale, int dtype,
                             void* stream) {
    if (!q || !k || !vt || !out) return D3R_ERR_INVALID;
    AttnParams a;
    a.q = q; a.k = k; a.vt = vt; a.out = out; a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.ldv = ldv; a.scale = scale;
    return rc_of(launch_attention(dtype, a, (hipStream_t)stream));
}

extern "C" int d3r_upsample2x_nhwc(const void* in, void* out, int B, int Hi, int Wi, int C, int Ho, int Wo, int dtype, void* stream) {
    if (!in || !out) return D3R_ERR_INVALID;
    return rc_of(launch_upsample2x(dtype, in, out, nullptr, B, Hi, Wi, C, C, Ho, Wo, (hipStream_t)stream));
}

// Diagnostics: per-block phase timestamps of every following GEMM launch into `buf` (8 x uint64 per block; NULL switches it off).
extern "C" int d3r_gemm_set_trace(void* buf, size_t capacity_blocks) {
    gemm_set_trace((unsigned long long*)buf, buf ? capacity_blocks : 0);
    return D3R_OK;
}

// 1: this library was compiled with -DD3R_PROBES (ablation kernels and probe-only environment switches present), 0: the default build
extern "C" int d3r_build_has_probes(void) { return kProbes ? 1 : 0; }

// Diagnostics, host only (no GPU, no launch): the tile configuration the heuristic of gemm.hip picks for an nn.Linear-shaped problem
// (same `epilogue` codes as d3r_linear; with_residual: an fp32 residual is added). The dispatch table of DESIGN.md section 4.1 as a function.
extern "C" int d3r_gemm_tile_config(int dtype, int M, int N, int K, int epilogue, int with_residual) {
    if (M <= 0 || N <= 0 || K <= 0 || N % 4 != 0) return D3R_ERR_INVALID;
    if (dtype != D3R_BF16 && dtype != D3R_F16 && dtype != D3R_F32 && dtype != D3R_F16X3 && dtype != D3R_F16F8 && dtype != D3R_F16X2F8) return D3R_ERR_INVALID;
    static const float dummy = 0.f;
    GemmParams p;
    p.lda = K; p.M = M; p.K = K; p.n_pad = rup(N, 128); p.n_rows = rup(N, 256); p.n_store = N;      // the engine's weight loader allocates rows the same way (engine.hip: Lin / ConvW)
    p.epi = epilogue == 1 ? EPI_F32 : (epilogue == 2 ? EPI_GELU : EPI_T);
    p.res1 = (epilogue == 1 && with_residual) ? &dummy : nullptr;
    p.act = &dummy; p.wgt = &dummy; p.out = const_cast<float*>(&dummy);      // (never dereferenced: the rule only asks whether the operands exist)
    return gemm_pick_config(p, dtype);
}

