/* dust3r_hip.h -- C ABI of libdust3r_hip.so: the MI355X (gfx950) engine behind the DUSt3R
 * inference-and-alignment hot path.
 *
 * Boundary. The reference (naver/dust3r) exposes this path as a PYTHON API; its only native
 * interface is croco's `curope` extension. The Python host package `dust3r_amd/` mirrors the
 * reference API (same names / arguments / outputs) and binds these entry points with ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add. Every function cites
 * the reference interface it replaces (paths relative to the reference repository root).
 *
 * Conventions: plain pointers and sizes, no framework types. Unless a parameter says "host",
 * pointers are DEVICE pointers owned by the caller (borrowed for the duration of the call, or
 * until destroy for handles that document it). `stream` is a hipStream_t passed as void*
 * (NULL = default stream); calls enqueue work and return without synchronising unless noted.
 * Return value: D3R_OK (0) or a negative D3R_ERR_* code / 1000 + hipError_t.
 * Threading: a handle must not be used from two threads at once; distinct handles are independent.
 */
#ifndef DUST3R_HIP_H
#define DUST3R_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3R_OK 0
#define D3R_ERR_INVALID (-1)
#define D3R_ERR_ALLOC (-2)
#define D3R_ERR_LAUNCH (-3)
#define D3R_ERR_UNKNOWN_KEY (-4)
#define D3R_ERR_SHAPE (-5)
#define D3R_ERR_STATE (-6)

/* arithmetic type of the matrix kernels (accumulation is always fp32) */
#define D3R_DTYPE_BF16 0 /* v_mfma_*_bf16: the throughput mode named by BASELINE.json */
#define D3R_DTYPE_F16 1  /* v_mfma_*_f16: same rate, 3 more mantissa bits */
#define D3R_DTYPE_F32 2  /* v_mfma_f32_*_f32: exact fp32 like the reference (dust3r/inference.py:44), 1/16 rate */
#define D3R_DTYPE_F16X3 3 /* split fp16 (hi + lo pairs, 3 f16 MFMAs per product): fp32-class accuracy at 1/3 of the 16-bit rate.
                           * Rows: 32-byte groups [hi fp16 x8][lo fp16 x8] (4 bytes per logical element). */
#define D3R_DTYPE_F16F8 4 /* fp16 + fp8: hi.hi on the f16 MFMA, the cross terms hi.lo + lo.hi on ONE K-concatenated e4m3 MFMA
                           * (v_mfma_scale_f32_16x16x128_f8f6f4, twice the 16-bit rate): 2 MFMA units per product instead of 3.
                           * Model engine: the transformer blocks' nn.Linear layers run in this layout, everything else in D3R_DTYPE_F16X3.
                           * Rows (K % 64 == 0): 256-byte super-groups [hi fp16 x64 | a8 e4m3 x64 | b8 e4m3 x64];
                           * activations a8 = e4m3(hi), b8 = e4m3(lo 2^11); weights a8 = e4m3(lo 2^17), b8 = e4m3(hi 2^6).
                           * d3r_layernorm writes activation rows; d3r_linear takes activation rows x weight rows (epilogue 0 / 2 write
                           * activation rows, N % 64 == 0; epilogue 1 fp32). */
#define D3R_DTYPE_F16X2F8 5 /* 2.5 MFMA units per product (round 4): hi.hi and hi.w_lo on the f16 MFMA -- the WEIGHTS keep 22 bits -- and only
                             * a_lo.w_hi on the e4m3 MFMA (K = 128). Model engine: like D3R_DTYPE_F16F8, the transformer blocks' nn.Linear layers.
                             * Activation rows: the D3R_DTYPE_F16F8 layout (the a8 copy is not read). Weight rows (K % 128 == 0): 5 K bytes, per 128 k
                             * five 128-byte chunks [w_hi k 0..63 fp16 | w_lo k 0..63 fp16 | w_hi k 64..127 | w_lo k 64..127 | e4m3(w_hi 2^6) k 0..127]. */

const char* d3r_version(void);
/* 0 when a gfx950 device is visible to the HIP runtime, else an error code (used to fail loudly) */
int d3r_device_check(void);

/* ------------------------------------------------------------------------------------------------
 * 2-D rotary embedding -- drop-in for the reference's only native op:
 *   croco/models/curope: rope_2d(Tensor tokens[B,N,H,D], Tensor positions[B,N,2] int64, float base, float F0)
 *   (pybind module `curope`, wrapped by cuRoPE2D; called from
 *   croco Attention/CrossAttention.forward which dust3r/model.py:136-137,180-186 drives).
 * In place on `tokens` (contiguous, D % 4 == 0); dtype is one of D3R_DTYPE_*.
 */
int d3r_rope2d(void* tokens, const int64_t* positions, int B, int N, int H, int D, float base, float F0, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Building-block kernels (exported so the parity tests can pin each one against PyTorch fp32).
 */
/* LayerNorm(eps) over the last dim: x fp32 [rows][C] -> out dtype [rows][C]   (croco norm_layer, eps=1e-6) */
int d3r_layernorm(const float* x, const float* gamma, const float* beta, void* out, int rows, int C, float eps, int dtype, void* stream);

/* out = epilogue(act[M][K] . wgt[N][K]^T + bias): nn.Linear of croco Mlp/Attention (dust3r/model.py:136-137,176-186).
 * act, wgt in `dtype`; wgt (and bias) must hold round_up(N,256) rows (extra rows zero: the widest tile is 256 columns);
 * K % (128/sizeof(dtype)) == 0.
 * epilogue: 0 store dtype | 1 fp32 out (+ optional fp32 residual, may alias out) | 2 GELU(erf) store dtype */
int d3r_linear(const void* act, const void* wgt, const float* bias, void* out, const float* residual, int M, int N, int K,
               int epilogue, int dtype, void* stream);

/* The same nn.Linear as the producer of a folded LayerNorm (split-fp16 only): out_rows [M][N] split-fp16 rows =
 * act . wgt^T + bias + residual_rows (split-fp16 rows [M][N] or NULL; may alias out_rows) -- croco Block: x = x + proj(attn) / x + fc2(...)
 * (dust3r/model.py:136-137 via croco blocks.py) -- and, when ln_part != NULL, ln_part[m][N / 32][2] = (sum, sum of squares) of each 32-column group of the
 * stored row, from which the next LayerNorm's statistics are formed. N % 8 == 0 (ln_part: N % 32 == 0). */
int d3r_linear_x3res(const void* act, const void* wgt, const float* bias, void* out_rows, const void* residual_rows, float* ln_part, int M, int N,
                     int K, void* stream);

/* 2-D convolution, NHWC, as implicit GEMM: in [B][Hin][Win][Cin] dtype, wgt [round_up(Cout,256)][k*k*Cin] dtype;
 * out [B][Hout][Wout][Cout] dtype = [relu](conv + bias + res1 + res2)   (DPT head convs, dust3r/heads/dpt_head.py:34-65).
 * K order of a weight row: with S = 128 / sizeof(dtype) channels per K step (Cin % S == 0),
 *   d3r_conv_k_slice_major() == 1 (default): k = (cin / S) * (k*k*S) + (ky*k + kx) * S + cin % S   (all taps of one channel slice
 *                                            back to back: the slice's input lines are re-read from L2, not from HBM)
 *   == 0 (D3R_CONV_KORDER=0, probe)        : k = (ky*k + kx) * Cin + cin.
 * zero_page: >= 256 bytes of zeros. */
int d3r_conv_k_slice_major(void);
int d3r_conv2d_nhwc(const void* in, const void* wgt, const float* bias, void* out, const void* res1, const void* res2,
                    void* out_relu_copy, int B, int Hin, int Win, int Cin, int Cout, int ksize, int stride, int pad, int relu,
                    const void* zero_page, int dtype, void* stream);

/* softmax(q k^T * scale) v: q [B][H][Nq][64], k [B][H][Nk][64], vt [B][H][64][ldv] (ldv % 64 == 0, pad zero),
 * out [B][Nq][H*64]; all `dtype`   (croco Attention / CrossAttention core) */
int d3r_attention(const void* q, const void* k, const void* vt, void* out, int B, int H, int Nq, int Nk, int ldv, float scale,
                  int dtype, void* stream);

/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) on NHWC, output cropped to (Ho, Wo) */
int d3r_upsample2x_nhwc(const void* in, void* out, int B, int Hi, int Wi, int C, int Ho, int Wo, int dtype, void* stream);
/* Diagnostics (no reference counterpart): per-block phase timestamps of every following GEMM / convolution launch are written to
 * `buf` (device memory, 8 x uint64 per block: wall-clock ticks at block entry, K-loop start, K-loop end, epilogue issued, stores
 * drained; then HW_ID, XCC_ID, blockIdx). `capacity_blocks` bounds the launches that are traced; buf = NULL switches it off. */
int d3r_gemm_set_trace(void* buf, size_t capacity_blocks);
/* Diagnostics, host only: 1 when the library was built with -DD3R_PROBES (`D3R_PROBES=1 python -m dust3r_amd.build`): the ablation kernels and the
 * probe-only D3R_* environment switches of the tools/ probes are compiled in. The default build (0) reads the documented switches only. */
int d3r_build_has_probes(void);
/* Diagnostics, host only (no device needed): the GEMM tile configuration the engine picks for an nn.Linear-shaped problem (epilogue codes of
 * d3r_linear; with_residual: an fp32 residual row is added). 0 = 128x128 (eight waves below 1100 tiles in split-fp16), 1 = 256x256,
 * 2 = 256x128, 3 = 512x128, 7 = 256x128 by four waves with a K step's weights in registers (two blocks per CU), 8 = 64x64 on a three-slot
 * ring (problems of fewer than 200 128x128 tiles, split-fp16), 9 = M 384 x N 192 by eight waves of 192 (n) x 48 (m) (split-fp16 nn.Linear launches without
 * attention heads whose tiles fill whole rounds of 256 CUs: the decoder's 24576-row GEMMs of the 32-pair step), 10 = the persistent kernel (gemm_p4.hip: M 256 x N 128, the
 * epilogue of a tile under the next tile's K loop), 11 = M 96 x N 64 on the three-slot ring (small-batch launches with K >= 2048 whose tiles come to 1.5 ... 2 per CU).
 * D3R_GEMM_CFG / D3R_GEMM_PERSIST apply. DESIGN.md section 4.1. */
int d3r_gemm_tile_config(int dtype, int M, int N, int K, int epilogue, int with_residual);

/* ------------------------------------------------------------------------------------------------
 * Model engine -- replaces AsymmetricCroCo3DStereo.forward (dust3r/model.py:199-211) including
 * _encode_image_pairs (:142-151), _decoder (:172-191), the DPT / linear heads
 * (dust3r/heads/dpt_head.py:34-115, linear_head.py:30-41) and postprocess (heads/postprocess.py:10-58).
 */
typedef struct d3r_model d3r_model;

typedef struct d3r_model_config {
    int enc_embed_dim, enc_depth, enc_num_heads; /* 1024 / 24 / 16 (README.md:318) */
    int dec_embed_dim, dec_depth, dec_num_heads; /* 768 / 12 / 12 */
    int patch_size;                              /* 16 */
    int head_type;                               /* 0 = linear (LinearPts3d), 1 = dpt */
    int dtype;                                   /* D3R_DTYPE_* */
    float rope_freq;                             /* 100 for pos_embed='RoPE100' */
    int dpt_skip_relu_inplace;                   /* 0: skip adds un-activated x (nn.ReLU(False)); see SURVEY.md A.5 */
} d3r_model_config;

/* Concurrency: the engines of one process share their helper HIP streams per device (decoder side 2, K | V ahead); a handle is not thread-safe, and the
 * enqueue of a forward (host side; the device work stays asynchronous) is a process-wide critical section inside the library -- engines driven from different
 * host threads are correct, but their side-stream halves run in enqueue order, not concurrently. One engine per device is the intended use. */
int d3r_model_create(d3r_model** out, const d3r_model_config* cfg);
int d3r_model_destroy(d3r_model* m);
/* Load one tensor of the reference checkpoint's state dict by its key, e.g.
 * "enc_blocks.3.attn.qkv.weight". data: HOST fp32, contiguous, PyTorch layout. Keys the engine does not use
 * (mask_token, aliased scratch.layerN_rn, ...) return D3R_OK and are ignored; unknown keys -> D3R_ERR_UNKNOWN_KEY.
 * dec_blocks.* also fills dec_blocks2.* until a dec_blocks2 key arrives (dust3r/model.py:91-98).
 * Split-fp16 engines (D3R_DTYPE_F16X3, the default; environment D3R_LN_FOLD=0 turns it off at creation) fold the blocks' LayerNorms
 * (croco Block.norm1 / norm2, DecoderBlock.norm1 / norm2 / norm3 / norm_y) into the nn.Linear behind each of them: the matrix is packed
 * as W diag(gamma) with the bias b + W beta, by the first forward / encode / decode call after a load, from fp32 copies of those matrices
 * that are released once packed. Consequence: after that call, a new value for one of these LayerNorm vectors, or for the bias of
 * attn.qkv / cross_attn.projq / projk / projv / mlp.fc1, must come together with the matrices it is folded into (ALL weight tensors of such
 * a matrix: projk and projv share one) -- otherwise the next forward returns D3R_ERR_STATE until they arrive. The bookkeeping is per matrix:
 * only the matrices whose inputs changed are re-folded, a consistently reloaded block never depends on the others. Loading a whole state
 * dict (what the Python mirror does) always satisfies this. */
int d3r_model_load_tensor(d3r_model* m, const char* key, const float* data_host, int ndim, const int64_t* shape);
/* same, but `data_dev` is a DEVICE fp32 tensor (e.g. a checkpoint already uploaded by the caller). Both variants convert
 * to the engine dtype / layout on the GPU; the call is ordered on the default stream and returns without synchronising. */
int d3r_model_load_tensor_device(d3r_model* m, const char* key, const float* data_dev, int ndim, const int64_t* shape);
/* number of tensors still missing (0 = ready) */
int d3r_model_missing(const d3r_model* m);
/* forward on B pairs of equal-size images (H, W multiples of patch_size):
 * img1, img2: fp32 [B][3][H][W] in [-1,1]; outputs fp32: pts1 [B][H][W][3], conf1 [B][H][W],
 * pts2 (= pred2['pts3d_in_other_view']) and conf2. Workspace is owned by the model and grown on demand. */
int d3r_model_forward(d3r_model* m, const float* img1, const float* img2, int B, int H, int W, float* pts1, float* conf1, float* pts2,
                      float* conf2, void* stream);
/* forward for a batch whose two views have DIFFERENT sizes (img1: B x 3 x H1 x W1, img2: B x 3 x H2 x W2), the else-branch of
 * dust3r/model.py:148-150 (_encode_image_pairs encodes the two views separately); outputs pts1/conf1 at (H1, W1), pts2/conf2 at (H2, W2).
 * Cross attention runs with Nq != Nk; with equal sizes it is d3r_model_forward. */
int d3r_model_forward_mixed(d3r_model* m, const float* img1, int H1, int W1, const float* img2, int H2, int W2, int B, float* pts1, float* conf1,
                            float* pts2, float* conf2, void* stream);

/* same forward, outputs interleaved per pixel: out8 fp32 [B][H][W][8] = (pts1 xyz, conf1, pts2 xyz, conf2) -- the single payload the
 * pair-sharded multi-GPU path all-gathers (dust3r_amd/parallel.py), written directly by the head epilogues */
int d3r_model_forward_packed(d3r_model* m, const float* img1, const float* img2, int B, int H, int W, float* out8, void* stream);
/* The same forward in two calls, so that a view shared by several pairs is encoded ONCE (the reference re-encodes it for every
 * pair, dust3r/model.py:142-151; make_pairs' complete graph over n views has n(n-1) pair slots but only n images):
 *   d3r_model_encode: patch-embed + encoder + enc_norm (model.py:128-140) over n images fp32 [n][3][H][W] -> feat_out, an
 *                     opaque device buffer of n * d3r_model_feature_bytes(m, H, W) bytes (engine dtype, [n][tokens][enc_dim]);
 *   d3r_model_decode: decoder + heads (model.py:172-211) over B pairs whose features the caller gathered into feat =
 *                     [view-1 features of the B pairs | view-2 features of the B pairs] (2 B feature blocks).
 * encode + gather + decode gives bit-identical outputs to d3r_model_forward on the same pairs. */
size_t d3r_model_feature_bytes(const d3r_model* m, int H, int W);
int d3r_model_encode(d3r_model* m, const float* img, int n, int H, int W, void* feat_out, void* stream);
int d3r_model_decode(d3r_model* m, const void* feat, int B, int H, int W, float* pts1, float* conf1, float* pts2, float* conf2,
                     void* stream);
/* d3r_model_decode with the outputs interleaved per pixel like d3r_model_forward_packed: out8 fp32 [B][H][W][8]. This is what a rank of
 * the pair-sharded path runs on its shard after encoding the distinct images of that shard once (dust3r_amd/parallel.py). */
int d3r_model_decode_packed(d3r_model* m, const void* feat, int B, int H, int W, float* out8, void* stream);
/* number of forwards served by a graph replay so far (tests, probes) */
long d3r_model_graph_replays(const d3r_model* m);
/* bytes of device memory currently held (weights + workspace) */
size_t d3r_model_device_bytes(const d3r_model* m);
/* Measurement hook (bench.py): with D3R_MODEL_OPT_PROFILE = 1 the next forwards record one HIP event before every
 * kernel launch on the caller's stream; d3r_model_profile_read then returns, for the LAST forward, the number of
 * launches, their summed duration in ms and their summed algorithmic work (flops) for one kernel class:
 * kind 0..7 = gemm_kernel launches of tile configuration `kind` on nn.Linear operands (0 = 128x128, 1 = 256x256,
 * 2 = 256x128, 3 = 512x128, 4 = 256x128 4-wave, 5 = 256x256 4-stage), 8..15 = the same configurations on implicit-GEMM
 * convolution operands, 16 = attention_kernel, 17 = all other kernels, 18 / 19 = tile configuration 8 (64x64, the small-batch
 * forwards of a split-fp16 engine) on nn.Linear / convolution operands, 21 = tile configuration 9 (M 384 x N 192, split-fp16 nn.Linear), 24..31 = gemm_kernel
 * launches on fp16 + fp8 operand rows (D3R_DTYPE_F16F8 / _F16X2F8 engines: the transformer blocks' linears) by tile configuration.
 * Profiling adds event overhead: never enable it inside a timed region. */
#define D3R_MODEL_OPT_PROFILE 1
#define D3R_MODEL_OPT_TWO_STREAMS 2 /* 1 (default): decoder side 2 and head 2 run on an engine-owned second HIP stream, joined back
                                     * into the caller's stream before d3r_model_forward's work completes; 0: everything on the caller's stream */
#define D3R_MODEL_OPT_SPLIT_K 4 /* 0 (default): every launch sums K in one block and a batch is bit-identical to its one-pair calls (what the parity tests pin).
                                * 1 (or D3R_SPLITK=1 at create): small-batch forwards (split-fp16 engine) split the K sum of an nn.Linear of the 64 x 64 tile over 2-8 blocks
                                * where the launch then still fits three blocks per CU -- the one-pair call's fc2 (1536 x 1024 x 4096), the decoder's fc2. The partial tiles
                                * travel through agent-scope stores / loads and are added in slice order by the last block to arrive: deterministic (the same call
                                * twice is bit-equal), but a pair run alone then differs from the same pair inside a large batch at fp32-rounding level.
                                * Round 6 built it for the one-pair latency and measured a LOSS on MI355X (9.86 -> 9.99 ms): kept as an option, not the default. */
#define D3R_MODEL_OPT_GRAPH_MAX_PAIRS 3 /* n > 0: whole forwards (d3r_model_forward / _mixed / _packed) of at most n pairs are replayed as a hipGraph
                                         * from the third call with the same (B, image sizes, output layout) on: ~700 launches become one
                                         * graph launch + input / output copies through engine-owned staging buffers (bit-identical results).
                                         * The D3R_* environment probes that change the launch plan (D3R_GEMM_*, D3R_HEAD_FUSE, D3R_ATTN_*) are read when a graph is CAPTURED: set them before
                                         * the first captured call (a captured graph replays the plan it was captured with).
                                         * DEFAULT 0 = off (or D3R_GRAPH_MAX_PAIRS at create): measured on MI355X, one 512x384 pair per call
                                         * took 14.83 ms replayed vs 14.86 ms eager (before the small-problem GEMM tile: 10.4 ms now) -- the one-pair forward is bound by the dependent chain of
                                         * ~700 partially filled kernels on the GPU, not by the host's launch rate (profiles/r03_a/latency.log);
                                         * the replay only frees the host thread. 0 also drops the captured graphs */
int d3r_model_set_option(d3r_model* m, int option, int value);
/* depth_mode / conf_mode of the heads' postprocess: the constructor keywords of dust3r/model.py:58-62, evaluated by
 * dust3r/heads/postprocess.py:23-58. depth_mode: 0 'exp' (released checkpoints), 1 'linear', 2 'square' (the reference asserts depth
 * bounds away, postprocess.py:29-30: always (-inf, inf)); conf_mode: 0 'exp' -> vmin + min(exp(x), vmax - vmin), 1 'sigmoid' ->
 * (vmax - vmin) sigmoid(x) + vmin (finite bounds required). Default (0, 0, 1, +inf). Synchronises the device (captured graphs are dropped).
 * Errors: D3R_ERR_INVALID for an unknown mode or vmin >= vmax, as the reference raises ValueError(f'bad {mode=}'). */
int d3r_model_set_postprocess(d3r_model* m, int depth_mode, int conf_mode, float conf_vmin, float conf_vmax);
int d3r_model_profile_read(d3r_model* m, int kind, int* launches, double* ms, double* work);
/* launch `index` of the last profiled forward: class, GEMM shape (attention: batch*heads, queries, keys), ms, flops;
 * D3R_ERR_STATE past the last launch */
int d3r_model_profile_launch(d3r_model* m, int index, int* kind, int* M, int* N, int* K, double* ms, double* work);
/* debug/parity hook: copy an internal activation of the last forward to `out_f32` (device fp32).
 * what: 0 = encoder output after enc_norm [2B*N][enc_dim] (img1 batch then img2 batch) */
int d3r_model_debug_read(d3r_model* m, int what, float* out_f32, size_t max_elems, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Global aligner -- replaces the hot loop of cloud_opt.PointCloudOptimizer:
 *   forward            dust3r/cloud_opt/optimizer.py:188-201 (+ base_opt.py:143-195, commons.py:62-80)
 *   loss.backward()    autograd
 *   Adam step + lr     dust3r/cloud_opt/base_opt.py:326-366 (betas (0.9, 0.9), cosine/linear schedule)
 * Parameter tensors use the reference's own parameterisation and names (state_dict(trainable=True)):
 *   pw_poses [E][8] = quat XYZW, signed-log translation, log scale;  pw_adaptors [E][2] (frozen);
 *   im_poses [n][7];  im_depthmaps [n][max_area] log-depth;  im_focals [n] = focal_break*log(f);  im_pp [n][2] (frozen)
 * They live in caller-owned device memory and are updated IN PLACE; the handle borrows them and the weight
 * tensors until destroy. pred_* [E][max_area][3] and w_* [E][max_area] = conf_trf(conf) (zero in padding), fp32, are read ONCE at create:
 * the handle keeps its own block-interleaved copy [side][E][max_area / 256][x | y | z | w][256], so that a wave of the hot loop streams one
 * contiguous 4 KiB run per edge side (environment D3R_ALIGNER_LAYOUT=0 at creation: a planar [E][3][max_area] copy of pred_* and the caller's
 * w_* rows, which are then borrowed until destroy like the parameters). ei/ej/img_h/img_w are HOST arrays. Alignment: pw_poses, im_depthmaps, pred_*, w_* 16 bytes, pw_adaptors 8
 * (D3R_ERR_INVALID otherwise; any torch allocation satisfies it). create enqueues its one-off work (clearing the Adam state, the planar
 * copy of pred_*) on `stream` and does not synchronise the device: pred_* must be complete on that stream, and may be freed once it has
 * drained; run / loss_grad on the same stream need no further ordering, and on ANOTHER stream they first wait for an event the create
 * call recorded behind its work (no caller-side synchronisation either way).
 */
typedef struct d3r_aligner d3r_aligner;
#define D3R_SCHEDULE_COSINE 0
#define D3R_SCHEDULE_LINEAR 1
#define D3R_ALIGNER_OPT_DPP_REDUCE 1 /* 1 (default): DPP wave reduction; 0: __shfl_xor butterfly */
#define D3R_ALIGNER_OPT_RESET_ADAM 2   /* clear the Adam moments, ordered on the stream of the next d3r_aligner_run */
#define D3R_ALIGNER_OPT_OPTIMIZE_PP 3 /* 1: im_pp is a trainable parameter (PointCloudOptimizer(optimize_pp=True), optimizer.py:22,34) */
#define D3R_ALIGNER_OPT_OPTIMIZE_ADAPTORS 4 /* 1: pw_adaptors are trainable (allow_pw_adaptors=True, base_opt.py:49,92) */
#define D3R_ALIGNER_OPT_GENERIC_SMALL 5 /* 1: per-iteration pose / focal step by the strided-loop kernel (any E, n) instead of the
                                          one-edge-per-thread kernel used for E, n <= 1024 (tests compare the two) */

int d3r_aligner_create(d3r_aligner** out, int n_imgs, int n_edges, const int* ei, const int* ej, const int* img_h, const int* img_w,
                       int max_area, const float* pred_i, const float* pred_j, const float* w_i, const float* w_j, float* pw_poses,
                       float* pw_adaptors, float* im_poses, float* im_depthmaps, float* im_focals, float* im_pp, float base_scale,
                       float pw_break, float focal_break, int dist_l2, int norm_pw_scale, int opt_im_poses, int opt_im_focals,
                       int max_iters_per_run, void* stream);
int d3r_aligner_destroy(d3r_aligner* a);
int d3r_aligner_set_option(d3r_aligner* a, int option, int value);
/* `niter` iterations of global_alignment_iter; iteration k uses lr = schedule((iter0 + k) / niter_total).
 * losses_out (device fp32 [niter], may be NULL) receives the loss evaluated BEFORE each step, as float(loss) does
 * at base_opt.py:366 -- but without the reference's per-iteration host synchronisation. */
int d3r_aligner_run(d3r_aligner* a, int niter, int iter0, int niter_total, float lr_base, float lr_min, int schedule,
                    float* losses_out, void* stream);
/* The alignment loop over SEVERAL GPUs (one process per GPU; new: the reference's loop, base_opt.py:326-366, is single-device; SURVEY.md 8(e) names it as the
 * optional next step). Every rank holds the whole scene (what the all-gather of the forward hands over) and the replicated parameters, and owns a contiguous range
 * of IMAGES: d3r_aligner_set_image_range(first, count). One iteration is then
 *     d3r_aligner_step_begin   [derived matrices on the first iteration] + the main pass over the owned images (their edge sides, their depth maps and the fused
 *                              Adam step of those) + the fixed-order fp64 reduction of the partial records
 *     all-reduce (sum) of the buffer d3r_aligner_reduced_sums returns (fp64 [2 E + n][16] on the device), by the caller's collective library
 *     d3r_aligner_step_end     the pose / focal / pairwise-pose step (replicated: same inputs, same arithmetic on every rank) and the loss of iteration k
 * A partial record belongs to exactly one image, so the other ranks contribute exact zeros to every sum: the all-reduced sums -- and with them the trajectory --
 * equal the single-GPU iteration bit for bit, whatever the number of ranks. The log-depth maps (and their Adam moments) of an image are updated on its owner
 * only; the caller exchanges the owned rows of im_depthmaps when the loop is over. k, iter0, niter_total, lr_base, lr_min, schedule as in d3r_aligner_run
 * (k < max_iters_per_run); d3r_aligner_read_losses copies the losses of iterations 0 .. niter-1 of the current run to a device array. */
int d3r_aligner_set_image_range(d3r_aligner* a, int first_image, int n_images);
int d3r_aligner_step_begin(d3r_aligner* a, int k, int iter0, int niter_total, float lr_base, float lr_min, int schedule, void* stream);
int d3r_aligner_step_end(d3r_aligner* a, int k, int iter0, int niter_total, float lr_base, float lr_min, int schedule, void* stream);
int d3r_aligner_reduced_sums(d3r_aligner* a, void** device_ptr, long long* n_doubles);
int d3r_aligner_read_losses(d3r_aligner* a, int niter, float* losses_out, void* stream);
/* one forward/backward without a step (parity tests): loss[1] and the gradients w.r.t. each parameter tensor (any may be NULL);
 * g_im_pp [n][2]: principal-point parameters (optimizer.py:141-142, trained when optimize_pp=True); g_pw_adaptors [E][2]: pairwise
 * xy / z adaptors (base_opt.py:143-149, trained when allow_pw_adaptors=True) */
int d3r_aligner_loss_grad(d3r_aligner* a, float* loss, float* g_pw_poses, float* g_im_poses, float* g_im_depthmaps,
                          float* g_im_focals, float* g_im_pp, float* g_pw_adaptors, void* stream);

/* clean_pointcloud (dust3r/cloud_opt/base_opt.py:369-405): a point of image i that projects in front of image j's depth map (by more than
 * tol) onto a pixel more confident than itself gets its confidence clipped to bad_conf. conf [n][max_area] is updated in place with
 * the reference's sequential semantics (image i sees the cleaned confidences of images j < i); depth [n][max_area],
 * pts3d [n][max_area][3] (world points), intrinsics [n][9], world2cam [n][16] are DEVICE fp32; img_h / img_w are DEVICE int arrays.
 * Enqueues n launches on `stream`; no allocation, no synchronisation. */
int d3r_clean_pointcloud(int n_imgs, float* conf, const float* depth, const float* pts3d, const float* intrinsics, const float* world2cam,
                         const int* img_h_dev, const int* img_w_dev, int max_area, float tol, float bad_conf, void* stream);

/* exhaustive 3-D nearest neighbour: idx_out[q] = argmin_r |query[q] - ref[r]|^2 (lowest index on ties); query [n_query][3],
 * ref [n_ref][3] DEVICE fp32, idx_out DEVICE int32. The building block of find_reciprocal_matches (dust3r/utils/geometry.py:345-361,
 * two SciPy KD-tree queries in the reference; caller: visloc.py:105). */
int d3r_nearest_neighbors(const float* query, int n_query, const float* ref, int n_ref, int* idx_out, void* stream);

/* ---- scene bootstrap: the one-shot initialisation of the aligner (csrc/bootstrap.hip) -----------------------------------------
 * Replaces the per-edge / per-image host loops of dust3r/cloud_opt/init_im_poses.py:67-287 (roma.rigid_points_registration at
 * :220-223, estimate_focal -> post_process.py:40-56, fast_pnp -> cv2.solvePnPRansac at :247-287) and pair_viewer.py:30-76. All
 * pointer TABLES (`*_ptrs`) are DEVICE arrays of device addresses; sums come back as DEVICE fp64. Nothing synchronises. */

/* out[r] = mean(x[r][0..cols)), x row stride ld (multiple of 4 floats): the edge confidence scores of commons.py:20-25. */
int d3r_row_means(const float* x, int rows, int cols, int ld, float* out, void* stream);

/* Weighted similarity-registration moments of n_jobs independent cloud pairs in one launch. Job j: source cloud src_ptrs[j]
 * ([npix[j]][3] fp32), target cloud tgt_ptrs[j], weights wgt_ptrs[j] ([npix[j]]). out[j][17] = { W, Sx[3], Sy[3], Sxy[3][3] (x_a y_b),
 * Sxx } with S = sum_p w_p (.); the caller finishes Umeyama (centre, 3x3 SVD, scale) on the host. npix: DEVICE int array. */
size_t d3r_similarity_moments_workspace(int n_jobs, int max_points);
int d3r_similarity_moments(int n_jobs, const void* src_ptrs, const void* tgt_ptrs, const void* wgt_ptrs, const int* npix, int max_points,
                           void* workspace, double* out, void* stream);

/* Weiszfeld focal of n_jobs pointmaps (map_ptrs[j]: [H][W][3]; heights / widths DEVICE int arrays), principal point at the image
 * centre, `iterations` re-weighting rounds after the closed-form start (the reference uses 10): post_process.py:40-56. */
int d3r_weiszfeld_focals(int n_jobs, const void* map_ptrs, const int* heights, const int* widths, int iterations, float* focals, void* stream);

/* out[i][p] = z = rows[i] . (map_i[p], 1); with take_log: log(z), 0 where z <= 0 (depth.log().nan_to_num(neginf=0),
 * optimizer.py:112-117); zero padded to max_area: the aligner's im_depthmaps straight from each image's anchor pointmap. */
int d3r_anchor_depth(int n_imgs, const void* map_ptrs, const float* rows, const int* npix, int max_area, int take_log, float* out, void* stream);

/* PnP support, batched over images. `jobs`: DEVICE array of d3r_pnp_job_bytes()-sized records
 *   { const float* map [H][W][3]; const float* conf [H][W]; float G[12] (3x4 applied to the map: world points); float f, ppx, ppy,
 *     conf_thr; int H, W }   (points with conf <= conf_thr are ignored)
 * d3r_pnp_score: counts[j][h] = inliers of hypothesis h (world->camera [R|t], 12 floats; hypotheses laid out
 *   [j][d3r_pnp_max_hypotheses()][12]) at reprojection error < reproj_err px, in front of the camera.
 * d3r_pnp_sums: per job, over the inliers of poses[j], the Gauss-Newton sums of the reprojection error over (rotation increment about
 *   the camera origin, translation increment): J^T J (21, upper triangle row-major), J^T r (6), cost, inlier count;
 *   out[j][d3r_pnp_sum_count()] fp64. */
int d3r_pnp_job_bytes(void);
int d3r_pnp_max_hypotheses(void);
int d3r_pnp_sum_count(void);
size_t d3r_pnp_workspace(int n_jobs);
int d3r_pnp_score(int n_jobs, const void* jobs, const float* hypotheses, int n_hyp, float reproj_err, int* counts, void* stream);
int d3r_pnp_sums(int n_jobs, const void* jobs, const float* poses, float reproj_err, void* workspace, double* out, void* stream);

/* Host-only self test of the analytic gradient formulas shared with the kernels (no GPU touched; all pointers HOST).
 * Not a compute path: the product never calls it. */
int d3r_selftest_aligner_math_host(int n_imgs, int n_edges, const int* ei, const int* ej, int H, int W, const float* pred_i,
                                   const float* pred_j, const float* w_i, const float* w_j, const float* pw_poses,
                                   const float* im_poses, const float* im_depthmaps, const float* im_focals, float base_scale,
                                   float focal_break, double* loss, double* g_pw_poses, double* g_im_poses, double* g_im_depthmaps,
                                   double* g_im_focals);

#ifdef __cplusplus
}
#endif
#endif /* DUST3R_HIP_H */
