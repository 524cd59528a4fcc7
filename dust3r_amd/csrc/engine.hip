// dust3r_amd -- model engine: weight packing + forward orchestration for
// AsymmetricCroCo3DStereo.forward (reference dust3r/model.py:199-211).
//
// The engine owns its device memory (weights packed once at load time into the layouts the
// kernels want; one workspace arena grown on demand) and enqueues the whole forward -- ~600
// kernel launches -- from C++ on the caller's stream, so the Python host makes ONE ctypes call
// per batch. Data layout in HBM:
//   residual streams  fp32  [tokens][C]           (LayerNorm, residual adds stay fp32)
//   GEMM operands     dtype [tokens][C]           (bf16 / f16 / f32 / split-fp16 per d3r_model_config.dtype; with D3R_DTYPE_F16F8 the
//                                                  inputs and weights of the transformer blocks' linears are fp16 + fp8 rows, the rest split-fp16)
//   q, k              dtype [B][H][N][64]  (RoPE applied), v^T dtype [B][H][64][ldv]
//   DPT feature maps  dtype NHWC, channel stride padded to the next K-tile multiple
//   outputs           fp32  pts3d [B][H][W][3], conf [B][H][W]   (the reference's output layout)
#include <math.h>
#include <string.h>

#include <map>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/dust3r_hip.h"
#include "kernels.hpp"

using namespace d3r;

namespace {

inline int rup(int a, int b) { return (a + b - 1) / b * b; }

enum PackKind { PK_VEC, PK_MAT, PK_CONV, PK_CONVT, PK_CONVT_BIAS, PK_IGNORE };

struct Slot {
    PackKind kind = PK_IGNORE;
    void* dst = nullptr;      // device destination (matrix in dtype or fp32 vector)
    int rows = 0, cols = 0;   // expected logical (N, K) / vector length in rows
    int row_off = 0;          // first destination row (concatenated matrices / biases)
    int dst_cols = 0;         // destination row length (K, possibly padded)
    int cin = 0, cin_pad = 0, ksize = 1, cout_pad = 0;
    int dt = -1;              // layout of a packed matrix (-1: the model's dtype; the transformer blocks' matrices may be fp16 + fp8 rows)
    bool loaded = false, explicit_loaded = false;
    std::string mirror;       // dec_blocks.* -> dec_blocks2.* duplication
    struct Lin* fold = nullptr;   // PK_MAT: this matrix has the LayerNorm in front of it folded in (ln_fold engines): staged in fp32, packed by finalize_fold
    bool refold = false;          // PK_VEC: a LayerNorm weight / bias or the bias of a folded nn.Linear -- a new value makes the fold stale
    std::vector<struct Lin*> refold_lins;   // ... of these matrices (only THEY are re-folded: round 6)
    unsigned fold_bit = 0;        // PK_MAT with a fold: this slot's bit in Lin::want_mask (a Lin fed by several tensors -- projk | projv -- is complete when all were staged)
};

struct LNp { float* g = nullptr; float* b = nullptr; };
// dt: operand layout of this GEMM. ln != nullptr (ln_fold engines): W is packed as W diag(gamma), b_fold = b + W beta, ln_s[n] = sum_k of the packed row n
struct Lin { void* w = nullptr; float* b = nullptr; int N = 0, K = 0, n_pad = 0, n_rows = 0, dt = 0;
             const LNp* ln = nullptr; float* ln_s = nullptr; float* b_fold = nullptr; float* w32 = nullptr;
             bool fold_dirty = false; unsigned want_mask = 0, have_mask = 0; };      // fold bookkeeping per matrix: stale? which of its weight tensors sit in w32?
struct EncBlk { LNp n1, n2; Lin qkv, proj, fc1, fc2; };
struct DecBlk { LNp n1, n2, n3, ny; Lin qkv, proj, cq, ckv, cproj, fc1, fc2; };
struct ConvW { void* w = nullptr; float* b = nullptr; int Cout = 0, Cin = 0, cin_pad = 0, k = 1, n_pad = 0, n_rows = 0, K = 0; };
struct Refine { ConvW r1c1, r1c2, r2c1, r2c2; Lin outc; };
struct DptHead {
    Lin act1x1[4];
    Lin convt[2];      // ConvTranspose k=4 / k=2 as GEMM; N = k*k*cout_pad
    int convt_k[2] = {4, 2}, convt_coutp[2] = {0, 0};
    ConvW act3conv;    // Conv2d(768,768,3,s2,p1)
    ConvW layer_rn[4];
    Refine rn[4];      // rn[0] = refinenet1 ... rn[3] = refinenet4
    ConvW head0, head2;
    float* head4_w = nullptr; float* head4_b = nullptr;
    int cstride[4] = {0, 0, 0, 0};  // channel stride of the 4 reassembled maps
};

}  // namespace

// The engines of a process share their helper streams (per device and role; never destroyed). A HIP stream is bound to one of a few hardware queues (4 by default,
// GPU_MAX_HW_QUEUES) in creation order: with a stream per engine, the n-th engine of a process could find its second stream on the caller's own hardware queue -- the
// two decoder sides then run one after the other (measured on MI355X: the THIRD engine created in a process took 13.1 ms per one-pair call against 10.1 ms for the
// first, second and fourth; profiles/r05_y/ln_inline3.log). One process-wide stream per role keeps every engine on the placement the first one got. Stream order only
// adds dependencies; what it costs: engines on one device are NOT independent any more -- their side-stream halves serialise, and the enqueue of a forward is a
// process-wide critical section (run_phases: a capture of one engine must not see another engine's launches). Documented in include/dust3r_hip.h (d3r_model_create).
static hipStream_t shared_stream(int role) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, hipStream_t> pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = pool.find({dev, role});
    if (it != pool.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    pool[{dev, role}] = s;
    return s;
}

struct d3r_model {
    d3r_model_config cfg;
    int dt = 0, ktile = 64;
    int bdt = 0;              // operand layout of the transformer blocks' linears (= dt, or D3R_F16F8 on top of dt = D3R_F16X3)
    // LayerNorm folded into the GEMMs around it (round 5; split-fp16 engines; kernels.hpp GemmParams::ln_*): norm1 / norm2 of the encoder blocks
    // and norm1 / norm2 / norm3 / norm_y of the decoder blocks are not launched -- the fp32-residual epilogue in front of them also stores the RAW
    // typed rows and per-row partial sums, the nn.Linear behind them is packed as W diag(gamma) and applies rstd / mean in its epilogue.
    // enc_norm / dec_norm (their outputs leave the engine or feed the heads) stay kernels. D3R_LN_FOLD=0|1 at model creation.
    int ln_inline_rows = 3072; // folded LayerNorms of at most this many rows (calls of one or two pairs; the decoder sides of four) get their statistics in the consumer GEMM's prologue
                              // (GemmParams::ln_part_in; kernels.hpp ln_row_stats = ln_finalize_kernel's arithmetic with four adjacent lanes per row) instead of by an ln_finalize
                              // launch: bit-identical, one pair 10.32 -> 10.10 ms, two 15.27 -> 15.09, four and eight equal (profiles/r05_y/ln_inline3.log; the first version, 32
                              // lanes per row with five fp64 exchange levels in front of every tile, was SLOWER: r05_v). D3R_LN_INLINE_ROWS=n at creation; 0 = always launch.
    int enc_split_max = 0;    // encoder of calls with <= this many images (two views): the two views as two concurrent chains on the two streams (D3R_ENC_SPLIT)
    bool ln_fold = false, fold_dirty = false;
    std::vector<Lin*> fold_lins;
    std::unordered_map<std::string, Slot> slots;
    std::vector<void*> allocs;
    size_t weight_bytes = 0;
    Lin patch;
    std::vector<EncBlk> enc;
    LNp enc_norm, dec_norm;
    Lin dec_embed;
    std::vector<DecBlk> dec[2];
    DptHead dpt[2];
    Lin lin_head[2];
    float* rope_table = nullptr;
    void* zero_page = nullptr;
    // split-K of the small-batch forwards (kernels.hpp GemmParams::splitk): partial-tile slabs and arrival counters, one set per stream of the forward (main, side)
    static constexpr size_t SK_SLAB_FLOATS = (size_t)8 << 20; static constexpr int SK_CNT = 1024;
    float* sk_slab[2] = {nullptr, nullptr}; unsigned* sk_cnt[2] = {nullptr, nullptr};
    bool splitk_on = true;       // D3R_SPLITK=0 at creation: never (every launch then sums K in one block: a batch is bit-identical to its one-pair calls)
    void* ws = nullptr; size_t ws_bytes = 0;
    void* stage = nullptr; size_t stage_bytes = 0;   // load-time staging of host tensors
    // the two decoder sides (and the two heads) are independent inside a layer: side 1 runs on this second stream
    hipStream_t side = nullptr;
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    bool two_streams = true;
    // Cross-attention K | V of a decoder block read the OTHER side's previous-layer output, not the side's own chain: with at most kv_ahead_rows rows per side (calls of
    // a few pairs: a chain of launches that each fill a fraction of the chip) they run on a third / fourth stream beside the block's self attention, into their own K / V^T
    // buffers, and the side's stream waits for them in front of the cross attention. Same kernels on the same rows: bit-identical. D3R_DEC_KV_AHEAD=rows (0 = off).
    hipStream_t kvs[2] = {nullptr, nullptr};
    hipEvent_t ev_kv_go[2] = {nullptr, nullptr}, ev_kv_done[2] = {nullptr, nullptr};
    int kv_ahead_rows = 0;
    PostMode post;                          // depth_mode / conf_mode of the heads (d3r_model_set_postprocess; default = the released checkpoints')
    int out_pstride = 3, out_cstride = 1;   // output element strides between pixels (8, 8 while d3r_model_forward_packed runs)
    // ---- hipGraph replay of small-batch forwards (d3r_model_forward* with B <= graph_max_pairs) ------------------------------------
    // One pair per call (dust3r/demo.py:156 batch_size=1, visloc.py:88) is ~700 launches of kernels that each fill a fraction of the
    // chip: the host-side enqueue and the two-stream event traffic, not the GPU, set the pace. The second call with the same
    // (entry point, B, image sizes, output layout, stream plan) is stream-captured on an engine-owned stream into a graph that works on
    // engine-owned input / output staging buffers; every later call is [copy inputs -> staging] + hipGraphLaunch + [copy staging ->
    // outputs] on the caller's stream. Same kernels, same order: bit-identical to the eager call.
    struct GraphKey {
        int B, H1, W1, H2, W2, pstride, two;
        bool operator==(const GraphKey& o) const { return B == o.B && H1 == o.H1 && W1 == o.W1 && H2 == o.H2 && W2 == o.W2 && pstride == o.pstride && two == o.two; }
    };
    struct GraphEntry { GraphKey key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int seen = 0; float* io = nullptr; size_t io_bytes = 0; };
    std::vector<GraphEntry> graphs;
    hipStream_t cap = nullptr;              // capture origin (the legacy default stream cannot be captured)
    int graph_max_pairs = 0;                // D3R_MODEL_OPT_GRAPH_MAX_PAIRS; 0 = off (default: measured no latency gain on MI355X, see the header)
    long graph_replays = 0;                 // statistics (tests)
    void drop_graphs() {
        for (auto& g : graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            if (g.graph) (void)hipGraphDestroy(g.graph);
            if (g.io) (void)hipFree(g.io);
        }
        graphs.clear();
    }
    // last forward (debug hook)
    const void* last_encn = nullptr; size_t last_encn_elems = 0;
    // optional per-launch HIP-event timing (d3r_model_set_option(D3R_MODEL_OPT_PROFILE)); off in timed runs
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    struct ProfRec { int kind; double work; int M, N, K; };
    std::vector<ProfRec> prof_rec;

    void* dalloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, bytes ? bytes : 16);
        allocs.push_back(p);
        weight_bytes += bytes;
        return p;
    }
};

namespace {

// ---- slot registration ---------------------------------------------------------------------------------
bool reg_vec(d3r_model* m, const std::string& key, float** dst, int n, int n_alloc = 0) {
    if (!*dst) {
        *dst = (float*)m->dalloc((size_t)(n_alloc ? n_alloc : n) * sizeof(float));
        if (!*dst) return false;
    }
    Slot s; s.kind = PK_VEC; s.dst = *dst; s.rows = n;
    m->slots[key] = s;
    return true;
}
bool reg_vec_at(d3r_model* m, const std::string& key, float* base, int off, int n) {
    Slot s; s.kind = PK_VEC; s.dst = base + off; s.rows = n;
    m->slots[key] = s;
    return true;
}
bool alloc_lin(d3r_model* m, Lin& L, int N, int K, bool bias = true, int dt = -1) {
    L.N = N; L.K = K; L.n_pad = rup(N, 128); L.n_rows = rup(N, 256);   // rows up to a 256-wide tile stay zero
    L.dt = dt < 0 ? m->dt : dt;
    L.w = m->dalloc((size_t)L.n_rows * K * wgt_bytes(L.dt));      // 2.5-unit rows: five bytes per element
    L.b = bias ? (float*)m->dalloc((size_t)L.n_rows * sizeof(float)) : nullptr;
    return L.w && (!bias || L.b);
}
void reg_mat(d3r_model* m, const std::string& key, const Lin& L, int rows, int row_off) {
    Slot s; s.kind = PK_MAT; s.dst = L.w; s.rows = rows; s.cols = L.K; s.row_off = row_off; s.dst_cols = L.K; s.dt = L.dt;
    m->slots[key] = s;
}
bool reg_linear(d3r_model* m, const std::string& prefix, Lin& L, int N, int K, int dt = -1) {
    if (!alloc_lin(m, L, N, K, true, dt)) return false;
    reg_mat(m, prefix + ".weight", L, N, 0);
    reg_vec_at(m, prefix + ".bias", L.b, 0, N);
    return true;
}
bool reg_ln(d3r_model* m, const std::string& prefix, LNp& p, int C) {
    return reg_vec(m, prefix + ".weight", &p.g, C) && reg_vec(m, prefix + ".bias", &p.b, C);
}
bool reg_conv(d3r_model* m, const std::string& prefix, ConvW& c, int Cout, int Cin, int k, bool bias) {
    c.Cout = Cout; c.Cin = Cin; c.k = k; c.cin_pad = rup(Cin, m->ktile); c.n_pad = rup(Cout, 128); c.n_rows = rup(Cout, 256); c.K = k * k * c.cin_pad;
    c.w = m->dalloc((size_t)c.n_rows * c.K * dt_bytes(m->dt));
    if (!c.w) return false;
    Slot s; s.kind = PK_CONV; s.dst = c.w; s.rows = Cout; s.cin = Cin; s.cin_pad = c.cin_pad; s.ksize = k; s.dst_cols = c.K;
    m->slots[prefix + ".weight"] = s;
    if (bias) {
        c.b = (float*)m->dalloc((size_t)c.n_rows * sizeof(float));
        if (!c.b) return false;
        reg_vec_at(m, prefix + ".bias", c.b, 0, Cout);
    }
    return true;
}
bool reg_convt(d3r_model* m, const std::string& prefix, Lin& L, int Cin, int Cout, int k, int cin_pad, int cout_pad) {
    L.N = k * k * cout_pad; L.K = cin_pad; L.n_pad = rup(L.N, 128); L.n_rows = rup(L.N, 256); L.dt = m->dt;
    L.w = m->dalloc((size_t)L.n_rows * L.K * dt_bytes(m->dt));
    L.b = (float*)m->dalloc((size_t)L.n_rows * sizeof(float));
    if (!L.w || !L.b) return false;
    Slot s; s.kind = PK_CONVT; s.dst = L.w; s.rows = Cin; s.cols = Cout; s.ksize = k; s.cin_pad = cin_pad; s.cout_pad = cout_pad; s.dst_cols = L.K;
    m->slots[prefix + ".weight"] = s;
    Slot b; b.kind = PK_CONVT_BIAS; b.dst = L.b; b.rows = Cout; b.ksize = k; b.cout_pad = cout_pad;
    m->slots[prefix + ".bias"] = b;
    return true;
}
// the nn.Linear `L` (weight slots `wkeys`, bias slots `bkeys`) consumes LayerNorm `ln` (slots prefix `lnkey`): fold it (ln_fold engines)
bool reg_fold(d3r_model* m, Lin& L, const LNp& ln, const std::string& lnkey, std::initializer_list<std::string> wkeys, std::initializer_list<std::string> bkeys) {
    if (!m->ln_fold) return true;
    L.ln = &ln;
    L.ln_s = (float*)m->dalloc((size_t)L.n_rows * sizeof(float));
    L.b_fold = (float*)m->dalloc((size_t)L.n_rows * sizeof(float));
    if (!L.ln_s || !L.b_fold) return false;
    unsigned bit = 1;
    for (auto& k : wkeys) { m->slots[k].fold = &L; m->slots[k].fold_bit = bit; L.want_mask |= bit; bit <<= 1; }
    for (auto& k : bkeys) { m->slots[k].refold = true; m->slots[k].refold_lins.push_back(&L); }
    for (const char* sfx : {".weight", ".bias"}) { Slot& ls = m->slots[lnkey + sfx]; ls.refold = true; ls.refold_lins.push_back(&L); }
    m->fold_lins.push_back(&L);
    return true;
}
void reg_ignore(d3r_model* m, const std::string& key) { Slot s; s.kind = PK_IGNORE; s.loaded = true; m->slots[key] = s; }

bool build_slots(d3r_model* m) {
    const d3r_model_config& c = m->cfg;
    const int Ce = c.enc_embed_dim, Cd = c.dec_embed_dim, ps = c.patch_size;
    const int bd = m->bdt;    // the 24 + 2 x 12 transformer blocks' matrices; patch embedding, decoder_embed and the heads keep m->dt
    if (!reg_linear(m, "patch_embed.proj", m->patch, Ce, 3 * ps * ps)) return false;
    reg_ignore(m, "mask_token");
    m->enc.resize(c.enc_depth);
    for (int l = 0; l < c.enc_depth; ++l) {
        const std::string p = "enc_blocks." + std::to_string(l);
        EncBlk& b = m->enc[l];
        if (!reg_ln(m, p + ".norm1", b.n1, Ce) || !reg_ln(m, p + ".norm2", b.n2, Ce) || !reg_linear(m, p + ".attn.qkv", b.qkv, 3 * Ce, Ce, bd) ||
            !reg_linear(m, p + ".attn.proj", b.proj, Ce, Ce, bd) || !reg_linear(m, p + ".mlp.fc1", b.fc1, 4 * Ce, Ce, bd) ||
            !reg_linear(m, p + ".mlp.fc2", b.fc2, Ce, 4 * Ce, bd))
            return false;
        if (!reg_fold(m, b.qkv, b.n1, p + ".norm1", {p + ".attn.qkv.weight"}, {p + ".attn.qkv.bias"}) ||
            !reg_fold(m, b.fc1, b.n2, p + ".norm2", {p + ".mlp.fc1.weight"}, {p + ".mlp.fc1.bias"}))
            return false;
    }
    if (!reg_ln(m, "enc_norm", m->enc_norm, Ce) || !reg_ln(m, "dec_norm", m->dec_norm, Cd) || !reg_linear(m, "decoder_embed", m->dec_embed, Cd, Ce))
        return false;
    for (int side = 0; side < 2; ++side) {
        m->dec[side].resize(c.dec_depth);
        for (int l = 0; l < c.dec_depth; ++l) {
            const std::string p = std::string(side ? "dec_blocks2." : "dec_blocks.") + std::to_string(l);
            DecBlk& b = m->dec[side][l];
            if (!reg_ln(m, p + ".norm1", b.n1, Cd) || !reg_ln(m, p + ".norm2", b.n2, Cd) || !reg_ln(m, p + ".norm3", b.n3, Cd) ||
                !reg_ln(m, p + ".norm_y", b.ny, Cd) || !reg_linear(m, p + ".attn.qkv", b.qkv, 3 * Cd, Cd, bd) ||
                !reg_linear(m, p + ".attn.proj", b.proj, Cd, Cd, bd) || !reg_linear(m, p + ".cross_attn.projq", b.cq, Cd, Cd, bd) ||
                !reg_linear(m, p + ".cross_attn.proj", b.cproj, Cd, Cd, bd) || !reg_linear(m, p + ".mlp.fc1", b.fc1, 4 * Cd, Cd, bd) ||
                !reg_linear(m, p + ".mlp.fc2", b.fc2, Cd, 4 * Cd, bd))
                return false;
            // projk and projv share their input: packed as one (2 Cd, Cd) matrix
            if (!alloc_lin(m, b.ckv, 2 * Cd, Cd, true, bd)) return false;
            reg_mat(m, p + ".cross_attn.projk.weight", b.ckv, Cd, 0);
            reg_mat(m, p + ".cross_attn.projv.weight", b.ckv, Cd, Cd);
            reg_vec_at(m, p + ".cross_attn.projk.bias", b.ckv.b, 0, Cd);
            reg_vec_at(m, p + ".cross_attn.projv.bias", b.ckv.b, Cd, Cd);
            if (!reg_fold(m, b.qkv, b.n1, p + ".norm1", {p + ".attn.qkv.weight"}, {p + ".attn.qkv.bias"}) ||
                !reg_fold(m, b.cq, b.n2, p + ".norm2", {p + ".cross_attn.projq.weight"}, {p + ".cross_attn.projq.bias"}) ||
                !reg_fold(m, b.ckv, b.ny, p + ".norm_y", {p + ".cross_attn.projk.weight", p + ".cross_attn.projv.weight"}, {p + ".cross_attn.projk.bias", p + ".cross_attn.projv.bias"}) ||
                !reg_fold(m, b.fc1, b.n3, p + ".norm3", {p + ".mlp.fc1.weight"}, {p + ".mlp.fc1.bias"}))
                return false;
        }
    }
    // dec_blocks.* duplicates into dec_blocks2.* until an explicit dec_blocks2 key arrives (model.py:91-98)
    {
        std::vector<std::string> keys;
        for (auto& kv : m->slots)
            if (kv.first.rfind("dec_blocks.", 0) == 0) keys.push_back(kv.first);
        for (auto& k : keys) m->slots[k].mirror = "dec_blocks2." + k.substr(strlen("dec_blocks."));
    }
    for (int hd = 0; hd < 2; ++hd) {
        const std::string hp = "downstream_head" + std::to_string(hd + 1);
        if (c.head_type == 0) {
            if (!reg_linear(m, hp + ".proj", m->lin_head[hd], 4 * ps * ps, Cd)) return false;
            continue;
        }
        DptHead& D = m->dpt[hd];
        const std::string dp = hp + ".dpt";
        const int ld[4] = {96, 192, 384, 768};
        const int din[4] = {Ce, Cd, Cd, Cd};
        for (int i = 0; i < 4; ++i) {
            D.cstride[i] = rup(ld[i], m->ktile);
            if (!alloc_lin(m, D.act1x1[i], ld[i], din[i])) return false;
            reg_mat(m, dp + ".act_postprocess." + std::to_string(i) + ".0.weight", D.act1x1[i], ld[i], 0);
            reg_vec_at(m, dp + ".act_postprocess." + std::to_string(i) + ".0.bias", D.act1x1[i].b, 0, ld[i]);
        }
        D.convt_coutp[0] = D.cstride[0]; D.convt_coutp[1] = D.cstride[1];
        if (!reg_convt(m, dp + ".act_postprocess.0.1", D.convt[0], ld[0], ld[0], 4, D.cstride[0], D.cstride[0])) return false;
        if (!reg_convt(m, dp + ".act_postprocess.1.1", D.convt[1], ld[1], ld[1], 2, D.cstride[1], D.cstride[1])) return false;
        if (!reg_conv(m, dp + ".act_postprocess.3.1", D.act3conv, ld[3], ld[3], 3, true)) return false;
        for (int i = 0; i < 4; ++i) {
            if (!reg_conv(m, dp + ".scratch.layer_rn." + std::to_string(i), D.layer_rn[i], 256, ld[i], 3, false)) return false;
            reg_ignore(m, dp + ".scratch.layer" + std::to_string(i + 1) + "_rn.weight");  // alias of layer_rn.i
            const std::string rp = dp + ".scratch.refinenet" + std::to_string(i + 1);
            Refine& R = D.rn[i];
            if (!reg_conv(m, rp + ".resConfUnit1.conv1", R.r1c1, 256, 256, 3, true) || !reg_conv(m, rp + ".resConfUnit1.conv2", R.r1c2, 256, 256, 3, true) ||
                !reg_conv(m, rp + ".resConfUnit2.conv1", R.r2c1, 256, 256, 3, true) || !reg_conv(m, rp + ".resConfUnit2.conv2", R.r2c2, 256, 256, 3, true))
                return false;
            if (!alloc_lin(m, R.outc, 256, 256)) return false;
            reg_mat(m, rp + ".out_conv.weight", R.outc, 256, 0);
            reg_vec_at(m, rp + ".out_conv.bias", R.outc.b, 0, 256);
        }
        // refinenet4.resConfUnit1 exists in the checkpoint but is never evaluated (single input): still required keys
        if (!reg_conv(m, dp + ".head.0", D.head0, 128, 256, 3, true) || !reg_conv(m, dp + ".head.2", D.head2, 128, 128, 3, true)) return false;
        if (!reg_vec(m, dp + ".head.4.weight", &D.head4_w, 4 * 128) || !reg_vec(m, dp + ".head.4.bias", &D.head4_b, 4)) return false;
    }
    return true;
}

// `data` is a DEVICE fp32 tensor in the checkpoint's (PyTorch) layout; the conversion to the engine's dtype and
// layout runs on the GPU (elementwise.hip: pack_weight_kernel), ordered on the default stream.
int pack_slot(d3r_model* m, Slot& s, const float* data, int ndim, const int64_t* shape) {
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    const size_t eb = dt_bytes(m->dt);
    PackParams pp;
    pp.src = data; pp.numel = numel; pp.dst = s.dst; pp.dst_cols = s.dst_cols;
    switch (s.kind) {
        case PK_IGNORE: return D3R_OK;
        case PK_VEC:
            if (numel != (size_t)s.rows) return D3R_ERR_SHAPE;
            if (s.refold) { m->fold_dirty = true; for (Lin* L : s.refold_lins) L->fold_dirty = true; }
            return hipMemcpyAsync(s.dst, data, numel * sizeof(float), hipMemcpyDeviceToDevice, nullptr) == hipSuccess ? D3R_OK : D3R_ERR_ALLOC;
        case PK_MAT:
            if (ndim < 2 || shape[0] != s.rows || numel != (size_t)s.rows * s.cols) return D3R_ERR_SHAPE;
            pp.kind = PACK_MAT; pp.cols = s.cols; pp.row_off = s.row_off;
            if (s.fold) {       // the LayerNorm in front of this matrix is folded into it: keep the fp32 rows until finalize_fold packs W diag(gamma)
                Lin& L = *s.fold;
                if (!L.w32) {
                    if (hipMalloc((void**)&L.w32, (size_t)L.N * L.K * sizeof(float)) != hipSuccess) { L.w32 = nullptr; return D3R_ERR_ALLOC; }
                    L.have_mask = 0;            // a fresh staging copy holds nothing yet
                }
                L.have_mask |= s.fold_bit;
                L.fold_dirty = true;
                m->fold_dirty = true;
                return hipMemcpyAsync(L.w32 + (size_t)s.row_off * L.K, data, numel * sizeof(float), hipMemcpyDeviceToDevice, nullptr) == hipSuccess ? D3R_OK : D3R_ERR_ALLOC;
            }
            break;
        case PK_CONV:
            if (ndim != 4 || shape[0] != s.rows || shape[1] != s.cin || shape[2] != s.ksize || shape[3] != s.ksize) return D3R_ERR_SHAPE;
            pp.kind = PACK_CONV; pp.cin = s.cin; pp.cin_pad = s.cin_pad; pp.ksize = s.ksize; pp.kslice_major = conv_k_slice_major() ? 1 : 0;
            break;
        case PK_CONVT:
            if (ndim != 4 || shape[0] != s.rows || shape[1] != s.cols || shape[2] != s.ksize || shape[3] != s.ksize) return D3R_ERR_SHAPE;
            pp.kind = PACK_CONVT; pp.cols = s.cols; pp.ksize = s.ksize; pp.cout_pad = s.cout_pad;
            break;
        case PK_CONVT_BIAS:
            if (numel != (size_t)s.rows) return D3R_ERR_SHAPE;
            return launch_pack_convt_bias(data, (float*)s.dst, s.rows, s.cout_pad, s.ksize * s.ksize, nullptr) == hipSuccess ? D3R_OK : D3R_ERR_LAUNCH;
    }
    (void)eb;
    return launch_pack_weight(s.dt < 0 ? m->dt : s.dt, pp, nullptr) == hipSuccess ? D3R_OK : D3R_ERR_LAUNCH;
}

// ---- launch helpers ------------------------------------------------------------------------------------------
struct Ctx {
    d3r_model* m;
    hipStream_t st;
    int rc = D3R_OK;
    hipStream_t st0 = nullptr;    // the caller's stream (st moves between it and the engine's helper streams)
    void chk(hipError_t e) { if (e != hipSuccess && rc == D3R_OK) rc = 1000 + (int)e; }
    // profiling: one event BEFORE every launch; a launch's duration is event[i+1] - event[i]
    void mark(int kind, double work, int M = 0, int N = 0, int K = 0) {
        if (!m->prof_on) return;
        const size_t i = m->prof_rec.size();
        if (i >= m->prof_ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            m->prof_ev.push_back(e);
        }
        (void)hipEventRecord(m->prof_ev[i], st);
        m->prof_rec.push_back({kind, work, M, N, K});
    }
};
// profile record kinds: GEMM-kernel launches carry their tile configuration: kind = cfg (0..7) + 8 for implicit-GEMM convolution
enum { PRF_GEMM = 0, PRF_CONV = 8, PRF_ATTN = 16, PRF_OTHER = 17, PRF_GEMM_64 = 18, PRF_CONV_64 = 19, PRF_END = 20, PRF_GEMM_384 = 21, PRF_GEMM_P4 = 22, PRF_GEMM_F8 = 24 };   // 24..31: fp16 + fp8 linear launches by tile configuration
// tile configuration 8 (the 64 x 64 tile of the small-batch forwards, split-fp16 only) has its own two kinds behind the 0..7 ranges
static inline int prf_kind(int base, int cfg) { return cfg == GEMM_CFG_P4 ? PRF_GEMM_P4 : cfg == GEMM_CFG_384x192 ? PRF_GEMM_384 : (cfg == GEMM_CFG_64 || cfg == GEMM_CFG_96x64) ? (base == PRF_CONV ? PRF_CONV_64 : PRF_GEMM_64) : base + cfg; }
#define D3R_OTHER(call) do { c.mark(PRF_OTHER, 0.0); c.chk(call); } while (0)

// folded LayerNorm (kernels.hpp GemmParams::ln_*): `stats` = the consumer side (rstd, -mean rstd of the input rows; the Lin carries column sums and
// folded bias), `part` = the producer side (partial sums of the rows this launch stores, next to their raw typed copy out2)
struct LnStats { const float* rstd = nullptr; const float* nmr = nullptr; const float* part_in = nullptr; };   // part_in: small problem, the consumer forms rstd / nmr itself (no ln_finalize launch)
void gemm_linear(Ctx& c, const void* act, int lda, const Lin& L, int M, int epi, void* out, int ldo, const void* res1 = nullptr,
                 void* out2 = nullptr, int ldo2 = 0, int n_store = -1, int flags = 0, LnStats stats = LnStats(), float* part = nullptr) {
    GemmParams p;
    p.act = act; p.lda = lda; p.wgt = L.w; p.bias = L.b; p.M = M; p.K = L.K; p.n_pad = L.n_pad; p.n_rows = L.n_rows;
    p.n_store = n_store >= 0 ? n_store : L.N;
    p.epi = epi; p.out = out; p.ldo = ldo; p.res1 = res1; p.ldr = ldo; p.out2 = out2; p.ldo2 = ldo2; p.flags = flags;
    if (stats.rstd) { p.ln_rstd = stats.rstd; p.ln_nmr = stats.nmr; p.ln_colsum = L.ln_s; p.bias = L.b_fold; p.ln_part_in = stats.part_in; p.ln_inv_c = 1.0f / (float)L.K; }
    p.ln_part = part;
    {   // split-K buffers of this launch's stream (launch_gemm decides whether the launch splits; never while profiling: one event per launch)
        const int sset = c.st == c.st0 ? 0 : (c.st == c.m->side ? 1 : -1);
        if (c.m->splitk_on && sset >= 0 && c.m->sk_slab[sset] && !c.m->prof_on) {
            p.splitk = 0; p.sk_slab = c.m->sk_slab[sset]; p.sk_cnt = c.m->sk_cnt[sset]; p.sk_slab_floats = d3r_model::SK_SLAB_FLOATS; p.sk_cnt_n = d3r_model::SK_CNT;
        }
    }
    c.mark(prf_kind((L.dt == D3R_F16F8 || L.dt == D3R_F16X2F8) ? PRF_GEMM_F8 : PRF_GEMM, gemm_pick_config(p, L.dt)), 2.0 * M * (double)L.N * L.K, M, L.N, L.K);
    c.chk(launch_gemm(L.dt, p, c.st));
}

void gemm_heads(Ctx& c, const void* act, int lda, const Lin& L, int M, int head_c, int nreg, const int* kinds, void* const* dsts, int heads,
                int ntok, int tok_w, int ldv, LnStats stats = LnStats()) {
    GemmParams p;
    p.act = act; p.lda = lda; p.wgt = L.w; p.bias = L.b; p.M = M; p.K = L.K; p.n_pad = L.n_pad; p.n_rows = L.n_rows; p.n_store = L.N;
    if (stats.rstd) { p.ln_rstd = stats.rstd; p.ln_nmr = stats.nmr; p.ln_colsum = L.ln_s; p.bias = L.b_fold; p.ln_part_in = stats.part_in; p.ln_inv_c = 1.0f / (float)L.K; }
    p.epi = EPI_HEADS; p.head_c = head_c;
    for (int i = 0; i < nreg; ++i) { p.head_kind[i] = kinds[i]; p.head_dst[i] = dsts[i]; }
    p.heads = heads; p.ntok = ntok; p.tok_w = tok_w; p.ldv = ldv; p.rope_table = c.m->rope_table;
    if constexpr (kProbes) {    // measurement aids, results INVALID (tools/launch_table.py): what the V^T scatter / the RoPE of the attention projections' epilogue cost --
        // D3R_PROBE_V_PLAIN=1 stores the V region row-major like K (the GEMM side of a design whose attention kernel transposes V with ds_read_b64_tr_b16),
        // D3R_PROBE_QK_PLAIN=1 stores q / k without the rotation
        const char* ev = probe_env("D3R_PROBE_V_PLAIN");
        const char* eq = probe_env("D3R_PROBE_QK_PLAIN");
        for (int i = 0; i < nreg; ++i) {
            if (ev && ev[0] == '1' && p.head_kind[i] == HEAD_VT) p.head_kind[i] = HEAD_PLAIN;
            if (eq && eq[0] == '1' && p.head_kind[i] == HEAD_ROPE) p.head_kind[i] = HEAD_PLAIN;
        }
    }
    c.mark(prf_kind((L.dt == D3R_F16F8 || L.dt == D3R_F16X2F8) ? PRF_GEMM_F8 : PRF_GEMM, gemm_pick_config(p, L.dt)), 2.0 * M * (double)L.N * L.K, M, L.N, L.K);
    c.chk(launch_gemm(L.dt, p, c.st));
}

void conv(Ctx& c, const void* in, int B, int Hin, int Win, int cstride, const ConvW& w, int stride, int pad, void* out, int ldo,
          int flags, const void* res1 = nullptr, const void* res2 = nullptr, void* out2 = nullptr, int n_store = -1) {
    GemmParams p;
    p.amode = AMODE_CONV; p.act = in; p.wgt = w.w; p.bias = w.b;
    p.Hin = Hin; p.Win = Win; p.Cin = w.cin_pad; p.cstride = cstride; p.ksize = w.k; p.stride = stride; p.pad = pad;
    p.Hout = (Hin + 2 * pad - w.k) / stride + 1; p.Wout = (Win + 2 * pad - w.k) / stride + 1;
    p.M = B * p.Hout * p.Wout; p.K = w.K; p.n_pad = w.n_pad; p.n_rows = w.n_rows; p.n_store = n_store >= 0 ? n_store : w.Cout;
    p.zero_page = c.m->zero_page;
    p.epi = EPI_T; p.flags = flags; p.out = out; p.ldo = ldo; p.res1 = res1; p.res2 = res2; p.ldr = ldo; p.out2 = out2; p.ldo2 = ldo;
    c.mark(prf_kind(PRF_CONV, gemm_pick_config(p, c.m->dt)), 2.0 * p.M * (double)w.Cout * w.k * w.k * w.Cin, p.M, w.Cout, w.k * w.k * w.Cin);
    c.chk(launch_gemm(c.m->dt, p, c.st));
}

// 3x3 convolution (stride 1, pad 1) + ReLU + Conv2d(Cout, 4, 1) + postprocess in one launch (EPI_HEAD4); false (nothing launched) when
// the problem does not get a tile shape whose waves hold all output channels
bool conv_head4(Ctx& c, const void* in, int B, int Hin, int Win, int cstride, const ConvW& w, const float* w4, const float* b4,
                float* pts, float* conf, int pstride, int cstride_conf) {
    GemmParams p;
    p.amode = AMODE_CONV; p.act = in; p.wgt = w.w; p.bias = w.b;
    p.Hin = Hin; p.Win = Win; p.Cin = w.cin_pad; p.cstride = cstride; p.ksize = w.k; p.stride = 1; p.pad = 1;
    p.Hout = Hin; p.Wout = Win;
    p.M = B * p.Hout * p.Wout; p.K = w.K; p.n_pad = w.n_pad; p.n_rows = w.n_rows; p.n_store = w.Cout;
    p.zero_page = c.m->zero_page;
    p.epi = EPI_HEAD4; p.flags = GF_RELU; p.out = pts; p.ldo = pstride; p.out2 = conf; p.ldo2 = cstride_conf; p.res1 = w4; p.res2 = b4; p.post = c.m->post;
    if (w.k != 3 || w.Cout > 128 || w.Cout % 4 != 0) return false;
    int cfg = gemm_pick_config(p, c.m->dt);
    if (cfg != GEMM_CFG_512x128) {      // fewer pixels (one or two images): the same tile by four waves stacked along m. ANY pixel count below the 512 x 128
        p.force_cfg = GEMM_CFG_256x128R;   // tile's takes this shape (round 5): with the heuristic's 128 x 128 choice for < 512 tiles the head of ONE 512 x 160
        cfg = gemm_pick_config(p, c.m->dt);   // pair took the two-kernel route and a batch of three the fused one -- 3e-6 apart, the only place where a batch was not
    }                                      // bit-equal to its one-pair calls (tests/test_timed_configs_gpu.py::test_full_size_released_resolutions_match_oracle)
    if (cfg != GEMM_CFG_512x128 && cfg != GEMM_CFG_256x128R) return false;
    c.mark(prf_kind(PRF_CONV, cfg), 2.0 * p.M * (double)w.Cout * w.k * w.k * w.Cin, p.M, w.Cout, w.k * w.k * w.Cin);
    c.chk(launch_gemm(c.m->dt, p, c.st));
    return true;
}

struct Arena {
    char* base; size_t off = 0, cap;
    Arena(void* b, size_t c) : base((char*)b), cap(c) {}
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

}  // namespace

// =========================================================================================================
extern "C" const char* d3r_version(void) { return "dust3r_amd 0.1 (gfx950)"; }

extern "C" int d3r_device_check(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return D3R_ERR_STATE;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return D3R_ERR_STATE;     // the CURRENT device is the one the engine will run on
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return D3R_ERR_STATE;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? D3R_OK : D3R_ERR_STATE;
}

extern "C" int d3r_model_create(d3r_model** out, const d3r_model_config* cfg) {
    if (!out || !cfg) return D3R_ERR_INVALID;
    if (cfg->dtype < 0 || cfg->dtype > D3R_F16X2F8 || cfg->patch_size % 4 != 0) return D3R_ERR_INVALID;
    if (cfg->dtype == D3R_F16X2F8 && (cfg->enc_embed_dim % 128 || cfg->dec_embed_dim % 128)) return D3R_ERR_INVALID;   // whole 128-k blocks of five weight chunks
    if (cfg->enc_embed_dim != cfg->enc_num_heads * 64 || cfg->dec_embed_dim != cfg->dec_num_heads * 64) return D3R_ERR_INVALID;  // head dim 64
    d3r_model* m = new (std::nothrow) d3r_model();
    if (!m) return D3R_ERR_ALLOC;
    // D3R_DTYPE_F16F8: the transformer blocks' linears on fp16 + fp8 operand rows, everything else (patch embedding, decoder_embed,
    // attention operands, DPT / linear heads) in split-fp16
    // D3R_DTYPE_F16X2F8: the same split of the network, the blocks' linears on the 2.5-unit arithmetic (activation rows as above, five-chunk weight rows)
    const bool f8blocks = cfg->dtype == D3R_F16F8 || cfg->dtype == D3R_F16X2F8;
    m->cfg = *cfg; m->dt = f8blocks ? D3R_F16X3 : cfg->dtype; m->bdt = f8blocks ? cfg->dtype : m->dt;
    m->ktile = 128 / (int)dt_bytes(m->dt);
    {   // LayerNorm folded into the neighbouring GEMMs (see d3r_model::ln_fold): split-fp16 engines, every channel count a multiple of 32
        // default ON since round 5 (same-box A/B of two engines in one process, tools/fold_probe.py, profiles/r05_d: 168.8 -> 164.9 ms per 32-pair forward,
        // "other" kernels 11.9 -> 5.3 ms, nn.Linear launches +1.8 ms); D3R_LN_FOLD=0: LayerNorm kernels and an fp32 residual stream (rounds 1-4)
        const char* e = getenv("D3R_LN_FOLD");
        const bool want = e ? e[0] != '0' : true;
        m->ln_fold = want && m->dt == D3R_F16X3 && m->bdt == D3R_F16X3 && cfg->enc_embed_dim % 32 == 0 && cfg->dec_embed_dim % 32 == 0;
    }
    if (cfg->enc_embed_dim % m->ktile || cfg->dec_embed_dim % m->ktile || (3 * cfg->patch_size * cfg->patch_size) % m->ktile ||
        (cfg->head_type == 1 && (cfg->dec_depth <= 9 || cfg->patch_size != 16))) { delete m; return D3R_ERR_INVALID; }   // run_dpt assumes 16 x th == H
    if (!build_slots(m)) { d3r_model_destroy(m); return D3R_ERR_ALLOC; }
    m->rope_table = (float*)m->dalloc(512 * 16 * 2 * sizeof(float));
    m->zero_page = m->dalloc(4096);
    // split-K of the small-batch forwards: implemented, deterministic, and measured a LOSS on MI355X (profiles/r06_c: one pair 9.86 ms without, 9.99 ms with it on
    // the 64 x 64 tile, 12.5-13.7 ms on 128 x 128 tiles split 4-8 ways) -- opt-in (D3R_SPLITK=1 at create, D3R_MODEL_OPT_SPLIT_K), off by default
    { const char* e = getenv("D3R_SPLITK"); m->splitk_on = e && e[0] == '1'; }
    for (int i = 0; i < 2 && m->dt == D3R_F16X3; ++i) {
        m->sk_slab[i] = (float*)m->dalloc(d3r_model::SK_SLAB_FLOATS * sizeof(float));
        m->sk_cnt[i] = (unsigned*)m->dalloc(d3r_model::SK_CNT * sizeof(unsigned));
        if (!m->sk_slab[i] || !m->sk_cnt[i] || hipMemset(m->sk_cnt[i], 0, d3r_model::SK_CNT * sizeof(unsigned)) != hipSuccess) { m->sk_slab[i] = nullptr; m->sk_cnt[i] = nullptr; }
    }
    if (!m->rope_table || !m->zero_page) { d3r_model_destroy(m); return D3R_ERR_ALLOC; }
    if (launch_rope_table(m->rope_table, 512, cfg->rope_freq, 1.0f, nullptr) != hipSuccess) { d3r_model_destroy(m); return D3R_ERR_LAUNCH; }
    m->side = shared_stream(0);
    if (!m->side || hipEventCreateWithFlags(&m->ev_main, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_side, hipEventDisableTiming) != hipSuccess) { d3r_model_destroy(m); return D3R_ERR_ALLOC; }
    if (const char* e = getenv("D3R_GRAPH_MAX_PAIRS")) m->graph_max_pairs = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = probe_env("D3R_ENC_SPLIT")) m->enc_split_max = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("D3R_LN_INLINE_ROWS")) m->ln_inline_rows = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = probe_env("D3R_DEC_KV_AHEAD")) m->kv_ahead_rows = atoi(e) > 0 ? atoi(e) : 0;
    if (m->kv_ahead_rows > 0)          // probe only: its two streams and four events exist when it is switched on
        for (int s = 0; s < 2; ++s) {
            m->kvs[s] = shared_stream(1 + s);
            if (!m->kvs[s] || hipEventCreateWithFlags(&m->ev_kv_go[s], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&m->ev_kv_done[s], hipEventDisableTiming) != hipSuccess) { d3r_model_destroy(m); return D3R_ERR_ALLOC; }
        }
    (void)hipDeviceSynchronize();
    *out = m;
    return D3R_OK;
}

extern "C" int d3r_model_destroy(d3r_model* m) {
    if (!m) return D3R_OK;
    for (void* p : m->allocs) (void)hipFree(p);
    for (Lin* L : m->fold_lins) if (L->w32) (void)hipFree(L->w32);
    for (hipEvent_t e : m->prof_ev) (void)hipEventDestroy(e);
    if (m->ev_main) (void)hipEventDestroy(m->ev_main);
    if (m->ev_side) (void)hipEventDestroy(m->ev_side);
    if (m->side) (void)hipStreamSynchronize(m->side);      // shared with the process' other engines (shared_stream): drained, never destroyed
    for (int s = 0; s < 2; ++s) {
        if (m->kvs[s]) (void)hipStreamSynchronize(m->kvs[s]);
        if (m->ev_kv_go[s]) (void)hipEventDestroy(m->ev_kv_go[s]);
        if (m->ev_kv_done[s]) (void)hipEventDestroy(m->ev_kv_done[s]);
    }
    m->drop_graphs();
    if (m->cap) (void)hipStreamDestroy(m->cap);
    if (m->ws) (void)hipFree(m->ws);
    if (m->stage) (void)hipFree(m->stage);
    delete m;
    return D3R_OK;
}

static int load_tensor_dev(d3r_model* m, const char* key, const float* data_dev, int ndim, const int64_t* shape) {
    auto it = m->slots.find(key);
    if (it == m->slots.end()) {
        const std::string k(key);
        if (k.find("pos_embed") != std::string::npos || k.rfind("prediction_head", 0) == 0) return D3R_OK;
        return D3R_ERR_UNKNOWN_KEY;
    }
    Slot& s = it->second;
    int rc = pack_slot(m, s, data_dev, ndim, shape);
    if (rc != D3R_OK) return rc;
    s.loaded = true;
    if (std::string(key).rfind("dec_blocks2.", 0) == 0) s.explicit_loaded = true;
    if (!s.mirror.empty()) {
        Slot& t = m->slots[s.mirror];
        if (!t.explicit_loaded) {
            rc = pack_slot(m, t, data_dev, ndim, shape);
            if (rc != D3R_OK) return rc;
            t.loaded = true;
        }
    }
    return D3R_OK;
}

extern "C" int d3r_model_load_tensor_device(d3r_model* m, const char* key, const float* data_dev, int ndim, const int64_t* shape) {
    if (!m || !key || !data_dev || ndim < 0 || ndim > 8) return D3R_ERR_INVALID;
    return load_tensor_dev(m, key, data_dev, ndim, shape);
}

extern "C" int d3r_model_load_tensor(d3r_model* m, const char* key, const float* data, int ndim, const int64_t* shape) {
    if (!m || !key || !data || ndim < 0 || ndim > 8) return D3R_ERR_INVALID;
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    const size_t bytes = (numel ? numel : 1) * sizeof(float);
    if (bytes > m->stage_bytes) {                 // host tensors are staged through one device buffer, packed on the GPU
        (void)hipDeviceSynchronize();
        if (m->stage) (void)hipFree(m->stage);
        m->stage = nullptr; m->stage_bytes = 0;
        if (hipMalloc(&m->stage, bytes) != hipSuccess) return D3R_ERR_ALLOC;
        m->stage_bytes = bytes;
    }
    // synchronous copy on the default stream: ordered after the previous tensor's pack kernels
    if (hipMemcpy(m->stage, data, numel * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return D3R_ERR_ALLOC;
    const int rc = load_tensor_dev(m, key, (const float*)m->stage, ndim, shape);
    if (hipStreamSynchronize(nullptr) != hipSuccess) return D3R_ERR_LAUNCH;   // the staging buffer is reused by the next call
    return rc;
}

extern "C" int d3r_model_missing(const d3r_model* m) {
    if (!m) return -1;
    int n = 0;
    for (auto& kv : m->slots)
        if (!kv.second.loaded) ++n;
    return n;
}

extern "C" size_t d3r_model_device_bytes(const d3r_model* m) { return m ? m->weight_bytes + m->ws_bytes : 0; }

extern "C" int d3r_model_debug_read(d3r_model* m, int what, float* out, size_t max_elems, void* stream) {
    if (!m || what != 0 || !m->last_encn) return D3R_ERR_INVALID;
    const size_t n = m->last_encn_elems < max_elems ? m->last_encn_elems : max_elems;
    if (m->dt == D3R_F32) return hipMemcpyAsync(out, m->last_encn, n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? D3R_OK : D3R_ERR_LAUNCH;
    return D3R_ERR_INVALID;  // 16-bit modes: read through the Python side (torch view of the raw buffer is not exposed)
}

extern "C" int d3r_model_set_option(d3r_model* m, int option, int value) {
    if (!m) return D3R_ERR_INVALID;
    if (option == D3R_MODEL_OPT_PROFILE) { m->prof_on = value != 0; m->prof_rec.clear(); return D3R_OK; }
    if (option == D3R_MODEL_OPT_TWO_STREAMS) { m->two_streams = value != 0; return D3R_OK; }
    if (option == D3R_MODEL_OPT_SPLIT_K) { hipDeviceSynchronize(); m->drop_graphs(); m->splitk_on = value != 0; return D3R_OK; }
    if (option == D3R_MODEL_OPT_GRAPH_MAX_PAIRS) {
        m->graph_max_pairs = value > 0 ? value : 0;
        if (value <= 0) { hipDeviceSynchronize(); m->drop_graphs(); }     // a replay of the previous call may still be in flight
        return D3R_OK;
    }
    return D3R_ERR_INVALID;
}

// depth_mode / conf_mode of the heads' postprocess (dust3r/heads/postprocess.py:23-58; model.py:58-62 constructor keywords)
extern "C" int d3r_model_set_postprocess(d3r_model* m, int depth_mode, int conf_mode, float conf_vmin, float conf_vmax) {
    if (!m || depth_mode < POST_DEPTH_EXP || depth_mode > POST_DEPTH_SQUARE || conf_mode < POST_CONF_EXP || conf_mode > POST_CONF_SIGMOID) return D3R_ERR_INVALID;
    if (!(conf_vmin < conf_vmax) || (conf_mode == POST_CONF_SIGMOID && !(conf_vmax < __builtin_huge_valf() && conf_vmin > -__builtin_huge_valf()))) return D3R_ERR_INVALID;
    hipDeviceSynchronize();     // a captured graph carries the old mode in its kernel arguments
    m->drop_graphs();
    m->post.depth = depth_mode; m->post.conf = conf_mode; m->post.cmin = conf_vmin; m->post.cmax = conf_vmax;
    return D3R_OK;
}

// Per-class totals of the LAST forward run with profiling on (kinds: include/dust3r_hip.h). Synchronises on the recorded events.
extern "C" int d3r_model_profile_read(d3r_model* m, int kind, int* launches, double* ms, double* work) {
    if (!m || kind < 0 || kind > 31 || m->prof_rec.size() < 2) return D3R_ERR_STATE;
    int n = 0; double t = 0.0, w = 0.0;
    if (hipEventSynchronize(m->prof_ev[m->prof_rec.size() - 1]) != hipSuccess) return D3R_ERR_LAUNCH;
    for (size_t i = 0; i + 1 < m->prof_rec.size(); ++i) {
        if (m->prof_rec[i].kind != kind) continue;
        float dt = 0.f;
        if (hipEventElapsedTime(&dt, m->prof_ev[i], m->prof_ev[i + 1]) != hipSuccess) return D3R_ERR_LAUNCH;
        ++n; t += dt; w += m->prof_rec[i].work;
    }
    if (launches) *launches = n;
    if (ms) *ms = t;
    if (work) *work = w;
    return D3R_OK;
}

// One launch of the LAST profiled forward: its class (as above), GEMM shape (M, N, K; attention: batch*heads, queries, keys; 0 for
// other kernels), duration and algorithmic work. Returns D3R_ERR_STATE past the last launch.
extern "C" int d3r_model_profile_launch(d3r_model* m, int index, int* kind, int* M, int* N, int* K, double* ms, double* work) {
    if (!m || index < 0 || (size_t)index + 1 >= m->prof_rec.size()) return D3R_ERR_STATE;
    if (hipEventSynchronize(m->prof_ev[m->prof_rec.size() - 1]) != hipSuccess) return D3R_ERR_LAUNCH;
    float dt = 0.f;
    if (hipEventElapsedTime(&dt, m->prof_ev[index], m->prof_ev[index + 1]) != hipSuccess) return D3R_ERR_LAUNCH;
    const auto& r = m->prof_rec[index];
    if (kind) *kind = r.kind;
    if (M) *M = r.M;
    if (N) *N = r.N;
    if (K) *K = r.K;
    if (ms) *ms = dt;
    if (work) *work = r.work;
    return D3R_OK;
}

// ---- the forward ---------------------------------------------------------------------------------------------
namespace {

void self_attention(Ctx& c, const void* xn, const Lin& qkv, int M, int C, int heads, int nimg, int ntok, int tok_w, int ldv, void* q, void* k,
                    void* vt, void* ao, LnStats stats = LnStats()) {
    const int kinds[3] = {HEAD_ROPE, HEAD_ROPE, HEAD_VT};
    void* dsts[3] = {q, k, vt};
    gemm_heads(c, xn, C, qkv, M, C, 3, kinds, dsts, heads, ntok, tok_w, ldv, stats);
    AttnParams a;
    a.q = q; a.k = k; a.vt = vt; a.out = ao; a.B = nimg; a.H = heads; a.Nq = ntok; a.Nk = ntok; a.ldv = ldv; a.scale = 0.125f;
    a.out_dt = c.m->bdt;      // rows for the proj GEMM
    c.mark(PRF_ATTN, 4.0 * nimg * heads * (double)ntok * ntok * 64, nimg * heads, ntok, ntok);
    c.chk(launch_attention(c.m->dt, a, c.st));
}

void run_dpt(Ctx& c, const DptHead& D, Arena ar, const void* const hooks[4], const int hook_c[4], int B, int th, int tw, float* pts, float* conf) {
    d3r_model* m = c.m;
    const size_t eb = dt_bytes(m->dt);
    const int N = th * tw;
    const int th2 = (th - 1) / 2 + 1, tw2 = (tw - 1) / 2 + 1;
    const int Hl[4] = {4 * th, 2 * th, th, th2}, Wl[4] = {4 * tw, 2 * tw, tw, tw2};
    // reassemble: 1x1 conv (+ ConvTranspose / strided conv)
    void* cmap[4];
    for (int i = 0; i < 4; ++i) cmap[i] = ar.take((size_t)B * Hl[i] * Wl[i] * D.cstride[i] * eb);
    void* t1 = ar.take((size_t)B * N * 768 * eb);
    for (int i = 0; i < 4; ++i) {
        const Lin& L = D.act1x1[i];
        const bool direct = (i == 2);
        gemm_linear(c, hooks[i], hook_c[i], L, B * N, EPI_T, direct ? cmap[2] : t1, D.cstride[i], nullptr, nullptr, 0, D.cstride[i]);
        if (i < 2) {
            GemmParams p;
            p.act = t1; p.lda = D.cstride[i]; p.wgt = D.convt[i].w; p.bias = D.convt[i].b; p.M = B * N; p.K = D.convt[i].K;
            p.n_pad = D.convt[i].n_pad; p.n_rows = D.convt[i].n_rows; p.n_store = D.convt[i].N; p.epi = EPI_CONVT; p.ksize = D.convt_k[i]; p.ct_cout = D.convt_coutp[i];
            p.Hin = th; p.Win = tw; p.out = cmap[i]; p.ldo = D.cstride[i];
            const int ldi[2] = {96, 192};
            c.mark(prf_kind(PRF_CONV, gemm_pick_config(p, m->dt)), 2.0 * p.M * (double)(D.convt_k[i] * D.convt_k[i] * ldi[i]) * ldi[i], p.M, D.convt_k[i] * D.convt_k[i] * ldi[i], ldi[i]);
            c.chk(launch_gemm(m->dt, p, c.st));
        } else if (i == 3) {
            conv(c, t1, B, th, tw, D.cstride[3], D.act3conv, 2, 1, cmap[3], D.cstride[3], 0);
        }
    }
    // layer_rn: 3x3 -> 256 (no bias), plus ReLU copy for the residual units' pre-activation
    void *r[4], *rr[4];
    for (int i = 0; i < 4; ++i) {
        const size_t bytes = (size_t)B * Hl[i] * Wl[i] * 256 * eb;
        r[i] = ar.take(bytes); rr[i] = ar.take(bytes);
        conv(c, cmap[i], B, Hl[i], Wl[i], D.cstride[i], D.layer_rn[i], 1, 1, r[i], 256, 0, nullptr, nullptr, rr[i]);
    }
    const size_t big = (size_t)B * Hl[0] * Wl[0] * 256 * eb;
    void *tA = ar.take(big), *tB = ar.take(big), *tC = ar.take(big);
    void* path = nullptr;  // output of the previous refinenet (already upsampled)
    for (int lvl = 3; lvl >= 0; --lvl) {
        const Refine& R = D.rn[lvl];
        const int H = Hl[lvl], W = Wl[lvl];
        const void* skip;      // input of resConfUnit2 (un-activated) ...
        const void* skip_relu; // ... and its ReLU
        if (lvl == 3) {
            skip = r[3]; skip_relu = rr[3];
        } else {
            conv(c, rr[lvl], B, H, W, 256, R.r1c1, 1, 1, tA, 256, GF_RELU);
            conv(c, tA, B, H, W, 256, R.r1c2, 1, 1, tB, 256, 0, m->cfg.dpt_skip_relu_inplace ? rr[lvl] : r[lvl], path, tC);
            skip = tB; skip_relu = tC;
        }
        conv(c, skip_relu, B, H, W, 256, R.r2c1, 1, 1, tA, 256, GF_RELU);
        void* o = (lvl == 3) ? tB : tC;
        conv(c, tA, B, H, W, 256, R.r2c2, 1, 1, o, 256, 0, m->cfg.dpt_skip_relu_inplace ? skip_relu : skip);
        // out_conv (1x1) commutes with the bilinear upsampling (both linear, weights sum to 1): run it at low resolution
        gemm_linear(c, o, 256, R.outc, B * H * W, EPI_T, tA, 256);
        const int Ho = lvl == 3 ? Hl[2] : 2 * H, Wo = lvl == 3 ? Wl[2] : 2 * W;  // dpt_head.py:57 crops refinenet4 to layer 3's size
        void* np = ar.take((size_t)B * Ho * Wo * 256 * eb);
        D3R_OTHER(launch_upsample2x(m->dt, tA, np, nullptr, B, H, W, 256, 256, Ho, Wo, c.st));
        path = np;
    }
    // head: 3x3 256->128, x2, 3x3 128->128 + ReLU, 1x1 128->4 + postprocess
    const int H8 = 8 * th, W8 = 8 * tw;
    void* h0 = ar.take((size_t)B * H8 * W8 * 128 * eb);
    conv(c, path, B, H8, W8, 256, D.head0, 1, 1, h0, 128, 0);
    void* h1 = ar.take((size_t)B * 4 * H8 * W8 * 128 * eb);
    D3R_OTHER(launch_upsample2x(m->dt, h0, h1, nullptr, B, H8, W8, 128, 128, 2 * H8, 2 * W8, c.st));
    void* h2 = ar.take((size_t)B * 4 * H8 * W8 * 128 * eb);
    // Split-fp16, enough pixels for the 512 x 128 tile (every wave then holds all 128 channels of its 64 pixels): the 1x1 convolution and
    // the postprocess run in the epilogue of the last 3x3 convolution, on its fp32 accumulators -- the 128-channel full-resolution map
    // (3.2 GB per head at 32 pairs) is neither written nor read back. D3R_HEAD_FUSE=0: the two-kernel route (test A/B).
    bool fused = false;
    if (m->dt == D3R_F16X3) {
        const char* e = getenv("D3R_HEAD_FUSE");
        if (!(e && e[0] == '0')) fused = conv_head4(c, h1, B, 2 * H8, 2 * W8, 128, D.head2, D.head4_w, D.head4_b, pts, conf, m->out_pstride, m->out_cstride);
    }
    if (!fused) {
        conv(c, h1, B, 2 * H8, 2 * W8, 128, D.head2, 1, 1, h2, 128, GF_RELU);
        D3R_OTHER(launch_head_final(m->dt, h2, 128, D.head4_w, D.head4_b, pts, conf, (size_t)B * 4 * H8 * W8, m->out_pstride, m->out_cstride, m->post, c.st));
    }
    if (ar.base && ar.off > ar.cap) c.rc = D3R_ERR_ALLOC;
}

// bytes of the DPT head arena for `bc` images (the allocation sequence of run_dpt)
size_t dpt_arena_bytes(const d3r_model* m, const DptHead& D, int bc, int th, int tw) {
    const size_t eb = dt_bytes(m->dt);
    const int N = th * tw;
    const int th2 = (th - 1) / 2 + 1, tw2 = (tw - 1) / 2 + 1;
    const int Hl[4] = {4 * th, 2 * th, th, th2}, Wl[4] = {4 * tw, 2 * tw, tw, tw2};
    Arena sub(nullptr, 0);
    for (int i = 0; i < 4; ++i) sub.take((size_t)bc * Hl[i] * Wl[i] * D.cstride[i] * eb);
    sub.take((size_t)bc * N * 768 * eb);
    for (int i = 0; i < 4; ++i) { sub.take((size_t)bc * Hl[i] * Wl[i] * 256 * eb); sub.take((size_t)bc * Hl[i] * Wl[i] * 256 * eb); }
    for (int i = 0; i < 3; ++i) sub.take((size_t)bc * Hl[0] * Wl[0] * 256 * eb);
    for (int lvl = 3; lvl >= 0; --lvl) {
        const int Ho = lvl == 3 ? Hl[2] : 2 * Hl[lvl], Wo = lvl == 3 ? Wl[2] : 2 * Wl[lvl];
        sub.take((size_t)bc * Ho * Wo * 256 * eb);
    }
    sub.take((size_t)bc * 64 * N * 128 * eb);
    sub.take((size_t)bc * 256 * N * 128 * eb);
    sub.take((size_t)bc * 256 * N * 128 * eb);
    return ((sub.off + 255) & ~(size_t)255) + 256;
}

// returns bytes needed when ws == nullptr (dry run), else runs
// Phases: the encoder runs over `nimg` images (img1 holds the first nimg1 of them, img2 the rest: forward passes B + B,
// d3r_model_encode passes everything in img1) and leaves the normalised features [nimg][N][Ce] in `feat`; the decoder +
// heads run over B pairs whose features are feat[0..B) (view 1) and feat[B..2B) (view 2). forward = both phases with
// feat inside the workspace; encode / decode run one phase with a caller-owned feature buffer.
enum { PH_ENCODE = 1, PH_DECODE = 2 };
// The two views of a pair may have different sizes (dust3r/model.py:148-150: the reference then encodes them separately): every
// per-side quantity of the decoder / heads is indexed by side -- token grid (th, tw), tokens per image N, rows M = B N, V^T row
// stride -- and cross attention runs Nq != Nk. With equal sizes this is the old schedule bit for bit.
struct SideDim { int H, W, th, tw, N, ldv; };
static SideDim side_dim(int H, int W, int ps) {
    SideDim d;
    d.H = H; d.W = W; d.th = H / ps; d.tw = W / ps; d.N = d.th * d.tw; d.ldv = rup(d.N, 64);
    return d;
}

size_t forward_impl(d3r_model* m, void* ws, size_t ws_cap, int phases, const float* img1, const float* img2, int nimg1, int nimg, void* feat_ext,
                    int B, SideDim d0, SideDim d1, float* pts1, float* conf1, float* pts2, float* conf2, hipStream_t st, int* rc_out) {
    const d3r_model_config& cf = m->cfg;
    const int ps = cf.patch_size;
    const SideDim D[2] = {d0, d1};
    const bool same = d0.H == d1.H && d0.W == d1.W;
    const int Ce = cf.enc_embed_dim, Cd = cf.dec_embed_dim, He = cf.enc_num_heads, Hd = cf.dec_num_heads;
    const int Cmax = Ce > Cd ? Ce : Cd, Hmax = He > Hd ? He : Hd;     // shared scratch is sized for the wider of encoder / decoder
    const bool do_enc = (phases & PH_ENCODE) != 0, do_dec = (phases & PH_DECODE) != 0;
    // encoder rows: one pass over all images when the views share a size, else one pass per view (view 1 rows first)
    const int Me = !do_enc ? 0 : (same ? nimg * d0.N : nimg1 * d0.N + (nimg - nimg1) * d1.N);
    const int Ms[2] = {do_dec ? B * d0.N : 0, do_dec ? B * d1.N : 0};     // decoder rows per side
    const int Roff[2] = {0, Ms[0]};                                       // first row of a side in the two-sided buffers
    const int M2d = Ms[0] + Ms[1];
    const int Mmax = Ms[0] > Ms[1] ? Ms[0] : Ms[1];                       // a side's scratch part holds its own rows OR the other view's (cross attention)
    const int M2 = Me > 2 * Mmax ? Me : 2 * Mmax;                         // rows of the shared scratch buffers
    const int ldv_max = d0.ldv > d1.ldv ? d0.ldv : d1.ldv;
    const int nvt = (do_enc ? nimg : 0) > 2 * (do_dec ? B : 0) ? nimg : 2 * B;   // images the v^T buffer must hold
    const size_t eb = dt_bytes(m->dt);
    const bool dry = ws == nullptr;
    Arena ar(ws, ws_cap);
    Ctx c{m, st};
    c.st0 = st;

    float* x = (float*)ar.take((size_t)Me * Ce * 4);
    void* xn = ar.take((size_t)M2 * Cmax * eb);
    void* q = ar.take((size_t)M2 * Cmax * eb);
    void* k = ar.take((size_t)M2 * Cmax * eb);
    void* vt = ar.take((size_t)nvt * Hmax * 64 * ldv_max * eb);
    void* ao = ar.take((size_t)M2 * Cmax * eb);
    void* hb = ar.take((size_t)M2 * 4 * Cmax * eb);   // MLP hidden; also holds the gathered patches
    void* encn = feat_ext ? feat_ext : ar.take((size_t)(Me > M2d ? Me : M2d) * Ce * eb);
    float* f[2] = {(float*)ar.take((size_t)M2d * Cd * 4), (float*)ar.take((size_t)M2d * Cd * 4)};
    void* yn = ar.take((size_t)2 * Mmax * Cd * eb);
    void* hook[2][3];
    for (int s = 0; s < 2; ++s)
        for (int j = 0; j < 3; ++j) hook[s][j] = ar.take((size_t)Ms[s] * Cd * eb);
    float* lin_out = cf.head_type == 0 ? (float*)ar.take((size_t)M2d * 4 * ps * ps * 4) : nullptr;
    // folded LayerNorm (d3r_model::ln_fold): partial sums [rows][C / 32][2], rstd / -mean rstd per row -- one set for the encoder (e_), one per
    // decoder LAYER OUTPUT and buffer (l_: both sides' rows, read by the own side's norm1 and the other side's norm_y) with the raw typed copy fr
    // of that output, one per side for the block-internal norm2 / norm3 (s_)
    const bool fold = m->ln_fold;
    const int Ge = Ce / 32, Gd = Cd / 32;
    // rows up to which a folded LayerNorm's statistics are formed in the consumer GEMM's prologue instead of by a launch (D3R_LN_INLINE_ROWS; 0 = always launch)
    const int inline_rows = m->ln_inline_rows;
    float *e_part = nullptr, *e_rs = nullptr, *e_nm = nullptr, *l_part[2] = {nullptr, nullptr}, *l_rs[2] = {nullptr, nullptr}, *l_nm[2] = {nullptr, nullptr};
    float *s_part[2] = {nullptr, nullptr}, *s_rs[2] = {nullptr, nullptr}, *s_nm[2] = {nullptr, nullptr};
    void* fr[2] = {nullptr, nullptr};
    void *ckb = nullptr, *cvtb = nullptr;      // cross-attention K / V^T of the two sides when they are projected ahead on their own streams (d3r_model::kvs)
    if (fold) {
        ckb = ar.take((size_t)2 * Mmax * Cd * eb);
        cvtb = ar.take((size_t)2 * B * Hd * 64 * ldv_max * eb);
        e_part = (float*)ar.take((size_t)Me * Ge * 8); e_rs = (float*)ar.take((size_t)Me * 4 + 16); e_nm = (float*)ar.take((size_t)Me * 4 + 16);
        for (int b = 0; b < 2; ++b) {
            fr[b] = ar.take((size_t)M2d * Cd * eb);
            l_part[b] = (float*)ar.take((size_t)M2d * Gd * 8); l_rs[b] = (float*)ar.take((size_t)M2d * 4 + 16); l_nm[b] = (float*)ar.take((size_t)M2d * 4 + 16);
            s_part[b] = (float*)ar.take((size_t)Mmax * Gd * 8); s_rs[b] = (float*)ar.take((size_t)Mmax * 4 + 16); s_nm[b] = (float*)ar.take((size_t)Mmax * 4 + 16);
        }
    }
    const size_t common_end = (ar.off + 255) & ~(size_t)255;
    const int chunk = B < 32 ? B : 32;   // 288 GB of HBM: batch the head as wide as the encoder (low-resolution stages need the rows)
    size_t head_arena = 0;
    if (cf.head_type == 1 && do_dec)
        for (int s = 0; s < 2; ++s) {
            const size_t a = dpt_arena_bytes(m, m->dpt[0], chunk, D[s].th, D[s].tw);
            head_arena = a > head_arena ? a : head_arena;
        }

    if (!dry) {
        // stream plan: encoder on the caller's stream; then side 0 stays there and side 1 runs on the model's second
        // stream, re-joined at every layer boundary (each side reads the other's previous-layer output) and at the end.
        // Each side has its own part of every scratch buffer and its own head arena.
        const bool two = m->two_streams && !m->prof_on && m->side != nullptr;
        hipStream_t S[2] = {st, two ? m->side : st};
        auto cross_sync = [&]() {
            if (!two) return;
            c.chk(hipEventRecord(m->ev_main, S[0]));
            c.chk(hipEventRecord(m->ev_side, S[1]));
            c.chk(hipStreamWaitEvent(S[0], m->ev_side, 0));
            c.chk(hipStreamWaitEvent(S[1], m->ev_main, 0));
        };
        if (d0.ldv != d0.N || d1.ldv != d1.N) D3R_OTHER(hipMemsetAsync(vt, 0, (size_t)nvt * Hmax * 64 * ldv_max * eb, st));
        const bool kva = two && fold && do_dec && m->kvs[0] && m->kvs[1] && Mmax <= m->kv_ahead_rows;
        if (kva && (d0.ldv != d0.N || d1.ldv != d1.N)) D3R_OTHER(hipMemsetAsync(cvtb, 0, (size_t)2 * B * Hd * 64 * ldv_max * eb, st));
        if (do_enc) {
            // ---- encoder: all images of the call in one pass when the views share a size (model.py:142-151 concatenates the
            // two views), else view 1's images then view 2's (model.py:148-150) -----------------------------------------------
            const size_t pk = 3 * (size_t)ps * ps;
            // Encoder work is a list of GROUPS of images of one size: one group (both views) when the views share a size, else one per view. Round 5: small calls
            // (<= enc_split_max images, two views, two streams on) run the two views as two groups CONCURRENTLY, view 2 on the model's second stream with its
            // own rows of every scratch buffer -- one pair per call is a dependent chain of kernels that each fill a fraction of the chip, and two
            // independent chains fill it better (what the decoder's two sides already do). Same kernels on the same rows: bit-identical.
            struct EncGroup { const float* im[2]; int n[2]; const SideDim* dd; size_t row0; int img0; hipStream_t s; bool off; };
            EncGroup groups[2];
            int ngroups = 0;
            const bool split = two && nimg1 > 0 && nimg > nimg1 && nimg <= m->enc_split_max;
            if (same && !split) {
                groups[ngroups++] = EncGroup{{img1, img2}, {nimg1, nimg - nimg1}, &D[0], 0, 0, st, false};
            } else {
                if (nimg1 > 0) groups[ngroups++] = EncGroup{{img1, nullptr}, {nimg1, 0}, &D[0], 0, 0, st, split};
                if (nimg > nimg1) groups[ngroups++] = EncGroup{{img2, nullptr}, {nimg - nimg1, 0}, &D[same ? 0 : 1], (size_t)nimg1 * d0.N, nimg1, split ? S[1] : st, split};
            }
            if (split) {      // view 2's stream starts behind everything the caller's stream holds (the images, the v^T clear)
                c.chk(hipEventRecord(m->ev_main, S[0]));
                c.chk(hipStreamWaitEvent(S[1], m->ev_main, 0));
            }
            // phase -1: patches + patch embedding; 0 .. depth - 1: block l; depth: enc_norm
            auto enc_phase = [&](const EncGroup& g, int phase) {
                const SideDim& dd = *g.dd;
                const int n_img = g.n[0] + g.n[1];
                const int Mp = n_img * dd.N;
                c.st = g.s;
                // a group that runs next to another one works in its own rows of the scratch buffers (sequential groups reuse them from row 0, as before)
                const size_t ro = g.off ? g.row0 : 0;
                float* xp = x + g.row0 * Ce;
                void* gxn = (char*)xn + ro * Ce * eb;
                void* gq = (char*)q + ro * Ce * eb;
                void* gk = (char*)k + ro * Ce * eb;
                void* gvt = (char*)vt + (size_t)(g.off ? g.img0 : 0) * He * 64 * dd.ldv * eb;
                void* gao = (char*)ao + ro * Ce * eb;
                void* ghb = (char*)hb + ro * 4 * Ce * eb;        // hidden rows; the patches of the group start at the same place
                float* gpart = fold ? e_part + ro * Ge * 2 : nullptr;
                float* grs = fold ? e_rs + ro : nullptr;
                float* gnm = fold ? e_nm + ro : nullptr;
                const bool inl = fold && Mp <= inline_rows;
                const LnStats es{grs, gnm, inl ? gpart : nullptr};
                if (phase < 0) {
                    if (g.n[0] > 0) D3R_OTHER(launch_patchify(m->dt, g.im[0], ghb, g.n[0], dd.H, dd.W, ps, c.st));
                    if (g.n[1] > 0) D3R_OTHER(launch_patchify(m->dt, g.im[1], (char*)ghb + (size_t)g.n[0] * dd.N * pk * eb, g.n[1], dd.H, dd.W, ps, c.st));
                    // fold: every residual epilogue stores the typed rows (the residual stream itself, GF_X3RES) and their partial sums; the LayerNorm is two fmas in
                    // the epilogue of the nn.Linear behind it (weights packed as W diag(gamma)) -- DESIGN 4.0
                    if (fold) gemm_linear(c, ghb, (int)pk, m->patch, Mp, EPI_F32, nullptr, Ce, nullptr, gxn, Ce, -1, GF_X3RES, LnStats(), gpart);
                    else gemm_linear(c, ghb, (int)pk, m->patch, Mp, EPI_F32, xp, Ce);
                } else if (phase < cf.enc_depth) {
                    const EncBlk& b = m->enc[phase];
                    if (fold) {
                        const bool last = phase + 1 == cf.enc_depth;        // enc_norm (a kernel) follows: no sums
                        if (!inl) D3R_OTHER(launch_ln_finalize(gpart, Mp, Ce, 1e-6f, grs, gnm, c.st));
                        self_attention(c, gxn, b.qkv, Mp, Ce, He, n_img, dd.N, dd.tw, dd.ldv, gq, gk, gvt, gao, es);
                        gemm_linear(c, gao, Ce, b.proj, Mp, EPI_F32, nullptr, Ce, gxn, gxn, Ce, -1, GF_X3RES, LnStats(), gpart);
                        if (!inl) D3R_OTHER(launch_ln_finalize(gpart, Mp, Ce, 1e-6f, grs, gnm, c.st));
                        gemm_linear(c, gxn, Ce, b.fc1, Mp, EPI_GELU, ghb, 4 * Ce, nullptr, nullptr, 0, -1, 0, es);
                        gemm_linear(c, ghb, 4 * Ce, b.fc2, Mp, EPI_F32, nullptr, Ce, gxn, gxn, Ce, -1, GF_X3RES, LnStats(), last ? nullptr : gpart);
                    } else {
                        D3R_OTHER(launch_layernorm(m->bdt, xp, b.n1.g, b.n1.b, gxn, Mp, Ce, 1e-6f, c.st));
                        self_attention(c, gxn, b.qkv, Mp, Ce, He, n_img, dd.N, dd.tw, dd.ldv, gq, gk, gvt, gao);
                        gemm_linear(c, gao, Ce, b.proj, Mp, EPI_F32, xp, Ce, xp);
                        D3R_OTHER(launch_layernorm(m->bdt, xp, b.n2.g, b.n2.b, gxn, Mp, Ce, 1e-6f, c.st));
                        gemm_linear(c, gxn, Ce, b.fc1, Mp, EPI_GELU, ghb, 4 * Ce);
                        gemm_linear(c, ghb, 4 * Ce, b.fc2, Mp, EPI_F32, xp, Ce, xp);
                    }
                } else {
                    void* dst = (char*)encn + g.row0 * Ce * eb;
                    if (fold) D3R_OTHER(launch_layernorm_x3in(gxn, m->enc_norm.g, m->enc_norm.b, dst, Mp, Ce, 1e-6f, c.st));
                    else D3R_OTHER(launch_layernorm(m->dt, xp, m->enc_norm.g, m->enc_norm.b, dst, Mp, Ce, 1e-6f, c.st));
                }
            };
            if (split) {       // layer by layer, so that both streams have work queued from the start
                for (int phase = -1; phase <= cf.enc_depth; ++phase)
                    for (int gi = 0; gi < ngroups; ++gi) enc_phase(groups[gi], phase);
                c.chk(hipEventRecord(m->ev_side, S[1]));           // the caller's stream continues behind view 2's encoder
                c.chk(hipStreamWaitEvent(S[0], m->ev_side, 0));
            } else {
                for (int gi = 0; gi < ngroups; ++gi)
                    for (int phase = -1; phase <= cf.enc_depth; ++phase) enc_phase(groups[gi], phase);
            }
            c.st = st;
            m->last_encn = encn; m->last_encn_elems = (size_t)Me * Ce;
        }
        if (do_dec) {
        // ---- decoder (model.py:172-191): side s reads the PREVIOUS layer's (f_s, f_other) ---------------------
        void* frp[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // fold: raw typed copy of layer output [buffer][side] (fr, or a DPT hook buffer at the hook layers)
        if (fold) {
            gemm_linear(c, encn, Ce, m->dec_embed, M2d, EPI_F32, nullptr, Cd, nullptr, fr[0], Cd, -1, GF_X3RES, LnStats(), l_part[0]);
            if (!(Ms[0] <= inline_rows && Ms[1] <= inline_rows)) D3R_OTHER(launch_ln_finalize(l_part[0], M2d, Cd, 1e-6f, l_rs[0], l_nm[0], st));
            for (int sd = 0; sd < 2; ++sd) frp[0][sd] = (char*)fr[0] + (size_t)Roff[sd] * Cd * eb;
        } else {
        gemm_linear(c, encn, Ce, m->dec_embed, M2d, EPI_F32, f[0], Cd);
        }
        if (two) {
            c.chk(hipEventRecord(m->ev_main, S[0]));
            c.chk(hipStreamWaitEvent(S[1], m->ev_main, 0));
        }
        int cur = 0;
        const int hk6 = cf.dec_depth * 2 / 4, hk9 = cf.dec_depth * 3 / 4;
        for (int l = 0; l < cf.dec_depth; ++l) {
            for (int s = 0; s < 2; ++s) {
                c.st = S[s];
                const SideDim& own = D[s];
                const SideDim& oth = D[1 - s];
                // this side's part of the scratch buffers (the encoder used them whole)
                void* sxn = (char*)xn + (size_t)s * Mmax * Cmax * eb;
                void* sq = (char*)q + (size_t)s * Mmax * Cmax * eb;
                void* sk = (char*)k + (size_t)s * Mmax * Cmax * eb;
                void* svt = (char*)vt + (size_t)s * B * Hmax * 64 * ldv_max * eb;
                void* sao = (char*)ao + (size_t)s * Mmax * Cmax * eb;
                void* shb = (char*)hb + (size_t)s * Mmax * 4 * Cmax * eb;
                void* syn = (char*)yn + (size_t)s * Mmax * Cd * eb;         // norm_y of the OTHER view's tokens (Ms[1 - s] <= Mmax rows)
                const DecBlk& b = m->dec[s][l];
                const float* xo = f[cur] + (size_t)Roff[s] * Cd;         // own stream (old)
                const float* yo = f[cur] + (size_t)Roff[1 - s] * Cd;     // other view (old)
                float* xw = f[cur ^ 1] + (size_t)Roff[s] * Cd;           // own stream (new)
                if (fold) {
                    // small problems: the consumers form rstd / nmr of their input rows themselves (the set of side t has Ms[t] rows whoever consumes it)
                    const bool inl_own = Ms[s] <= inline_rows, inl_oth = Ms[1 - s] <= inline_rows;
                    const LnStats sx{l_rs[cur] + Roff[s], l_nm[cur] + Roff[s], inl_own ? l_part[cur] + (size_t)Roff[s] * Gd * 2 : nullptr},
                                  sy{l_rs[cur] + Roff[1 - s], l_nm[cur] + Roff[1 - s], inl_oth ? l_part[cur] + (size_t)Roff[1 - s] * Gd * 2 : nullptr},
                                  ss{s_rs[s], s_nm[s], inl_own ? s_part[s] : nullptr};
                    void* xk = sk;
                    void* xvt = svt;
                    if (kva) {
                        // S[s] stands behind the layer boundary here (the other side's rows and statistics are complete) and behind this side's previous cross attention
                        // (the last reader of xk / xvt): the projection may start now, beside the self attention
                        xk = (char*)ckb + (size_t)s * Mmax * Cd * eb;
                        xvt = (char*)cvtb + (size_t)s * B * Hd * 64 * ldv_max * eb;
                        c.chk(hipEventRecord(m->ev_kv_go[s], S[s]));
                        c.chk(hipStreamWaitEvent(m->kvs[s], m->ev_kv_go[s], 0));
                        c.st = m->kvs[s];
                        const int kkv[2] = {HEAD_ROPE, HEAD_VT};
                        void* dkv[2] = {xk, xvt};
                        gemm_heads(c, frp[cur][1 - s], Cd, b.ckv, Ms[1 - s], Cd, 2, kkv, dkv, Hd, oth.N, oth.tw, oth.ldv, sy);
                        c.chk(hipEventRecord(m->ev_kv_done[s], m->kvs[s]));
                        c.st = S[s];
                    }
                    self_attention(c, frp[cur][s], b.qkv, Ms[s], Cd, Hd, B, own.N, own.tw, own.ldv, sq, sk, svt, sao, sx);      // norm1 folded
                    // the residual stream is the typed rows themselves (GF_X3RES): layer input frp[cur][s] -> sxn (after self attention, then in place after
                    // cross attention) -> the next layer's input fr[cur ^ 1] (or a DPT hook buffer)
                    gemm_linear(c, sao, Cd, b.proj, Ms[s], EPI_F32, nullptr, Cd, frp[cur][s], sxn, Cd, -1, GF_X3RES, LnStats(), s_part[s]);
                    if (!inl_own) D3R_OTHER(launch_ln_finalize(s_part[s], Ms[s], Cd, 1e-6f, s_rs[s], s_nm[s], c.st));
                    {
                        const int kq[1] = {HEAD_ROPE};
                        void* dq[1] = {sq};
                        gemm_heads(c, sxn, Cd, b.cq, Ms[s], Cd, 1, kq, dq, Hd, own.N, own.tw, own.ldv, ss);                     // norm2 folded
                        if (kva) c.chk(hipStreamWaitEvent(S[s], m->ev_kv_done[s], 0));
                        else {
                        const int kkv[2] = {HEAD_ROPE, HEAD_VT};
                        void* dkv[2] = {sk, svt};
                        gemm_heads(c, frp[cur][1 - s], Cd, b.ckv, Ms[1 - s], Cd, 2, kkv, dkv, Hd, oth.N, oth.tw, oth.ldv, sy);  // norm_y folded: the other side's raw rows and statistics
                        }
                        AttnParams a;
                        a.q = sq; a.k = xk; a.vt = xvt; a.out = sao; a.B = B; a.H = Hd; a.Nq = own.N; a.Nk = oth.N; a.ldv = oth.ldv; a.scale = 0.125f;
                        a.out_dt = m->bdt;
                        c.mark(PRF_ATTN, 4.0 * B * Hd * (double)own.N * oth.N * 64, B * Hd, own.N, oth.N);
                        c.chk(launch_attention(m->dt, a, c.st));
                    }
                    gemm_linear(c, sao, Cd, b.cproj, Ms[s], EPI_F32, nullptr, Cd, sxn, sxn, Cd, -1, GF_X3RES, LnStats(), s_part[s]);
                    if (!inl_own) D3R_OTHER(launch_ln_finalize(s_part[s], Ms[s], Cd, 1e-6f, s_rs[s], s_nm[s], c.st));
                    gemm_linear(c, sxn, Cd, b.fc1, Ms[s], EPI_GELU, shb, 4 * Cd, nullptr, nullptr, 0, -1, 0, ss);                 // norm3 folded
                    const int layer_no = l + 1;
                    void* hcopy = (cf.head_type == 1 && (layer_no == hk6 || layer_no == hk9)) ? hook[s][layer_no == hk6 ? 0 : 1] : nullptr;
                    // the typed layer output IS a DPT hook at the hook layers; after the last layer dec_norm (a kernel) follows: no sums
                    void* raw = hcopy ? hcopy : (void*)((char*)fr[cur ^ 1] + (size_t)Roff[s] * Cd * eb);
                    float* lp = layer_no == cf.dec_depth ? nullptr : l_part[cur ^ 1] + (size_t)Roff[s] * Gd * 2;
                    gemm_linear(c, shb, 4 * Cd, b.fc2, Ms[s], EPI_F32, nullptr, Cd, sxn, raw, Cd, -1, GF_X3RES, LnStats(), lp);
                    if (lp && !inl_own) D3R_OTHER(launch_ln_finalize(lp, Ms[s], Cd, 1e-6f, l_rs[cur ^ 1] + Roff[s], l_nm[cur ^ 1] + Roff[s], c.st));
                    frp[cur ^ 1][s] = raw;
                    continue;
                }
                D3R_OTHER(launch_layernorm(m->bdt, xo, b.n1.g, b.n1.b, sxn, Ms[s], Cd, 1e-6f, c.st));
                self_attention(c, sxn, b.qkv, Ms[s], Cd, Hd, B, own.N, own.tw, own.ldv, sq, sk, svt, sao);
                gemm_linear(c, sao, Cd, b.proj, Ms[s], EPI_F32, xw, Cd, xo);
                // cross attention: q from norm2(x), k/v from norm_y(y): Nk = the other view's token count
                D3R_OTHER(launch_layernorm(m->bdt, yo, b.ny.g, b.ny.b, syn, Ms[1 - s], Cd, 1e-6f, c.st));
                D3R_OTHER(launch_layernorm(m->bdt, xw, b.n2.g, b.n2.b, sxn, Ms[s], Cd, 1e-6f, c.st));
                {
                    const int kq[1] = {HEAD_ROPE};
                    void* dq[1] = {sq};
                    gemm_heads(c, sxn, Cd, b.cq, Ms[s], Cd, 1, kq, dq, Hd, own.N, own.tw, own.ldv);
                    const int kkv[2] = {HEAD_ROPE, HEAD_VT};
                    void* dkv[2] = {sk, svt};
                    gemm_heads(c, syn, Cd, b.ckv, Ms[1 - s], Cd, 2, kkv, dkv, Hd, oth.N, oth.tw, oth.ldv);
                    AttnParams a;
                    a.q = sq; a.k = sk; a.vt = svt; a.out = sao; a.B = B; a.H = Hd; a.Nq = own.N; a.Nk = oth.N; a.ldv = oth.ldv; a.scale = 0.125f;
                    a.out_dt = m->bdt;
                    c.mark(PRF_ATTN, 4.0 * B * Hd * (double)own.N * oth.N * 64, B * Hd, own.N, oth.N);
                    c.chk(launch_attention(m->dt, a, c.st));
                }
                gemm_linear(c, sao, Cd, b.cproj, Ms[s], EPI_F32, xw, Cd, xw);
                D3R_OTHER(launch_layernorm(m->bdt, xw, b.n3.g, b.n3.b, sxn, Ms[s], Cd, 1e-6f, c.st));
                gemm_linear(c, sxn, Cd, b.fc1, Ms[s], EPI_GELU, shb, 4 * Cd);
                const int layer_no = l + 1;
                void* hcopy = (cf.head_type == 1 && (layer_no == hk6 || layer_no == hk9)) ? hook[s][layer_no == hk6 ? 0 : 1] : nullptr;
                gemm_linear(c, shb, 4 * Cd, b.fc2, Ms[s], EPI_F32, xw, Cd, xw, hcopy, Cd);
            }
            cross_sync();
            cur ^= 1;
        }
        // ---- dec_norm + heads, each side on its own stream ---------------------------------------------------
        float* pts[2] = {pts1, pts2};
        float* cnf[2] = {conf1, conf2};
        for (int s = 0; s < 2; ++s) {
            c.st = S[s];
            const SideDim& own = D[s];
            if (fold) D3R_OTHER(launch_layernorm_x3in(frp[cur][s], m->dec_norm.g, m->dec_norm.b, hook[s][2], Ms[s], Cd, 1e-6f, c.st));
            else
            D3R_OTHER(launch_layernorm(m->dt, f[cur] + (size_t)Roff[s] * Cd, m->dec_norm.g, m->dec_norm.b, hook[s][2], Ms[s], Cd, 1e-6f, c.st));
            if (cf.head_type == 0) {
                float* lo = lin_out + (size_t)Roff[s] * 4 * ps * ps;
                gemm_linear(c, hook[s][2], Cd, m->lin_head[s], Ms[s], EPI_F32, lo, 4 * ps * ps);
                D3R_OTHER(launch_linear_head_post(lo, pts[s], cnf[s], B, own.th, own.tw, ps, m->out_pstride, m->out_cstride, m->post, c.st));
            } else {
                for (int b0 = 0; b0 < B; b0 += chunk) {
                    const int bc = (B - b0) < chunk ? (B - b0) : chunk;
                    Arena sub((char*)ws + common_end + (size_t)s * head_arena, head_arena);
                    const void* hooks[4] = {(const char*)encn + ((size_t)Roff[s] + (size_t)b0 * own.N) * Ce * eb, (const char*)hook[s][0] + (size_t)b0 * own.N * Cd * eb,
                                            (const char*)hook[s][1] + (size_t)b0 * own.N * Cd * eb, (const char*)hook[s][2] + (size_t)b0 * own.N * Cd * eb};
                    const int hc[4] = {Ce, Cd, Cd, Cd};
                    run_dpt(c, m->dpt[s], sub, hooks, hc, bc, own.th, own.tw, pts[s] + (size_t)b0 * own.H * own.W * m->out_pstride,
                            cnf[s] + (size_t)b0 * own.H * own.W * m->out_cstride);
                }
            }
        }
        c.st = st;
        if (two) {   // join: the caller's stream continues only after side 1 has finished
            c.chk(hipEventRecord(m->ev_side, S[1]));
            c.chk(hipStreamWaitEvent(S[0], m->ev_side, 0));
        }
        }  // do_dec
        c.mark(PRF_END, 0.0);
    }
    if (rc_out) *rc_out = c.rc;
    return common_end + 2 * head_arena + 256;
}

}  // namespace

// ln_fold engines: (re)pack the nn.Linear matrices behind a folded LayerNorm whose inputs changed as W diag(gamma) and form their column sums / folded bias,
// from the fp32 rows staged by pack_slot. Runs at the first forward after a load; the staging copies are released afterwards. Bookkeeping is PER MATRIX
// (round 6; before, one stale vector made every one of the ~100 folded matrices demand a reload): a matrix is re-folded when one of its weight tensors, its
// bias or its LayerNorm's vectors were loaded, and it needs ALL of its weight tensors staged for that (projk | projv: both). A vector-only update of a
// matrix whose staging copy is gone cannot be folded: D3R_ERR_STATE until that matrix's weights are loaded again (include/dust3r_hip.h, d3r_model_load_tensor);
// the other matrices are unaffected.
static int finalize_fold(d3r_model* m) {
    if (!m->ln_fold || !m->fold_dirty) return D3R_OK;
    for (Lin* L : m->fold_lins) if (L->fold_dirty && (!L->w32 || L->have_mask != L->want_mask)) return D3R_ERR_STATE;
    for (Lin* L : m->fold_lins) {
        if (!L->fold_dirty) continue;
        PackParams pp;
        pp.src = L->w32; pp.numel = (size_t)L->N * L->K; pp.dst = L->w; pp.dst_cols = L->K; pp.kind = PACK_MAT; pp.cols = L->K; pp.row_off = 0;
        pp.kscale = L->ln->g;
        if (launch_pack_weight(L->dt, pp, nullptr) != hipSuccess) return D3R_ERR_LAUNCH;
        if (launch_ln_fold_vectors(L->dt, L->w32, L->ln->g, L->ln->b, L->b, L->ln_s, L->b_fold, L->N, L->K, nullptr) != hipSuccess) return D3R_ERR_LAUNCH;
    }
    if (hipDeviceSynchronize() != hipSuccess) return D3R_ERR_LAUNCH;
    for (Lin* L : m->fold_lins) {
        if (!L->fold_dirty) continue;
        (void)hipFree(L->w32);
        L->w32 = nullptr;
        L->have_mask = 0;
        L->fold_dirty = false;
    }
    m->fold_dirty = false;
    return D3R_OK;
}

static int run_phases(d3r_model* m, int phases, const float* img1, const float* img2, int nimg1, int nimg, void* feat, int B, int H1, int W1,
                      int H2, int W2, float* pts1, float* conf1, float* pts2, float* conf2, hipStream_t st) {
    // The engines of a process share their helper streams (shared_stream above): the ENQUEUE of a forward -- host side only, the device work stays asynchronous --
    // is therefore one critical section per process. Engines driven from different host threads stay correct (one engine's stream capture cannot swallow another
    // engine's launches on the shared side stream; fork / join events of two forwards do not interleave); their side-stream halves run in enqueue order.
    static std::mutex enqueue_mu;
    std::lock_guard<std::mutex> enqueue_lock(enqueue_mu);
    const int ps = m->cfg.patch_size;
    for (int v = 0; v < 2; ++v) {
        const int H = v ? H2 : H1, W = v ? W2 : W1;
        if (H % ps || W % ps || H <= 0 || W <= 0 || H / ps > 511 || W / ps > 511) return D3R_ERR_SHAPE;
    }
    if (d3r_model_missing(m) != 0) return D3R_ERR_STATE;
    { const int frc = finalize_fold(m); if (frc != D3R_OK) return frc; }
    const SideDim d0 = side_dim(H1, W1, ps), d1 = side_dim(H2, W2, ps);
    const size_t need = forward_impl(m, nullptr, 0, phases, img1, img2, nimg1, nimg, feat, B, d0, d1, pts1, conf1, pts2, conf2, st, nullptr);
    if (need > m->ws_bytes) {
        (void)hipStreamSynchronize(st);
        if (m->side) (void)hipStreamSynchronize(m->side);
        m->drop_graphs();                         // captured launches point into the old workspace
        if (m->ws) (void)hipFree(m->ws);
        m->ws = nullptr; m->ws_bytes = 0;
        if (hipMalloc(&m->ws, need) != hipSuccess) return D3R_ERR_ALLOC;
        m->ws_bytes = need;
    }
    int rc = D3R_OK;
    m->prof_rec.clear();
    // ---- small whole forwards: replay a captured graph (see d3r_model::GraphEntry) ---------------------------------------------------
    const bool graphable = phases == (PH_ENCODE | PH_DECODE) && m->graph_max_pairs > 0 && B <= m->graph_max_pairs && !m->prof_on && nimg1 == B && nimg == 2 * B;
    if (graphable) {
        const d3r_model::GraphKey key{B, H1, W1, H2, W2, m->out_pstride, (m->two_streams && m->side) ? 1 : 0};
        d3r_model::GraphEntry* ge = nullptr;
        for (auto& g : m->graphs)
            if (g.key == key) { ge = &g; break; }
        if (!ge) {
            if (m->graphs.size() >= 16) {                       // a handful of (batch, size) combinations is the expected use
                (void)hipStreamSynchronize(st);                 // a replay enqueued by the previous call (and its staging copies) may still be in flight
                if (m->side) (void)hipStreamSynchronize(m->side);
                if (m->cap) (void)hipStreamSynchronize(m->cap);
                m->drop_graphs();
            }
            m->graphs.push_back(d3r_model::GraphEntry());
            ge = &m->graphs.back();
            ge->key = key;
        }
        const size_t n_in1 = (size_t)B * 3 * H1 * W1, n_in2 = (size_t)B * 3 * H2 * W2;
        const size_t a1 = (size_t)B * H1 * W1, a2 = (size_t)B * H2 * W2;
        const bool packed = m->out_pstride == 8;                 // one [B][H][W][8] payload (H1 == H2, W1 == W2)
        const size_t n_out = packed ? a1 * 8 : 4 * (a1 + a2);
        // staging layout (floats): img1 | img2 | pts1 conf1 pts2 conf2 (or the packed payload)
        float* io = ge->io;
        auto s_img1 = [&]() { return io; };
        auto s_img2 = [&]() { return io + n_in1; };
        auto s_out = [&]() { return io + n_in1 + n_in2; };
        if (ge->seen >= 1 && !ge->exec && ge->seen < 1000) {     // second call: capture
            bool ok = true;
            if (!m->cap) ok = hipStreamCreateWithFlags(&m->cap, hipStreamNonBlocking) == hipSuccess;
            const size_t bytes = (n_in1 + n_in2 + n_out) * sizeof(float);
            if (ok && !ge->io) { ok = hipMalloc((void**)&ge->io, bytes) == hipSuccess; ge->io_bytes = bytes; io = ge->io; }
            if (ok) {
                float *o1 = s_out(), *c1, *o2, *c2;
                if (packed) { c1 = o1 + 3; o2 = o1 + 4; c2 = o1 + 7; }
                else { c1 = o1 + 3 * a1; o2 = c1 + a1; c2 = o2 + 3 * a2; }
                ok = hipStreamBeginCapture(m->cap, hipStreamCaptureModeRelaxed) == hipSuccess;
                if (ok) {
                    int crc = D3R_OK;
                    forward_impl(m, m->ws, m->ws_bytes, phases, s_img1(), s_img2(), nimg1, nimg, feat, B, d0, d1, o1, c1, o2, c2, m->cap, &crc);
                    hipGraph_t g = nullptr;
                    const bool ended = hipStreamEndCapture(m->cap, &g) == hipSuccess && g != nullptr;
                    ok = ended && crc == D3R_OK && hipGraphInstantiate(&ge->exec, g, nullptr, nullptr, 0) == hipSuccess;
                    if (ok) ge->graph = g;
                    else if (g) (void)hipGraphDestroy(g);
                }
            }
            if (!ok) {                        // this shape stays eager
                (void)hipGetLastError();
                ge->exec = nullptr;
                ge->seen = 1000;
            }
        }
        if (ge->exec) {
            io = ge->io;
            bool ok = hipMemcpyAsync(s_img1(), img1, n_in1 * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess &&
                      hipMemcpyAsync(s_img2(), img2, n_in2 * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess &&
                      hipGraphLaunch(ge->exec, st) == hipSuccess;
            const float* o1 = s_out();
            if (ok && packed) ok = hipMemcpyAsync(pts1, o1, n_out * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess;
            else if (ok)
                ok = hipMemcpyAsync(pts1, o1, 3 * a1 * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess &&
                     hipMemcpyAsync(conf1, o1 + 3 * a1, a1 * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess &&
                     hipMemcpyAsync(pts2, o1 + 4 * a1, 3 * a2 * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess &&
                     hipMemcpyAsync(conf2, o1 + 4 * a1 + 3 * a2, a2 * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess;
            if (ok) { ++m->graph_replays; return D3R_OK; }
            return D3R_ERR_LAUNCH;
        }
        if (ge->seen < 1000) ++ge->seen;
    }
    forward_impl(m, m->ws, m->ws_bytes, phases, img1, img2, nimg1, nimg, feat, B, d0, d1, pts1, conf1, pts2, conf2, st, &rc);
    return rc;
}

extern "C" long d3r_model_graph_replays(const d3r_model* m) { return m ? m->graph_replays : -1; }

extern "C" int d3r_model_forward(d3r_model* m, const float* img1, const float* img2, int B, int H, int W, float* pts1, float* conf1,
                                 float* pts2, float* conf2, void* stream) {
    if (!m || !img1 || !img2 || B <= 0 || !pts1 || !conf1 || !pts2 || !conf2) return D3R_ERR_INVALID;
    return run_phases(m, PH_ENCODE | PH_DECODE, img1, img2, B, 2 * B, nullptr, B, H, W, H, W, pts1, conf1, pts2, conf2, (hipStream_t)stream);
}

extern "C" int d3r_model_forward_mixed(d3r_model* m, const float* img1, int H1, int W1, const float* img2, int H2, int W2, int B, float* pts1,
                                       float* conf1, float* pts2, float* conf2, void* stream) {
    if (!m || !img1 || !img2 || B <= 0 || !pts1 || !conf1 || !pts2 || !conf2) return D3R_ERR_INVALID;
    return run_phases(m, PH_ENCODE | PH_DECODE, img1, img2, B, 2 * B, nullptr, B, H1, W1, H2, W2, pts1, conf1, pts2, conf2, (hipStream_t)stream);
}

extern "C" int d3r_model_forward_packed(d3r_model* m, const float* img1, const float* img2, int B, int H, int W, float* out8, void* stream) {
    if (!m || !img1 || !img2 || B <= 0 || !out8) return D3R_ERR_INVALID;
    m->out_pstride = 8; m->out_cstride = 8;
    const int rc = run_phases(m, PH_ENCODE | PH_DECODE, img1, img2, B, 2 * B, nullptr, B, H, W, H, W, out8, out8 + 3, out8 + 4, out8 + 7, (hipStream_t)stream);
    m->out_pstride = 3; m->out_cstride = 1;
    return rc;
}

extern "C" size_t d3r_model_feature_bytes(const d3r_model* m, int H, int W) {
    if (!m || H <= 0 || W <= 0) return 0;
    const int ps = m->cfg.patch_size;
    return (size_t)(H / ps) * (W / ps) * m->cfg.enc_embed_dim * dt_bytes(m->dt);
}

extern "C" int d3r_model_encode(d3r_model* m, const float* img, int n, int H, int W, void* feat_out, void* stream) {
    if (!m || !img || n <= 0 || !feat_out) return D3R_ERR_INVALID;
    return run_phases(m, PH_ENCODE, img, nullptr, n, n, feat_out, 0, H, W, H, W, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int d3r_model_decode(d3r_model* m, const void* feat, int B, int H, int W, float* pts1, float* conf1, float* pts2, float* conf2,
                                void* stream) {
    if (!m || !feat || B <= 0 || !pts1 || !conf1 || !pts2 || !conf2) return D3R_ERR_INVALID;
    return run_phases(m, PH_DECODE, nullptr, nullptr, 0, 0, const_cast<void*>(feat), B, H, W, H, W, pts1, conf1, pts2, conf2, (hipStream_t)stream);
}

extern "C" int d3r_model_decode_packed(d3r_model* m, const void* feat, int B, int H, int W, float* out8, void* stream) {
    if (!m || !feat || B <= 0 || !out8) return D3R_ERR_INVALID;
    m->out_pstride = 8; m->out_cstride = 8;
    const int rc = run_phases(m, PH_DECODE, nullptr, nullptr, 0, 0, const_cast<void*>(feat), B, H, W, H, W, out8, out8 + 3, out8 + 4, out8 + 7, (hipStream_t)stream);
    m->out_pstride = 3; m->out_cstride = 1;
    return rc;
}
