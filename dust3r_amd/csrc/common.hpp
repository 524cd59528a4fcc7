// dust3r_amd -- common device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Every matrix kernel is written once against `Traits<DT>`; DT selects the MFMA family:
//   D3R_BF16 / D3R_F16 : v_mfma_f32_{16x16x32,32x32x16}_{bf16,f16}   (2.5 PFLOP/s dense peak)
//   D3R_F32            : v_mfma_f32_{16x16x4,32x32x2}_f32            (exact f32, 157 TFLOP/s)
//   D3R_F16X3          : the f16 MFMAs on split operands x = hi + lo (hi = fp16(x), lo = fp16(x - hi), 22
//                        significand bits together), three MFMAs per product (lo*lo dropped, 2^-22 relative):
//                        fp32-class results at 1/3 of the 16-bit MFMA rate = 5.3x the exact-f32 MFMA rate.
//                        Storage: a row of K logical elements is K*4 bytes made of 32-byte groups
//                        [8 x hi fp16][8 x lo fp16], i.e. the same 16-byte-chunk geometry as every other mode.
//   D3R_F16F8          : the same split, but only hi*hi runs on the f16 MFMA; the cross terms hi*lo + lo*hi run on
//                        v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 copies of the four factors (twice the 16-bit rate,
//                        both cross terms K-concatenated into one instruction): 2 MFMA units per product instead of 3,
//                        ~15-16 significand bits per operand. Storage: see Traits<D3R_F16F8>.
// The f32 instantiation is the "reference-exact" precision mode (the reference runs fp32,
// dust3r/inference.py:44); it shares tiles, LDS images and epilogues with the 16-bit modes
// because all of them move operands as 16-byte chunks (8 x 16-bit or 4 x f32) and the MFMA
// contraction index may be permuted freely as long as both operands use the same map.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// Probe switches. The default build (no -DD3R_PROBES) reads only the documented product / A-B switches from the environment and does not
// instantiate the kernels that exist for ablations only; `D3R_PROBES=1 python -m dust3r_amd.build` (tools/, profiles/ probes) compiles them back in.
// d3r_build_has_probes() (include/dust3r_hip.h) reports which build is loaded; tests of probe-only variants skip on a default build.
#ifdef D3R_PROBES
constexpr bool kProbes = true;
static inline const char* probe_env(const char* name) { return getenv(name); }
#else
constexpr bool kProbes = false;
static inline const char* probe_env(const char*) { return nullptr; }
#endif

#define D3R_BF16 0
#define D3R_F16 1
#define D3R_F32 2
#define D3R_F16X3 3   // split fp16: every value is a (hi, lo) fp16 pair, products use 3 MFMAs (hi*hi + hi*lo + lo*hi)
#define D3R_F16F8 4   // fp16 + fp8: hi*hi on the f16 MFMA, the two cross terms hi*lo + lo*hi on ONE K-concatenated fp8 (e4m3) MFMA at twice the rate
#define D3R_F16X2F8 5 // 2.5 MFMA units per product: hi*hi and hi*w_lo on the f16 MFMA (the WEIGHTS keep their 22 bits), only a_lo*w_hi on the e4m3 MFMA.
                      // Activation rows: the D3R_F16F8 layout (hi fp16 + b8 = e4m3(lo 2^11); the a8 copy is not read). Weight rows: Traits<D3R_F16X2F8>.

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

#define D3R_DEV __device__ __forceinline__

template <int DT> struct Traits;

template <> struct Traits<D3R_BF16> {
    static constexpr int EB = 2;   // bytes per element
    static constexpr int CH = 8;   // elements per 16-byte chunk
    D3R_DEV static void mma16(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
    D3R_DEV static void mma32(f32x16_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
    D3R_DEV static uint32_t pack2(float lo, float hi) {
        typedef __attribute__((ext_vector_type(2))) __bf16 v2;
        v2 t = {(__bf16)lo, (__bf16)hi};
        return __builtin_bit_cast(uint32_t, t);
    }
    D3R_DEV static float unpack_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
    D3R_DEV static float unpack_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
};

template <> struct Traits<D3R_F16> {
    static constexpr int EB = 2;
    static constexpr int CH = 8;
    D3R_DEV static void mma16(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
    D3R_DEV static void mma32(f32x16_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
    D3R_DEV static uint32_t pack2(float lo, float hi) {
        typedef __attribute__((ext_vector_type(2))) _Float16 v2;
        v2 t = {(_Float16)lo, (_Float16)hi};
        return __builtin_bit_cast(uint32_t, t);
    }
    D3R_DEV static float unpack_lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xFFFFu)); }
    D3R_DEV static float unpack_hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
};

template <> struct Traits<D3R_F32> {
    static constexpr int EB = 4;
    static constexpr int CH = 4;
    // one 16-byte chunk = 4 consecutive k; MFMA #j consumes element j of every lane's chunk, so
    // k-slot (lane>>4) of MFMA j is global k = 4*(lane>>4)+j on BOTH operands (consistent permutation).
    D3R_DEV static void mma16(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
    }
    D3R_DEV static void mma32(f32x16_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
    }
};

template <> struct Traits<D3R_F16X3> {
    static constexpr int EB = 4;   // bytes per LOGICAL element (2 hi + 2 lo)
    static constexpr int CH = 4;
    D3R_DEV static f16x8_t h8(const uint4& a) { return __builtin_bit_cast(f16x8_t, a); }
    // a, b: hi chunks; al, bl: lo chunks (8 consecutive k each). Small terms first.
    D3R_DEV static void mma16x3(f32x4_t& acc, const uint4& a, const uint4& al, const uint4& b, const uint4& bl) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(al), h8(b), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(a), h8(bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(a), h8(b), acc, 0, 0, 0);
    }
    // one of the three terms of mma16x3 (same order: 0 = lo*hi, 1 = hi*lo, 2 = hi*hi), for loops that interleave accumulators
    D3R_DEV static void mma16_term(int term, f32x4_t& acc, const uint4& a, const uint4& al, const uint4& b, const uint4& bl) {
        if (term == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(al), h8(b), acc, 0, 0, 0);
        else if (term == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(a), h8(bl), acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(h8(a), h8(b), acc, 0, 0, 0);
    }
    D3R_DEV static void mma32x3(f32x16_t& acc, const uint4& a, const uint4& al, const uint4& b, const uint4& bl) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(al), h8(b), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a), h8(bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(a), h8(b), acc, 0, 0, 0);
    }
    // split two floats into packed (hi, hi) and (lo, lo) fp16 pairs; inputs saturate at the fp16 range. Written on 2-vectors so that hipcc
    // emits the packed forms (v_cvt_pk_f16_f32, v_pk_add_f32 with a negated operand: 5 VALU per pair instead of 8); same roundings.
    typedef float v2f_t __attribute__((ext_vector_type(2)));
    typedef _Float16 v2h_t __attribute__((ext_vector_type(2)));
    D3R_DEV static void split2_inrange(float x, float y, uint32_t& hi, uint32_t& lo) {
        const v2f_t xv = {x, y};
        const v2h_t h = __builtin_convertvector(xv, v2h_t);
        const v2f_t d = xv - __builtin_convertvector(h, v2f_t);
        const v2h_t l = __builtin_convertvector(d, v2h_t);
        hi = __builtin_bit_cast(uint32_t, h);
        lo = __builtin_bit_cast(uint32_t, l);
    }
    D3R_DEV static void split2(float x, float y, uint32_t& hi, uint32_t& lo) {
        x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);   // one v_med3_f32 (fminf(fmaxf()) costs two canonicalising v_max on top)
        y = __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);
        split2_inrange(x, y, hi, lo);
    }
    D3R_DEV static float join_lo(uint32_t hi, uint32_t lo) {
        return (float)__builtin_bit_cast(_Float16, (uint16_t)(hi & 0xFFFFu)) + (float)__builtin_bit_cast(_Float16, (uint16_t)(lo & 0xFFFFu));
    }
    D3R_DEV static float join_hi(uint32_t hi, uint32_t lo) {
        return (float)__builtin_bit_cast(_Float16, (uint16_t)(hi >> 16)) + (float)__builtin_bit_cast(_Float16, (uint16_t)(lo >> 16));
    }
    // byte offset of logical element e inside a tensor whose rows start at multiples of 8 elements
    D3R_DEV static size_t boff(size_t e) { return (e >> 3) * 32 + (e & 7) * 2; }
};

// fp16 + fp8 operands. A row of K logical elements (K % 64 == 0) is K*4 bytes made of 256-byte super-groups of 64 elements:
//     [hi fp16 x64 (128 B) | a8 e4m3 x64 (64 B) | b8 e4m3 x64 (64 B)]
//   activations:  a8 = e4m3(hi),               b8 = e4m3(lo * 2^11)        (hi = fp16(x), lo = x - hi)
//   weights:      a8 = e4m3(lo * 2^(11 + 6)),  b8 = e4m3(hi * 2^6)         (2^6: |w| of O(0.01) sits in e4m3's normal range)
// so that, slot by slot, a8x*a8w + b8x*b8w = (hi_x*lo_w + lo_x*hi_w) * 2^17: one fp8 MFMA over the concatenated [a8 | b8] halves
// yields both cross terms, and its E8M0 scale operand (2^-17) undoes the factor inside the instruction. A GEMM K step (128 bytes of
// every row) is alternately the fp16 half of a super-group (two 16x16x32 f16 MFMAs per fragment pair) and its fp8 half (one
// 16x16x128 fp8 MFMA): per 64 logical k 64 MFMA cycles instead of the 96 of three f16 MFMAs (measured 1.47x, tools/f8_probe.hip).
// v_cvt_pk_fp8_f32 rounds to nearest even, keeps subnormals and returns NaN above 464: the sources are clamped to +-448 first.
template <> struct Traits<D3R_F16F8> {
    static constexpr int EB = 4;   // bytes per LOGICAL element
    static constexpr int CH = 4;
    static constexpr int SCALE_P = 0x6E6E6E6E;   // E8M0 2^(110 - 127) = 2^-17 in every byte (whichever op_sel)
    static constexpr int SCALE_Q = 0x7F7F7F7F;   // 1.0
    D3R_DEV static void mma16_hi(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
    }
    // a0 / b0: the lane's 16 a8 bytes, a1 / b1: its 16 b8 bytes (same 16 logical k on both operands)
    D3R_DEV static void mma16_f8(f32x4_t& acc, const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
        // register-tuple concatenation (a shufflevector of two 4-vectors: no moves; an element-wise initialiser list compiles to shuffles)
        typedef __attribute__((ext_vector_type(4))) int i32x4_t;
        const i32x8_t A = __builtin_shufflevector(__builtin_bit_cast(i32x4_t, a0), __builtin_bit_cast(i32x4_t, a1), 0, 1, 2, 3, 4, 5, 6, 7);
        const i32x8_t B = __builtin_shufflevector(__builtin_bit_cast(i32x4_t, b0), __builtin_bit_cast(i32x4_t, b1), 0, 1, 2, 3, 4, 5, 6, 7);
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 0, 0, 0, SCALE_P, 0, SCALE_Q);
    }
    D3R_DEV static float clamp8(float x) { return __builtin_amdgcn_fmed3f(x, -448.f, 448.f); }
    // four consecutive values -> 4 x fp16 (hi), 4 x e4m3 (a), 4 x e4m3 (b); WGT: the weight encoding
    template <bool WGT> D3R_DEV static void enc4(float v0, float v1, float v2, float v3, uint2& hi, uint32_t& a, uint32_t& b) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
        v0 = __builtin_amdgcn_fmed3f(v0, -65504.f, 65504.f); v1 = __builtin_amdgcn_fmed3f(v1, -65504.f, 65504.f);
        v2 = __builtin_amdgcn_fmed3f(v2, -65504.f, 65504.f); v3 = __builtin_amdgcn_fmed3f(v3, -65504.f, 65504.f);
        const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1, h2 = (_Float16)v2, h3 = (_Float16)v3;
        const h2_t p01 = {h0, h1}, p23 = {h2, h3};
        hi.x = __builtin_bit_cast(uint32_t, p01);
        hi.y = __builtin_bit_cast(uint32_t, p23);
        const float f0 = (float)h0, f1 = (float)h1, f2 = (float)h2, f3 = (float)h3;
        constexpr float SH = WGT ? 64.f : 1.f, SL = WGT ? 131072.f : 2048.f;
        const float q0 = clamp8(f0 * SH), q1 = clamp8(f1 * SH), q2 = clamp8(f2 * SH), q3 = clamp8(f3 * SH);
        const float r0 = clamp8((v0 - f0) * SL), r1 = clamp8((v1 - f1) * SL), r2 = clamp8((v2 - f2) * SL), r3 = clamp8((v3 - f3) * SL);
        int th = __builtin_amdgcn_cvt_pk_fp8_f32(q0, q1, 0, false);
        th = __builtin_amdgcn_cvt_pk_fp8_f32(q2, q3, th, true);
        int tl = __builtin_amdgcn_cvt_pk_fp8_f32(r0, r1, 0, false);
        tl = __builtin_amdgcn_cvt_pk_fp8_f32(r2, r3, tl, true);
        a = (uint32_t)(WGT ? tl : th);
        b = (uint32_t)(WGT ? th : tl);
    }
    // byte offsets of logical element e (rows start at multiples of 64 elements)
    D3R_DEV static size_t off_hi(size_t e) { return (e >> 6) * 256 + (e & 63) * 2; }
    D3R_DEV static size_t off_a(size_t e) { return (e >> 6) * 256 + 128 + (e & 63); }
    D3R_DEV static size_t off_b(size_t e) { return (e >> 6) * 256 + 192 + (e & 63); }
    // value of an ACTIVATION element: hi + lo8 * 2^-11 (15-16 bits; device code only reads this format back in tests / fallbacks)
    D3R_DEV static float dec(uint16_t h, uint32_t bword, int byte) {
        float l;
        switch (byte) {
            case 0: l = __builtin_amdgcn_cvt_f32_fp8((int)bword, 0); break;
            case 1: l = __builtin_amdgcn_cvt_f32_fp8((int)bword, 1); break;
            case 2: l = __builtin_amdgcn_cvt_f32_fp8((int)bword, 2); break;
            default: l = __builtin_amdgcn_cvt_f32_fp8((int)bword, 3); break;
        }
        return (float)__builtin_bit_cast(_Float16, h) + l * (1.0f / 2048.f);
    }
    // one WEIGHT element (load-time packing: one thread per source element)
    D3R_DEV static void store1_wgt(void* base, size_t e, float v) {
        uint2 hi; uint32_t a, b;
        enc4<true>(v, 0.f, 0.f, 0.f, hi, a, b);
        char* p = reinterpret_cast<char*>(base);
        *reinterpret_cast<uint16_t*>(p + off_hi(e)) = (uint16_t)(hi.x & 0xFFFFu);
        *reinterpret_cast<uint8_t*>(p + off_a(e)) = (uint8_t)(a & 0xFFu);
        *reinterpret_cast<uint8_t*>(p + off_b(e)) = (uint8_t)(b & 0xFFu);
    }
};

// 2.5-unit arithmetic (round 4). The numerics study (tools/precision_fp8cross.py a25, profiles/r04_cpu) attributes the error of the fp16 + fp8
// scheme to ONE of its two e4m3 cross terms: a_hi * w_lo -- the same rounded weight residue meets every token, a systematic error --, while
// a_lo * w_hi (the rounding residue of an activation is noise-like) costs 1/3 of it. Here the harmful term stays on the f16 MFMA:
//     x.w  ~=  a_hi * w_hi  +  a_hi * w_lo   (f16 MFMAs, fp16 w_lo: 22-bit weights)   +   e4m3(a_lo 2^11) * e4m3(w_hi 2^6) 2^-17   (fp8 MFMA)
// A weight row of K logical elements (K % 128 == 0) is 5 K bytes: per 128 k five 128-byte chunks
//     [w_hi k 0..63 fp16 | w_lo k 0..63 fp16 | w_hi k 64..127 | w_lo k 64..127 | h8 k 0..127 e4m3(w_hi 2^6)]
// and the K loop walks them as five K steps per 128 k (gemm.hip): four steps of two f16 MFMA k-steps and one fp8 step whose 16x16x128 MFMA takes the
// 128 k of b8 (activation rows, gathered from two super-groups by the DMA's source addresses) against the h8 chunk: 80 MFMA cycles per 64 k
// instead of 96 (fp16x3) / 64 (fp16f8).
template <> struct Traits<D3R_F16X2F8> : Traits<D3R_F16F8> {
    D3R_DEV static size_t wrow_bytes(size_t K) { return K * 5; }
    // one WEIGHT element (load-time packing): row r of a matrix with K logical columns, column c
    D3R_DEV static void store1_wgt5(void* base, size_t r, size_t c, size_t K, float v) {
        v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
        const _Float16 h = (_Float16)v;
        const float hf = (float)h;
        const _Float16 l = (_Float16)(v - hf);
        const int h8 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp8(hf * 64.f), 0.f, 0, false);
        char* p = reinterpret_cast<char*>(base) + r * (K * 5) + (c >> 7) * 640;
        const size_t j = c & 127, half = j >> 6, jj = j & 63;
        *reinterpret_cast<_Float16*>(p + (2 * half) * 128 + jj * 2) = h;
        *reinterpret_cast<_Float16*>(p + (2 * half + 1) * 128 + jj * 2) = l;
        *reinterpret_cast<uint8_t*>(p + 512 + j) = (uint8_t)(h8 & 0xFF);
    }
};
// activation-row layout of a GEMM dtype: the 2.5-unit mode shares the fp16 + fp8 rows
__host__ __device__ constexpr int d3r_act_dt(int dt) { return dt == D3R_F16X2F8 ? D3R_F16F8 : dt; }

// ---- typed 4-element (row-contiguous) loads / stores used by every epilogue -----------------
template <int DT> D3R_DEV void store4(void* base, size_t elem_off, float a, float b, float c, float d) {
    if constexpr (DT == D3R_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + elem_off) = make_float4(a, b, c, d);
    } else if constexpr (DT == D3R_F16X3) {   // elem_off % 4 == 0: the 4 elements share one 8-group
        using TX = Traits<D3R_F16X3>;
        uint2 h, l;
        TX::split2(a, b, h.x, l.x);
        TX::split2(c, d, h.y, l.y);
        char* p = reinterpret_cast<char*>(base) + TX::boff(elem_off);
        *reinterpret_cast<uint2*>(p) = h;
        *reinterpret_cast<uint2*>(p + 16) = l;
    } else if constexpr (DT == D3R_F16F8 || DT == D3R_F16X2F8) {   // activation encoding; elem_off % 4 == 0
        using TF = Traits<D3R_F16F8>;
        uint2 h; uint32_t a8, b8;
        TF::enc4<false>(a, b, c, d, h, a8, b8);
        char* p = reinterpret_cast<char*>(base);
        *reinterpret_cast<uint2*>(p + TF::off_hi(elem_off)) = h;
        *reinterpret_cast<uint32_t*>(p + TF::off_a(elem_off)) = a8;
        *reinterpret_cast<uint32_t*>(p + TF::off_b(elem_off)) = b8;
    } else {
        uint2 v;
        v.x = Traits<DT>::pack2(a, b);
        v.y = Traits<DT>::pack2(c, d);
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + elem_off) = v;
    }
}
template <int DT> D3R_DEV float4 load4(const void* base, size_t elem_off) {
    if constexpr (DT == D3R_F32) {
        return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem_off);
    } else if constexpr (DT == D3R_F16X3) {
        using TX = Traits<D3R_F16X3>;
        const char* p = reinterpret_cast<const char*>(base) + TX::boff(elem_off);
        const uint2 h = *reinterpret_cast<const uint2*>(p), l = *reinterpret_cast<const uint2*>(p + 16);
        return make_float4(TX::join_lo(h.x, l.x), TX::join_hi(h.x, l.x), TX::join_lo(h.y, l.y), TX::join_hi(h.y, l.y));
    } else if constexpr (DT == D3R_F16F8 || DT == D3R_F16X2F8) {
        using TF = Traits<D3R_F16F8>;
        const char* p = reinterpret_cast<const char*>(base);
        const uint2 h = *reinterpret_cast<const uint2*>(p + TF::off_hi(elem_off));
        const uint32_t b8 = *reinterpret_cast<const uint32_t*>(p + TF::off_b(elem_off));
        return make_float4(TF::dec((uint16_t)(h.x & 0xFFFFu), b8, 0), TF::dec((uint16_t)(h.x >> 16), b8, 1), TF::dec((uint16_t)(h.y & 0xFFFFu), b8, 2),
                           TF::dec((uint16_t)(h.y >> 16), b8, 3));
    } else {
        uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem_off);
        return make_float4(Traits<DT>::unpack_lo(v.x), Traits<DT>::unpack_hi(v.x), Traits<DT>::unpack_lo(v.y),
                           Traits<DT>::unpack_hi(v.y));
    }
}
// 8 consecutive elements, elem_off % 8 == 0: ONE 16-byte access for the 16-bit types (two for fp32 and for the split-fp16 rows)
template <int DT> D3R_DEV void store8(void* base, size_t elem_off, const float (&v)[8]) {
    if constexpr (DT == D3R_BF16 || DT == D3R_F16) {
        uint4 u;
        u.x = Traits<DT>::pack2(v[0], v[1]); u.y = Traits<DT>::pack2(v[2], v[3]);
        u.z = Traits<DT>::pack2(v[4], v[5]); u.w = Traits<DT>::pack2(v[6], v[7]);
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + elem_off) = u;
    } else if constexpr (DT == D3R_F16X3) {
        using TX = Traits<D3R_F16X3>;
        uint4 h, l;
        TX::split2(v[0], v[1], h.x, l.x); TX::split2(v[2], v[3], h.y, l.y);
        TX::split2(v[4], v[5], h.z, l.z); TX::split2(v[6], v[7], h.w, l.w);
        char* p = reinterpret_cast<char*>(base) + TX::boff(elem_off);
        *reinterpret_cast<uint4*>(p) = h;
        *reinterpret_cast<uint4*>(p + 16) = l;
    } else {
        store4<DT>(base, elem_off, v[0], v[1], v[2], v[3]);
        store4<DT>(base, elem_off + 4, v[4], v[5], v[6], v[7]);
    }
}
template <int DT> D3R_DEV void load8(const void* base, size_t elem_off, float (&v)[8]) {
    if constexpr (DT == D3R_BF16 || DT == D3R_F16) {
        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + elem_off);
        v[0] = Traits<DT>::unpack_lo(u.x); v[1] = Traits<DT>::unpack_hi(u.x); v[2] = Traits<DT>::unpack_lo(u.y); v[3] = Traits<DT>::unpack_hi(u.y);
        v[4] = Traits<DT>::unpack_lo(u.z); v[5] = Traits<DT>::unpack_hi(u.z); v[6] = Traits<DT>::unpack_lo(u.w); v[7] = Traits<DT>::unpack_hi(u.w);
    } else if constexpr (DT == D3R_F16X3) {
        using TX = Traits<D3R_F16X3>;
        const char* p = reinterpret_cast<const char*>(base) + TX::boff(elem_off);
        const uint4 h = *reinterpret_cast<const uint4*>(p), l = *reinterpret_cast<const uint4*>(p + 16);
        v[0] = TX::join_lo(h.x, l.x); v[1] = TX::join_hi(h.x, l.x); v[2] = TX::join_lo(h.y, l.y); v[3] = TX::join_hi(h.y, l.y);
        v[4] = TX::join_lo(h.z, l.z); v[5] = TX::join_hi(h.z, l.z); v[6] = TX::join_lo(h.w, l.w); v[7] = TX::join_hi(h.w, l.w);
    } else {
        const float4 a = load4<DT>(base, elem_off), b = load4<DT>(base, elem_off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}
template <int DT> D3R_DEV void store1(void* base, size_t elem_off, float a) {
    if constexpr (DT == D3R_F32) reinterpret_cast<float*>(base)[elem_off] = a;
    else if constexpr (DT == D3R_F16X3) {
        using TX = Traits<D3R_F16X3>;
        uint32_t h, l;
        TX::split2(a, 0.f, h, l);
        char* p = reinterpret_cast<char*>(base) + TX::boff(elem_off);
        *reinterpret_cast<uint16_t*>(p) = (uint16_t)(h & 0xFFFFu);
        *reinterpret_cast<uint16_t*>(p + 16) = (uint16_t)(l & 0xFFFFu);
    } else if constexpr (DT == D3R_F16F8 || DT == D3R_F16X2F8) {
        using TF = Traits<D3R_F16F8>;
        uint2 h; uint32_t a8, b8;
        TF::enc4<false>(a, 0.f, 0.f, 0.f, h, a8, b8);
        char* p = reinterpret_cast<char*>(base);
        *reinterpret_cast<uint16_t*>(p + TF::off_hi(elem_off)) = (uint16_t)(h.x & 0xFFFFu);
        *reinterpret_cast<uint8_t*>(p + TF::off_a(elem_off)) = (uint8_t)(a8 & 0xFFu);
        *reinterpret_cast<uint8_t*>(p + TF::off_b(elem_off)) = (uint8_t)(b8 & 0xFFu);
    } else reinterpret_cast<uint16_t*>(base)[elem_off] = (uint16_t)(Traits<DT>::pack2(a, 0.f) & 0xFFFFu);
}
template <int DT> D3R_DEV float load1(const void* base, size_t elem_off) {
    if constexpr (DT == D3R_F32) return reinterpret_cast<const float*>(base)[elem_off];
    else if constexpr (DT == D3R_F16X3) {
        const char* p = reinterpret_cast<const char*>(base) + Traits<D3R_F16X3>::boff(elem_off);
        return (float)*reinterpret_cast<const _Float16*>(p) + (float)*reinterpret_cast<const _Float16*>(p + 16);
    } else if constexpr (DT == D3R_F16F8 || DT == D3R_F16X2F8) {
        using TF = Traits<D3R_F16F8>;
        const char* p = reinterpret_cast<const char*>(base);
        return TF::dec(*reinterpret_cast<const uint16_t*>(p + TF::off_hi(elem_off)), (uint32_t)*reinterpret_cast<const uint8_t*>(p + TF::off_b(elem_off)), 0);
    } else return Traits<DT>::unpack_lo((uint32_t)reinterpret_cast<const uint16_t*>(base)[elem_off]);
}

// ---- LDS-DMA (global_load_lds_dwordx4) issued from inline asm ---------------------------------------------
// hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first ds_read that follows a
// __builtin_amdgcn_global_load_lds (it cannot prove the read does not alias the DMA's LDS destination), which
// serialises a double-buffered K loop: the next tile's loads are drained before the current tile is computed.
// Issued from asm the loads are invisible to the compiler's counters; the kernels wait for them explicitly
// (d3r_wait_vm0) in front of the barrier that publishes the tile. lds_dst: wave-uniform LDS byte address (the DMA
// writes lane l's 16 bytes at lds_dst + 16 l); gsrc: this lane's source address.
D3R_DEV void glds16(const void* gsrc, uint32_t lds_dst) {
    // M0 declared clobbered (not saved / restored around the load: two scalar instructions per piece, 16 pieces per K step)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
// The same DMA with a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit unsigned byte offset: half the address registers
// of glds16 (a 256 x 256 tile keeps 16 row addresses per lane alive through its K loop).
D3R_DEV void glds16_so(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    // M0 is declared clobbered instead of saved and restored around the load: two scalar instructions less per piece (16 pieces per K step)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
D3R_DEV void d3r_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
D3R_DEV uint32_t lds_addr(const void* p) {
    return (uint32_t)(size_t)(__attribute__((address_space(3))) const char*)p;
}

D3R_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. below the rounding of every 16-bit and split-16-bit
// operand type): one v_rcp, one v_exp and six FMAs instead of libm's branchy erff -- the GELU epilogue runs once per
// accumulator element, 128 times per lane per 256x256 tile. The exact-fp32 mode keeps erff.
D3R_DEV float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __builtin_amdgcn_exp2f(-1.44269504088896340736f * ax * ax);
    return __builtin_copysignf(r, x);
}
template <int DT> D3R_DEV float gelu(float x) {
    if constexpr (DT == D3R_F32) return gelu_erf(x);
    else return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}
// Two GELUs on packed fp32 (v_pk_mul / v_pk_fma: two lanes' worth per issue slot): the epilogue of fc1 runs 128 times per lane per 256 x 256
// tile and is VALU-bound (15 us of an 85 us tile in round 2). Same erf approximation as erf_as (Abramowitz-Stegun 7.1.26), rearranged so that
// no sign has to be copied: gelu(x) = x/2 + |x|/2 erf(|x| / sqrt 2), erf(|z|) = 1 - poly(t) exp(-z^2), t = 1 / (1 + p |z|).
typedef float d3r_v2f_t __attribute__((ext_vector_type(2)));
D3R_DEV d3r_v2f_t gelu_pk(d3r_v2f_t x) {
    typedef d3r_v2f_t v2;
    const v2 z = x * (v2){0.70710678118654752440f, 0.70710678118654752440f};
    v2 az;
    az[0] = fabsf(z[0]); az[1] = fabsf(z[1]);
    const v2 d = __builtin_elementwise_fma(az, (v2){0.3275911f, 0.3275911f}, (v2){1.0f, 1.0f});
    v2 t;
    t[0] = __builtin_amdgcn_rcpf(d[0]); t[1] = __builtin_amdgcn_rcpf(d[1]);
    v2 poly = __builtin_elementwise_fma(t, (v2){1.061405429f, 1.061405429f}, (v2){-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(poly, t, (v2){1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(poly, t, (v2){-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(poly, t, (v2){0.254829592f, 0.254829592f});
    poly = poly * t;
    const v2 e2 = (az * (v2){-1.44269504088896340736f, -1.44269504088896340736f}) * az;
    v2 ex;
    ex[0] = __builtin_amdgcn_exp2f(e2[0]); ex[1] = __builtin_amdgcn_exp2f(e2[1]);
    const v2 r = __builtin_elementwise_fma(-poly, ex, (v2){1.0f, 1.0f});          // erf(|z|) >= 0
    const v2 h = x * (v2){0.5f, 0.5f};
    return __builtin_elementwise_fma(az * (v2){0.70710678118654752440f, 0.70710678118654752440f}, r, h);   // |x| / 2 = |z| / sqrt 2
}
// four accumulator values at once (what every epilogue holds per fragment): exact erf for the fp32 mode, the packed form otherwise
template <int DT> D3R_DEV void gelu4(float& v0, float& v1, float& v2, float& v3) {
    if constexpr (DT == D3R_F32) {
        v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
    } else {
        const d3r_v2f_t a = gelu_pk((d3r_v2f_t){v0, v1}), b = gelu_pk((d3r_v2f_t){v2, v3});
        v0 = a[0]; v1 = a[1]; v2 = b[0]; v3 = b[1];
    }
}

// XCD-aware bijective block remap (8 XCDs, block b runs on XCD b % 8): gives each XCD a
// contiguous range of logical tile ids so neighbouring tiles share operand panels in one L2.
D3R_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ------------------------------------------------------------------------------ head epilogues
// postprocess (dust3r/heads/postprocess.py:10-58). depth_mode (reg_dense_depth :23-47; the reference asserts the bounds away, :29-30):
//   'exp'    pts = xyz / max(|xyz|, 1e-8) * expm1(|xyz|)      (the released checkpoints, model.py:61)
//   'linear' pts = xyz
//   'square' pts = xyz / max(|xyz|, 1e-8) * |xyz|^2
// conf_mode (reg_dense_conf :50-58):
//   'exp'     conf = vmin + min(exp(x), vmax - vmin)          (released: ('exp', 1, inf) -> 1 + exp(x))
//   'sigmoid' conf = (vmax - vmin) * sigmoid(x) + vmin
// pts / conf element strides between pixels: (3, 1) = the reference's separate pts3d / conf tensors; (8, 8) = the packed
// [pixel][pts1 conf1 pts2 conf2] record that the multi-GPU path all-gathers as one payload.
enum { POST_DEPTH_EXP = 0, POST_DEPTH_LINEAR = 1, POST_DEPTH_SQUARE = 2, POST_CONF_EXP = 0, POST_CONF_SIGMOID = 1 };
struct PostMode {
    int depth = POST_DEPTH_EXP, conf = POST_CONF_EXP;
    float cmin = 1.0f, cmax = __builtin_huge_valf();
};
D3R_DEV void postprocess_store(float x, float y, float z, float cl, float* pts, float* conf, size_t pix, int ps, int cs, const PostMode pm) {
    float sc = 1.0f;
    if (pm.depth != POST_DEPTH_LINEAR) {        // wave-uniform
        const float d = sqrtf(x * x + y * y + z * z);
        sc = (pm.depth == POST_DEPTH_EXP ? expm1f(d) : d * d) / fmaxf(d, 1e-8f);
    }
    pts[ps * pix + 0] = x * sc;
    pts[ps * pix + 1] = y * sc;
    pts[ps * pix + 2] = z * sc;
    conf[cs * pix] = pm.conf == POST_CONF_EXP ? pm.cmin + fminf(expf(cl), pm.cmax - pm.cmin)
                                              : (pm.cmax - pm.cmin) * (1.0f / (1.0f + expf(-cl))) + pm.cmin;
}

// sum over the four 16-lane rows of a wave, per column (lane & 15), every lane gets the total: v_permlane32_swap exchanges rows {2, 3} of
// its first operand with rows {0, 1} of its second, v_permlane16_swap rows {1, 3} with rows {0, 2}; with both operands = v the two
// results add up to v[l] + v[l ^ 32] (then ^ 16). Inline asm: hipcc 7.2's permlane swap builtins return their first result twice.
D3R_DEV float rows_sum4(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    v = a + b;
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t dt_bytes(int dt) { return (dt == D3R_F32 || dt == D3R_F16X3 || dt == D3R_F16F8 || dt == D3R_F16X2F8) ? 4 : 2; }   // bytes per logical element of an ACTIVATION row
static inline size_t wgt_bytes(int dt) { return dt == D3R_F16X2F8 ? 5 : dt_bytes(dt); }                                                  // ... of a WEIGHT row
