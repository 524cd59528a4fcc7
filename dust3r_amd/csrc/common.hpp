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
// The f32 instantiation is the "reference-exact" precision mode (the reference runs fp32,
// dust3r/inference.py:44); it shares tiles, LDS images and epilogues with the 16-bit modes
// because all of them move operands as 16-byte chunks (8 x 16-bit or 4 x f32) and the MFMA
// contraction index may be permuted freely as long as both operands use the same map.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define D3R_BF16 0
#define D3R_F16 1
#define D3R_F32 2
#define D3R_F16X3 3   // split fp16: every value is a (hi, lo) fp16 pair, products use 3 MFMAs (hi*hi + hi*lo + lo*hi)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

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
    // split two floats into packed (hi, hi) and (lo, lo) fp16 pairs; inputs saturate at the fp16 range
    D3R_DEV static void split2(float x, float y, uint32_t& hi, uint32_t& lo) {
        typedef __attribute__((ext_vector_type(2))) _Float16 v2;
        x = fminf(fmaxf(x, -65504.f), 65504.f);
        y = fminf(fmaxf(y, -65504.f), 65504.f);
        const _Float16 hx = (_Float16)x, hy = (_Float16)y;
        v2 h = {hx, hy};
        v2 l = {(_Float16)(x - (float)hx), (_Float16)(y - (float)hy)};
        hi = __builtin_bit_cast(uint32_t, h);
        lo = __builtin_bit_cast(uint32_t, l);
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
    } else reinterpret_cast<uint16_t*>(base)[elem_off] = (uint16_t)(Traits<DT>::pack2(a, 0.f) & 0xFFFFu);
}
template <int DT> D3R_DEV float load1(const void* base, size_t elem_off) {
    if constexpr (DT == D3R_F32) return reinterpret_cast<const float*>(base)[elem_off];
    else if constexpr (DT == D3R_F16X3) {
        const char* p = reinterpret_cast<const char*>(base) + Traits<D3R_F16X3>::boff(elem_off);
        return (float)*reinterpret_cast<const _Float16*>(p) + (float)*reinterpret_cast<const _Float16*>(p + 16);
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
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
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

// XCD-aware bijective block remap (8 XCDs, block b runs on XCD b % 8): gives each XCD a
// contiguous range of logical tile ids so neighbouring tiles share operand panels in one L2.
D3R_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t dt_bytes(int dt) { return (dt == D3R_F32 || dt == D3R_F16X3) ? 4 : 2; }
