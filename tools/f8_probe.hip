// Hardware probe (not product, not a test): semantics of the gfx950 instructions the fp16+fp8 operand mode relies on.
//   v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 operands, E8M0 scales): result for all-ones operands, effect of the scale bytes and
//   of op_sel, row / column maps of A, B and D, and that k-slot (lane group, byte) of A pairs with the SAME slot of B;
//   v_cvt_pk_fp8_f32: encoding (OCP e4m3fn), rounding, behaviour out of range.
// Build: hipcc --offload-arch=gfx950 -O2 tools/f8_probe.hip -o tools/f8_probe.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

__global__ void mfma_k(const unsigned char* a, const unsigned char* b, float* d, int sa, int sb, int mode) {
    v8i A, B;
    const int* ai = reinterpret_cast<const int*>(a) + threadIdx.x * 8;
    const int* bi = reinterpret_cast<const int*>(b) + threadIdx.x * 8;
    for (int i = 0; i < 8; ++i) { A[i] = ai[i]; B[i] = bi[i]; }
    v4f c = {0, 0, 0, 0};
    if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0, 0, 0, sa, 0, sb);
    else if (mode == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0, 0, 1, sa, 2, sb);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0, 0, 3, sa, 3, sb);
    for (int i = 0; i < 4; ++i) d[threadIdx.x * 4 + i] = c[i];
}
__global__ void cvt_k(const float* x, unsigned char* o, int n) {
    for (int i = 0; i + 1 < n; i += 2) {
        int r = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], x[i + 1], 0, false);
        o[i] = r & 0xFF; o[i + 1] = (r >> 8) & 0xFF;
    }
}

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
// issue-rate microbenchmark: 8 independent accumulators per wave, 8 waves per CU, every CU busy; random operands (power-realistic)
template <int MODE> __global__ __launch_bounds__(512) void rate_k(const int* src, float* out, int iters) {
    v8i A, B; h8_t ha, hb;
    const int* s = src + (threadIdx.x & 63) * 8;
    for (int i = 0; i < 8; ++i) { A[i] = s[i]; B[i] = s[i + 512]; }
    { typedef __attribute__((ext_vector_type(4))) int v4i; v4i t = {s[0] & 0x3BFF3BFF, s[1] & 0x3BFF3BFF, s[2] & 0x3BFF3BFF, s[3] & 0x3BFF3BFF}; ha = __builtin_bit_cast(h8_t, t);
      v4i u = {s[4] & 0x3BFF3BFF, s[5] & 0x3BFF3BFF, s[6] & 0x3BFF3BFF, s[7] & 0x3BFF3BFF}; hb = __builtin_bit_cast(h8_t, u); }
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4f){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0 || MODE == 2) { acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0); acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, acc[i], 0, 0, 0); }
            if (MODE == 1 || MODE == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[i], 0, 0, 0, 0x6E6E6E6E, 0, 0x7F7F7F7F);
            if (MODE == 3) { acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0); acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, acc[i], 0, 0, 0);
                             acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, ha, acc[i], 0, 0, 0); }
        }
    }
    float t = 0;
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 123.456f) out[0] = t;
}
template <int MODE> static void rate(const char* name, const int* src, float* out, double flop_per_iter_wave) {
    const int iters = 4000, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_k<MODE>, dim3(blocks), dim3(512), 0, 0, src, out, 100); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_k<MODE>, dim3(blocks), dim3(512), 0, 0, src, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 8;
    printf("RATE %-34s %8.3f ms  %8.1f TFLOP/s of MFMA work  (%.1f cycles per 8-accumulator round per SIMD at 2.4 GHz)\n", name, ms, waves * iters * flop_per_iter_wave / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (iters * (waves / (256.0 * 4))));
}
static float e4m3_val(unsigned char c) {
    const int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
    float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    if (e == 15 && m == 7) v = NAN;
    return s ? -v : v;
}
static unsigned char e4m3_enc_int(int v) {   // exact encoding of the integers 1..16
    for (int c = 0; c < 127; ++c) if (e4m3_val((unsigned char)c) == (float)v) return (unsigned char)c;
    return 0;
}
int main() {
    unsigned char *a, *b; float* d;
    hipMallocManaged(&a, 64 * 32); hipMallocManaged(&b, 64 * 32); hipMallocManaged(&d, 64 * 4 * sizeof(float));
    auto run = [&](int sa, int sb, int mode) { hipLaunchKernelGGL(mfma_k, dim3(1), dim3(64), 0, 0, a, b, d, sa, sb, mode); hipDeviceSynchronize(); };
    // T1: all ones
    memset(a, 0x38, 2048); memset(b, 0x38, 2048);
    run(0x7F7F7F7F, 0x7F7F7F7F, 0); printf("T1 ones, scales 127/127: d[0]=%g d[255]=%g (expect 128)\n", d[0], d[255]);
    run(0x7F7F7F6E, 0x7F7F7F7F, 0); printf("T2 scale_a byte0=110 (opsel 0): d[0]=%g (expect 128*2^-17=%g)\n", d[0], 128.0 / 131072.0);
    run(0x7F7F7F7F, 0x7F7F7F6E, 0); printf("T2 scale_b byte0=110 (opsel 0): d[0]=%g\n", d[0]);
    run(0x7F7F6E7F, 0x7F6E7F7F, 1); printf("T2 opsel a=1 b=2, bytes 1/2 = 110: d[0]=%g (expect 128*2^-34=%g)\n", d[0], 128.0 / 131072.0 / 131072.0);
    run(0x6E7F7F7F, 0x7F7F7F7F, 2); printf("T2 opsel a=3, byte3=110: d[0]=%g\n", d[0]);
    run(0, 0, 0); printf("T2 scales 0/0: d[0]=%g\n", d[0]);
    run(0x6E6E6E6E, 0x7F7F7F7F, 0); printf("T2 scale_a all bytes 110: d[0]=%g\n", d[0]);
    // T3: row / column maps: A row i = i+1 (every slot), B row j = j+1
    for (int l = 0; l < 64; ++l) { memset(a + l * 32, e4m3_enc_int((l & 15) + 1), 32); memset(b + l * 32, e4m3_enc_int(1), 32); }
    run(0x7F7F7F7F, 0x7F7F7F7F, 0);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (d[l * 4 + r] != 128.0f * (4 * (l >> 4) + r + 1)) ++bad;
    printf("T3 A rows: D[lane][r] == 128*(4*(lane>>4)+r+1) mismatches %d (d[0..3]=%g %g %g %g, lane16: %g)\n", bad, d[0], d[1], d[2], d[3], d[64]);
    for (int l = 0; l < 64; ++l) { memset(a + l * 32, e4m3_enc_int(1), 32); memset(b + l * 32, e4m3_enc_int((l & 15) + 1), 32); }
    run(0x7F7F7F7F, 0x7F7F7F7F, 0);
    bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (d[l * 4 + r] != 128.0f * ((l & 15) + 1)) ++bad;
    printf("T3 B rows: D[lane][r] == 128*((lane&15)+1) mismatches %d (lane1: %g)\n", bad, d[4]);
    // T4: slot pairing: A = 1 at one slot (g, t) of every row, B = 1 at the same slot, 2 at every other slot
    bad = 0;
    for (int g = 0; g < 4; ++g) for (int t = 0; t < 32; ++t) {
        memset(a, 0, 2048); memset(b, 0x40, 2048);   // 0x40 = 2.0
        for (int r = 0; r < 16; ++r) { a[(g * 16 + r) * 32 + t] = 0x38; b[(g * 16 + r) * 32 + t] = 0x38; }
        run(0x7F7F7F7F, 0x7F7F7F7F, 0);
        for (int i = 0; i < 256; ++i) if (d[i] != 1.0f) { if (bad < 4) printf("   slot (%d,%d): d[%d]=%g\n", g, t, i, d[i]); ++bad; }
    }
    printf("T4 same-slot pairing mismatches %d\n", bad);
    // T5: conversions
    float* x; unsigned char* o;
    const float vals[] = {1.0f, 0.3f, 448.f, 449.f, 464.f, 465.f, 500.f, 1e4f, -1000.f, 0.015625f, 0.0078125f, 0.001953125f, 0.0009765625f, 1e-3f, 0.f, -0.f,
                          1.0625f, 1.1875f, 1.125f, 3e-4f, INFINITY, -INFINITY, 240.f, 0.1f};
    const int n = sizeof(vals) / sizeof(float);
    hipMallocManaged(&x, n * 4); hipMallocManaged(&o, n);
    memcpy(x, vals, n * 4);
    hipLaunchKernelGGL(cvt_k, dim3(1), dim3(1), 0, 0, x, o, n); hipDeviceSynchronize();
    for (int i = 0; i < n; ++i) printf("T5 cvt %12g -> 0x%02X = %g\n", vals[i], o[i], e4m3_val(o[i]));
    {   // issue rates
        int* src; hipMallocManaged(&src, 1024 * 4);
        unsigned seed = 12345u;
        for (int i = 0; i < 1024; ++i) { seed = seed * 1664525u + 1013904223u; src[i] = (int)(seed & 0x77777777u); }   // fp8 bytes with exponent < 15: finite
        const double f16 = 2.0 * 16 * 16 * 32, f8 = 2.0 * 16 * 16 * 128;
        rate<0>("f16 16x16x32 x2 per acc", src, d, 8 * 2 * f16);
        rate<1>("fp8 16x16x128 x1 per acc", src, d, 8 * f8);
        rate<2>("f16 x2 + fp8 x1 per acc (64 k)", src, d, 8 * (2 * f16 + f8));
        rate<3>("f16 x3 per acc (32 k, fp16x3)", src, d, 8 * 3 * f16);
    }
    return 0;
}
