// Hardware probe (not product, not a test): can a gfx950 wave run VALU work under its own MFMAs, and can two waves of a SIMD overlap one's
// MFMAs with the other's VALU? Round 3's pipelined attention kernel alternates one MFMA with ~8 VALU instructions in program order and
// still measured MFMA-busy + VALU-busy ~ 86 % of the SIMD cycles -- as if the two pipes excluded each other. This probe times, per
// iteration, {NM MFMAs 32x32x16 f16 on DEP-chained or independent accumulators} + {NV fp32 FMAs (or NT v_exp_f32)} pinned in program
// order, with 1 or 2 waves per SIMD, every CU busy, random operands. If the pipes overlap: max(32 NM, 4 NV) cycles per iteration and wave
// slot; if they exclude each other: 32 NM + 4 NV.
// Build: hipcc --offload-arch=gfx950 -O2 tools/issue_probe.hip -o tools/issue_probe.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(16))) float f16v_t;
typedef __attribute__((ext_vector_type(4))) float f4v_t;

// NM MFMAs then NV VALU per slot; DEP: all MFMAs of a slot chain on one accumulator, else they rotate over 4
template <int NM, int NV, int NT, bool DEP, bool SMALL>
__global__ __launch_bounds__(512) void probe_k(const int* src, float* out, int iters, long long* cyc) {
    h8_t a, b;
    {
        typedef __attribute__((ext_vector_type(4))) int v4i;
        const int* s = src + (threadIdx.x & 63) * 8;
        v4i t = {s[0] & 0x3BFF3BFF, s[1] & 0x3BFF3BFF, s[2] & 0x3BFF3BFF, s[3] & 0x3BFF3BFF};
        v4i u = {s[4] & 0x3BFF3BFF, s[5] & 0x3BFF3BFF, s[6] & 0x3BFF3BFF, s[7] & 0x3BFF3BFF};
        a = __builtin_bit_cast(h8_t, t); b = __builtin_bit_cast(h8_t, u);
    }
    f16v_t acc[4];
    f4v_t sacc[4];
    for (int i = 0; i < 4; ++i) { acc[i] = (f16v_t){0}; sacc[i] = (f4v_t){0, 0, 0, 0}; }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
    const float ka = 0.999f, kb = 1e-4f;
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int slot = 0; slot < 8; ++slot) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int ai = DEP ? 0 : ((slot * NM + m) & 3);   // DEP: every MFMA accumulates onto the same registers
                if (SMALL) sacc[ai] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, sacc[ai], 0, 0, 0);
                else acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[ai], 0, 0, 0);
            }
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(sacc[0]), "+v"(sacc[1]), "+v"(sacc[2]), "+v"(sacc[3]),
                         "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
#pragma unroll
            for (int v = 0; v < NV; ++v) x[v & 7] = __builtin_fmaf(x[v & 7], ka, kb);
#pragma unroll
            for (int v = 0; v < NT; ++v) x[v & 7] = __builtin_amdgcn_exp2f(x[v & 7]);
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(sacc[0]), "+v"(sacc[1]), "+v"(sacc[2]), "+v"(sacc[3]),
                         "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
        }
    }
    const long long t1 = wall_clock64();
    float t = 0;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 16; ++j) t += acc[i][j]; for (int j = 0; j < 4; ++j) t += sacc[i][j]; }
    for (int i = 0; i < 8; ++i) t += x[i];
    if (t == 123.456f) out[0] = t;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NM, int NV, int NT, bool DEP, bool SMALL>
static void run(const char* name, const int* src, float* out, long long* cyc, int threads) {
    const int iters = 2000, blocks = 256 * 2;        // 2 resident blocks per CU at 256 threads, 1 at 512: launch enough for every CU either way
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_k<NM, NV, NT, DEP, SMALL>), dim3(256), dim3(threads), 0, 0, src, out, 50, cyc); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe_k<NM, NV, NT, DEP, SMALL>), dim3(256), dim3(threads), 0, 0, src, out, iters, cyc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    (void)blocks;
    const double slots = (double)iters * 8;
    const double ns_slot = ms * 1e6 / slots;                         // one slot of every resident wave (they run concurrently)
    const int wps = threads / 256;                                   // waves per SIMD (one block per CU)
    const double mfma_cyc = (SMALL ? 16.0 : 32.0) * NM, valu_cyc = 4.0 * NV + 16.0 * NT;
    printf("PROBE %-44s %d wave(s)/SIMD: %7.2f ns per slot  = %6.1f cycles @2.4 GHz per slot;  per SIMD: MFMA %5.0f + VALU %5.0f cycles of work x %d wave(s)  -> overlap bound %5.0f, exclusive bound %5.0f\n",
           name, wps, ns_slot, ns_slot * 2.4, mfma_cyc, valu_cyc, wps, (mfma_cyc > valu_cyc ? mfma_cyc : valu_cyc) * wps, (mfma_cyc + valu_cyc) * wps);
}

int main() {
    int* src; float* out; long long* cyc;
    hipMalloc(&src, 4096 * sizeof(int)); hipMalloc(&out, 64); hipMalloc(&cyc, 64);
    int h[4096];
    unsigned s = 12345u;
    for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)s; }
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    for (int threads = 256; threads <= 512; threads += 256) {
        run<1, 0, 0, false, false>("1 MFMA 32x32x16 (independent acc)", src, out, cyc, threads);
        run<1, 0, 0, true, false>("1 MFMA 32x32x16 (chained acc)", src, out, cyc, threads);
        run<0, 8, 0, false, false>("8 FMA", src, out, cyc, threads);
        run<0, 0, 2, false, false>("2 v_exp", src, out, cyc, threads);
        run<1, 2, 0, true, false>("1 MFMA + 2 FMA", src, out, cyc, threads);
        run<1, 4, 0, true, false>("1 MFMA + 4 FMA", src, out, cyc, threads);
        run<1, 8, 0, true, false>("1 MFMA + 8 FMA", src, out, cyc, threads);
        run<1, 12, 0, true, false>("1 MFMA + 12 FMA", src, out, cyc, threads);
        run<1, 16, 0, true, false>("1 MFMA + 16 FMA", src, out, cyc, threads);
        run<1, 4, 2, true, false>("1 MFMA + 4 FMA + 2 v_exp", src, out, cyc, threads);
        run<3, 8, 0, true, false>("3 MFMA (chained) + 8 FMA", src, out, cyc, threads);
        run<3, 24, 0, true, false>("3 MFMA (chained) + 24 FMA", src, out, cyc, threads);
        run<3, 24, 0, false, false>("3 MFMA (independent) + 24 FMA", src, out, cyc, threads);
        run<2, 0, 0, true, true>("2 MFMA 16x16x32 (chained)", src, out, cyc, threads);
        run<2, 8, 0, true, true>("2 MFMA 16x16x32 (chained) + 8 FMA", src, out, cyc, threads);
        run<6, 8, 0, false, true>("6 MFMA 16x16x32 (independent) + 8 FMA", src, out, cyc, threads);
    }
    return 0;
}
