// On-GPU probe (not part of the library): does the row pitch of a GEMM operand matter for the rate at which tiles can be pulled out of L2 / MALL?
// Every block plays a 64 x 64 tile of an M x N x K split-fp16 GEMM: per K step it reads 128 bytes from each of its 64 activation rows and 64 weight rows
// (row pitch = `pitch` bytes), 16 bytes per lane, three steps in flight -- the access stream of gemm_kernel<Cfg64s3> without LDS or MFMA.
// With pitch = K * 4 bytes (a power of two for K = 1024 / 4096) the 128 row chunks of a step differ only in address bits >= 12/14.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2probe tools/l2_stride_probe.hip && /tmp/l2probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ __launch_bounds__(256) void probe(const char* a, const char* w, size_t pitch, int kbytes, int tiles_m, unsigned* sink) {
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int t = threadIdx.x, row = t >> 3, chunk = t & 7;
    const char* pa[2];
    const char* pw[2];
    for (int q = 0; q < 2; ++q) {
        pa[q] = a + (size_t)(tm * 64 + q * 32 + row) * pitch + chunk * 16;
        pw[q] = w + (size_t)(tn * 64 + q * 32 + row) * pitch + chunk * 16;
    }
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 4
    for (int k = 0; k < kbytes; k += 128) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint4 x = *reinterpret_cast<const uint4*>(pa[q] + k);
            const uint4 y = *reinterpret_cast<const uint4*>(pw[q] + k);
            acc.x ^= x.x ^ y.x; acc.y ^= x.y ^ y.y; acc.z ^= x.z ^ y.z; acc.w ^= x.w ^ y.w;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
    struct Shape { int M, N, K; } shapes[] = {{1536, 1024, 4096}, {1536, 1024, 1024}, {768, 768, 768}, {768, 768, 3072}, {12288, 1024, 4096}};
    const int pads[] = {0, 128, 256, 512, 1024, 2048, 4096 + 128};
    unsigned* sink;
    hipMalloc(&sink, 4);
    for (auto s : shapes) {
        for (int pad : pads) {
            const size_t pitch = (size_t)s.K * 4 + pad;
            char *a, *w;
            hipMalloc(&a, pitch * s.M);
            hipMalloc(&w, pitch * s.N);
            hipMemset(a, 1, pitch * s.M);
            hipMemset(w, 2, pitch * s.N);
            const int tiles_m = s.M / 64, tiles = tiles_m * (s.N / 64);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe, dim3(tiles), dim3(256), 0, 0, a, w, pitch, s.K * 4, tiles_m, sink);
            hipEventRecord(e0);
            const int reps = 20;
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe, dim3(tiles), dim3(256), 0, 0, a, w, pitch, s.K * 4, tiles_m, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= reps;
            const double bytes = (double)tiles * 128 * s.K * 4;
            printf("M %5d N %5d K %5d  %4d tiles  pitch K*4+%-5d : %7.1f us  %6.2f TB/s tile ingest\n", s.M, s.N, s.K, tiles, pad, ms * 1e3, bytes / ms / 1e9);
            fflush(stdout);
            hipFree(a);
            hipFree(w);
        }
    }
    return 0;
}
