/* A plain-C consumer of the drop-in boundary (include/dust3r_hip.h): what a maintainer's FFI stub sees.
 * Built and run by tests/test_host_cpu.py with gcc (no hipcc, no torch, no GPU): the header must be valid C99, every
 * entry point must link, and the argument checks must answer with D3R_ERR_* codes instead of touching a device. */
#include <stdio.h>
#include <string.h>

#include "dust3r_hip.h"

int main(void) {
    int fails = 0;
    const char* v = d3r_version();
    if (!v || !strlen(v)) { printf("FAIL version\n"); ++fails; }
    /* null handles / pointers are rejected before any HIP call */
    if (d3r_model_forward(NULL, NULL, NULL, 1, 32, 32, NULL, NULL, NULL, NULL, NULL) == D3R_OK) { printf("FAIL model_forward(NULL)\n"); ++fails; }
    if (d3r_model_set_option(NULL, D3R_MODEL_OPT_PROFILE, 1) != D3R_ERR_INVALID) { printf("FAIL set_option(NULL)\n"); ++fails; }
    if (d3r_model_profile_read(NULL, 0, NULL, NULL, NULL) == D3R_OK) { printf("FAIL profile_read(NULL)\n"); ++fails; }
    if (d3r_model_profile_launch(NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL) == D3R_OK) { printf("FAIL profile_launch(NULL)\n"); ++fails; }
    if (d3r_aligner_run(NULL, 1, 0, 1, 0.01f, 1e-6f, D3R_SCHEDULE_COSINE, NULL, NULL) == D3R_OK) { printf("FAIL aligner_run(NULL)\n"); ++fails; }
    if (d3r_aligner_set_option(NULL, D3R_ALIGNER_OPT_DPP_REDUCE, 1) != D3R_ERR_INVALID) { printf("FAIL aligner_set_option(NULL)\n"); ++fails; }
    if (d3r_aligner_set_image_range(NULL, 0, 1) != D3R_ERR_INVALID) { printf("FAIL aligner_set_image_range(NULL)\n"); ++fails; }
    if (d3r_aligner_step_begin(NULL, 0, 0, 1, 0.01f, 1e-6f, D3R_SCHEDULE_COSINE, NULL) != D3R_ERR_INVALID) { printf("FAIL aligner_step_begin(NULL)\n"); ++fails; }
    if (d3r_aligner_step_end(NULL, 0, 0, 1, 0.01f, 1e-6f, D3R_SCHEDULE_COSINE, NULL) != D3R_ERR_INVALID) { printf("FAIL aligner_step_end(NULL)\n"); ++fails; }
    { void* sums = NULL; long long count = 0;
      if (d3r_aligner_reduced_sums(NULL, &sums, &count) != D3R_ERR_INVALID) { printf("FAIL aligner_reduced_sums(NULL)\n"); ++fails; } }
    if (d3r_aligner_read_losses(NULL, 1, NULL, NULL) != D3R_ERR_INVALID) { printf("FAIL aligner_read_losses(NULL)\n"); ++fails; }
    if (d3r_layernorm(NULL, NULL, NULL, NULL, 1, 64, 1e-6f, D3R_DTYPE_F32, NULL) != D3R_ERR_INVALID) { printf("FAIL layernorm(NULL)\n"); ++fails; }
    /* the host-only gradient self test runs without a device: one edge between two 2x2 images */
    {
        const int ei[1] = {0}, ej[1] = {1};
        float pred_i[12], pred_j[12], w_i[4], w_j[4], pw[8] = {0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f};
        float imp[14] = {0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.1f, 0.f, 0.f}, dep[8], foc[2] = {20.f, 20.f};
        double loss = -1.0, g_pw[8], g_imp[14], g_dep[8], g_foc[2];
        int k;
        for (k = 0; k < 12; ++k) { pred_i[k] = 0.1f * (float)(k + 1); pred_j[k] = 0.2f * (float)(k + 1); }
        for (k = 0; k < 4; ++k) { w_i[k] = 1.f; w_j[k] = 0.5f; }
        for (k = 0; k < 8; ++k) dep[k] = 0.f;
        if (d3r_selftest_aligner_math_host(2, 1, ei, ej, 2, 2, pred_i, pred_j, w_i, w_j, pw, imp, dep, foc, 0.5f, 20.f, &loss, g_pw, g_imp, g_dep,
                                           g_foc) != D3R_OK || !(loss > 0.0)) {
            printf("FAIL selftest loss=%g\n", loss);
            ++fails;
        }
    }
    if (fails) printf("c_abi probe: %d failures\n", fails);
    else printf("c_abi probe ok (%s)\n", v);
    return fails ? 1 : 0;
}
