// Latency anatomy of the decode-step projection (1024 x 512 x K): links one build of gemm.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/capdec.h"
#include "../../include/capdec_tuning.h"   // link against libcapdec_tune.so
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const bool one = argc > 1;
    const int M = 1024, N = 512, KMAX = 4096;
    float *x, *W, *b, *R, *C, *C2;
    CK(hipMalloc(&x, (size_t)M * KMAX * 4)); CK(hipMalloc(&W, (size_t)N * KMAX * 4)); CK(hipMalloc(&b, N * 4));
    CK(hipMalloc(&R, (size_t)M * N * 4)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&C2, (size_t)M * N * 4));
    CK(hipMemset(x, 0, (size_t)M * KMAX * 4)); CK(hipMemset(W, 0, (size_t)N * KMAX * 4)); CK(hipMemset(b, 0, N * 4)); CK(hipMemset(R, 0, (size_t)M * N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int Ks[5] = {256, 512, 1024, 2048, 4096};
    for (int cfg : {32, 34})
    for (int ki = 0; ki < 5; ++ki)
        for (int ep = 0; ep < 2; ++ep) {
            const int K = Ks[ki];
            if (one && !(cfg == 32 && K == 1024 && ep == 0)) continue;
            capdec_gemm_force_config(cfg);
            // two alternating launches (C -> C2 chains through R) so that consecutive kernels depend on each other
            auto run = [&](int n) {
                capdec_gemm_group g{};
                g.P = x; g.Q = W; g.C = (n & 1) ? C2 : C; g.ldp = K; g.ldq = K; g.ldci = N; g.ldcj = 1; g.I = M; g.J = N;
                if (ep) { g.bias_j = b; g.R = (n & 1) ? C : C2; g.ldri = N; g.ldrj = 1; }
                return capdec_gemm(&g, 1, K, 0, nullptr);
            };
            for (int i = 0; i < 10; ++i) if (run(i) != 0) { printf("launch failed\n"); return 1; }
            CK(hipDeviceSynchronize());
            const int reps = one ? 5 : 200;
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < reps; ++i) run(i);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("cfg %3d K=%4d %s: %6.2f us per launch\n", cfg, K, ep ? "bias+residual" : "plain        ", ms * 1e3 / reps);
        }
    return 0;
}
