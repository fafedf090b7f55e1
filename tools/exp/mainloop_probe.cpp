// Times the fused vocabulary projection (M x 10000 x 512) and the plain logits contraction; links one build of gemm.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/capdec.h"
#include "../../include/capdec_tuning.h"   // link against libcapdec_tune.so
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int V = 10000, K = 512;
    const int Ms[2] = {1024, 10240};
    float *x, *W, *b, *pv, *C; int32_t* pi;
    const int nt = capdec_vocab_tiles(V);
    CK(hipMalloc(&x, (size_t)10240 * K * 4)); CK(hipMalloc(&W, (size_t)V * K * 4)); CK(hipMalloc(&b, V * 4));
    CK(hipMalloc(&pv, (size_t)10240 * nt * 4)); CK(hipMalloc(&pi, (size_t)10240 * nt * 4));
    CK(hipMalloc(&C, (size_t)10240 * V * 4));
    std::vector<float> h((size_t)10240 * K);
    srand(1);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)V * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, h.data(), V * 4, hipMemcpyHostToDevice));
    {   // both staging forms must agree bit for bit (same order of sums)
        const int M = 1000;
        std::vector<float> v1((size_t)M * nt), v0((size_t)M * nt), c1((size_t)M * V), c0((size_t)M * V);
        std::vector<int32_t> i1((size_t)M * nt), i0((size_t)M * nt);
        capdec_gemm_group g{};
        g.P = x; g.Q = W; g.C = C; g.bias_j = b; g.ldp = K; g.ldq = K; g.ldci = V; g.ldcj = 1; g.I = M; g.J = V;
        for (int st = 1; st >= 0; --st) {
            capdec_gemm_force_staging(st);
            if (capdec_vocab_argmax_partial(x, K, W, b, M, V, K, pv, pi, nullptr) != 0 || capdec_gemm(&g, 1, K, 0, nullptr) != 0) { printf("launch failed\n"); return 1; }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(st ? v1.data() : v0.data(), pv, v1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(st ? (void*)i1.data() : (void*)i0.data(), pi, i1.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(st ? c1.data() : c0.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
        }
        size_t bad = 0, badc = 0;
        for (size_t n = 0; n < v1.size(); ++n) bad += (v1[n] != v0[n]) || (i1[n] != i0[n]);
        for (size_t n = 0; n < c1.size(); ++n) badc += (c1[n] != c0[n]);
        printf("staging forms differ in %zu of %zu partial results, %zu of %zu logits\n", bad, v1.size(), badc, c1.size());
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int st = 1; st >= 0; --st)
    for (int mi = 0; mi < 2; ++mi) {
        capdec_gemm_force_staging(st);
        const int M = Ms[mi];
        for (int mode = 0; mode < 5; ++mode) {
            if (mode >= 1 && M == 1024) continue;
            const int cfgs[5] = {0, 116, 11, 216, 22};
            capdec_gemm_force_config(cfgs[mode]);
            auto run = [&]() {
                if (mode == 0) return capdec_vocab_argmax_partial(x, K, W, b, M, V, K, pv, pi, nullptr);
                capdec_gemm_group g{};
                g.P = x; g.Q = W; g.C = C; g.bias_j = b; g.ldp = K; g.ldq = K; g.ldci = V; g.ldcj = 1; g.I = M; g.J = V;
                return capdec_gemm(&g, 1, K, 0, nullptr);
            };
            for (int i = 0; i < 3; ++i) if (run() != 0) { printf("launch failed\n"); return 1; }
            CK(hipDeviceSynchronize());
            const int reps = M == 1024 ? 40 : 8;
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < reps; ++i) run();
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            printf("staging %d %s M=%5d: %8.1f us  %6.1f TFLOP/s\n", st, mode == 0 ? "vocab-argmax" : (mode == 1 ? "logits 64x64 BK16" : mode == 2 ? "logits 64x64 BK32" : mode == 3 ? "logits 128x128 BK16" : "logits 128x128 BK32"), M, us, 2.0 * M * V * K / us / 1e6);
        }
    }
    return 0;
}
