// Times the training-step contraction shapes (forward, dgrad, wgrad) of one build of gemm.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/capdec.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
int main() {
    const int M = 10240, E = 512, V = 10000;
    float *x, *dy, *W, *Wv, *dl, *C, *Cv;
    CK(hipMalloc(&x, (size_t)M * E * 4)); CK(hipMalloc(&dy, (size_t)M * E * 4)); CK(hipMalloc(&W, (size_t)E * E * 4));
    CK(hipMalloc(&Wv, (size_t)V * E * 4)); CK(hipMalloc(&dl, (size_t)M * V * 4)); CK(hipMalloc(&C, (size_t)M * E * 4));
    CK(hipMalloc(&Cv, (size_t)V * E * 4));
    std::vector<float> h((size_t)M * E);
    srand(1);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)E * E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Wv, h.data(), (size_t)V * E * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 20; ++i) CK(hipMemcpy(dl + (size_t)i * (M / 20) * V, h.data(), (size_t)(M / 20) * V * 4 < h.size() * 4 ? (size_t)(M / 20) * V * 4 : h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int kind; } cases[] = {{"fwd   10240x512x512", 0}, {"dgrad 10240x512x512", 1}, {"wgrad 512x512x10240 (split-k)", 2},
                                                            {"dgrad vocab 10240x512x10000", 3}, {"wgrad vocab 10000x512x10240 (split-k)", 4}};
    for (auto& c : cases) {
        auto run = [&]() {
            capdec_gemm_group g{};
            int K = E, flags = 0;
            switch (c.kind) {
            case 0: g.P = x; g.Q = W; g.C = C; g.ldp = E; g.ldq = E; g.ldci = E; g.ldcj = 1; g.I = M; g.J = E; break;
            case 1: g.P = dy; g.Q = W; g.C = C; g.ldp = E; g.ldq = E; g.ldci = E; g.ldcj = 1; g.I = M; g.J = E; flags = CAPDEC_GEMM_TQ; break;
            case 2: g.P = dy; g.Q = x; g.C = W; g.ldp = E; g.ldq = E; g.ldci = E; g.ldcj = 1; g.I = E; g.J = E; K = M; flags = CAPDEC_GEMM_TP | CAPDEC_GEMM_TQ | CAPDEC_GEMM_ATOMIC; break;
            case 3: g.P = dl; g.Q = Wv; g.C = C; g.ldp = V; g.ldq = E; g.ldci = E; g.ldcj = 1; g.I = M; g.J = E; K = V; flags = CAPDEC_GEMM_TQ; break;
            default: g.P = dl; g.Q = x; g.C = Cv; g.ldp = V; g.ldq = E; g.ldci = E; g.ldcj = 1; g.I = V; g.J = E; K = M; flags = CAPDEC_GEMM_TP | CAPDEC_GEMM_TQ | CAPDEC_GEMM_ATOMIC; break;
            }
            return capdec_gemm(&g, 1, K, flags, nullptr);
        };
        for (int i = 0; i < 3; ++i) if (run() != 0) { printf("launch failed (%s)\n", c.name); return 1; }
        CK(hipDeviceSynchronize());
        const int reps = c.kind >= 3 ? 8 : 40;
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        const double fl = c.kind >= 3 ? 2.0 * M * E * V : 2.0 * M * E * E;
        printf("%-40s %8.1f us  %6.1f TFLOP/s\n", c.name, us, fl / us / 1e6);
    }
    return 0;
}
