// Host-side argument validation of the C ABI under AddressSanitizer (host code only; no kernel is ever launched, so
// this runs on a machine without a GPU).  Every call below must be rejected with a negative CAPDEC_E* code before the
// library touches a pointer; ASan aborts the process if the host code of an entry point reads or writes out of bounds
// on the way (for instance in the model / group structs it copies).  Built and run by tools/host_asan.sh.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/capdec.h"

static int failures = 0;
#define EXPECT(expr, want)                                                                  \
    do {                                                                                    \
        const long long got__ = (long long)(expr);                                          \
        if (got__ != (long long)(want)) {                                                   \
            std::printf("FAIL %s:%d  %s = %lld, expected %lld\n", __FILE__, __LINE__, #expr, got__, (long long)(want)); \
            ++failures;                                                                     \
        }                                                                                   \
    } while (0)

int main() {
    // a block of host memory stands in for "device" pointers: validation only inspects addresses
    std::vector<float> mem(4096 + 8);
    float* a = mem.data();
    while ((reinterpret_cast<uintptr_t>(a) & 15u) != 0) ++a;      // 16-byte aligned
    float* una = a + 1;                                            // deliberately misaligned
    int32_t* ai = reinterpret_cast<int32_t*>(a);
    int64_t* al = reinterpret_cast<int64_t*>(a);

    EXPECT(capdec_abi_version(), CAPDEC_ABI_VERSION);
    EXPECT(capdec_strerror(CAPDEC_EINVAL) != nullptr, 1);
    EXPECT(capdec_strerror(-12345) != nullptr, 1);

    // contraction: null group array, zero groups, too many groups, K < 1, null operands, short leading dimensions
    capdec_gemm_group g{};
    EXPECT(capdec_gemm(nullptr, 1, 8, 0, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_gemm(&g, 0, 8, 0, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_gemm(&g, 5, 8, 0, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_gemm(&g, 1, 0, 0, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_gemm(&g, 1, 8, 0, nullptr), CAPDEC_EINVAL);                    // P, Q, C null
    g.P = a; g.Q = a; g.C = a; g.I = 4; g.J = 4; g.ldp = 4; g.ldq = 8; g.ldci = 4; g.ldcj = 1;
    EXPECT(capdec_gemm(&g, 1, 8, 0, nullptr), CAPDEC_EINVAL);                    // ldp < K
    g.ldp = 8;
    g.scale_j = a;                                                               // scale without shift
    EXPECT(capdec_gemm(&g, 1, 8, 0, nullptr), CAPDEC_EINVAL);
    g.scale_j = nullptr;
    EXPECT(capdec_gemm(&g, 1, 8, CAPDEC_GEMM_TP, nullptr), CAPDEC_EINVAL);       // TP without TQ
    EXPECT(capdec_gemm(&g, 1, 8, CAPDEC_GEMM_ATOMIC | CAPDEC_GEMM_RELU, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_gemm(&g, 1, 8, CAPDEC_GEMM_SPLITK, nullptr), CAPDEC_EINVAL);   // split-K without a workspace
    capdec_gemm_extra ex{};
    ex.splitk_ws = a; ex.splitk_ws_elems = 3;                                    // smaller than one 4 x 4 slab
    EXPECT(capdec_gemm_ex(&g, 1, 8, CAPDEC_GEMM_SPLITK, &ex, nullptr), CAPDEC_ENOMEM);
    ex = capdec_gemm_extra{};
    ex.lse = a;                                                                  // lse without targets / scale
    EXPECT(capdec_gemm_ex(&g, 1, 8, 0, &ex, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_gemm_splitk_workspace_elems(0, 4, 8), 0);
    EXPECT(capdec_gemm_splitk_workspace_elems(1024, 512, 86436) >= (size_t)1024 * 512, 1);

    // vocabulary projections
    EXPECT(capdec_vocab_tiles(10000), 157);
    EXPECT(capdec_vocab_partial_elems(-1, 10), 0);
    EXPECT(capdec_vocab_argmax_partial(nullptr, 8, a, a, 4, 16, 8, a, ai, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_vocab_argmax_partial(a, 4, a, a, 4, 16, 8, a, ai, nullptr), CAPDEC_EINVAL);     // ldx < K
    EXPECT(capdec_argmax_finalize(a, ai, 4, 0, ai, nullptr, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_vocab_argmax_static(a, 8, a, a, 1 << 20, 10000, 8, ai, nullptr, nullptr), CAPDEC_E2BIG);
    EXPECT(capdec_vocab_argmax_static(a, 8, a, a, 4, 16, 8, nullptr, nullptr, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_nll_finalize(a, a, 4, 1, a, 8, a, a, 8, nullptr, 0, a, a, a, nullptr), CAPDEC_EINVAL);

    // row kernels: shapes, alignment
    EXPECT(capdec_embed_pe(al, 0, 4, a, a, 0, 2, 4, 6, 10, a, nullptr), CAPDEC_EINVAL);           // E % 4 != 0
    EXPECT(capdec_embed_pe(al, 0, 2, a, a, 0, 2, 4, 8, 10, a, nullptr), CAPDEC_EINVAL);           // ldid < T
    EXPECT(capdec_embed_pe(al, 0, 4, una, a, 0, 2, 4, 8, 10, a, nullptr), CAPDEC_EALIGN);
    EXPECT(capdec_embed_train_fwd(al, 3, nullptr, a, a, 8, 4, 8, 10, 0.1f, 1, nullptr, a, nullptr, nullptr), CAPDEC_EINVAL);   // ldid < T
    EXPECT(capdec_embed_train_fwd(al, 4, nullptr, a, a, 8, 4, 8, 10, 1.0f, 1, nullptr, a, nullptr, nullptr), CAPDEC_EINVAL);   // p must be < 1
    EXPECT(capdec_embed_train_fwd(al, 4, nullptr, una, a, 8, 4, 8, 10, 0.f, 1, nullptr, a, nullptr, nullptr), CAPDEC_EALIGN);
    EXPECT(capdec_embed_train_fwd(al, 4, nullptr, a, a, 0, 4, 8, 10, 0.f, 1, nullptr, a, nullptr, nullptr), 0);              // no slots: nothing launched
    EXPECT(capdec_embed_train_bwd(nullptr, a, 4, 8, 10, 0.f, 1, nullptr, a, nullptr, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_embed_train_bwd(al, a, 0, 8, 10, 0.5f, 1, nullptr, a, nullptr, nullptr), 0);
    EXPECT(capdec_compact_prefix_strided(al, al, 3, 2, 4, 0, al, al, ai, ai, ai, al, ai, nullptr), CAPDEC_EINVAL);           // ld < T
    EXPECT(capdec_add_layernorm(nullptr, nullptr, 1, a, a, 1e-5f, 4, 64, a, nullptr, nullptr, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_attention(nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 1, 1, 1, 1, 64, 0, 0, nullptr, 0, nullptr,
                            nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_dropout(a, 16, 1.0f, 1, nullptr, a, nullptr), CAPDEC_EINVAL);                   // p must be < 1
    EXPECT(capdec_layernorm_bwd_workspace_floats(0, 64), 0);
    EXPECT(capdec_layernorm_bwd_workspace_floats(1024, 512) >= (size_t)3 * 512, 1);
    EXPECT(capdec_layernorm_bwd_staged(a, a, a, a, a, 4, 64, 0.f, 1, nullptr, a, a, a, a, a, nullptr, a, 1, nullptr),
           CAPDEC_ENOMEM);                                                                        // workspace too small
    EXPECT(capdec_layernorm_bwd_staged(nullptr, a, a, a, a, 4, 64, 0.f, 1, nullptr, a, a, a, a, a, nullptr, nullptr, 0,
                                       nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_dropout(nullptr, 16, 0.1f, 1, nullptr, a, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_add_layernorm_train_rows(a, 0, nullptr, a, a, a, 1e-5f, 4, 64, 0.1f, 1, nullptr, a, a, a, a, nullptr,
                                           nullptr), CAPDEC_EINVAL);                              // x_div < 1
    EXPECT(capdec_add_layernorm_train_rows(una, 1, nullptr, a, a, a, 1e-5f, 4, 64, 0.1f, 1, nullptr, a, a, a, a, nullptr,
                                           nullptr), CAPDEC_EALIGN);
    EXPECT(capdec_layernorm_bwd_rows(a, a, a, a, a, 4, 62, 0.f, 0, nullptr, a, a, a, a, nullptr, nullptr, nullptr),
           CAPDEC_EINVAL);                                                                        // E % 4 != 0
    EXPECT(capdec_colsum(nullptr, 8, 4, 8, a, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_bn_train_backward(nullptr, a, a, a, a, 4, 8, a, a, a, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_adam_step(a, a, a, a, 16, 1e-3f, 0.9f, 0.999f, 1e-8f, 0, 1.f, nullptr), CAPDEC_EINVAL);   // step >= 1
    EXPECT(capdec_adam_step_dev(a, a, a, a, 16, 0.9f, 0.999f, 1e-8f, nullptr, 1.f, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_train_state_advance(nullptr, 0.9f, 0.999f, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_train_state_advance(reinterpret_cast<capdec_train_state*>(a), 1.5f, 0.999f, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_lstm_step(nullptr, 8, nullptr, 0, nullptr, a, a, a, a, 4, 8, 8, a, 8, a, nullptr, nullptr), CAPDEC_EINVAL);
    // per-sample contractions and row softmax
    capdec_bgemm_desc bd{};
    EXPECT(capdec_bgemm(nullptr, CAPDEC_BGEMM_NT, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_bgemm(&bd, CAPDEC_BGEMM_NT, nullptr), CAPDEC_EINVAL);                           // null operands
    bd.P = a; bd.Q = a; bd.C = a; bd.ldp = 8; bd.ldq = 8; bd.ldc = 8; bd.dense_stride = 32;
    bd.mult = 1; bd.seg = 4; bd.B = 1; bd.I = 4; bd.J = 4; bd.K = 8; bd.scale = 1.f;
    EXPECT(capdec_bgemm(&bd, 3, nullptr), CAPDEC_EINVAL);                                         // unknown form
    bd.K = 6;
    EXPECT(capdec_bgemm(&bd, CAPDEC_BGEMM_NT, nullptr), CAPDEC_EINVAL);                           // K % 4 != 0
    bd.K = 8; bd.ldp = 6;
    EXPECT(capdec_bgemm(&bd, CAPDEC_BGEMM_NT, nullptr), CAPDEC_EALIGN);                           // ldp % 4 != 0
    bd.ldp = 8; bd.P = una;
    EXPECT(capdec_bgemm(&bd, CAPDEC_BGEMM_NT, nullptr), CAPDEC_EALIGN);
    bd.P = a; bd.P2 = a;
    EXPECT(capdec_bgemm(&bd, CAPDEC_BGEMM_TN, nullptr), CAPDEC_EINVAL);                           // P2 without Q2
    bd.P2 = nullptr; bd.J = 6;
    EXPECT(capdec_bgemm(&bd, CAPDEC_BGEMM_NN, nullptr), CAPDEC_EINVAL);                           // J % 4 != 0
    bd.J = 4; bd.B = 0;
    EXPECT(capdec_bgemm(&bd, CAPDEC_BGEMM_NN, nullptr), 0);                                       // empty batch: nothing to do
    EXPECT(capdec_softmax_rows(nullptr, 8, 4, 5, 8, nullptr, 0, 1, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_softmax_rows(a, 8, 4, 9, 8, nullptr, 0, 1, nullptr), CAPDEC_EINVAL);            // pad_to < cols
    EXPECT(capdec_softmax_rows(a, 4, 4, 5, 8, nullptr, 0, 1, nullptr), CAPDEC_EINVAL);            // ld < pad_to
    EXPECT(capdec_softmax_rows(a, 8, 0, 5, 8, nullptr, 0, 1, nullptr), 0);
    EXPECT(capdec_softmax_rows_bwd(a, nullptr, 8, 4, 5, 8, 1.f, nullptr, 0, 1, nullptr), CAPDEC_EINVAL);
    // cross-attention over two memory rows, training form
    EXPECT(capdec_pair_diff(nullptr, a, a, 1, 8, 2, a, a, a, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_pair_diff(a, a, a, 1, 12, 2, a, a, a, nullptr), CAPDEC_EINVAL);                  // E % (4 H) != 0
    EXPECT(capdec_pair_diff(una, a, a, 1, 8, 2, a, a, a, nullptr), CAPDEC_EALIGN);
    EXPECT(capdec_pair_diff(a, a, a, 0, 8, 2, a, a, a, nullptr), 0);
    EXPECT(capdec_pair_diff_bwd(a, a, a, a, a, a, 1, 8, 2, a, nullptr, a, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_pair_diff_bwd(a, a, a, a, a, a, 1, 8, 2, a, una, a, nullptr), CAPDEC_EALIGN);
    EXPECT(capdec_pair_gate_fwd(a, a, a, a, a, nullptr, 0, 4, 8, 2, 1.f, a, a, nullptr, nullptr), CAPDEC_EINVAL);   // T < 1
    EXPECT(capdec_pair_gate_fwd(a, a, a, a, a, nullptr, 4, 4, 4096, 2, 1.f, a, a, nullptr, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_pair_gate_fwd(a, una, a, a, a, nullptr, 4, 4, 8, 2, 1.f, a, a, nullptr, nullptr), CAPDEC_EALIGN);
    EXPECT(capdec_pair_gate_fwd(a, a, a, a, a, nullptr, 4, 0, 8, 2, 1.f, a, a, nullptr, nullptr), 0);
    EXPECT(capdec_pair_gate_bwd(a, a, a, a, a, nullptr, 1, 4, 2048, 2, 1.f, a, a, a, a, a, nullptr), CAPDEC_EINVAL);
    EXPECT(capdec_pair_gate_bwd(a, a, a, a, a, nullptr, 1, 4, 8, 2, 1.f, una, a, a, a, a, nullptr), CAPDEC_EALIGN);
    EXPECT(capdec_pair_gate_bwd(a, a, a, a, a, nullptr, 0, 4, 8, 2, 1.f, a, a, a, a, a, nullptr), 0);
    EXPECT(capdec_topk_rows(a, 4, 65, 3, a, ai, nullptr), CAPDEC_EINVAL);                          // n > 64
    EXPECT(capdec_topk_rows(a, 4, 9, 0, a, ai, nullptr), CAPDEC_EINVAL);

    // whole-loop entry points: inconsistent model structs (copied and inspected on the host)
    capdec_tf_model m{};
    m.E = 10; m.H = 3; m.L = 1; m.V = 5; m.max_pos = 4;
    EXPECT(capdec_tf_decode_workspace_bytes(&m, 4, 1, 4, 0), 0);                                   // E % H != 0
    EXPECT(capdec_tf_decode_workspace_bytes(nullptr, 4, 1, 4, 0), 0);
    m.E = 64; m.H = 4; m.L = 2; m.V = 50; m.max_pos = 50;                                          // pointers still null
    EXPECT(capdec_tf_greedy_decode(&m, a, 4, 1, 8, 1, 2, 0, 0, a, 1 << 20, al, ai, ai, nullptr, nullptr), CAPDEC_EINVAL);
    std::vector<capdec_tf_layer> layers(2);                                                        // all-null layer structs
    m.layers = layers.data(); m.emb = a; m.pe = a; m.fc2_w = a; m.fc2_b = a;
    EXPECT(capdec_tf_greedy_decode(&m, a, 4, 1, 51, 1, 2, 0, 0, a, 1 << 20, al, ai, ai, nullptr, nullptr), CAPDEC_EINVAL);   // max_len > max_pos
    EXPECT(capdec_tf_greedy_decode(&m, a, 4, 1, 8, 1, 2, 0, 0, a, 16, al, ai, ai, nullptr, nullptr), CAPDEC_ENOMEM);        // workspace too small
    EXPECT(capdec_tf_greedy_decode(&m, una, 4, 1, 8, 1, 2, 0, 0, a, (size_t)1 << 30, al, ai, ai, nullptr, nullptr), CAPDEC_EALIGN);
    m.emb = una;                                                                                   // step 0's rows are read 16 bytes at a time
    EXPECT(capdec_tf_greedy_decode(&m, a, 4, 1, 8, 1, 2, 0, 0, a, (size_t)1 << 30, al, ai, ai, nullptr, nullptr), CAPDEC_EALIGN);
    m.emb = a;
    EXPECT(capdec_tf_beam_search(&m, a, 4, 9, 1, 8, 1, 2, 0, 8, a, 1 << 20, al, ai, a, ai, nullptr), CAPDEC_EINVAL);         // W > 8
    EXPECT(capdec_tf_beam_search(&m, a, 4, 3, 1, 8, 1, 2, 5, 3, a, 1 << 20, al, ai, a, ai, nullptr), CAPDEC_EINVAL);         // t_begin >= t_end
    EXPECT(capdec_tf_forward_logits(&m, a, nullptr, 4, 1, 8, 0, a, 1 << 20, a, nullptr), CAPDEC_EINVAL);
    capdec_rnn_model r{};
    EXPECT(capdec_rnn_decode_workspace_bytes(&r, 4, 8), 0);
    EXPECT(capdec_rnn_greedy_decode(&r, a, 4, 8, 1, 2, a, 1 << 20, al, ai, nullptr), CAPDEC_EINVAL);
    capdec_satt_model sm{};
    EXPECT(capdec_satt_decode_workspace_bytes(&sm, 4, 8), 0);

    if (failures) {
        std::printf("%d validation checks failed\n", failures);
        return 1;
    }
    std::printf("host validation under AddressSanitizer: all checks passed\n");
    return 0;
}
