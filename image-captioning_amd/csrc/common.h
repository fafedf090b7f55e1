// Shared device/host helpers for libcapdec (gfx950 only: 64-lane wavefronts, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/capdec.h"

#define CAPDEC_WAVE 64

#define CAPDEC_REQUIRE(cond)            \
    do {                                \
        if (!(cond)) return CAPDEC_EINVAL; \
    } while (0)

#define CAPDEC_LAUNCH_CHECK()                       \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

#define CAPDEC_TRY(expr)          \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != 0) return rc__; \
    } while (0)

// Kernel-selection overrides exist only in the tuning build of the library (libcapdec_tune.so, compiled with
// -DCAPDEC_TUNING_HOOKS for the tests and benchmarks that compare kernel variants; see include/capdec_tuning.h).  In
// the product build they are compile-time zeros, so the library holds no mutable state that a call could depend on.
#ifdef CAPDEC_TUNING_HOOKS
#include "../../include/capdec_tuning.h"
#define CAPDEC_HOOK_VAR(name) int name = 0
#else
#define CAPDEC_HOOK_VAR(name) constexpr int name = 0
#endif

static inline bool capdec_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Per-tile partial results of the fused vocabulary projections: blocks of 64 rows, [row block][vocabulary tile][64],
// so that the 64 or 128 rows a tile finishes are stored as contiguous runs instead of one 4-byte store per row stride.
__host__ __device__ inline int64_t capdec_part_off(int row, int tile, int ntiles) {
    return ((int64_t)(row >> 6) * ntiles + tile) * 64 + (row & 63);
}
static inline size_t capdec_part_elems(int M, int ntiles) { return (size_t)((M + 63) / 64) * 64 * (size_t)ntiles; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- wavefront (64-lane) reductions -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// arg-max with torch.argmax tie rule (lowest index wins)
__device__ __forceinline__ void argmax_merge(float& v, int& i, float v2, int i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float v2 = __shfl_xor(v, o, 64);
        int i2 = __shfl_xor(i, o, 64);
        argmax_merge(v, i, v2, i2);
    }
}

// internal launchers shared between translation units (all return 0 / CAPDEC_E* / hipError_t)
int capdec_launch_gemm(const capdec_gemm_group* groups, int ngroups, int K, int flags, hipStream_t s,
                       const capdec_gemm_extra* extra = nullptr);
// capdec_vocab_argmax_partial over the first rows_dev[0] rows only (nullable device scalar; whole 64- or 128-row
// tiles: the partials of rows beyond it are left untouched)
int capdec_launch_vocab_argmax_rows(const float* x, int64_t ldx, const float* W, const float* bias, int M, int V, int K,
                                    float* part_val, int32_t* part_idx, const int32_t* rows_dev, hipStream_t s);
// capdec_lstm_step that does nothing when live[0] is zero (nullable device scalar)
int capdec_launch_lstm_step(const float* x, int64_t ldx, const float* h0, int64_t ldh0, const float* c0, const float* w_ih,
                            const float* w_hh, const float* b_ih, const float* b_hh, int B, int E, int Hd, float* h1,
                            int64_t ldh1, float* c1, float* gates_out, const int32_t* live, hipStream_t stream);
int capdec_launch_layernorm_rows(const float* x, const float* add, int add_div, const int32_t* add_map,
                                 const float* gamma, const float* beta, float eps, int Mcap, int E, float* y,
                                 const int32_t* rows_dev, hipStream_t s, const float* gamma2 = nullptr,
                                 const float* beta2 = nullptr, const float* add2 = nullptr, int add2_div = 1,
                                 const int32_t* add2_map = nullptr);
// (gamma2 != NULL: y = LN2(LN1(x + add) + add2[add2_map[row] / add2_div]) in one pass over the row)
// Two memory rows per sample: y = LN2(x1 + c1[b] + sum_h sigmoid((x1 . G[b,h] + sb[b,h]) * scale) * D[b,h]), x1 = LN1(x),
// b = row_map[row] / map_div; G, D [B][H][E], c1 [B][E], sb [B][H] (see pair_memory_constants in engine.hip)
int capdec_launch_layernorm_pair_rows(const float* x, const float* gamma1, const float* beta1, const float* gamma2,
                                      const float* beta2, float eps, int Mcap, int E, float* y, const int32_t* rows_dev,
                                      const int32_t* row_map, int map_div, const float* G, const float* D, const float* c1,
                                      const float* sb, int H, float scale, hipStream_t s);
// Decode-step attention of Rcap rows over the raw memory of their samples ([B*S][E], E in {64, 128, 256, 512}): qp, ctx
// [Rcap][H*E] (head h in columns [h*E, (h+1)*E)); see attention_memory_kernel in ops.hip
inline bool capdec_attention_memory_supported(int E) { return E == 64 || E == 128 || E == 256 || E == 512; }
int capdec_launch_attention_memory_rows(const float* qp, int64_t ldq, const float* mem, int S, int E, int H, float scale,
                                        float* ctx, int64_t ldc, int Rcap, const int32_t* row_map, int map_div,
                                        const int32_t* rows_dev, hipStream_t s);
// ---- all decoder layers of a cached decode step in one launch (decode_fused.hip) ---------------------------------------
// Rows are cut into blocks of CAPDEC_FUSED_ROWS, each walked through the layers by a team of CAPDEC_FUSED_TEAM workgroups;
// a team carries up to CAPDEC_FUSED_SIDE_ROWS earlier <PAD> positions of its samples along (two layers only).
#define CAPDEC_FUSED_TEAM 8
#define CAPDEC_FUSED_ROWS 16
#define CAPDEC_FUSED_SIDE_ROWS 16
#define CAPDEC_FUSED_MAX_LAYERS 2
struct capdec_fused_layer {
    const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *w1, *b1, *w2, *b2;
    const float *n1g, *n1b, *n2g, *n2b, *n3g, *n3b;
    const float* cc;          // [B][E] the cross-attention block of a single memory row, per sample
    const float *pg, *pd, *psb;   // two memory rows (nullable): G, D [B][H][E] and sb [B][H] of the logistic closed form; cc is then c1
    float *kc, *vc;           // [B][max_len][E] self-attention cache of this layer
};
struct capdec_fused_args {
    capdec_fused_layer ly[CAPDEC_FUSED_MAX_LAYERS];
    int L, B, max_len, Tk;    // Tk = keys of this step (step index + 1)
    int teams;                // ceil(B / CAPDEC_FUSED_ROWS)
    int cache_bytes;          // B * max_len * E * 4
    int cc_div;               // row b uses the cross-attention constants of sample b / cc_div (beam search: the width)
    float* x;                 // [B][E] row-major: the step's input rows; receives the last LayerNorm's rows
    float* q;                 // [teams][32][E] row-major queries
    float *f_ctx, *f_y, *f_h, *f_z;   // [teams][E/4][32][4] fragment-major exchange buffers
    float *s_ctx, *s_y, *s_h, *s_z;   // [teams][E/4][16][4] the same of the side rows
    const int32_t* tok; int64_t ldtok;
    const int32_t* nrows;     // device scalar: 0 once every caption has finished
    const int32_t* done;      // [B]
    const int32_t* plist;     // [teams][16] sample * max_len + position of a generated <PAD>, -1: free slot
    const float *emb, *pe;
    // step t >= 1 also picks the tokens of step t-1 (pval != NULL): per-tile arg-max winners of the previous vocabulary
    // projection in, the reference's bookkeeping out (what cached_advance_kernel does as a launch of its own otherwise)
    const float* pval; const int32_t* pidx; int ntiles, V, eos;
    int32_t *tok_w, *done_w, *out_len, *dirty, *fin, *nrows_w, *pcount_w, *plist_w;
    int64_t* out_ids;
    unsigned* ctr;            // [teams][32] arrival counters of this step (one per 128-byte line), zero before the launch
    unsigned* err;            // raised by a workgroup whose bounded wait ran out
    unsigned long long* trace;   // probe builds only (CAPDEC_FUSED_TRACE): [8][64] stage timestamps of team trace_team
    int trace_team;
    unsigned* hwid;              // probe builds only: [grid] hardware id register of each workgroup's first wavefront
};
bool capdec_fused_layers_supported(int E, int H, int L, int64_t B, int max_len);
int capdec_launch_fused_layers(const capdec_fused_args& a, int E, hipStream_t s);

int capdec_launch_attention_rows(const float* q, int64_t ldq, const float* k, int64_t ldk, int64_t sk, const float* v,
                                 int64_t ldv, int64_t sv, float* out, int64_t ldo, int Rcap, int H, int Tk, int dk,
                                 int causal, const int32_t* ids, int64_t ldid, const int32_t* row_map, int map_div,
                                 const int32_t* rows_dev, hipStream_t s);
