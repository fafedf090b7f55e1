// Whole-loop greedy decoders and the teacher-forced forward for gfx950.
//
// One host call enqueues every kernel of every decode step on the caller's stream; nothing here
// synchronises the host or allocates.  Token ids, done flags and output lengths live on the device;
// the arg-max of the vocabulary projection is fused into the projection (logits are never written)
// and the kernel that picks the winner also writes the next step's input row (embedding + position).
#include "common.h"
#include <limits.h>
#include <math.h>

namespace {

constexpr size_t ALIGN = 256;
inline size_t up(size_t n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }

// bump allocator over the caller's workspace; with base == nullptr it only measures
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    template <typename T>
    T* take(size_t n) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += up(n * sizeof(T));
        return p;
    }
};

// ---- bookkeeping kernels ------------------------------------------------------------------------
__global__ void greedy_init_kernel(int32_t* tok, int64_t ldtok, int32_t* done, int64_t* out_ids, int32_t* out_len,
                                   int32_t* dirty, int B, int max_len, int sos, int32_t* fin = nullptr,
                                   int32_t* live = nullptr) {
    const int n = B * max_len;
    if (fin)      // per-step finish counts and row counts of a loop that ends on the device (greedy_advance_kernel)
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t <= max_len; t += gridDim.x * blockDim.x) {
            if (t < max_len) fin[t] = 0;
            live[t] = B;
        }
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) out_ids[idx] = -1;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        tok[(int64_t)b * ldtok] = sos;
        done[b] = 0;
        out_len[b] = 0;
        if (dirty) dirty[b] = 0;
    }
}

// One wavefront per row: merge the per-tile arg-max winners, record the token exactly as the
// reference's host loop does (transformer.py:155-171 / cnnrnn_model.py:90-106: a finished row keeps
// feeding its prediction back but records nothing), and write the next step's input row.
__global__ __launch_bounds__(256) void greedy_advance_kernel(const float* __restrict__ pv, const int* __restrict__ pi,
                                                             int B, int ntiles, int t, int max_len, int eos,
                                                             int32_t* tok, int64_t ldtok, int32_t* done,
                                                             int64_t* out_ids, int32_t* out_len, int32_t* dirty,
                                                             const float* __restrict__ emb,
                                                             const float* __restrict__ pe, int E, int V,
                                                             float* __restrict__ x_next, int64_t ldx_next = 0,
                                                             int32_t* fin = nullptr, int32_t* live_next = nullptr) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (fin) {
        // fin[s]: captions whose <EOS> came at step s.  Once all B finished in the steps before this one (counts no
        // workgroup of this launch changes) nothing is recorded any more and the next step's kernels find a row count
        // of zero: the reference's "all done" exit (cnnrnn_model.py:101-103) without a host synchronisation.
        int fin_before = 0;
        for (int s = lane; s < t; s += 64) fin_before += fin[s];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) fin_before += __shfl_xor(fin_before, o, 64);
        if (fin_before >= B) {
            if (blockIdx.x == 0 && threadIdx.x == 0 && live_next) live_next[0] = 0;
            return;
        }
    }
    if (b >= B) return;
    float v = -INFINITY;
    int i = INT_MAX;
    for (int n = lane; n < ntiles; n += 64) argmax_merge(v, i, pv[capdec_part_off(b, n, ntiles)], pi[capdec_part_off(b, n, ntiles)]);
    wave_argmax(v, i);
    if (i < 0 || i >= V) i = 0;   // all-NaN row: no finite winner
    if (lane == 0) {
        tok[(int64_t)b * ldtok + t + 1] = i;
        if (!done[b]) {
            if (i == eos) {
                done[b] = 1;
                if (fin) atomicAdd(&fin[t], 1);
            } else {
                const int n = out_len[b];
                out_ids[(int64_t)b * max_len + n] = i;
                out_len[b] = n + 1;
                if (i == 0 && dirty) dirty[b] = 1;
            }
        }
    }
    if (x_next) {
        const f32x4* src = reinterpret_cast<const f32x4*>(emb + (int64_t)i * E);
        const f32x4* pp = pe ? reinterpret_cast<const f32x4*>(pe + (int64_t)(t + 1) * E) : nullptr;
        f32x4* dst = reinterpret_cast<f32x4*>(x_next + (int64_t)b * (ldx_next ? ldx_next : (int64_t)E));
        for (int c = lane; c < E / 4; c += 64) {
            f32x4 w = src[c];
            if (pp) w += pp[c];
            dst[c] = w;
        }
    }
}

// ---- additive (Bahdanau) attention over the memory rows of one sample -----------------------------------------
//   e_s = sum_a v_a tanh(u_a + pf[s,a]),  alpha = softmax_s(e),  ctx = sum_s alpha_s enc[s,:]
// One workgroup per sample.  The scores and the weights live in LDS; pf and enc are each read once, 16 bytes per lane.
// (No counterpart in the reference: the soft-attention LSTM decoder of Show-Attend-Tell.)
__global__ __launch_bounds__(256) void additive_attention_kernel(const float* __restrict__ u, int64_t ldu,
                                                                 const float* __restrict__ pf,
                                                                 const float* __restrict__ v,
                                                                 const float* __restrict__ enc, int S, int A, int E,
                                                                 float* __restrict__ ctx, int64_t ldctx,
                                                                 float* __restrict__ alpha) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [S] scores, then [4][E] partial contexts
    float* sc = lds;
    float* part = lds + ((S + 3) / 4) * 4;
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* ub = u + (int64_t)b * ldu;
    const float* pfb = pf + (int64_t)b * S * A;
    const float* eb = enc + (int64_t)b * S * E;
    for (int s = wave; s < S; s += 4) {
        float e = 0.f;
        for (int c = lane; c < A / 4; c += 64) {
            const f32x4 p4 = *reinterpret_cast<const f32x4*>(pfb + (int64_t)s * A + 4 * c);
            const f32x4 u4 = *reinterpret_cast<const f32x4*>(ub + 4 * c);
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(v + 4 * c);
#pragma unroll
            for (int k = 0; k < 4; ++k) e += v4[k] * tanhf(u4[k] + p4[k]);
        }
        e = wave_sum(e);
        if (lane == 0) sc[s] = e;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int s = lane; s < S; s += 64) mx = fmaxf(mx, sc[s]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int s = lane; s < S; s += 64) sm += expf(sc[s] - mx);
    sm = wave_sum(sm);
    // each wavefront accumulates its share of the rows over all columns; the four partial contexts are summed in LDS
    for (int c = lane; c < E / 4; c += 64) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s = wave; s < S; s += 4)
            acc += *reinterpret_cast<const f32x4*>(eb + (int64_t)s * E + 4 * c) * (expf(sc[s] - mx) / sm);
        *reinterpret_cast<f32x4*>(part + wave * E + 4 * c) = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256)
        ctx[(int64_t)b * ldctx + c] = (part[c] + part[E + c]) + (part[2 * E + c] + part[3 * E + c]);
    if (alpha)
        for (int s = threadIdx.x; s < S; s += 256) alpha[(int64_t)b * S + s] = expf(sc[s] - mx) / sm;
}

// Backward of additive_attention_kernel for one decode/training step.  One workgroup per sample; enc and pf are read
// once, the accumulators denc / dpf / dvb (which collect the contributions of all steps of a caption) are read and
// written once.  With a = alpha (saved by the forward pass), g = dctx:
//   dalpha_s = g . enc_s ;  denc_s += a_s g ;  de_s = a_s (dalpha_s - sum_s' a_s' dalpha_s')
//   th = tanh(u + pf_s) ;  dz = de_s v (1 - th^2) ;  du = sum_s dz ;  dpf_s += dz ;  dvb += sum_s de_s th
// dvb is per sample ([B,A]; its column sum over the batch is the gradient of v), so that no two workgroups add to
// the same address.
__global__ __launch_bounds__(256) void additive_attention_bwd_kernel(const float* __restrict__ u, int64_t ldu,
                                                                     const float* __restrict__ pf,
                                                                     const float* __restrict__ v,
                                                                     const float* __restrict__ enc,
                                                                     const float* __restrict__ alpha,
                                                                     const float* __restrict__ dctx, int64_t lddctx,
                                                                     int S, int A, int E, float* __restrict__ du,
                                                                     int64_t lddu, float* __restrict__ dpf,
                                                                     float* __restrict__ denc, float* __restrict__ dvb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [S] alpha, [S] dalpha -> de, [NSG][A] du, [NSG][A] dv
    const int S4 = ((S + 3) / 4) * 4;
    float* al = lds;
    float* de = lds + S4;
    float* pdu = lds + 2 * S4;
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ncol = A / 4;                                        // float4 columns of the attention space
    const int nsg = ncol >= 256 ? 1 : 256 / ncol;                  // row groups working on the same columns
    float* pdv = pdu + nsg * A;
    const float* ub = u + (int64_t)b * ldu;
    const float* pfb = pf + (int64_t)b * S * A;
    const float* eb = enc + (int64_t)b * S * E;
    const float* gb = dctx + (int64_t)b * lddctx;
    float* dpfb = dpf + (int64_t)b * S * A;
    float* dencb = denc + (int64_t)b * S * E;
    for (int s = threadIdx.x; s < S; s += 256) al[s] = alpha[(int64_t)b * S + s];
    __syncthreads();
    // pass 1 over enc: dalpha_s, and denc_s += alpha_s g in the same sweep
    for (int s = wave; s < S; s += 4) {
        const float a = al[s];
        float d = 0.f;
        for (int c = lane; c < E / 4; c += 64) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gb + 4 * c);
            const f32x4 e4 = *reinterpret_cast<const f32x4*>(eb + (int64_t)s * E + 4 * c);
            d += g4[0] * e4[0] + g4[1] * e4[1] + g4[2] * e4[2] + g4[3] * e4[3];
            f32x4* o = reinterpret_cast<f32x4*>(dencb + (int64_t)s * E + 4 * c);
            *o = *o + g4 * a;
        }
        d = wave_sum(d);
        if (lane == 0) de[s] = d;
    }
    __syncthreads();
    float dot = 0.f;
    for (int s = lane; s < S; s += 64) dot += al[s] * de[s];
    dot = wave_sum(dot);                                            // every wavefront computes the same sum
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += 256) de[s] = al[s] * (de[s] - dot);
    __syncthreads();
    // pass 2 over pf: a thread owns float4 columns c, c + 256 * ..., and the rows s = sg, sg + nsg, ...
    const int sg = threadIdx.x / ncol < nsg ? threadIdx.x / ncol : nsg;      // (threads beyond nsg * ncol idle when ncol does not divide 256)
    const int c0 = threadIdx.x - sg * ncol;
    if (sg < nsg) {
        for (int c = c0; c < ncol; c += (ncol >= 256 ? 256 : ncol)) {
            const f32x4 u4 = *reinterpret_cast<const f32x4*>(ub + 4 * c);
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(v + 4 * c);
            f32x4 adu = {0.f, 0.f, 0.f, 0.f}, adv = {0.f, 0.f, 0.f, 0.f};
            for (int s = sg; s < S; s += nsg) {
                const f32x4 p4 = *reinterpret_cast<const f32x4*>(pfb + (int64_t)s * A + 4 * c);
                const float d = de[s];
                f32x4 dz;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float th = tanhf(u4[k] + p4[k]);
                    dz[k] = d * v4[k] * (1.f - th * th);
                    adv[k] += d * th;
                }
                adu += dz;
                f32x4* o = reinterpret_cast<f32x4*>(dpfb + (int64_t)s * A + 4 * c);
                *o = *o + dz;
            }
            *reinterpret_cast<f32x4*>(pdu + sg * A + 4 * c) = adu;
            *reinterpret_cast<f32x4*>(pdv + sg * A + 4 * c) = adv;
        }
    }
    __syncthreads();
    for (int a = threadIdx.x; a < A; a += 256) {
        float su = 0.f, sv = 0.f;
        for (int g = 0; g < nsg; ++g) {
            su += pdu[g * A + a];
            sv += pdv[g * A + a];
        }
        du[(int64_t)b * lddu + a] = su;
        dvb[(int64_t)b * A + a] += sv;
    }
}

// ---- beam search bookkeeping (transformer.py:177-244, cnnrnn_model.py:112-209) --------------------------------
constexpr int BEAM_MAX_W = 8;

// One wavefront per hypothesis row: log-sum-exp of the row and its W largest logits (ties: lowest id first, as
// torch.topk returns them on the CPU), written as log-probabilities.  CAP = capacity of a lane's sorted list (W <= CAP);
// the row is swept with 16-byte loads when it is aligned for them.
template <int CAP, bool VEC>
__global__ __launch_bounds__(256) void beam_topk_kernel(const float* __restrict__ logits, int64_t ld, int N, int V, int W,
                                                        float* __restrict__ top_lp, int32_t* __restrict__ top_id) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* lr = logits + (int64_t)row * ld;
    float bv[CAP];
    int bi[CAP];
#pragma unroll
    for (int k = 0; k < CAP; ++k) { bv[k] = -INFINITY; bi[k] = INT_MAX; }
    float mx = -INFINITY, sm = 0.f;
    // (ids reach a lane in rising order, so a strict compare keeps the lowest id first among equals)
    auto take = [&](float v, int c) {
        if (v > mx) { sm *= __expf(mx - v); mx = v; }
        sm += __expf(v - mx);
        if (!(v > bv[CAP - 1])) return;        // not among this lane's best CAP: the common case
        float cv = v;
        int ci = c;
#pragma unroll
        for (int k = 0; k < CAP; ++k) {
            if (cv > bv[k]) {
                const float tv = bv[k]; const int ti = bi[k];
                bv[k] = cv; bi[k] = ci;
                cv = tv; ci = ti;
            }
        }
    };
    if (VEC) {
        // eight 16-byte loads of a lane in flight at a time (one memory round trip per 8 KiB of the row, not per 1 KiB:
        // the row is a dependent chain otherwise, 40 us whatever the batch); the values are taken in the same order
        const f32x4* l4 = reinterpret_cast<const f32x4*>(lr);
        const int n4 = V / 4;
        for (int c0 = lane; c0 < n4; c0 += 64 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + 64 * u;
                v[u] = l4[c < n4 ? c : n4 - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + 64 * u;
                if (c < n4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) take(v[u][e], 4 * c + e);
                }
            }
        }
    } else {
        for (int c = lane; c < V; c += 64) take(lr[c], c);
    }
    const float wm = wave_max(mx);
    sm = wave_sum(mx > -INFINITY ? sm * __expf(mx - wm) : 0.f);
    const float lse = wm + logf(sm);
    for (int k = 0; k < W; ++k) {
        float v = bv[0];
        int i = bi[0];
        wave_argmax(v, i);                      // best remaining head over the wave
        if (bi[0] == i) {                       // the owning lane pops it
#pragma unroll
            for (int q = 0; q + 1 < CAP; ++q) { bv[q] = bv[q + 1]; bi[q] = bi[q + 1]; }
            bv[CAP - 1] = -INFINITY; bi[CAP - 1] = INT_MAX;
        }
        if (lane == 0) {
            top_lp[(int64_t)row * W + k] = v - lse;
            top_id[(int64_t)row * W + k] = (i >= 0 && i < V) ? i : 0;
        }
    }
}

// The same with a whole workgroup per row, for batches too small to fill the chip with one wavefront per row (a
// wavefront walks its row alone: 156 values per lane at V = 10000, about 30 us whatever the batch): the four
// wavefronts sweep contiguous quarters of the row, their (max, sum exp) pairs and W best candidates meet in LDS and
// the first wavefront merges them -- the sums in quarter order, the candidates by (value, lowest id).
template <int CAP, bool VEC>
__global__ __launch_bounds__(256) void beam_topk_row_kernel(const float* __restrict__ logits, int64_t ld, int N, int V, int W,
                                                            float* __restrict__ top_lp, int32_t* __restrict__ top_id) {
    __shared__ float s_mx[4], s_sm[4], s_v[4 * BEAM_MAX_W];
    __shared__ int s_i[4 * BEAM_MAX_W];
    const int row = blockIdx.x, part = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* lr = logits + (int64_t)row * ld;
    float bv[CAP];
    int bi[CAP];
#pragma unroll
    for (int k = 0; k < CAP; ++k) { bv[k] = -INFINITY; bi[k] = INT_MAX; }
    float mx = -INFINITY, sm = 0.f;
    auto take = [&](float v, int c) {
        if (v > mx) { sm *= __expf(mx - v); mx = v; }
        sm += __expf(v - mx);
        if (!(v > bv[CAP - 1])) return;
        float cv = v;
        int ci = c;
#pragma unroll
        for (int k = 0; k < CAP; ++k) {
            if (cv > bv[k]) {
                const float tv = bv[k]; const int ti = bi[k];
                bv[k] = cv; bi[k] = ci;
                cv = tv; ci = ti;
            }
        }
    };
    if (VEC) {
        const f32x4* l4 = reinterpret_cast<const f32x4*>(lr);
        const int n4 = V / 4, lo = (int)((int64_t)n4 * part / 4), hi = (int)((int64_t)n4 * (part + 1) / 4);
        for (int c0 = lo + lane; c0 < hi; c0 += 64 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + 64 * u;
                v[u] = l4[c < hi ? c : hi - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + 64 * u;
                if (c < hi) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) take(v[u][e], 4 * c + e);
                }
            }
        }
    } else {
        const int lo = (int)((int64_t)V * part / 4), hi = (int)((int64_t)V * (part + 1) / 4);
        for (int c = lo + lane; c < hi; c += 64) take(lr[c], c);
    }
    const float wm = wave_max(mx);
    sm = wave_sum(mx > -INFINITY ? sm * __expf(mx - wm) : 0.f);
    if (lane == 0) { s_mx[part] = wm; s_sm[part] = sm; }
    for (int k = 0; k < W; ++k) {
        float v = bv[0];
        int i = bi[0];
        wave_argmax(v, i);
        if (bi[0] == i) {
#pragma unroll
            for (int q = 0; q + 1 < CAP; ++q) { bv[q] = bv[q + 1]; bi[q] = bi[q + 1]; }
            bv[CAP - 1] = -INFINITY; bi[CAP - 1] = INT_MAX;
        }
        if (lane == 0) { s_v[part * BEAM_MAX_W + k] = v; s_i[part * BEAM_MAX_W + k] = i; }
    }
    __syncthreads();
    if (part != 0) return;
    float M = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    float S = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) S += s_mx[q] > -INFINITY ? s_sm[q] * __expf(s_mx[q] - M) : 0.f;
    const float lse = M + logf(S);
    // lane q*W + k holds candidate k of quarter q; W rounds of "best remaining"
    float cv = -INFINITY;
    int ci = INT_MAX;
    if (lane < 4 * W) { cv = s_v[(lane / W) * BEAM_MAX_W + lane % W]; ci = s_i[(lane / W) * BEAM_MAX_W + lane % W]; }
    for (int k = 0; k < W; ++k) {
        float v = cv;
        int i = ci;
        wave_argmax(v, i);
        if (ci == i) { cv = -INFINITY; ci = INT_MAX; }      // (ids are distinct: exactly one lane holds the winner)
        if (lane == 0) {
            top_lp[(int64_t)row * W + k] = v - lse;
            top_id[(int64_t)row * W + k] = (i >= 0 && i < V) ? i : 0;
        }
    }
}

struct BeamState {
    int32_t* tok; float* score; int32_t* done;     // [N][ldtok], [N], [N]
};

__global__ void beam_init_kernel(BeamState st, int64_t ldtok, int N, int sos, int32_t* __restrict__ unfinished,
                                 int max_len) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    for (int n = g; n < N; n += gridDim.x * blockDim.x) {
        st.tok[(int64_t)n * ldtok] = sos;
        st.score[n] = 0.f;
        st.done[n] = 0;
    }
    for (int t = g; t < max_len; t += gridDim.x * blockDim.x) unfinished[t] = 0;
}

// ---- torch.topk's order among equal values on the CPU, for rows of at most 64 values --------------------------------
// The reference selects the surviving hypotheses with new_scores.topk(beam_width) (transformer.py:211,
// cnnrnn_model.py:172), and a hypothesis that has just produced <EOS> contributes W candidates with exactly the same
// score, so which of several EQUAL candidates survive shapes the rest of the search.  ATen's CPU kernel
// (topk_impl_loop: k * 64 > n) runs std::nth_element(begin, begin + k - 1, end, greater) and then
// std::sort(begin, begin + k - 1, greater) over (value, index) pairs; with GNU libstdc++ that is introselect --
// median of three moved to the front, unguarded Hoare partition, insertion sort below four elements, heap select once
// the depth limit 2 * floor(log2 n) is used up -- followed by one insertion sort (fewer than 16 elements).  The
// functions below perform the same element moves, so the result is the same permutation, ties included.
// A thread's row lives in LDS, element a at [a][thread] (the algorithm indexes its row with run-time indices: as a
// private array it would sit in scratch memory, a memory round trip per access).  Workgroups of TOPK_THREADS threads.
constexpr int TOPK_THREADS = 64, TOPK_MAX_N = BEAM_MAX_W * BEAM_MAX_W;
struct TopkVals { float* p; __device__ float& operator[](int a) const { return p[a * TOPK_THREADS]; } };
struct TopkIdxs { int* p; __device__ int& operator[](int a) const { return p[a * TOPK_THREADS]; } };
struct TopkRow {
    TopkVals v;
    TopkIdxs i;
    __device__ void swap(int a, int b) {
        const float tv = v[a]; v[a] = v[b]; v[b] = tv;
        const int ti = i[a]; i[a] = i[b]; i[b] = ti;
    }
    __device__ void move(int dst, int src) { v[dst] = v[src]; i[dst] = i[src]; }
};
// ATen's comparator for largest = true: NaN first, otherwise strictly greater
__device__ inline bool topk_gt(float x, float y) { return ((x != x) && !(y != y)) || (x > y); }

__device__ inline int topk_partition_pivot(TopkRow& q, int first, int last) {
    const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
    int m;
    if (topk_gt(q.v[a], q.v[b])) m = topk_gt(q.v[b], q.v[c]) ? b : (topk_gt(q.v[a], q.v[c]) ? c : a);
    else m = topk_gt(q.v[a], q.v[c]) ? a : (topk_gt(q.v[b], q.v[c]) ? c : b);
    q.swap(first, m);
    int lo = first + 1, hi = last;
    const float piv = q.v[first];
    for (;;) {
        while (topk_gt(q.v[lo], piv)) ++lo;
        --hi;
        while (topk_gt(piv, q.v[hi])) --hi;
        if (!(lo < hi)) return lo;
        q.swap(lo, hi);
        ++lo;
    }
}

__device__ inline void topk_insertion_sort(TopkRow& q, int first, int last) {
    for (int n = first + 1; n < last; ++n) {
        const float val = q.v[n];
        const int vi = q.i[n];
        int j = n;
        if (topk_gt(val, q.v[first])) {
            for (; j > first; --j) q.move(j, j - 1);
        } else {
            while (topk_gt(val, q.v[j - 1])) { q.move(j, j - 1); --j; }
        }
        q.v[j] = val; q.i[j] = vi;
    }
}

__device__ inline void topk_adjust_heap(TopkRow& q, int first, int hole, int len, float val, int vi) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (topk_gt(q.v[first + child], q.v[first + child - 1])) --child;
        q.move(first + hole, first + child);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        q.move(first + hole, first + child - 1);
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && topk_gt(q.v[first + parent], val)) {
        q.move(first + hole, first + parent);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    q.v[first + hole] = val; q.i[first + hole] = vi;
}

__device__ inline void topk_heap_select(TopkRow& q, int first, int middle, int last) {
    const int len = middle - first;
    if (len >= 2)
        for (int parent = (len - 2) / 2; parent >= 0; --parent)
            topk_adjust_heap(q, first, parent, len, q.v[first + parent], q.i[first + parent]);
    for (int n = middle; n < last; ++n)
        if (topk_gt(q.v[n], q.v[first])) {
            const float val = q.v[n];
            const int vi = q.i[n];
            q.move(n, first);
            topk_adjust_heap(q, first, 0, len, val, vi);
        }
}

// The k best of q.v[0..n) end up in q.v / q.i [0..k), best first, as torch.topk(values, k) returns them on the CPU
// (1 <= k <= n <= 64, k * 64 > n; q.i must hold 0..n-1 on entry).
__device__ inline void topk_cpu_order(TopkRow& q, int n, int k) {
    int first = 0, last = n;
    const int nth = k - 1;
    int depth = 2 * (31 - __clz(n));
    bool selected = false;
    while (last - first > 3) {
        if (depth == 0) {
            topk_heap_select(q, first, nth + 1, last);
            q.swap(first, nth);
            selected = true;
            break;
        }
        --depth;
        const int cut = topk_partition_pivot(q, first, last);
        if (cut <= nth) first = cut; else last = cut;
    }
    if (!selected) topk_insertion_sort(q, first, last);
    topk_insertion_sort(q, 0, k - 1);
}

__global__ void topk_rows_kernel(const float* __restrict__ x, int rows, int n, int k, float* __restrict__ out_val,
                                 int32_t* __restrict__ out_idx) {
    __shared__ float sv[TOPK_MAX_N * TOPK_THREADS];
    __shared__ int si[TOPK_MAX_N * TOPK_THREADS];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    TopkRow q{{sv + threadIdx.x}, {si + threadIdx.x}};
    for (int c = 0; c < n; ++c) { q.v[c] = x[(int64_t)r * n + c]; q.i[c] = c; }
    topk_cpu_order(q, n, k);
    for (int c = 0; c < k; ++c) { out_val[(int64_t)r * k + c] = q.v[c]; out_idx[(int64_t)r * k + c] = q.i[c]; }
}

// One thread per sample: the W*W candidates of its hypotheses, best W kept in torch.topk's order.  Writes the next
// state (token prefixes copied from the parents) and the parent map.
// `done` holds "this hypothesis' last token is <EOS>" and nothing else: the reference's `mask[:, :, 0] = False`
// (transformer.py:199-201, cnnrnn_model.py:161-163) writes through a stride-0 view and clears its `done` tensor at
// every step, so a hypothesis is frozen (adds (1 - 1) * log-prob) for the one step after its <EOS>, all W of its
// proposals stay eligible with that same score, and it accumulates log-probabilities again afterwards; proposals are
// masked to -inf at the first step only (hypotheses 1.. : all W start identical).
// unfinished[t] counts the samples that received a token other than <EOS> at step t.  The reference leaves its loop
// (`if done.all(): break`, transformer.py:231) when that count is 0, i.e. when all B*W hypotheses produced <EOS> in
// the same step; from then on this kernel passes the state through unchanged (the host may have enqueued further
// steps without looking), appending <EOS> so that later kernels still read valid token ids.
__global__ void beam_select_kernel(int step, int B, int W, int eos, BeamState cur, BeamState nxt, int64_t ldtok,
                                   const float* __restrict__ top_lp, const int32_t* __restrict__ top_id,
                                   int32_t* __restrict__ parent, int32_t* __restrict__ unfinished,
                                   int32_t* __restrict__ pad_seen = nullptr) {
    __shared__ float sv[TOPK_MAX_N * TOPK_THREADS];
    __shared__ int si[TOPK_MAX_N * TOPK_THREADS];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // a hypothesis' token prefix, eight loads in flight at a time (the two states never alias; a load-store chain per
    // token made this kernel grow from 10 to 30 us over 20 steps)
    auto copy_prefix = [&](int dst, int src) {
        const int32_t* sp = cur.tok + (int64_t)src * ldtok;
        int32_t* dp = nxt.tok + (int64_t)dst * ldtok;
        for (int p0 = 0; p0 <= step; p0 += 8) {
            int32_t tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tmp[u] = sp[min(p0 + u, step)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + u <= step) dp[p0 + u] = tmp[u];
        }
    };
    if (step > 0 && unfinished[step - 1] == 0) {           // the reference's loop has ended: keep everything
        for (int w = 0; w < W; ++w) {
            const int n = b * W + w;
            copy_prefix(n, n);
            nxt.tok[(int64_t)n * ldtok + step + 1] = eos;
            nxt.score[n] = cur.score[n];
            nxt.done[n] = 1;
            parent[n] = n;
        }
        return;                                            // unfinished[step] stays 0
    }
    TopkRow q{{sv + threadIdx.x}, {si + threadIdx.x}};
    for (int w = 0; w < W; ++w) {
        const int n = b * W + w;
        const float keep = cur.done[n] != 0 ? 0.f : 1.f;
        for (int k = 0; k < W; ++k) {
            float v = cur.score[n] + keep * top_lp[(int64_t)n * W + k];
            if (step == 0 && w > 0) v = -INFINITY;
            q.v[w * W + k] = v;
            q.i[w * W + k] = w * W + k;
        }
    }
    topk_cpu_order(q, W * W, W);
    bool every = true;
    for (int r = 0; r < W; ++r) {
        const int best = q.i[r];
        const int pw = best / W, pk = best - pw * W;
        const int src = b * W + pw, dst = b * W + r;
        const int tokn = top_id[(int64_t)src * W + pk];
        copy_prefix(dst, src);
        nxt.tok[(int64_t)dst * ldtok + step + 1] = tokn;
        nxt.score[dst] = q.v[r];
        const int fin = tokn == eos ? 1 : 0;
        nxt.done[dst] = fin;
        parent[dst] = src;
        if (pad_seen && tokn == 0) pad_seen[b] = 1;      // a <PAD> query row attends uniformly: no cache can hold that
        every = every && fin;
    }
    if (!every) atomicAdd(unfinished + step, 1);
}

// dst[n,:] = src[parent[n],:]
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ parent,
                                                          int div, int N, int D, float* __restrict__ dst) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const int from = parent ? parent[row] : row / div;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src + (int64_t)from * D);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst + (int64_t)row * D);
    for (int c = lane; c < D / 4; c += 64) d4[c] = s4[c];
}

// Key/value caches follow their hypotheses: dst[a][n][p][:] = src[a][parent[n]][p][:] for p < len, for every cache
// array a (layer x {K,V}) of [N][max_len][E] floats.  One wavefront per (n, p) row.
__global__ __launch_bounds__(256) void beam_cache_gather_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                const int32_t* __restrict__ parent, int N, int len,
                                                                int max_len, int E) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long)N * len) return;
    const int n = (int)(row / len), p = (int)(row - (long)n * len);
    const size_t arr = (size_t)blockIdx.y * N * max_len * E;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src + arr + ((size_t)parent[n] * max_len + p) * E);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst + arr + ((size_t)n * max_len + p) * E);
    for (int c = lane; c < E / 4; c += 64) d4[c] = s4[c];
}

// Best hypothesis of every sample without <SOS>, cut at its first <EOS> (transformer.py:233-241)
__global__ void beam_result_kernel(BeamState st, int64_t ldtok, int B, int W, int steps, int max_len, int eos,
                                   int64_t* __restrict__ out_ids, int32_t* __restrict__ out_len,
                                   float* __restrict__ out_scores, const int32_t* __restrict__ unfinished,
                                   int32_t* __restrict__ n_unfinished) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (b == 0) n_unfinished[0] = unfinished[steps - 1];
    const int32_t* row = st.tok + (int64_t)b * W * ldtok;
    int n = 0;
    for (int p = 1; p <= steps; ++p) {
        const int tk = row[p];
        if (tk == eos) break;
        out_ids[(int64_t)b * max_len + n++] = tk;
    }
    for (int p = n; p < max_len; ++p) out_ids[(int64_t)b * max_len + p] = -1;
    out_len[b] = n;
    if (out_scores)
        for (int w = 0; w < W; ++w) out_scores[b * W + w] = st.score[b * W + w];
}

int check_model(const capdec_tf_model* m) {
    CAPDEC_REQUIRE(m && m->E >= 16 && m->E % 4 == 0 && m->H >= 1 && m->E % m->H == 0 && m->L >= 0 && m->V >= 1 &&
                   m->max_pos >= 1);
    const int dk = m->E / m->H;
    CAPDEC_REQUIRE(dk == 16 || dk == 32 || dk == 64 || dk == 128 || dk == 256);
    CAPDEC_REQUIRE(m->emb && m->pe && m->fc2_w && m->fc2_b && (m->L == 0 || m->layers));
    return 0;
}

capdec_gemm_group linear_group(const float* x, int64_t ldx, int M, const float* W, const float* b, int N, float* C,
                               int64_t ldc, const float* R = nullptr, int64_t ldr = 0) {
    capdec_gemm_group g{};
    g.P = x; g.Q = W; g.C = C; g.bias_j = b; g.R = R;
    g.ldp = ldx; g.ldq = 0 /* set by caller */; g.ldci = ldc; g.ldcj = 1; g.ldri = ldr; g.ldrj = 1;
    g.I = M; g.J = N;
    return g;
}

// y = x W^T + b (+ R), W [N,K]
int linear(const float* x, int64_t ldx, int M, const float* W, const float* b, int N, int K, float* C, int64_t ldc,
           const float* R, int64_t ldr, int flags, hipStream_t s) {
    capdec_gemm_group g = linear_group(x, ldx, M, W, b, N, C, ldc, R, ldr);
    g.ldq = K;
    return capdec_launch_gemm(&g, 1, K, flags, s);
}

// ---- cross-attention memory: K/V projections of the encoder rows, once per call -----------------
int project_memory(const capdec_tf_model* m, const float* enc, int B, int S, float* ck, float* cv, hipStream_t s) {
    const int E = m->E;
    for (int l = 0; l < m->L; ++l) {
        const capdec_attn_w& w = m->layers[l].cross_attn;
        capdec_gemm_group g[2];
        g[0] = linear_group(enc, E, B * S, w.wk, w.bk, E, ck + (size_t)l * B * S * E, E);
        g[1] = linear_group(enc, E, B * S, w.wv, w.bv, E, cv + (size_t)l * B * S * E, E);
        g[0].ldq = g[1].ldq = E;
        CAPDEC_TRY(capdec_launch_gemm(g, 2, E, 0, s));
    }
    return 0;
}

// ---- two memory rows: the query-independent parts of the cross-attention block, once per call -----
// dK = k0 - k1, dV = v0 - v1 of each sample's projected memory rows, and sb[b,h] = b_q,h . dK_h
__global__ __launch_bounds__(256) void pair_diff_kernel(const float* __restrict__ ck, const float* __restrict__ cv,
                                                        const float* __restrict__ bq, int B, int E, int H,
                                                        float* __restrict__ dk_out, float* __restrict__ dv_out,
                                                        float* __restrict__ sb) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int nv = E / 4, hv = nv / H;
    const f32x4* k0 = reinterpret_cast<const f32x4*>(ck + (int64_t)(2 * b) * E);
    const f32x4* v0 = reinterpret_cast<const f32x4*>(cv + (int64_t)(2 * b) * E);
    const f32x4* bq4 = reinterpret_cast<const f32x4*>(bq);
    f32x4* dk4 = reinterpret_cast<f32x4*>(dk_out + (int64_t)b * E);
    f32x4* dv4 = reinterpret_cast<f32x4*>(dv_out + (int64_t)b * E);
    for (int c = lane; c < nv; c += 64) {
        dk4[c] = k0[c] - k0[nv + c];
        dv4[c] = v0[c] - v0[nv + c];
    }
    for (int h = 0; h < H; ++h) {
        float part = 0.f;
        for (int c = h * hv + lane; c < (h + 1) * hv; c += 64) {
            const f32x4 d = k0[c] - k0[nv + c], q = bq4[c];
            part += d[0] * q[0] + d[1] * q[1] + d[2] * q[2] + d[3] * q[3];
        }
        for (int o = 32; o; o >>= 1) part += __shfl_xor(part, o, 64);
        if (lane == 0) sb[(int64_t)b * H + h] = part;
    }
}

// Per layer: G [B][H][E] = W_q,h^T dK_h, D [B][H][E] = W_o[:, h] dV_h, c1 [B][E] = W_o v1 + b_o, sb [B][H] (the operands
// of capdec_launch_layernorm_pair_rows).  ck, cv: the projected memory, [L][B*2][E]; dkb, dvb: scratch [B][E] each.
int pair_memory_constants(const capdec_tf_model* m, const float* ck, const float* cv, int B, float* dkb, float* dvb,
                          float* G, float* D, float* c1, float* sb, hipStream_t s) {
    const int E = m->E, H = m->H, dk = E / H;
    const size_t BE = (size_t)B * E;
    for (int l = 0; l < m->L; ++l) {
        const capdec_attn_w& a = m->layers[l].cross_attn;
        const float* ckl = ck + (size_t)l * 2 * BE;
        const float* cvl = cv + (size_t)l * 2 * BE;
        hipLaunchKernelGGL(pair_diff_kernel, dim3((B + 3) / 4), dim3(256), 0, s, ckl, cvl, a.bq, B, E, H, dkb, dvb,
                           sb + (size_t)l * B * H);
        CAPDEC_LAUNCH_CHECK();
        for (int h0 = 0; h0 < H; h0 += 4) {
            const int n = H - h0 < 4 ? H - h0 : 4;
            capdec_gemm_group gg[4], gd[4];
            for (int i = 0; i < n; ++i) {
                const int h = h0 + i;
                // G_h[b, e] = sum_d dK[b, h*dk + d] W_q[h*dk + d, e]: the rows h*dk.. of W_q are its [k][j] image
                gg[i] = linear_group(dkb + h * dk, E, B, a.wq + (size_t)h * dk * E, nullptr, E,
                                     G + (size_t)l * BE * H + (size_t)h * E, (int64_t)H * E);
                gg[i].ldq = E;
                // D_h[b, e] = sum_d dV[b, h*dk + d] W_o[e, h*dk + d]
                gd[i] = linear_group(dvb + h * dk, E, B, a.wo + h * dk, nullptr, E, D + (size_t)l * BE * H + (size_t)h * E,
                                     (int64_t)H * E);
                gd[i].ldq = E;
            }
            CAPDEC_TRY(capdec_launch_gemm(gg, n, dk, CAPDEC_GEMM_TQ, s));
            CAPDEC_TRY(capdec_launch_gemm(gd, n, dk, 0, s));
        }
        CAPDEC_TRY(linear(cvl + E, 2 * E, B, a.wo, a.bo, E, E, c1 + (size_t)l * BE, E, nullptr, 0, 0, s));
    }
    return 0;
}

// ---- long memories in decode: key / value projections folded into the query and output projections --------------
// scores_h = ((x1 W_q^T + b_q)_h W_k,h) mem^T / sqrt(dk) = (x1 W1_h^T + b1_h) mem^T / sqrt(dk),  W1_h = W_k,h^T W_q,h  [E,E]
// block   = sum_h (P_h mem) (W_o,h W_v,h)^T + W_o b_v + b_o = ctx W2^T + b2,  W2[:, h] = W_o[:, h] W_v,h  [E,E] per head
// (b_k shifts every score of a row equally and drops out of the softmax; b_v passes through because the probabilities
//  of a row sum to one; transformer.py:21-27,45-60,86).  Weights only: computed once per call, four small launches a
//  layer.  W1 [L][H*E][E], b1 [L][H*E], W2 [L][E][H*E], b2 [L][E].
int fold_memory_weights(const capdec_tf_model* m, float* W1, float* b1, float* W2, float* b2, hipStream_t s) {
    const int E = m->E, H = m->H, dk = E / H;
    const size_t EE = (size_t)E * E;
    for (int l = 0; l < m->L; ++l) {
        const capdec_attn_w& a = m->layers[l].cross_attn;
        for (int h0 = 0; h0 < H; h0 += 4) {
            const int n = H - h0 < 4 ? H - h0 : 4;
            capdec_gemm_group g1[4], gb[4], g2[4];
            for (int i = 0; i < n; ++i) {
                const int h = h0 + i;
                const float* wkh = a.wk + (size_t)h * dk * E;      // rows h*dk.. of W_k: its [k][j] image for this head
                const float* wqh = a.wq + (size_t)h * dk * E;
                g1[i] = linear_group(wkh, E, E, wqh, nullptr, E, W1 + (size_t)l * H * EE + (size_t)h * EE, E);
                g1[i].ldq = E;
                gb[i] = linear_group(a.bq + h * dk, dk, 1, wkh, nullptr, E, b1 + (size_t)l * H * E + (size_t)h * E, E);
                gb[i].ldq = E;
                g2[i] = linear_group(a.wo + h * dk, E, E, a.wv + (size_t)h * dk * E, nullptr, E,
                                     W2 + (size_t)l * H * EE + (size_t)h * E, (int64_t)H * E);
                g2[i].ldq = E;
            }
            CAPDEC_TRY(capdec_launch_gemm(g1, n, dk, CAPDEC_GEMM_TP | CAPDEC_GEMM_TQ, s));
            CAPDEC_TRY(capdec_launch_gemm(gb, n, dk, CAPDEC_GEMM_TQ, s));
            CAPDEC_TRY(capdec_launch_gemm(g2, n, dk, CAPDEC_GEMM_TQ, s));
        }
        CAPDEC_TRY(linear(a.bv, E, 1, a.wo, a.bo, E, E, b2 + (size_t)l * E, E, nullptr, 0, 0, s));
    }
    return 0;
}

// ---- decoder stack over a block of rows ------------------------------------------------------
// Full-prefix form: X [B*T,E] in, X out (post-LN); every position of every sample is recomputed.
struct FullBufs {
    float *X, *Y, *X1, *X2, *Qb, *Kb, *Vb, *AO, *Hb;
};

int decoder_stack_full(const capdec_tf_model* m, const FullBufs& w, const float* ck, const float* cv, int B, int S,
                       int T, const int32_t* tok, int64_t ldtok, hipStream_t s) {
    const int E = m->E, H = m->H, dk = E / H, M = B * T;
    void* vs = (void*)s;
    for (int l = 0; l < m->L; ++l) {
        const capdec_tf_layer& ly = m->layers[l];
        capdec_gemm_group g[3];
        g[0] = linear_group(w.X, E, M, ly.self_attn.wq, ly.self_attn.bq, E, w.Qb, E);
        g[1] = linear_group(w.X, E, M, ly.self_attn.wk, ly.self_attn.bk, E, w.Kb, E);
        g[2] = linear_group(w.X, E, M, ly.self_attn.wv, ly.self_attn.bv, E, w.Vb, E);
        g[0].ldq = g[1].ldq = g[2].ldq = E;
        CAPDEC_TRY(capdec_launch_gemm(g, 3, E, 0, s));
        CAPDEC_TRY(capdec_attention(w.Qb, E, w.Kb, E, (int64_t)T * E, w.Vb, E, (int64_t)T * E, w.AO, E, B, H, T, T, dk,
                                    1, 0, tok, ldtok, nullptr, vs));
        CAPDEC_TRY(linear(w.AO, E, M, ly.self_attn.wo, ly.self_attn.bo, E, E, w.Y, E, w.X, E, 0, s));
        CAPDEC_TRY(capdec_add_layernorm(w.Y, nullptr, 1, ly.n1_g, ly.n1_b, 1e-5f, M, E, w.X1, nullptr, nullptr, vs));
        // cross attention over the S memory rows (never masked)
        CAPDEC_TRY(linear(w.X1, E, M, ly.cross_attn.wq, ly.cross_attn.bq, E, E, w.Qb, E, nullptr, 0, 0, s));
        CAPDEC_TRY(capdec_attention(w.Qb, E, ck + (size_t)l * B * S * E, E, (int64_t)S * E,
                                    cv + (size_t)l * B * S * E, E, (int64_t)S * E, w.AO, E, B, H, T, S, dk, 0, 0,
                                    nullptr, 0, nullptr, vs));
        CAPDEC_TRY(linear(w.AO, E, M, ly.cross_attn.wo, ly.cross_attn.bo, E, E, w.Y, E, w.X1, E, 0, s));
        CAPDEC_TRY(capdec_add_layernorm(w.Y, nullptr, 1, ly.n2_g, ly.n2_b, 1e-5f, M, E, w.X2, nullptr, nullptr, vs));
        // feed forward, d_ff = E
        CAPDEC_TRY(linear(w.X2, E, M, ly.ff_w1, ly.ff_b1, E, E, w.Hb, E, nullptr, 0, CAPDEC_GEMM_RELU, s));
        CAPDEC_TRY(linear(w.Hb, E, M, ly.ff_w2, ly.ff_b2, E, E, w.Y, E, w.X2, E, 0, s));
        CAPDEC_TRY(capdec_add_layernorm(w.Y, nullptr, 1, ly.n3_g, ly.n3_b, 1e-5f, M, E, w.X, nullptr, nullptr, vs));
    }
    return 0;
}

}  // namespace

extern "C" {

// =================================================================================================
// KV-cached greedy decode
// =================================================================================================
// Rows of a step.  Row r < B is the new position t of sample r.  Rows r >= B re-run earlier positions of the few
// samples that have generated <PAD>: the reference masks a PAD *query row*, which then attends uniformly over the
// CURRENT length, so from the first PAD position p0 on, every position p0..t-1 of that sample changes with each
// step (transformer.py:117-123) and is recomputed here through the same kernels, its k/v written back to the cache.
// rowmap[t][r] = b * max_len + position; nrows[t] = B + number of such rows (device side: the host never sees it).
struct DecodeWs {
    float *x, *y, *x1, *x2, *q, *ao, *hb;   // [Mcap][E], Mcap = B + extra_cap
    float *kc, *vc;          // [L][B][max_len][E]
    float *ck, *cv;          // [L][B*S][E]
    float *cc;               // [L][B][E]  S == 1: the whole cross-attention block output per sample (S == 2: c1)
    float *pg, *pd, *psb;    // S == 2: [L][B][H][E] x2 and [L][B][H], see pair_memory_constants
    float *fw1, *fb1, *fw2, *fb2;   // S >= 3: folded cross-attention weights, see fold_memory_weights
    float *fq, *fc;          // [Mcap][H*E]: folded queries and contexts of a step
    float* pval; int32_t* pidx;
    int32_t *tok, *done, *first_pad;
    int32_t *rowmap;         // [max_len][Mcap]
    int32_t *nrows;          // [max_len + 1]
    int32_t *fin;            // [max_len] captions whose <EOS> came at step t
    // all layers of a step in one launch (decode_fused.hip; S <= 2 and a supported shape only, otherwise null)
    float *tq, *tctx, *ty, *th, *tz;    // [teams][32][E] exchange buffers of the teams
    float *sctx, *sy, *sh, *sz;         // [teams][16][E] the same of their side rows (earlier <PAD> positions)
    unsigned* ctr;           // [max_len][teams][32] arrival counters, then 32 words whose first is the error word
    int32_t *pcount, *plist; // [teams], [teams][16]: <PAD> positions generated by the team's samples
    int fused_teams;
};

static int default_extra_cap(int B) { return ((B / 4 > 64 ? B / 4 : 64) + 63) / 64 * 64; }

static size_t carve_decode(const capdec_tf_model* m, int B, int S, int max_len, int extra_cap, void* base,
                           DecodeWs* w) {
    Carver c(base);
    const size_t BE = (size_t)B * m->E;
    const size_t ME = (size_t)(B + extra_cap) * m->E;
    DecodeWs d;
    d.x = c.take<float>(ME); d.y = c.take<float>(ME); d.x1 = c.take<float>(ME); d.x2 = c.take<float>(ME);
    d.q = c.take<float>(ME); d.ao = c.take<float>(ME); d.hb = c.take<float>(ME);
    d.kc = c.take<float>((size_t)m->L * BE * max_len);
    d.vc = c.take<float>((size_t)m->L * BE * max_len);
    d.ck = c.take<float>((size_t)m->L * BE * S);
    d.cv = c.take<float>((size_t)m->L * BE * S);
    d.cc = c.take<float>((size_t)m->L * BE);
    d.pg = d.pd = d.psb = nullptr;
    if (S == 2) {
        d.pg = c.take<float>((size_t)m->L * BE * m->H);
        d.pd = c.take<float>((size_t)m->L * BE * m->H);
        d.psb = c.take<float>((size_t)m->L * B * m->H);
    }
    d.fw1 = d.fb1 = d.fw2 = d.fb2 = d.fq = d.fc = nullptr;
    if (S >= 3 && capdec_attention_memory_supported(m->E)) {
        const size_t HEE = (size_t)m->H * m->E * m->E;
        d.fw1 = c.take<float>((size_t)m->L * HEE);
        d.fb1 = c.take<float>((size_t)m->L * m->H * m->E);
        d.fw2 = c.take<float>((size_t)m->L * HEE);
        d.fb2 = c.take<float>((size_t)m->L * m->E);
        d.fq = c.take<float>(ME * m->H);
        d.fc = c.take<float>(ME * m->H);
    }
    const size_t nt = (size_t)capdec_vocab_tiles(m->V);
    d.pval = c.take<float>(capdec_part_elems(B, (int)nt));
    d.pidx = c.take<int32_t>(capdec_part_elems(B, (int)nt));
    d.tok = c.take<int32_t>((size_t)B * (max_len + 1));
    d.done = c.take<int32_t>((size_t)B);
    d.first_pad = c.take<int32_t>((size_t)B);
    d.rowmap = c.take<int32_t>((size_t)max_len * (B + extra_cap));
    d.nrows = c.take<int32_t>((size_t)max_len + 1);
    d.fin = c.take<int32_t>((size_t)max_len);
    d.tq = d.tctx = d.ty = d.th = d.tz = d.sctx = d.sy = d.sh = d.sz = nullptr;
    d.ctr = nullptr;
    d.pcount = d.plist = nullptr;
    d.fused_teams = 0;
    if ((S == 1 || S == 2) && capdec_fused_layers_supported(m->E, m->H, m->L, B, max_len)) {
        d.fused_teams = (B + CAPDEC_FUSED_ROWS - 1) / CAPDEC_FUSED_ROWS;
        const size_t TE = (size_t)d.fused_teams * CAPDEC_FUSED_ROWS * m->E;
        const size_t SE = (size_t)d.fused_teams * CAPDEC_FUSED_SIDE_ROWS * m->E;
        d.tq = c.take<float>(TE); d.tctx = c.take<float>(TE); d.ty = c.take<float>(TE);
        d.th = c.take<float>(TE); d.tz = c.take<float>(TE);
        d.sctx = c.take<float>(SE); d.sy = c.take<float>(SE); d.sh = c.take<float>(SE); d.sz = c.take<float>(SE);
        d.ctr = c.take<unsigned>(((size_t)max_len * d.fused_teams + 1) * 32);
        d.pcount = c.take<int32_t>((size_t)d.fused_teams);
        d.plist = c.take<int32_t>((size_t)d.fused_teams * CAPDEC_FUSED_SIDE_ROWS);
    }
    if (w) *w = d;
    return c.off;
}

__global__ void cached_init_kernel(int32_t* tok, int64_t ldtok, int32_t* done, int32_t* first_pad, int64_t* out_ids,
                                   int32_t* out_len, int32_t* dirty, int32_t* rowmap, int32_t* nrows,
                                   int32_t* fin, int B, int max_len, int sos, int32_t* pcount, int32_t* plist, int nteams,
                                   const float* __restrict__ emb, const float* __restrict__ pe, int E, float* __restrict__ x,
                                   unsigned* ctr, int ctr_n) {
    const int n = B * max_len;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    // step 0's input rows: every caption starts at <SOS>, position 0 (the sum embed_pe_kernel would form, row by row)
    {
        const int e4 = E / 4;
        const f32x4* src = reinterpret_cast<const f32x4*>(emb + (int64_t)sos * E);
        const f32x4* pp = reinterpret_cast<const f32x4*>(pe);
        f32x4* dst = reinterpret_cast<f32x4*>(x);
        for (int idx = tid; idx < B * e4; idx += nth) {
            const int c = idx % e4;
            f32x4 v = src[c];
            v += pp[c];
            dst[idx] = v;
        }
    }
    // arrival counters of the one-launch layers (every step's start at zero) and their error word
    for (int idx = tid; idx < ctr_n; idx += nth) ctr[idx] = 0;
    for (int idx = tid; idx < n; idx += nth) out_ids[idx] = -1;
    for (int b = tid; b < B; b += nth) {
        tok[(int64_t)b * ldtok] = sos;
        done[b] = 0;
        first_pad[b] = INT_MAX;
        out_len[b] = 0;
        if (dirty) dirty[b] = 0;
        rowmap[b] = b * max_len;            // step 0: position 0 of every sample
    }
    for (int t = tid; t <= max_len; t += nth) nrows[t] = B;
    for (int t = tid; t < max_len; t += nth) fin[t] = 0;
    if (pcount) {
        for (int t = tid; t < nteams; t += nth) pcount[t] = 0;
        for (int t = tid; t < nteams * CAPDEC_FUSED_SIDE_ROWS; t += nth) plist[t] = -1;      // free slots of the <PAD> lists
    }
}

// One wavefront per sample: pick the token (as greedy_advance_kernel), then prepare step t+1: the sample's new row and,
// if it holds a <PAD> and is still running, one re-run row per position from the first PAD on.
// fin[s] counts the captions whose <EOS> came at step s.  The reference leaves its loop once every caption has one
// (transformer.py:166-168); here the steps are enqueued ahead of time, so the loop ends on the device instead: a
// launch that finds all B captions finished in the steps BEFORE its own (counts that no workgroup of this launch
// changes, so every workgroup decides alike) records nothing - whatever a finished caption would predict is
// discarded anyway - and sets the next step's row count to zero, which every kernel of that step (projections,
// attention, LayerNorm, the vocabulary projection) reads before it does anything.  The step after the last <EOS>
// still runs in full: one step is the price of not synchronising with the host.
__global__ __launch_bounds__(256) void cached_advance_kernel(const float* __restrict__ pv, const int* __restrict__ pi,
                                                             int B, int ntiles, int t, int max_len, int eos,
                                                             int32_t* tok, int64_t ldtok, int32_t* done,
                                                             int32_t* first_pad, int64_t* out_ids, int32_t* out_len,
                                                             int32_t* dirty, const float* __restrict__ emb,
                                                             const float* __restrict__ pe, int E, int V,
                                                             float* __restrict__ x_next, int32_t* rowmap_next,
                                                             int32_t* nrows_next, int extra_cap, int32_t* fin,
                                                             int32_t* pcount, int32_t* plist, const unsigned* fuse_err) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (fuse_err && fuse_err[0] && dirty && b < B && lane == 0) dirty[b] = 2;   // a wait of the one-launch layers ran out: redo the sample exactly
    int fin_before = 0;
    for (int s = lane; s < t; s += 64) fin_before += fin[s];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fin_before += __shfl_xor(fin_before, o, 64);
    if (fin_before >= B) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && nrows_next) {
            nrows_next[0] = 0;
            rowmap_next[0] = 0;          // waves beyond the row count still look up row 0's sample (see attention_kernel)
        }
        return;
    }
    if (b >= B) return;
    float v = -INFINITY;
    int i = INT_MAX;
    for (int n = lane; n < ntiles; n += 64) argmax_merge(v, i, pv[capdec_part_off(b, n, ntiles)], pi[capdec_part_off(b, n, ntiles)]);
    wave_argmax(v, i);
    if (i < 0 || i >= V) i = 0;   // all-NaN row: no finite winner
    int p0 = INT_MAX, running = 0, base = 0;
    if (lane == 0) {
        tok[(int64_t)b * ldtok + t + 1] = i;
        if (!done[b]) {
            if (i == eos) {
                done[b] = 1;
                atomicAdd(&fin[t], 1);
            } else {
                const int n = out_len[b];
                out_ids[(int64_t)b * max_len + n] = i;
                out_len[b] = n + 1;
                if (i == 0 && first_pad[b] == INT_MAX) first_pad[b] = t + 1;
                if (i == 0 && pcount) {
                    // one launch for all layers: the <PAD> position joins the side rows of its sample's team
                    const int team = b / CAPDEC_FUSED_ROWS;
                    const int slot = atomicAdd(&pcount[team], 1);
                    if (slot < CAPDEC_FUSED_SIDE_ROWS) plist[team * CAPDEC_FUSED_SIDE_ROWS + slot] = b * max_len + t + 1;
                    else if (dirty && dirty[b] == 0) dirty[b] = 1;
                }
            }
        }
        p0 = first_pad[b];
        running = !done[b];          // a finished row's later tokens are discarded: no need to keep it exact
        if (x_next) {
            rowmap_next[b] = b * max_len + t + 1;
            if (!pcount && p0 != INT_MAX && running && t + 1 > p0) base = atomicAdd(nrows_next, t + 1 - p0) - B;
        }
    }
    if (!x_next) return;
    p0 = __shfl(p0, 0, 64);
    running = __shfl(running, 0, 64);
    base = __shfl(base, 0, 64);
    const int nv = E / 4;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(emb + (int64_t)i * E);
        const f32x4* pp = reinterpret_cast<const f32x4*>(pe + (int64_t)(t + 1) * E);
        f32x4* dst = reinterpret_cast<f32x4*>(x_next + (int64_t)b * E);
        for (int c = lane; c < nv; c += 64) dst[c] = src[c] + pp[c];
    }
    if (pcount || p0 == INT_MAX || !running || t + 1 <= p0) return;
    const int cnt = t + 1 - p0;      // positions p0 .. t are re-run at step t+1
    if (base + cnt > extra_cap && lane == 0 && dirty && dirty[b] == 0) dirty[b] = 1;   // list full: the host re-decodes this row exactly
    for (int k = 0; k < cnt; ++k) {
        const int slot = base + k;
        if (slot >= extra_cap) break;
        const int p = p0 + k;
        if (lane == 0) rowmap_next[B + slot] = b * max_len + p;
        const int id = tok[(int64_t)b * ldtok + p];
        const f32x4* src = reinterpret_cast<const f32x4*>(emb + (int64_t)id * E);
        const f32x4* pp = reinterpret_cast<const f32x4*>(pe + (int64_t)p * E);
        f32x4* dst = reinterpret_cast<f32x4*>(x_next + (int64_t)(B + slot) * E);
        for (int c = lane; c < nv; c += 64) dst[c] = src[c] + pp[c];
    }
}

size_t capdec_tf_decode_workspace_bytes(const capdec_tf_model* m, int B, int S, int max_len, int extra_cap) {
    if (check_model(m) != 0 || B < 0 || S < 1 || max_len < 1 || extra_cap < 0) return 0;
    return carve_decode(m, B, S, max_len, extra_cap ? extra_cap : default_extra_cap(B), nullptr, nullptr);
}

int capdec_event_create(void** ev) {
    CAPDEC_REQUIRE(ev);
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return (int)rc;
    *ev = (void*)e;
    return 0;
}
int capdec_event_destroy(void* ev) { return ev ? (int)hipEventDestroy((hipEvent_t)ev) : 0; }
int capdec_event_record(void* ev, void* stream) {
    CAPDEC_REQUIRE(ev);
    return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
}
int capdec_event_elapsed_ms(void* begin, void* end, float* ms) {
    CAPDEC_REQUIRE(begin && end && ms);
    return (int)hipEventElapsedTime(ms, (hipEvent_t)begin, (hipEvent_t)end);
}

// y = x W^T + b (+ R) over the first rows_dev[0] rows of an Mcap-row slab
static int linear_rows(const float* x, int Mcap, const float* W, const float* b, int E, float* C, const float* R,
                       int flags, const int32_t* rows_dev, hipStream_t s) {
    capdec_gemm_group g = linear_group(x, E, Mcap, W, b, E, C, E, R, E);
    g.ldq = E;
    capdec_gemm_extra ex{};
    ex.rows_dev = rows_dev;
    return capdec_launch_gemm(&g, 1, E, flags, s, &ex);
}

// the same for a [Mcap,K] x [N,K]^T contraction with its own widths
static int linear_rows_nk(const float* x, int Mcap, const float* W, const float* b, int N, int K, float* C, const float* R,
                          const int32_t* rows_dev, hipStream_t s) {
    capdec_gemm_group g = linear_group(x, K, Mcap, W, b, N, C, N, R, N);
    g.ldq = K;
    capdec_gemm_extra ex{};
    ex.rows_dev = rows_dev;
    return capdec_launch_gemm(&g, 1, K, 0, s, &ex);
}

int capdec_tf_greedy_decode(const capdec_tf_model* m, const float* enc, int B, int S, int max_len, int sos, int eos,
                            int flags, int extra_cap, void* workspace, size_t workspace_bytes, int64_t* out_ids,
                            int32_t* out_len, int32_t* out_dirty, const capdec_probe* probe, void* stream) {
    CAPDEC_TRY(check_model(m));
    CAPDEC_REQUIRE(enc && out_ids && out_len && out_dirty && B >= 0 && S >= 1 && S <= 2048 && max_len >= 1);
    CAPDEC_REQUIRE(max_len <= m->max_pos);           // the reference's pe buffer has max_pos rows
    CAPDEC_REQUIRE(sos >= 0 && sos < m->V && extra_cap >= 0);
    CAPDEC_REQUIRE((int64_t)B * max_len < INT_MAX);
    if (B == 0) return 0;
    if (!workspace || !capdec_aligned16(workspace) || !capdec_aligned16(enc)) return CAPDEC_EALIGN;
    if (extra_cap == 0) extra_cap = default_extra_cap(B);
    DecodeWs w;
    if (carve_decode(m, B, S, max_len, extra_cap, workspace, &w) > workspace_bytes) return CAPDEC_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    const int E = m->E, H = m->H, dk = E / H, L = m->L, V = m->V;
    const int Mcap = B + extra_cap;
    const int64_t ldtok = max_len + 1;
    const size_t BE = (size_t)B * E;
    const bool s1 = (S == 1) && !(flags & CAPDEC_DECODE_NO_S1_SHORTCUT);
    const bool s2 = (S == 2) && !(flags & CAPDEC_DECODE_NO_S1_SHORTCUT);
    // long memories: the raw memory is attended to directly (see fold_memory_weights); measured at B=1024, E=512, L=2,
    // len 20: 9.7 against 9.9 ms at 48 rows, 10.0 against 10.7 at 64, 10.5 against 12.3 at 96, 12.7 against 20.1 at 261
    const bool fold = w.fw1 && ((flags & CAPDEC_DECODE_FOLD_MEMORY) ||
                                (S >= CAPDEC_DECODE_FOLD_MIN_ROWS && !(flags & CAPDEC_DECODE_NO_S1_SHORTCUT)));
    CAPDEC_REQUIRE(!(flags & CAPDEC_DECODE_FOLD_MEMORY) || fold);      // forced but unavailable (S < 3 or E not built)

    // all layers of a step in one launch (one or two memory rows, supported shape): the arrival counters of every step start at zero
    const bool fused = (s1 || s2) && w.ctr && !(flags & CAPDEC_DECODE_NO_FUSED_LAYERS);
    const size_t ctr_step = (size_t)w.fused_teams * 32;
    unsigned* fuse_err = fused ? w.ctr + (size_t)max_len * ctr_step : nullptr;
    const size_t ctr_n = fused ? (size_t)max_len * ctr_step + 32 : 0;
    CAPDEC_REQUIRE(ctr_n < INT_MAX && (int64_t)B * (E / 4) < INT_MAX);
    if (!capdec_aligned16(m->emb) || !capdec_aligned16(m->pe)) return CAPDEC_EALIGN;        // rows are read 16 bytes at a time
    // one launch: the bookkeeping arrays, step 0's embedded rows (<SOS> at position 0) and the counters
    hipLaunchKernelGGL(cached_init_kernel, dim3(256), dim3(256), 0, s, w.tok, ldtok, w.done, w.first_pad, out_ids,
                       out_len, out_dirty, w.rowmap, w.nrows, w.fin, B, max_len, sos, w.pcount, w.plist, w.fused_teams,
                       m->emb, m->pe, E, w.x, fused ? w.ctr : nullptr, (int)ctr_n);
    CAPDEC_LAUNCH_CHECK();

    if (s1) {
        // softmax over one key is exactly 1, so the block's output is W_o(W_v enc + b_v) + b_o for every
        // query: one vector per sample and layer, computed once
        // (up to four layers per grouped launch; w.ao, w.hb, w.q, w.y are free until the first step)
        float* tmp[4] = {w.ao, w.hb, w.q, w.y};
        for (int l0 = 0; l0 < L; l0 += 4) {
            const int n = L - l0 < 4 ? L - l0 : 4;
            capdec_gemm_group g[4];
            for (int k = 0; k < n; ++k) {
                const capdec_attn_w& a = m->layers[l0 + k].cross_attn;
                g[k] = linear_group(enc, E, B, a.wv, a.bv, E, tmp[k], E);
                g[k].ldq = E;
            }
            CAPDEC_TRY(capdec_launch_gemm(g, n, E, 0, s));
            for (int k = 0; k < n; ++k) {
                const capdec_attn_w& a = m->layers[l0 + k].cross_attn;
                g[k] = linear_group(tmp[k], E, B, a.wo, a.bo, E, w.cc + (size_t)(l0 + k) * BE, E);
                g[k].ldq = E;
            }
            CAPDEC_TRY(capdec_launch_gemm(g, n, E, 0, s));
        }
    } else if (fold) {
        CAPDEC_TRY(fold_memory_weights(m, w.fw1, w.fb1, w.fw2, w.fb2, s));
    } else {
        CAPDEC_TRY(project_memory(m, enc, B, S, w.ck, w.cv, s));
        // two keys: the block is a per-sample constant plus H logistic gates (w.q / w.ao are free until the first step)
        if (s2) CAPDEC_TRY(pair_memory_constants(m, w.ck, w.cv, B, w.q, w.ao, w.pg, w.pd, w.cc, w.psb, s));
    }

    const int nt = capdec_vocab_tiles(V);
    for (int t = 0; t < max_len; ++t) {
        const int32_t* rows = w.nrows + t;                       // device: B + re-run rows of this step
        const int32_t* map = w.rowmap + (size_t)t * Mcap;        // device: sample * max_len + position per row
        if (fused) {
            capdec_fused_args fa{};
            for (int l = 0; l < L; ++l) {
                const capdec_tf_layer& ly = m->layers[l];
                capdec_fused_layer& f = fa.ly[l];
                f.wq = ly.self_attn.wq; f.bq = ly.self_attn.bq; f.wk = ly.self_attn.wk; f.bk = ly.self_attn.bk;
                f.wv = ly.self_attn.wv; f.bv = ly.self_attn.bv; f.wo = ly.self_attn.wo; f.bo = ly.self_attn.bo;
                f.w1 = ly.ff_w1; f.b1 = ly.ff_b1; f.w2 = ly.ff_w2; f.b2 = ly.ff_b2;
                f.n1g = ly.n1_g; f.n1b = ly.n1_b; f.n2g = ly.n2_g; f.n2b = ly.n2_b; f.n3g = ly.n3_g; f.n3b = ly.n3_b;
                f.cc = w.cc + l * BE;
                if (s2) {
                    f.pg = w.pg + (size_t)l * BE * H;
                    f.pd = w.pd + (size_t)l * BE * H;
                    f.psb = w.psb + (size_t)l * B * H;
                }
                f.kc = w.kc + (size_t)l * BE * max_len;
                f.vc = w.vc + (size_t)l * BE * max_len;
            }
            fa.L = L; fa.B = B; fa.max_len = max_len; fa.Tk = t + 1;
            fa.teams = w.fused_teams;
            fa.cache_bytes = (int)(BE * max_len * sizeof(float));
            fa.cc_div = 1;
            fa.x = w.x; fa.q = w.tq; fa.f_ctx = w.tctx; fa.f_y = w.ty; fa.f_h = w.th; fa.f_z = w.tz;
            fa.s_ctx = w.sctx; fa.s_y = w.sy; fa.s_h = w.sh; fa.s_z = w.sz;
            fa.tok = w.tok; fa.ldtok = ldtok; fa.nrows = rows; fa.done = w.done;
            fa.plist = w.plist; fa.emb = m->emb; fa.pe = m->pe;
            if (t >= 1) {
                // this launch also picks the tokens of step t-1 (no token-pick launch between the steps)
                fa.pval = w.pval; fa.pidx = w.pidx; fa.ntiles = nt; fa.V = V; fa.eos = eos;
                fa.tok_w = w.tok; fa.done_w = w.done; fa.out_ids = out_ids; fa.out_len = out_len; fa.dirty = out_dirty;
                fa.fin = w.fin; fa.nrows_w = w.nrows + t; fa.pcount_w = w.pcount; fa.plist_w = w.plist;
            }
            fa.ctr = w.ctr + (size_t)t * ctr_step; fa.err = fuse_err;
            CAPDEC_TRY(capdec_launch_fused_layers(fa, E, s));
        }
        for (int l = 0; l < L && !fused; ++l) {
            const capdec_tf_layer& ly = m->layers[l];
            float* kc = w.kc + (size_t)l * BE * max_len;
            float* vc = w.vc + (size_t)l * BE * max_len;
            // q, and every row's k/v straight into its cache row
            capdec_gemm_group g[3];
            g[0] = linear_group(w.x, E, Mcap, ly.self_attn.wq, ly.self_attn.bq, E, w.q, E);
            g[1] = linear_group(w.x, E, Mcap, ly.self_attn.wk, ly.self_attn.bk, E, kc, E);
            g[2] = linear_group(w.x, E, Mcap, ly.self_attn.wv, ly.self_attn.bv, E, vc, E);
            g[0].ldq = g[1].ldq = g[2].ldq = E;
            g[1].c_rows = g[2].c_rows = map;
            capdec_gemm_extra ex{};
            ex.rows_dev = rows;
            CAPDEC_TRY(capdec_launch_gemm(g, 3, E, 0, s, &ex));
            CAPDEC_TRY(capdec_launch_attention_rows(w.q, E, kc, E, (int64_t)max_len * E, vc, E, (int64_t)max_len * E,
                                                    w.ao, E, Mcap, H, t + 1, dk, 1, w.tok, ldtok, map, max_len, rows, s));
            CAPDEC_TRY(linear_rows(w.ao, Mcap, ly.self_attn.wo, ly.self_attn.bo, E, w.y, w.x, 0, rows, s));
            if (s1) {
                // norm1 and norm2 in one pass: the cross-attention sub-layer is the per-sample constant cc
                CAPDEC_TRY(capdec_launch_layernorm_rows(w.y, nullptr, 1, nullptr, ly.n1_g, ly.n1_b, 1e-5f, Mcap, E, w.x2,
                                                        rows, s, ly.n2_g, ly.n2_b, w.cc + l * BE, max_len, map));
            } else if (s2) {
                // norm1, the two-key cross-attention block and norm2 in one pass over the row
                CAPDEC_TRY(capdec_launch_layernorm_pair_rows(w.y, ly.n1_g, ly.n1_b, ly.n2_g, ly.n2_b, 1e-5f, Mcap, E, w.x2,
                                                             rows, map, max_len, w.pg + (size_t)l * BE * H,
                                                             w.pd + (size_t)l * BE * H, w.cc + l * BE,
                                                             w.psb + (size_t)l * B * H, H, 1.0f / sqrtf((float)dk), s));
            } else if (fold) {
                const size_t HEE = (size_t)H * E * E;
                CAPDEC_TRY(capdec_launch_layernorm_rows(w.y, nullptr, 1, nullptr, ly.n1_g, ly.n1_b, 1e-5f, Mcap, E, w.x1,
                                                        rows, s));
                CAPDEC_TRY(linear_rows_nk(w.x1, Mcap, w.fw1 + l * HEE, w.fb1 + (size_t)l * H * E, H * E, E, w.fq, nullptr,
                                          rows, s));
                CAPDEC_TRY(capdec_launch_attention_memory_rows(w.fq, (int64_t)H * E, enc, S, E, H, 1.0f / sqrtf((float)dk),
                                                               w.fc, (int64_t)H * E, Mcap, map, max_len, rows, s));
                CAPDEC_TRY(linear_rows_nk(w.fc, Mcap, w.fw2 + l * HEE, w.fb2 + (size_t)l * E, E, H * E, w.y, w.x1, rows, s));
                CAPDEC_TRY(capdec_launch_layernorm_rows(w.y, nullptr, 1, nullptr, ly.n2_g, ly.n2_b, 1e-5f, Mcap, E,
                                                        w.x2, rows, s));
            } else {
                CAPDEC_TRY(capdec_launch_layernorm_rows(w.y, nullptr, 1, nullptr, ly.n1_g, ly.n1_b, 1e-5f, Mcap, E, w.x1,
                                                        rows, s));
                CAPDEC_TRY(linear_rows(w.x1, Mcap, ly.cross_attn.wq, ly.cross_attn.bq, E, w.q, nullptr, 0, rows, s));
                CAPDEC_TRY(capdec_launch_attention_rows(w.q, E, w.ck + (size_t)l * BE * S, E, (int64_t)S * E,
                                                        w.cv + (size_t)l * BE * S, E, (int64_t)S * E, w.ao, E, Mcap, H, S,
                                                        dk, 0, nullptr, 0, map, max_len, rows, s));
                CAPDEC_TRY(linear_rows(w.ao, Mcap, ly.cross_attn.wo, ly.cross_attn.bo, E, w.y, w.x1, 0, rows, s));
                CAPDEC_TRY(capdec_launch_layernorm_rows(w.y, nullptr, 1, nullptr, ly.n2_g, ly.n2_b, 1e-5f, Mcap, E,
                                                        w.x2, rows, s));
            }
            CAPDEC_TRY(linear_rows(w.x2, Mcap, ly.ff_w1, ly.ff_b1, E, w.hb, nullptr, CAPDEC_GEMM_RELU, rows, s));
            CAPDEC_TRY(linear_rows(w.hb, Mcap, ly.ff_w2, ly.ff_b2, E, w.y, w.x2, 0, rows, s));
            CAPDEC_TRY(capdec_launch_layernorm_rows(w.y, nullptr, 1, nullptr, ly.n3_g, ly.n3_b, 1e-5f, Mcap, E, w.x,
                                                    rows, s));
        }
        // only the B new rows predict a token
        const bool probed = probe && t < probe->n && probe->begin[t] && probe->end[t];
        if (probed) CAPDEC_TRY((int)hipEventRecord((hipEvent_t)probe->begin[t], s));
        CAPDEC_TRY(capdec_launch_vocab_argmax_rows(w.x, E, m->fc2_w, m->fc2_b, B, V, E, w.pval, w.pidx, rows, s));
        if (probed) CAPDEC_TRY((int)hipEventRecord((hipEvent_t)probe->end[t], s));
        const bool last = (t + 1 == max_len);
        if (fused && !last) continue;              // the next step's launch picks this step's tokens
        hipLaunchKernelGGL(cached_advance_kernel, dim3((B + 3) / 4), dim3(256), 0, s, w.pval, w.pidx, B, nt, t, max_len,
                           eos, w.tok, ldtok, w.done, w.first_pad, out_ids, out_len, out_dirty, m->emb, m->pe, E, V,
                           last ? nullptr : w.x, last ? nullptr : w.rowmap + (size_t)(t + 1) * Mcap,
                           last ? nullptr : w.nrows + t + 1, extra_cap, w.fin, fused ? w.pcount : nullptr, w.plist, fuse_err);
        CAPDEC_LAUNCH_CHECK();
    }
    return 0;
}

// =================================================================================================
// Full-prefix paths: exact greedy decode and teacher-forced logits
// =================================================================================================
struct FullWs {
    FullBufs f;
    float *ck, *cv;
    float* pval; int32_t* pidx;
    int32_t *tok, *done;
};

static size_t carve_full(const capdec_tf_model* m, int B, int S, int T, bool decode, void* base, FullWs* w) {
    Carver c(base);
    const size_t n = (size_t)B * T * m->E;
    FullWs d;
    d.f.X = c.take<float>(n); d.f.Y = c.take<float>(n); d.f.X1 = c.take<float>(n); d.f.X2 = c.take<float>(n);
    d.f.Qb = c.take<float>(n); d.f.Kb = c.take<float>(n); d.f.Vb = c.take<float>(n); d.f.AO = c.take<float>(n);
    d.f.Hb = c.take<float>(n);
    d.ck = c.take<float>((size_t)m->L * B * S * m->E);
    d.cv = c.take<float>((size_t)m->L * B * S * m->E);
    const size_t nt = (size_t)capdec_vocab_tiles(m->V);
    d.pval = c.take<float>(decode ? capdec_part_elems(B, (int)nt) : 1);
    d.pidx = c.take<int32_t>(decode ? capdec_part_elems(B, (int)nt) : 1);
    d.tok = c.take<int32_t>((size_t)B * (T + 1));
    d.done = c.take<int32_t>((size_t)B);
    if (w) *w = d;
    return c.off;
}

size_t capdec_tf_exact_workspace_bytes(const capdec_tf_model* m, int B, int S, int max_len) {
    if (check_model(m) != 0 || B < 0 || S < 1 || max_len < 1) return 0;
    return carve_full(m, B, S, max_len, true, nullptr, nullptr);
}

int capdec_tf_greedy_decode_exact(const capdec_tf_model* m, const float* enc, int B, int S, int max_len, int sos,
                                  int eos, void* workspace, size_t workspace_bytes, int64_t* out_ids, int32_t* out_len,
                                  void* stream) {
    CAPDEC_TRY(check_model(m));
    CAPDEC_REQUIRE(enc && out_ids && out_len && B >= 0 && S >= 1 && S <= 2048 && max_len >= 1);
    CAPDEC_REQUIRE(max_len <= m->max_pos && sos >= 0 && sos < m->V);
    if (B == 0) return 0;
    if (!workspace || !capdec_aligned16(workspace) || !capdec_aligned16(enc)) return CAPDEC_EALIGN;
    FullWs w;
    if (carve_full(m, B, S, max_len, true, workspace, &w) > workspace_bytes) return CAPDEC_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    const int E = m->E, V = m->V;
    const int64_t ldtok = max_len + 1;
    hipLaunchKernelGGL(greedy_init_kernel, dim3(256), dim3(256), 0, s, w.tok, ldtok, w.done, out_ids, out_len,
                       (int32_t*)nullptr, B, max_len, sos);
    CAPDEC_LAUNCH_CHECK();
    CAPDEC_TRY(project_memory(m, enc, B, S, w.ck, w.cv, s));
    const int nt = capdec_vocab_tiles(V);
    for (int t = 0; t < max_len; ++t) {
        const int T = t + 1;   // the whole prefix, every step (transformer.py:152)
        CAPDEC_TRY(capdec_embed_pe(w.tok, 1, ldtok, m->emb, m->pe, 0, B, T, E, V, w.f.X, stream));
        CAPDEC_TRY(decoder_stack_full(m, w.f, w.ck, w.cv, B, S, T, w.tok, ldtok, s));
        // take_last: row (b, T-1) of X
        CAPDEC_TRY(capdec_vocab_argmax_partial(w.f.X + (size_t)(T - 1) * E, (int64_t)T * E, m->fc2_w, m->fc2_b, B, V, E,
                                               w.pval, w.pidx, stream));
        hipLaunchKernelGGL(greedy_advance_kernel, dim3((B + 3) / 4), dim3(256), 0, s, w.pval, w.pidx, B, nt, t, max_len,
                           eos, w.tok, ldtok, w.done, out_ids, out_len, (int32_t*)nullptr, m->emb, m->pe, E, V,
                           (float*)nullptr);
        CAPDEC_LAUNCH_CHECK();
    }
    return 0;
}

size_t capdec_tf_forward_workspace_bytes(const capdec_tf_model* m, int B, int S, int T) {
    if (check_model(m) != 0 || B < 0 || S < 1 || T < 1) return 0;
    return carve_full(m, B, S, T, false, nullptr, nullptr);
}

__global__ void ids_to_i32_kernel(const int64_t* __restrict__ in, int32_t* __restrict__ out, int B, int T, int64_t ldo) {
    const int n = B * T;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int b = idx / T, i = idx - b * T;
        out[(int64_t)b * ldo + i] = (int32_t)in[idx];
    }
}

int capdec_tf_forward_logits(const capdec_tf_model* m, const float* enc, const int64_t* ids, int B, int S, int T,
                             int take_last, void* workspace, size_t workspace_bytes, float* logits, void* stream) {
    CAPDEC_TRY(check_model(m));
    CAPDEC_REQUIRE(enc && ids && logits && B >= 0 && S >= 1 && S <= 2048 && T >= 1 && T <= m->max_pos && T <= 2048);
    if (B == 0) return 0;
    if (!workspace || !capdec_aligned16(workspace) || !capdec_aligned16(enc) || !capdec_aligned16(logits))
        return CAPDEC_EALIGN;
    FullWs w;
    if (carve_full(m, B, S, T, false, workspace, &w) > workspace_bytes) return CAPDEC_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    const int E = m->E, V = m->V;
    const int64_t ldtok = T + 1;
    hipLaunchKernelGGL(ids_to_i32_kernel, dim3(256), dim3(256), 0, s, ids, w.tok, B, T, ldtok);
    CAPDEC_LAUNCH_CHECK();
    CAPDEC_TRY(capdec_embed_pe(ids, 0, T, m->emb, m->pe, 0, B, T, E, V, w.f.X, stream));
    CAPDEC_TRY(project_memory(m, enc, B, S, w.ck, w.cv, s));
    CAPDEC_TRY(decoder_stack_full(m, w.f, w.ck, w.cv, B, S, T, w.tok, ldtok, s));
    if (take_last)
        return linear(w.f.X + (size_t)(T - 1) * E, (int64_t)T * E, B, m->fc2_w, m->fc2_b, V, E, logits, V, nullptr, 0, 0, s);
    return linear(w.f.X, E, B * T, m->fc2_w, m->fc2_b, V, E, logits, V, nullptr, 0, 0, s);
}

// =================================================================================================
// CNN-RNN greedy decode
// =================================================================================================
struct RnnWs {
    float *x, *h[2], *c[2];
    float* pval; int32_t* pidx;
    int32_t *tok, *done;
    int32_t *fin, *live;     // [max_len], [max_len + 1]: see greedy_advance_kernel
};

static size_t carve_rnn(const capdec_rnn_model* m, int B, int max_len, void* base, RnnWs* w) {
    Carver c(base);
    RnnWs d;
    d.x = c.take<float>((size_t)B * m->E);
    for (int n = 0; n < 2; ++n) {
        d.h[n] = c.take<float>((size_t)B * m->Hd);
        d.c[n] = c.take<float>((size_t)B * m->Hd);
    }
    const size_t nt = (size_t)capdec_vocab_tiles(m->V);
    d.pval = c.take<float>(capdec_part_elems(B, (int)nt));
    d.pidx = c.take<int32_t>(capdec_part_elems(B, (int)nt));
    d.tok = c.take<int32_t>((size_t)B * (max_len + 1));
    d.done = c.take<int32_t>((size_t)B);
    d.fin = c.take<int32_t>((size_t)max_len);
    d.live = c.take<int32_t>((size_t)max_len + 1);
    if (w) *w = d;
    return c.off;
}

static int check_rnn(const capdec_rnn_model* m) {
    CAPDEC_REQUIRE(m && m->E >= 4 && m->E % 4 == 0 && m->Hd >= 4 && m->Hd % 4 == 0 && m->V >= 1);
    CAPDEC_REQUIRE(m->emb && m->w_ih && m->w_hh && m->b_ih && m->b_hh && m->fc2_w && m->fc2_b);
    return 0;
}

size_t capdec_rnn_decode_workspace_bytes(const capdec_rnn_model* m, int B, int max_len) {
    if (check_rnn(m) != 0 || B < 0 || max_len < 1) return 0;
    return carve_rnn(m, B, max_len, nullptr, nullptr);
}

int capdec_rnn_greedy_decode(const capdec_rnn_model* m, const float* x0, int B, int max_len, int sos, int eos,
                             void* workspace, size_t workspace_bytes, int64_t* out_ids, int32_t* out_len,
                             void* stream) {
    CAPDEC_TRY(check_rnn(m));
    CAPDEC_REQUIRE(x0 && out_ids && out_len && B >= 0 && max_len >= 1 && sos >= 0 && sos < m->V);
    if (B == 0) return 0;
    if (!workspace || !capdec_aligned16(workspace)) return CAPDEC_EALIGN;
    RnnWs w;
    if (carve_rnn(m, B, max_len, workspace, &w) > workspace_bytes) return CAPDEC_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    const int E = m->E, Hd = m->Hd, V = m->V;
    const int64_t ldtok = max_len + 1;
    hipLaunchKernelGGL(greedy_init_kernel, dim3(256), dim3(256), 0, s, w.tok, ldtok, w.done, out_ids, out_len,
                       (int32_t*)nullptr, B, max_len, sos, w.fin, w.live);
    CAPDEC_LAUNCH_CHECK();
    // the image is time-step 0 of the LSTM, from the zero state (cnnrnn_model.py:75)
    CAPDEC_TRY(capdec_lstm_step(x0, E, nullptr, 0, nullptr, m->w_ih, m->w_hh, m->b_ih, m->b_hh, B, E, Hd, w.h[0], Hd,
                                w.c[0], nullptr, stream));
    CAPDEC_TRY(capdec_embed_pe(w.tok, 1, ldtok, m->emb, nullptr, 0, B, 1, E, V, w.x, stream));
    const int nt = capdec_vocab_tiles(V);
    int cur = 0;
    for (int t = 0; t < max_len; ++t) {
        // (w.live[t]: B, or zero once every caption has finished; every kernel of the step reads it first)
        CAPDEC_TRY(capdec_launch_lstm_step(w.x, E, w.h[cur], Hd, w.c[cur], m->w_ih, m->w_hh, m->b_ih, m->b_hh, B, E, Hd,
                                           w.h[cur ^ 1], Hd, w.c[cur ^ 1], nullptr, w.live + t, s));
        cur ^= 1;
        CAPDEC_TRY(capdec_launch_vocab_argmax_rows(w.h[cur], Hd, m->fc2_w, m->fc2_b, B, V, Hd, w.pval, w.pidx, w.live + t, s));
        const bool last = (t + 1 == max_len);
        hipLaunchKernelGGL(greedy_advance_kernel, dim3((B + 3) / 4), dim3(256), 0, s, w.pval, w.pidx, B, nt, t, max_len,
                           eos, w.tok, ldtok, w.done, out_ids, out_len, (int32_t*)nullptr, m->emb,
                           (const float*)nullptr, E, V, last ? nullptr : w.x, (int64_t)0, w.fin,
                           last ? nullptr : w.live + t + 1);
        CAPDEC_LAUNCH_CHECK();
    }
    return 0;
}

// =================================================================================================
// Beam search (full-prefix recompute of all B*W hypotheses every step, like the reference)
// =================================================================================================
struct BeamBufs {
    float *logits, *top_lp;
    int32_t *top_id, *parent;
    int32_t* unfinished;      // [max_len]: samples that received a token other than <EOS> at step t
    BeamState st[2];
};

static void carve_beam_bufs(Carver& c, int N, int W, int V, int max_len, BeamBufs* d) {
    d->logits = c.take<float>((size_t)N * V);
    d->top_lp = c.take<float>((size_t)N * W);
    d->top_id = c.take<int32_t>((size_t)N * W);
    d->parent = c.take<int32_t>((size_t)N);
    d->unfinished = c.take<int32_t>((size_t)max_len);
    for (int n = 0; n < 2; ++n) {
        d->st[n].tok = c.take<int32_t>((size_t)N * (max_len + 1));
        d->st[n].score = c.take<float>((size_t)N);
        d->st[n].done = c.take<int32_t>((size_t)N);
    }
}

static int beam_pick(const BeamBufs& d, int t, int B, int W, int V, int eos, int64_t ldtok, hipStream_t s,
                     int32_t* pad_seen = nullptr) {
    const int N = B * W;
    const bool vec = V % 4 == 0 && capdec_aligned16(d.logits);
    // up to 1024 hypothesis rows: a workgroup per row (one wavefront per row leaves most of the chip idle there)
    const bool by_row = N <= 1024 && V >= 1024;
#define CAPDEC_TOPK(CAP_)                                                                                               \
    do {                                                                                                                \
        if (by_row && vec) hipLaunchKernelGGL((beam_topk_row_kernel<CAP_, true>), dim3(N), dim3(256), 0, s, d.logits,   \
                                              (int64_t)V, N, V, W, d.top_lp, d.top_id);                                 \
        else if (by_row) hipLaunchKernelGGL((beam_topk_row_kernel<CAP_, false>), dim3(N), dim3(256), 0, s, d.logits,    \
                                            (int64_t)V, N, V, W, d.top_lp, d.top_id);                                   \
        else if (vec) hipLaunchKernelGGL((beam_topk_kernel<CAP_, true>), dim3((N + 3) / 4), dim3(256), 0, s, d.logits,  \
                                    (int64_t)V, N, V, W, d.top_lp, d.top_id);                                           \
        else hipLaunchKernelGGL((beam_topk_kernel<CAP_, false>), dim3((N + 3) / 4), dim3(256), 0, s, d.logits,          \
                                (int64_t)V, N, V, W, d.top_lp, d.top_id);                                               \
    } while (0)
    if (W <= 2) CAPDEC_TOPK(2);
    else if (W <= 4) CAPDEC_TOPK(4);
    else CAPDEC_TOPK(BEAM_MAX_W);
#undef CAPDEC_TOPK
    CAPDEC_LAUNCH_CHECK();
    hipLaunchKernelGGL(beam_select_kernel, dim3((B + 63) / 64), dim3(64), 0, s, t, B, W, eos, d.st[t & 1],
                       d.st[(t + 1) & 1], ldtok, d.top_lp, d.top_id, d.parent, d.unfinished, pad_seen);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

struct TfBeamWs {
    FullWs f;
    float* encx;
    BeamBufs b;
};

static size_t carve_tf_beam(const capdec_tf_model* m, int B, int W, int S, int max_len, void* base, TfBeamWs* w) {
    const int N = B * W;
    TfBeamWs d;
    const size_t head = carve_full(m, N, S, max_len, false, base, &d.f);
    Carver c(base ? (char*)base + head : nullptr);
    d.encx = c.take<float>((size_t)N * S * m->E);
    carve_beam_bufs(c, N, W, m->V, max_len, &d.b);
    if (w) *w = d;
    return head + c.off;
}

int capdec_topk_rows(const float* x, int rows, int n, int k, float* out_val, int32_t* out_idx, void* stream) {
    CAPDEC_REQUIRE(x && out_val && out_idx && rows >= 0 && n >= 1 && n <= BEAM_MAX_W * BEAM_MAX_W && k >= 1 && k <= n &&
                   k <= BEAM_MAX_W && k * 64 > n);
    if (rows == 0) return 0;
    hipLaunchKernelGGL(topk_rows_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, rows, n, k, out_val,
                       out_idx);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

size_t capdec_tf_beam_workspace_bytes(const capdec_tf_model* m, int B, int W, int S, int max_len) {
    if (check_model(m) != 0 || B < 0 || W < 1 || W > BEAM_MAX_W || S < 1 || max_len < 1) return 0;
    return carve_tf_beam(m, B, W, S, max_len, nullptr, nullptr);
}

int capdec_tf_beam_search(const capdec_tf_model* m, const float* enc, int B, int W, int S, int max_len, int sos, int eos,
                          int t_begin, int t_end, void* workspace, size_t workspace_bytes, int64_t* out_ids,
                          int32_t* out_len, float* out_scores, int32_t* n_unfinished, void* stream) {
    CAPDEC_TRY(check_model(m));
    CAPDEC_REQUIRE(enc && out_ids && out_len && n_unfinished && B >= 0 && W >= 1 && W <= BEAM_MAX_W && S >= 1 && S <= 2048);
    CAPDEC_REQUIRE(max_len >= 1 && max_len <= m->max_pos && sos >= 0 && sos < m->V && m->V >= W);
    CAPDEC_REQUIRE(0 <= t_begin && t_begin < t_end && t_end <= max_len);
    if (B == 0) return 0;
    if (!workspace || !capdec_aligned16(workspace) || !capdec_aligned16(enc)) return CAPDEC_EALIGN;
    TfBeamWs w;
    if (carve_tf_beam(m, B, W, S, max_len, workspace, &w) > workspace_bytes) return CAPDEC_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    const int E = m->E, V = m->V, N = B * W;
    const int64_t ldtok = max_len + 1;
    if (t_begin == 0) {
        hipLaunchKernelGGL(beam_init_kernel, dim3(64), dim3(256), 0, s, w.b.st[0], ldtok, N, sos, w.b.unfinished, max_len);
        CAPDEC_LAUNCH_CHECK();
        // every hypothesis of a sample attends to the sample's memory rows (cnnattn_model.py:65-66)
        hipLaunchKernelGGL(gather_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, s, enc, (const int32_t*)nullptr, W, N,
                           S * E, w.encx);
        CAPDEC_LAUNCH_CHECK();
        CAPDEC_TRY(project_memory(m, w.encx, N, S, w.f.ck, w.f.cv, s));
    }
    for (int t = t_begin; t < t_end; ++t) {
        const int T = t + 1;
        const BeamState& cur = w.b.st[t & 1];
        CAPDEC_TRY(capdec_embed_pe(cur.tok, 1, ldtok, m->emb, m->pe, 0, N, T, E, V, w.f.f.X, stream));
        CAPDEC_TRY(decoder_stack_full(m, w.f.f, w.f.ck, w.f.cv, N, S, T, cur.tok, ldtok, s));
        CAPDEC_TRY(linear(w.f.f.X + (size_t)(T - 1) * E, (int64_t)T * E, N, m->fc2_w, m->fc2_b, V, E, w.b.logits, V,
                          nullptr, 0, 0, s));
        CAPDEC_TRY(beam_pick(w.b, t, B, W, V, eos, ldtok, s));
    }
    hipLaunchKernelGGL(beam_result_kernel, dim3((B + 63) / 64), dim3(64), 0, s, w.b.st[t_end & 1], ldtok, B, W, t_end,
                       max_len, eos, out_ids, out_len, out_scores, w.b.unfinished, n_unfinished);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

// ---- KV-cached beam search: one new position per hypothesis and step; the caches are re-gathered after every
// selection so that each surviving hypothesis owns its parent's keys and values.  The reference lets a <PAD> query row
// attend uniformly over the CURRENT length, which changes that position's output at every later step.  Where the
// one-launch layers run (decode_fused.hip) the <PAD> positions of a hypothesis' prefix follow along as side rows and
// only a team with more of them than side rows reports its samples in pad_seen; the launch-per-operation layers report
// every sample in which a <PAD> token is generated.  Reported samples must be redone with the full-prefix form.
struct TfBeamCacheWs {
    float *x, *y, *x2, *x1, *q, *ao, *hb;     // [N][E]
    float *cc, *ck, *cv, *encx;
    float *pg, *pd, *psb;                     // S == 2: G, D [L][B][H][E] and sb [L][B][H] of the two-row closed form
    float* cache[2];                          // each [L][2][N][max_len][E]
    // all layers of a step in one launch (decode_fused.hip; S <= 2 and a supported shape, otherwise null): exchange
    // blocks, the hand-off flags of every step, and what that kernel reads of a greedy decode's state - a row count, the
    // finished flags (all zero here) and the <PAD> lists (rebuilt every step from the hypotheses' prefixes)
    float *tq, *tctx, *ty, *th, *tz, *sctx, *sy, *sh, *sz;
    unsigned* ctr;
    int32_t *fz_nrows, *fz_done, *fz_plist;
    int fused_teams;
    BeamBufs b;
};

CAPDEC_HOOK_VAR(g_beam_no_fused);          // tuning build: 1 keeps the launch-per-operation layers

__global__ void beam_fused_init_kernel(int32_t* __restrict__ nrows, int32_t* __restrict__ done, int32_t* __restrict__ plist,
                                       int N, int nlist) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (g == 0) nrows[0] = N;
    for (int i = g; i < N; i += nth) done[i] = 0;
    for (int i = g; i < nlist; i += nth) plist[i] = -1;
}
// The <PAD> positions in the prefixes of a team's 16 hypothesis rows, for the side rows of the one-launch layers (which
// refresh those positions' layer-1 keys and values for the current length: the reference lets a <PAD> query row attend
// uniformly over all positions so far).  Rebuilt every step, because the selection permutes the hypotheses.  One
// wavefront per team; row r's entries follow those of the rows before it; a team with more than 16 such positions
// reports the samples of the rows that did not fit (pad_seen: redone with the full-prefix form).
__global__ __launch_bounds__(64) void beam_plist_kernel(const int32_t* __restrict__ tok, int64_t ldtok, int N, int t,
                                                        int max_len, int W, int32_t* __restrict__ plist,
                                                        int32_t* __restrict__ pad_seen) {
    const int team = blockIdx.x, lane = threadIdx.x;
    const int n = team * CAPDEC_FUSED_ROWS + (lane & 15);
    const bool act = lane < CAPDEC_FUSED_ROWS && n < N;
    int cnt = 0;
    if (act)
        for (int p = 1; p < t; ++p) cnt += tok[(int64_t)n * ldtok + p] == 0;
    int off = cnt;                                   // inclusive prefix sum over the 16 rows
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const int o = __shfl_up(off, d, 64);
        if ((lane & 15) >= d) off += o;
    }
    off -= cnt;
    int32_t* pl = plist + (size_t)team * CAPDEC_FUSED_SIDE_ROWS;
    if (lane < CAPDEC_FUSED_SIDE_ROWS) pl[lane] = -1;
    __syncthreads();
    if (act && cnt) {
        int k = off;
        for (int p = 1; p < t; ++p)
            if (tok[(int64_t)n * ldtok + p] == 0) {
                if (k < CAPDEC_FUSED_SIDE_ROWS) pl[k] = n * max_len + p;
                else pad_seen[n / W] = 1;
                ++k;
            }
    }
}
// a wait of the one-launch layers that ran out: every sample is reported for the full-prefix form
__global__ void beam_fused_err_kernel(const unsigned* __restrict__ err, int32_t* __restrict__ pad_seen, int B) {
    if (err[0] == 0) return;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) pad_seen[b] = 1;
}

static size_t carve_tf_beam_cached(const capdec_tf_model* m, int B, int W, int S, int max_len, void* base,
                                   TfBeamCacheWs* w) {
    const int N = B * W;
    const size_t NE = (size_t)N * m->E;
    Carver c(base);
    TfBeamCacheWs d;
    d.x = c.take<float>(NE); d.y = c.take<float>(NE); d.x2 = c.take<float>(NE); d.x1 = c.take<float>(NE);
    d.q = c.take<float>(NE); d.ao = c.take<float>(NE); d.hb = c.take<float>(NE);
    // one and two memory rows: per-sample constants shared by a sample's W hypotheses; longer memories: the projected
    // memory once per hypothesis
    const size_t BE = (size_t)B * m->E;
    d.cc = c.take<float>(S <= 2 ? (size_t)m->L * BE : 1);
    d.ck = c.take<float>(S == 1 ? 1 : (S == 2 ? (size_t)m->L * BE * 2 : (size_t)m->L * NE * S));
    d.cv = c.take<float>(S == 1 ? 1 : (S == 2 ? (size_t)m->L * BE * 2 : (size_t)m->L * NE * S));
    d.encx = c.take<float>(S <= 2 ? 1 : NE * S);
    d.pg = c.take<float>(S == 2 ? (size_t)m->L * BE * m->H : 1);
    d.pd = c.take<float>(S == 2 ? (size_t)m->L * BE * m->H : 1);
    d.psb = c.take<float>(S == 2 ? (size_t)m->L * B * m->H : 1);
    for (int n = 0; n < 2; ++n) d.cache[n] = c.take<float>((size_t)m->L * 2 * NE * max_len);
    d.tq = d.tctx = d.ty = d.th = d.tz = d.sctx = d.sy = d.sh = d.sz = nullptr;
    d.ctr = nullptr;
    d.fz_nrows = d.fz_done = d.fz_plist = nullptr;
    d.fused_teams = 0;
    if (S <= 2 && capdec_fused_layers_supported(m->E, m->H, m->L, N, max_len)) {
        d.fused_teams = (N + CAPDEC_FUSED_ROWS - 1) / CAPDEC_FUSED_ROWS;
        const size_t TE = (size_t)d.fused_teams * CAPDEC_FUSED_ROWS * m->E;
        d.tq = c.take<float>(TE); d.tctx = c.take<float>(TE); d.ty = c.take<float>(TE);
        d.th = c.take<float>(TE); d.tz = c.take<float>(TE);
        const size_t SE = (size_t)d.fused_teams * CAPDEC_FUSED_SIDE_ROWS * m->E;
        d.sctx = c.take<float>(SE); d.sy = c.take<float>(SE); d.sh = c.take<float>(SE); d.sz = c.take<float>(SE);
        d.ctr = c.take<unsigned>(((size_t)max_len * d.fused_teams + 1) * 32);
        d.fz_nrows = c.take<int32_t>(4);
        d.fz_done = c.take<int32_t>((size_t)N);
        d.fz_plist = c.take<int32_t>((size_t)d.fused_teams * CAPDEC_FUSED_SIDE_ROWS);
    }
    carve_beam_bufs(c, N, W, m->V, max_len, &d.b);
    if (w) *w = d;
    return c.off;
}

size_t capdec_tf_beam_cached_workspace_bytes(const capdec_tf_model* m, int B, int W, int S, int max_len) {
    if (check_model(m) != 0 || B < 0 || W < 1 || W > BEAM_MAX_W || S < 1 || max_len < 1) return 0;
    return carve_tf_beam_cached(m, B, W, S, max_len, nullptr, nullptr);
}

int capdec_tf_beam_search_cached(const capdec_tf_model* m, const float* enc, int B, int W, int S, int max_len, int sos,
                                 int eos, int t_begin, int t_end, void* workspace, size_t workspace_bytes,
                                 int64_t* out_ids, int32_t* out_len, float* out_scores, int32_t* n_unfinished,
                                 int32_t* pad_seen, void* stream) {
    CAPDEC_TRY(check_model(m));
    CAPDEC_REQUIRE(enc && out_ids && out_len && n_unfinished && pad_seen && B >= 0 && W >= 1 && W <= BEAM_MAX_W && S >= 1 &&
                   S <= 2048);
    CAPDEC_REQUIRE(max_len >= 1 && max_len <= m->max_pos && sos >= 0 && sos < m->V && m->V >= W);
    CAPDEC_REQUIRE(0 <= t_begin && t_begin < t_end && t_end <= max_len);
    if (B == 0) return 0;
    if (!workspace || !capdec_aligned16(workspace) || !capdec_aligned16(enc)) return CAPDEC_EALIGN;
    TfBeamCacheWs w;
    if (carve_tf_beam_cached(m, B, W, S, max_len, workspace, &w) > workspace_bytes) return CAPDEC_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    const int E = m->E, V = m->V, H = m->H, L = m->L, dk = E / H, N = B * W;
    const size_t NE = (size_t)N * E, BE = (size_t)B * E;
    const int64_t ldtok = max_len + 1, cstride = (int64_t)max_len * E;
    const bool s1 = S == 1, s2 = S == 2;
    const bool fused = w.ctr && !g_beam_no_fused;
    const size_t ctr_step = (size_t)w.fused_teams * 32;
    unsigned* fuse_err = fused ? w.ctr + (size_t)max_len * ctr_step : nullptr;
    if (t_begin == 0) {
        hipLaunchKernelGGL(beam_init_kernel, dim3(64), dim3(256), 0, s, w.b.st[0], ldtok, N, sos, w.b.unfinished, max_len);
        CAPDEC_LAUNCH_CHECK();
        if (fused) {
            CAPDEC_TRY((int)hipMemsetAsync(w.ctr, 0, ((size_t)max_len * ctr_step + 32) * sizeof(unsigned), s));
            hipLaunchKernelGGL(beam_fused_init_kernel, dim3(64), dim3(256), 0, s, w.fz_nrows, w.fz_done, w.fz_plist, N,
                               w.fused_teams * CAPDEC_FUSED_SIDE_ROWS);
            CAPDEC_LAUNCH_CHECK();
        }
        if (hipMemsetAsync(pad_seen, 0, (size_t)B * sizeof(int32_t), s) != hipSuccess) return (int)hipGetLastError();
        if (s1) {
            for (int l = 0; l < L; ++l) {
                const capdec_attn_w& a = m->layers[l].cross_attn;
                CAPDEC_TRY(linear(enc, E, B, a.wv, a.bv, E, E, w.ao, E, nullptr, 0, 0, s));
                CAPDEC_TRY(linear(w.ao, E, B, a.wo, a.bo, E, E, w.cc + l * BE, E, nullptr, 0, 0, s));
            }
        } else if (s2) {
            // two keys: a per-sample constant plus H logistic gates (see pair_memory_constants); w.q / w.ao are free
            // until the first step
            CAPDEC_TRY(project_memory(m, enc, B, 2, w.ck, w.cv, s));
            CAPDEC_TRY(pair_memory_constants(m, w.ck, w.cv, B, w.q, w.ao, w.pg, w.pd, w.cc, w.psb, s));
        } else {
            hipLaunchKernelGGL(gather_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, s, enc, (const int32_t*)nullptr, W, N,
                               S * E, w.encx);
            CAPDEC_LAUNCH_CHECK();
            CAPDEC_TRY(project_memory(m, w.encx, N, S, w.ck, w.cv, s));
        }
        CAPDEC_TRY(capdec_embed_pe(w.b.st[0].tok, 1, ldtok, m->emb, m->pe, 0, N, 1, E, V, w.x, stream));
    }
    for (int t = t_begin; t < t_end; ++t) {
        float* cur = w.cache[t & 1];
        float* nxt = w.cache[(t + 1) & 1];
        if (fused) {
            // every layer of the step as one launch over the N hypothesis rows (row n: the constants of sample n / W)
            capdec_fused_args fa{};
            for (int l = 0; l < L; ++l) {
                const capdec_tf_layer& ly = m->layers[l];
                capdec_fused_layer& f = fa.ly[l];
                f.wq = ly.self_attn.wq; f.bq = ly.self_attn.bq; f.wk = ly.self_attn.wk; f.bk = ly.self_attn.bk;
                f.wv = ly.self_attn.wv; f.bv = ly.self_attn.bv; f.wo = ly.self_attn.wo; f.bo = ly.self_attn.bo;
                f.w1 = ly.ff_w1; f.b1 = ly.ff_b1; f.w2 = ly.ff_w2; f.b2 = ly.ff_b2;
                f.n1g = ly.n1_g; f.n1b = ly.n1_b; f.n2g = ly.n2_g; f.n2b = ly.n2_b; f.n3g = ly.n3_g; f.n3b = ly.n3_b;
                f.cc = w.cc + l * BE;
                if (s2) {
                    f.pg = w.pg + (size_t)l * BE * H;
                    f.pd = w.pd + (size_t)l * BE * H;
                    f.psb = w.psb + (size_t)l * B * H;
                }
                f.kc = cur + (size_t)(2 * l) * NE * max_len;
                f.vc = cur + (size_t)(2 * l + 1) * NE * max_len;
            }
            fa.L = L; fa.B = N; fa.max_len = max_len; fa.Tk = t + 1;
            fa.teams = w.fused_teams;
            fa.cache_bytes = (int)(NE * max_len * sizeof(float));
            fa.cc_div = W;
            fa.x = w.x; fa.q = w.tq; fa.f_ctx = w.tctx; fa.f_y = w.ty; fa.f_h = w.th; fa.f_z = w.tz;
            fa.s_ctx = w.sctx; fa.s_y = w.sy; fa.s_h = w.sh; fa.s_z = w.sz;
            fa.tok = w.b.st[t & 1].tok; fa.ldtok = ldtok; fa.nrows = w.fz_nrows; fa.done = w.fz_done;
            fa.plist = w.fz_plist; fa.emb = m->emb; fa.pe = m->pe;
            fa.ctr = w.ctr + (size_t)t * ctr_step; fa.err = fuse_err;
            if (L >= 2 && t >= 2) {
                hipLaunchKernelGGL(beam_plist_kernel, dim3(w.fused_teams), dim3(64), 0, s, w.b.st[t & 1].tok, ldtok, N, t,
                                   max_len, W, w.fz_plist, pad_seen);
                CAPDEC_LAUNCH_CHECK();
            }
            CAPDEC_TRY(capdec_launch_fused_layers(fa, E, s));
        }
        for (int l = 0; l < L && !fused; ++l) {
            const capdec_tf_layer& ly = m->layers[l];
            float* kc = cur + (size_t)(2 * l) * NE * max_len;
            float* vc = cur + (size_t)(2 * l + 1) * NE * max_len;
            // q, and k/v of the new position straight into row t of every hypothesis' cache
            capdec_gemm_group g[3];
            g[0] = linear_group(w.x, E, N, ly.self_attn.wq, ly.self_attn.bq, E, w.q, E);
            g[1] = linear_group(w.x, E, N, ly.self_attn.wk, ly.self_attn.bk, E, kc + (size_t)t * E, cstride);
            g[2] = linear_group(w.x, E, N, ly.self_attn.wv, ly.self_attn.bv, E, vc + (size_t)t * E, cstride);
            g[0].ldq = g[1].ldq = g[2].ldq = E;
            CAPDEC_TRY(capdec_launch_gemm(g, 3, E, 0, s));
            CAPDEC_TRY(capdec_attention(w.q, E, kc, E, cstride, vc, E, cstride, w.ao, E, N, H, 1, t + 1, dk, 0, 0, nullptr,
                                        0, nullptr, stream));
            CAPDEC_TRY(linear(w.ao, E, N, ly.self_attn.wo, ly.self_attn.bo, E, E, w.y, E, w.x, E, 0, s));
            if (s1) {
                CAPDEC_TRY(capdec_launch_layernorm_rows(w.y, nullptr, 1, nullptr, ly.n1_g, ly.n1_b, 1e-5f, N, E, w.x2, nullptr,
                                                        s, ly.n2_g, ly.n2_b, w.cc + l * BE, W, nullptr));
            } else if (s2) {
                // norm1, the two-key cross-attention block and norm2 in one pass; hypothesis n belongs to sample n / W
                CAPDEC_TRY(capdec_launch_layernorm_pair_rows(w.y, ly.n1_g, ly.n1_b, ly.n2_g, ly.n2_b, 1e-5f, N, E, w.x2,
                                                             nullptr, nullptr, W, w.pg + (size_t)l * BE * H,
                                                             w.pd + (size_t)l * BE * H, w.cc + l * BE,
                                                             w.psb + (size_t)l * B * H, H, 1.0f / sqrtf((float)dk), s));
            } else {
                CAPDEC_TRY(capdec_launch_layernorm_rows(w.y, nullptr, 1, nullptr, ly.n1_g, ly.n1_b, 1e-5f, N, E, w.x1, nullptr,
                                                        s));
                CAPDEC_TRY(linear(w.x1, E, N, ly.cross_attn.wq, ly.cross_attn.bq, E, E, w.q, E, nullptr, 0, 0, s));
                CAPDEC_TRY(capdec_attention(w.q, E, w.ck + (size_t)l * NE * S, E, (int64_t)S * E, w.cv + (size_t)l * NE * S, E,
                                            (int64_t)S * E, w.ao, E, N, H, 1, S, dk, 0, 0, nullptr, 0, nullptr, stream));
                CAPDEC_TRY(linear(w.ao, E, N, ly.cross_attn.wo, ly.cross_attn.bo, E, E, w.y, E, w.x1, E, 0, s));
                CAPDEC_TRY(capdec_launch_layernorm_rows(w.y, nullptr, 1, nullptr, ly.n2_g, ly.n2_b, 1e-5f, N, E, w.x2, nullptr,
                                                        s));
            }
            CAPDEC_TRY(linear(w.x2, E, N, ly.ff_w1, ly.ff_b1, E, E, w.hb, E, nullptr, 0, CAPDEC_GEMM_RELU, s));
            CAPDEC_TRY(linear(w.hb, E, N, ly.ff_w2, ly.ff_b2, E, E, w.y, E, w.x2, E, 0, s));
            CAPDEC_TRY(capdec_launch_layernorm_rows(w.y, nullptr, 1, nullptr, ly.n3_g, ly.n3_b, 1e-5f, N, E, w.x, nullptr, s));
        }
        CAPDEC_TRY(linear(w.x, E, N, m->fc2_w, m->fc2_b, V, E, w.b.logits, V, nullptr, 0, 0, s));
        // (with the one-launch layers a <PAD> hypothesis is followed by side rows, not reported)
        CAPDEC_TRY(beam_pick(w.b, t, B, W, V, eos, ldtok, s, fused ? nullptr : pad_seen));
        if (t + 1 < max_len) {
            const long rows = (long)N * (t + 1);
            hipLaunchKernelGGL(beam_cache_gather_kernel, dim3((unsigned)((rows + 3) / 4), 2 * L), dim3(256), 0, s, cur, nxt,
                               w.b.parent, N, t + 1, max_len, E);
            CAPDEC_LAUNCH_CHECK();
            CAPDEC_TRY(capdec_embed_pe(w.b.st[(t + 1) & 1].tok + (t + 1), 1, ldtok, m->emb, m->pe, t + 1, N, 1, E, V, w.x,
                                       stream));
        }
    }
    hipLaunchKernelGGL(beam_result_kernel, dim3((B + 63) / 64), dim3(64), 0, s, w.b.st[t_end & 1], ldtok, B, W, t_end,
                       max_len, eos, out_ids, out_len, out_scores, w.b.unfinished, n_unfinished);
    CAPDEC_LAUNCH_CHECK();
    if (fused) {
        hipLaunchKernelGGL(beam_fused_err_kernel, dim3(16), dim3(256), 0, s, fuse_err, pad_seen, B);
        CAPDEC_LAUNCH_CHECK();
    }
    return 0;
}

#ifdef CAPDEC_TUNING_HOOKS
int capdec_beam_force_unfused(int on) {
    g_beam_no_fused = on ? 1 : 0;
    return 0;
}
#endif

struct RnnBeamWs {
    float *x, *h[2], *c[2];
    BeamBufs b;
};

static size_t carve_rnn_beam(const capdec_rnn_model* m, int B, int W, int max_len, void* base, RnnBeamWs* w) {
    const int N = B * W;
    Carver c(base);
    RnnBeamWs d;
    d.x = c.take<float>((size_t)N * m->E);
    for (int n = 0; n < 2; ++n) {
        d.h[n] = c.take<float>((size_t)N * m->Hd);
        d.c[n] = c.take<float>((size_t)N * m->Hd);
    }
    carve_beam_bufs(c, N, W, m->V, max_len, &d.b);
    if (w) *w = d;
    return c.off;
}

size_t capdec_rnn_beam_workspace_bytes(const capdec_rnn_model* m, int B, int W, int max_len) {
    if (check_rnn(m) != 0 || B < 0 || W < 1 || W > BEAM_MAX_W || max_len < 1) return 0;
    return carve_rnn_beam(m, B, W, max_len, nullptr, nullptr);
}

int capdec_rnn_beam_search(const capdec_rnn_model* m, const float* x0, int B, int W, int max_len, int sos, int eos,
                           int t_begin, int t_end, void* workspace, size_t workspace_bytes, int64_t* out_ids,
                           int32_t* out_len, float* out_scores, int32_t* n_unfinished, void* stream) {
    CAPDEC_TRY(check_rnn(m));
    CAPDEC_REQUIRE(x0 && out_ids && out_len && n_unfinished && B >= 0 && W >= 1 && W <= BEAM_MAX_W && max_len >= 1);
    CAPDEC_REQUIRE(sos >= 0 && sos < m->V && m->V >= W && 0 <= t_begin && t_begin < t_end && t_end <= max_len);
    if (B == 0) return 0;
    if (!workspace || !capdec_aligned16(workspace) || !capdec_aligned16(x0)) return CAPDEC_EALIGN;
    RnnBeamWs w;
    if (carve_rnn_beam(m, B, W, max_len, workspace, &w) > workspace_bytes) return CAPDEC_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    const int E = m->E, Hd = m->Hd, V = m->V, N = B * W;
    const int64_t ldtok = max_len + 1;
    if (t_begin == 0) {
        hipLaunchKernelGGL(beam_init_kernel, dim3(64), dim3(256), 0, s, w.b.st[0], ldtok, N, sos, w.b.unfinished, max_len);
        CAPDEC_LAUNCH_CHECK();
        // the image is time-step 0 of the LSTM (cnnrnn_model.py:129); every hypothesis starts from that state (:133-134)
        CAPDEC_TRY(capdec_lstm_step(x0, E, nullptr, 0, nullptr, m->w_ih, m->w_hh, m->b_ih, m->b_hh, B, E, Hd, w.h[1], Hd,
                                    w.c[1], nullptr, stream));
        hipLaunchKernelGGL(gather_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w.h[1], (const int32_t*)nullptr, W, N,
                           Hd, w.h[0]);
        hipLaunchKernelGGL(gather_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w.c[1], (const int32_t*)nullptr, W, N,
                           Hd, w.c[0]);
        CAPDEC_LAUNCH_CHECK();
    }
    for (int t = t_begin; t < t_end; ++t) {
        const BeamState& cur = w.b.st[t & 1];
        CAPDEC_TRY(capdec_embed_pe(cur.tok + t, 1, ldtok, m->emb, nullptr, 0, N, 1, E, V, w.x, stream));
        CAPDEC_TRY(capdec_lstm_step(w.x, E, w.h[0], Hd, w.c[0], m->w_ih, m->w_hh, m->b_ih, m->b_hh, N, E, Hd, w.h[1], Hd,
                                    w.c[1], nullptr, stream));
        CAPDEC_TRY(linear(w.h[1], Hd, N, m->fc2_w, m->fc2_b, V, Hd, w.b.logits, V, nullptr, 0, 0, s));
        CAPDEC_TRY(beam_pick(w.b, t, B, W, V, eos, ldtok, s));
        // the state of every surviving hypothesis follows its parent (cnnrnn_model.py:186-188)
        hipLaunchKernelGGL(gather_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w.h[1], w.b.parent, 1, N, Hd, w.h[0]);
        hipLaunchKernelGGL(gather_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w.c[1], w.b.parent, 1, N, Hd, w.c[0]);
        CAPDEC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(beam_result_kernel, dim3((B + 63) / 64), dim3(64), 0, s, w.b.st[t_end & 1], ldtok, B, W, t_end,
                       max_len, eos, out_ids, out_len, out_scores, w.b.unfinished, n_unfinished);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// Soft-attention LSTM decoder (additive attention over the memory rows feeding an LSTM cell).
// The reference has no such model; this follows the Show-Attend-Tell decoder named by the project's goals.
// =================================================================================================
int capdec_additive_attention(const float* u, int64_t ldu, const float* pf, const float* v, const float* enc, int B, int S,
                              int A, int E, float* ctx, int64_t ldctx, float* alpha, void* stream) {
    CAPDEC_REQUIRE(u && pf && v && enc && ctx && B >= 0 && S >= 1 && S <= 4096 && A >= 4 && A % 4 == 0 && E >= 4 && E % 4 == 0);
    CAPDEC_REQUIRE(ldu >= A && ldctx >= E && ldu % 4 == 0);
    if (!capdec_aligned16(u) || !capdec_aligned16(pf) || !capdec_aligned16(v) || !capdec_aligned16(enc)) return CAPDEC_EALIGN;
    if (B == 0) return 0;
    const size_t smem = ((size_t)((S + 3) / 4) * 4 + 4 * (size_t)E) * sizeof(float);
    CAPDEC_REQUIRE(smem <= 64 * 1024);
    hipLaunchKernelGGL(additive_attention_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, u, ldu, pf, v, enc, S, A, E,
                       ctx, ldctx, alpha);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_additive_attention_bwd(const float* u, int64_t ldu, const float* pf, const float* v, const float* enc,
                                  const float* alpha, const float* dctx, int64_t lddctx, int B, int S, int A, int E,
                                  float* du, int64_t lddu, float* dpf, float* denc, float* dvb, void* stream) {
    CAPDEC_REQUIRE(u && pf && v && enc && alpha && dctx && du && dpf && denc && dvb && B >= 0 && S >= 1 && S <= 4096 &&
                   A >= 4 && A % 4 == 0 && E >= 4 && E % 4 == 0);
    CAPDEC_REQUIRE(ldu >= A && lddu >= A && lddctx >= E && ldu % 4 == 0 && lddctx % 4 == 0);
    if (!capdec_aligned16(u) || !capdec_aligned16(pf) || !capdec_aligned16(v) || !capdec_aligned16(enc) ||
        !capdec_aligned16(dctx) || !capdec_aligned16(dpf) || !capdec_aligned16(denc))
        return CAPDEC_EALIGN;
    if (B == 0) return 0;
    const int ncol = A / 4, nsg = ncol >= 256 ? 1 : 256 / ncol;
    const size_t smem = (2 * (size_t)((S + 3) / 4) * 4 + 2 * (size_t)nsg * A) * sizeof(float);
    CAPDEC_REQUIRE(smem <= 64 * 1024);
    hipLaunchKernelGGL(additive_attention_bwd_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, u, ldu, pf, v, enc, alpha,
                       dctx, lddctx, S, A, E, du, lddu, dpf, denc, dvb);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

struct SattWs {
    float *xcat, *emb0, *u, *h[2], *c[2];
    float* pval; int32_t* pidx;
    int32_t *tok, *done;
};

static size_t carve_satt(const capdec_satt_model* m, int B, int max_len, void* base, SattWs* w) {
    Carver c(base);
    SattWs d;
    d.xcat = c.take<float>((size_t)B * 2 * m->E);
    d.emb0 = c.take<float>((size_t)B * m->E);
    d.u = c.take<float>((size_t)B * m->A);
    for (int n = 0; n < 2; ++n) {
        d.h[n] = c.take<float>((size_t)B * m->Hd);
        d.c[n] = c.take<float>((size_t)B * m->Hd);
    }
    const size_t nt = (size_t)capdec_vocab_tiles(m->V);
    d.pval = c.take<float>(capdec_part_elems(B, (int)nt));
    d.pidx = c.take<int32_t>(capdec_part_elems(B, (int)nt));
    d.tok = c.take<int32_t>((size_t)B * (max_len + 1));
    d.done = c.take<int32_t>((size_t)B);
    if (w) *w = d;
    return c.off;
}

static int check_satt(const capdec_satt_model* m) {
    CAPDEC_REQUIRE(m && m->E >= 4 && m->E % 4 == 0 && m->Hd >= 4 && m->Hd % 4 == 0 && m->A >= 4 && m->A % 4 == 0 && m->V >= 1);
    CAPDEC_REQUIRE(m->emb && m->w_ih && m->w_hh && m->b_ih && m->b_hh && m->att_h_w && m->att_h_b && m->att_v && m->fc2_w &&
                   m->fc2_b);
    return 0;
}

size_t capdec_satt_decode_workspace_bytes(const capdec_satt_model* m, int B, int max_len) {
    if (check_satt(m) != 0 || B < 0 || max_len < 1) return 0;
    return carve_satt(m, B, max_len, nullptr, nullptr);
}

int capdec_satt_greedy_decode(const capdec_satt_model* m, const float* enc, const float* pf, const float* h0,
                              const float* c0, int B, int S, int max_len, int sos, int eos, void* workspace,
                              size_t workspace_bytes, int64_t* out_ids, int32_t* out_len, void* stream) {
    CAPDEC_TRY(check_satt(m));
    CAPDEC_REQUIRE(enc && pf && h0 && c0 && out_ids && out_len && B >= 0 && S >= 1 && max_len >= 1 && sos >= 0 && sos < m->V);
    if (B == 0) return 0;
    if (!workspace || !capdec_aligned16(workspace)) return CAPDEC_EALIGN;
    SattWs w;
    if (carve_satt(m, B, max_len, workspace, &w) > workspace_bytes) return CAPDEC_ENOMEM;
    hipStream_t s = (hipStream_t)stream;
    const int E = m->E, Hd = m->Hd, A = m->A, V = m->V;
    const int64_t ldtok = max_len + 1;
    hipLaunchKernelGGL(greedy_init_kernel, dim3(256), dim3(256), 0, s, w.tok, ldtok, w.done, out_ids, out_len,
                       (int32_t*)nullptr, B, max_len, sos);
    CAPDEC_LAUNCH_CHECK();
    const int nt = capdec_vocab_tiles(V);
    // <SOS> rows into the left half of the LSTM input rows (row stride 2E); the right half receives the context
    CAPDEC_TRY(capdec_embed_pe(w.tok, 1, ldtok, m->emb, nullptr, 0, B, 1, E, V, w.emb0, stream));
    if (hipMemcpy2DAsync(w.xcat, 2 * (size_t)E * sizeof(float), w.emb0, (size_t)E * sizeof(float), (size_t)E * sizeof(float),
                         B, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return (int)hipGetLastError();
    const float* h = h0;
    const float* c = c0;
    for (int t = 0; t < max_len; ++t) {
        // u = att_h(h); attention context -> xcat[:, E:]
        CAPDEC_TRY(linear(h, Hd, B, m->att_h_w, m->att_h_b, A, Hd, w.u, A, nullptr, 0, 0, s));
        CAPDEC_TRY(capdec_additive_attention(w.u, A, pf, m->att_v, enc, B, S, A, E, w.xcat + E, 2 * E, nullptr, stream));
        const int nx = t & 1;
        CAPDEC_TRY(capdec_lstm_step(w.xcat, 2 * E, h, Hd, c, m->w_ih, m->w_hh, m->b_ih, m->b_hh, B, 2 * E, Hd, w.h[nx], Hd,
                                    w.c[nx], nullptr, stream));
        h = w.h[nx];
        c = w.c[nx];
        CAPDEC_TRY(capdec_vocab_argmax_partial(h, Hd, m->fc2_w, m->fc2_b, B, V, Hd, w.pval, w.pidx, stream));
        const bool last = (t + 1 == max_len);
        hipLaunchKernelGGL(greedy_advance_kernel, dim3((B + 3) / 4), dim3(256), 0, s, w.pval, w.pidx, B, nt, t, max_len,
                           eos, w.tok, ldtok, w.done, out_ids, out_len, (int32_t*)nullptr, m->emb,
                           (const float*)nullptr, E, V, last ? nullptr : w.xcat, (int64_t)2 * E);
        CAPDEC_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
