// Row-wise kernels of the caption decoder for gfx950: embedding gather + positional encoding,
// residual-add LayerNorm, fused (masked / cross) attention.  All are HBM-bound; each uses one 64-lane
// wavefront per row with 16-byte accesses and shuffle reductions (no LDS except the score row).
#include "common.h"
#include <math.h>

namespace {

// ---- x[b*T+i,:] = emb[ids[b,i],:] + pe[pos0+i,:] ------------------------------------------------
template <typename IdT>
__global__ __launch_bounds__(256) void embed_pe_kernel(const IdT* __restrict__ ids, int64_t ldid,
                                                       const float* __restrict__ emb, const float* __restrict__ pe,
                                                       int pos0, int rows, int T, int E, int V,
                                                       float* __restrict__ x) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / T, i = row - b * T;
    long id = (long)ids[(int64_t)b * ldid + i];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const f32x4* src = reinterpret_cast<const f32x4*>(emb + id * (int64_t)E);
    const f32x4* pp = pe ? reinterpret_cast<const f32x4*>(pe + (int64_t)(pos0 + i) * E) : nullptr;
    f32x4* dst = reinterpret_cast<f32x4*>(x + (int64_t)row * E);
    for (int c = lane; c < E / 4; c += 64) {
        f32x4 v = src[c];
        if (pp) v += pp[c];
        dst[c] = v;
    }
}

// ---- y = LayerNorm(x + add) * gamma + beta ------------------------------------------------------
constexpr int LN_MAXV = 8;   // float4 per lane: E <= 2048

// NV = float4 per lane (E <= 256 * NV): sized to the row so that the row's registers do not limit occupancy
#define CAPDEC_LN_DISPATCH(E_, CALL)              \
    do {                                           \
        const int nvl_ = ((E_) / 4 + 63) / 64;     \
        if (nvl_ <= 1) { CALL(1); }                \
        else if (nvl_ <= 2) { CALL(2); }           \
        else if (nvl_ <= 4) { CALL(4); }           \
        else { CALL(8); }                          \
    } while (0)
// NM = head dimension / 16 (float4 per lane of the 4-lanes-per-key score sweep)
#define CAPDEC_DK_DISPATCH(dk_, CALL)              \
    do {                                           \
        switch ((dk_) / 16) {                      \
            case 1: CALL(1); break;                \
            case 2: CALL(2); break;                \
            case 4: CALL(4); break;                \
            case 8: CALL(8); break;                \
            default: CALL(16); break;              \
        }                                          \
    } while (0)

template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                            int add_div, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int M, int E,
                                                            float* __restrict__ y, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out,
                                                            const int32_t* __restrict__ add_map,
                                                            const int32_t* __restrict__ rows_dev,
                                                            const float* __restrict__ gamma2 = nullptr,
                                                            const float* __restrict__ beta2 = nullptr,
                                                            const float* __restrict__ add2 = nullptr, int add2_div = 1,
                                                            const int32_t* __restrict__ add2_map = nullptr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M || (rows_dev && row >= rows_dev[0])) return;
    const int nv = E / 4;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (int64_t)row * E);
    // the addend's row: row / add_div, or (add_map[row] / add_div) when rows are listed explicitly
    const int arow = (add_map ? add_map[row] : row) / add_div;
    const f32x4* ar = add ? reinterpret_cast<const f32x4*>(add + (int64_t)arow * E) : nullptr;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            v[u] = xr[c];
            if (ar) v[u] += ar[c];
            s += v[u][0] + v[u][1] + v[u][2] + v[u][3];
        }
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        if (c < nv) {
            v[u] -= mean;
            q += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
        }
    }
    const float var = wave_sum(q) / (float)E;
    const float rstd = 1.0f / sqrtf(var + eps);
    f32x4* yr = reinterpret_cast<f32x4*>(y + (int64_t)row * E);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
    if (gamma2) {
        // second normalisation on the same row, still in registers: y = LN2(LN1(x + add) + add2); the intermediate is
        // rounded exactly as if it had been stored, so the result equals two separate launches bit for bit
        const int arow2 = (add2_map ? add2_map[row] : row) / add2_div;
        const f32x4* a2 = reinterpret_cast<const f32x4*>(add2 + (int64_t)arow2 * E);
        float s2 = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = lane + 64 * u;
            if (c < nv) {
                v[u] = (v[u] * rstd * g4[c] + b4[c]) + a2[c];
                s2 += v[u][0] + v[u][1] + v[u][2] + v[u][3];
            }
        }
        const float mean2 = wave_sum(s2) / (float)E;
        float q2 = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = lane + 64 * u;
            if (c < nv) {
                v[u] -= mean2;
                q2 += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
            }
        }
        const float rstd2 = 1.0f / sqrtf(wave_sum(q2) / (float)E + eps);
        const f32x4* g24 = reinterpret_cast<const f32x4*>(gamma2);
        const f32x4* b24 = reinterpret_cast<const f32x4*>(beta2);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = lane + 64 * u;
            if (c < nv) yr[c] = v[u] * rstd2 * g24[c] + b24[c];
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        if (c < nv) yr[c] = v[u] * rstd * g4[c] + b4[c];
    }
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// Cross-attention over exactly two memory rows, folded into the two LayerNorms around it.  The softmax over two keys is
// a logistic function of the score difference, so per sample and layer everything that does not depend on the query is
// a constant: with dK = k0 - k1, dV = v0 - v1 (projected memory rows),
//   p_h  = sigmoid(((W_q x1 + b_q)_h . dK_h) / sqrt(dk)) = sigmoid((x1 . G_h + sb_h) / sqrt(dk)),  G_h = W_q,h^T dK_h
//   attn = W_o (v1 + p (.) dV) + b_o = c1 + sum_h p_h D_h,   c1 = W_o v1 + b_o,  D_h = W_o[:, h] dV_h
// (transformer.py:21-27,45-60,86 with Tk = 2).  One wavefront per row: x1 = LN1(x) stays in registers.
template <int NV>
__global__ __launch_bounds__(256) void ln_pair_kernel(const float* __restrict__ x, const float* __restrict__ gamma1,
                                                      const float* __restrict__ beta1, const float* __restrict__ gamma2,
                                                      const float* __restrict__ beta2, float eps, int M, int E,
                                                      float* __restrict__ y, const int32_t* __restrict__ rows_dev,
                                                      const int32_t* __restrict__ row_map, int map_div,
                                                      const float* __restrict__ G, const float* __restrict__ D,
                                                      const float* __restrict__ c1, const float* __restrict__ sb, int H,
                                                      float scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M || (rows_dev && row >= rows_dev[0])) return;
    const int nv = E / 4;
    const int b = (row_map ? row_map[row] : row) / map_div;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (int64_t)row * E);
    f32x4 v[NV], a2[NV];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            v[u] = xr[c];
            s += v[u][0] + v[u][1] + v[u][2] + v[u][3];
        }
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        if (c < nv) {
            v[u] -= mean;
            q += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + eps);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma1);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(beta1);
    const f32x4* c4 = reinterpret_cast<const f32x4*>(c1 + (int64_t)b * E);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        a2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            v[u] = v[u] * rstd * g4[c] + b4[c];          // x1
            a2[u] = c4[c];
        }
    }
    for (int h = 0; h < H; ++h) {
        const f32x4* gh = reinterpret_cast<const f32x4*>(G + ((int64_t)b * H + h) * E);
        const f32x4* dh = reinterpret_cast<const f32x4*>(D + ((int64_t)b * H + h) * E);
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = lane + 64 * u;
            if (c < nv) {
                const f32x4 g = gh[c];
                part += v[u][0] * g[0] + v[u][1] * g[1] + v[u][2] * g[2] + v[u][3] * g[3];
            }
        }
        const float z = (wave_sum(part) + sb[(int64_t)b * H + h]) * scale;
        const float p = 1.0f / (1.0f + expf(-z));
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = lane + 64 * u;
            if (c < nv) a2[u] += dh[c] * p;
        }
    }
    float s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        if (c < nv) {
            v[u] += a2[u];
            s2 += v[u][0] + v[u][1] + v[u][2] + v[u][3];
        }
    }
    const float mean2 = wave_sum(s2) / (float)E;
    float q2 = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        if (c < nv) {
            v[u] -= mean2;
            q2 += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
        }
    }
    const float rstd2 = 1.0f / sqrtf(wave_sum(q2) / (float)E + eps);
    const f32x4* g24 = reinterpret_cast<const f32x4*>(gamma2);
    const f32x4* b24 = reinterpret_cast<const f32x4*>(beta2);
    f32x4* yr = reinterpret_cast<f32x4*>(y + (int64_t)row * E);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        if (c < nv) yr[c] = v[u] * rstd2 * g24[c] + b24[c];
    }
}

// ---- fused attention, one wavefront per (b, h, query row) ---------------------------------------
struct AttnArgs {
    const float* q; int64_t ldq;
    const float* k; int64_t ldk, sk;
    const float* v; int64_t ldv, sv;
    float* out; int64_t ldo;
    int B, H, Tq, Tk, dk;
    int causal, causal_off;
    const int32_t* ids; int64_t ldid;
    float* lse;
    // row-list form (decode steps): Tq == 1, query/output row r, sample and position from row_map[r]
    const int32_t* row_map; int map_div;     // b = row_map[r] / map_div, position = row_map[r] % map_div
    const int32_t* rows_dev;                 // nullable device scalar: only rows r < rows_dev[0] are live
    // ragged batch (short kernel only): sample b is the rows [seq_off[b], seq_off[b+1]) of q, out and ids -- and of k, v
    // too (self-attention) unless ragged_q_only, where k, v stay Tk dense rows per sample (cross-attention)
    const int32_t* seq_off;
    int ragged_q_only;
};

template <int NM>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long total = (long)a.B * a.H * a.Tq;
    long idx = (long)blockIdx.x * 4 + wave;
    bool live = idx < total;
    if (!live) idx = total - 1;            // keep the wave in the barriers; it stores nothing
    int i = (int)(idx % a.Tq);             // position of the query inside its sample (mask / causal limit)
    const int h = (int)((idx / a.Tq) % a.H);
    int b = (int)(idx / ((long)a.Tq * a.H));
    int64_t qrow = (int64_t)b * a.Tq + i;  // row of q / out
    if (a.row_map) {
        const int r = (int)(idx / a.H);
        if (a.rows_dev && r >= a.rows_dev[0]) live = false;
        const int m = a.row_map[live ? r : 0];
        b = m / a.map_div;
        i = m - b * a.map_div;
        qrow = r;
    }
    float* sc = lds + wave * a.Tk;         // this wave's score row

    const bool pad_row = a.ids && a.ids[(int64_t)b * a.ldid + i] == 0;
    int nkeys = a.Tk;
    if (a.causal && !pad_row) {
        nkeys = i + a.causal_off + 1;
        nkeys = nkeys < 1 ? 1 : (nkeys > a.Tk ? a.Tk : nkeys);
    }
    const int dk = a.dk;
    const float* kb = a.k + (int64_t)b * a.sk + (int64_t)h * dk;
    const float* vb = a.v + (int64_t)b * a.sv + (int64_t)h * dk;

    float mx = 0.f, sum = 1.f;
    if (!pad_row) {
        // scores: 4 lanes per key, each lane owns dims {4*part + 16*m}
        const int part = lane & 3, kl = lane >> 2;
        const float* qp = a.q + qrow * a.ldq + (int64_t)h * dk + 4 * part;
        f32x4 qreg[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) qreg[m] = *reinterpret_cast<const f32x4*>(qp + 16 * m);
        const float scale = sqrtf((float)dk);
        for (int j0 = 0; j0 < nkeys; j0 += 16) {
            const int j = j0 + kl;
            float d = 0.f;
            if (j < nkeys) {
                const float* kp = kb + (int64_t)j * a.ldk + 4 * part;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + 16 * m);
                    d += qreg[m][0] * kk[0] + qreg[m][1] * kk[1] + qreg[m][2] * kk[2] + qreg[m][3] * kk[3];
                }
            }
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            if (part == 0 && j < nkeys) sc[j] = d / scale;
        }
    }
    __syncthreads();
    if (!pad_row) {
        float m = -INFINITY;
        for (int j = lane; j < nkeys; j += 64) m = fmaxf(m, sc[j]);
        mx = wave_max(m);
        float s = 0.f;
        for (int j = lane; j < nkeys; j += 64) {
            const float e = expf(sc[j] - mx);
            sc[j] = e;
            s += e;
        }
        sum = wave_sum(s);
    } else {
        // every score is the same fill value: exp(0) = 1 for each of the Tk keys
        for (int j = lane; j < nkeys; j += 64) sc[j] = 1.f;
        mx = -1e9f;
        sum = (float)nkeys;
    }
    __syncthreads();
    // P.V: dk/4 lanes per key (one float4 of the head each), 64/(dk/4) keys in flight
    const int lpv = dk / 4;
    const int grp = lane / lpv, dl = lane - grp * lpv, ngrp = 64 / lpv;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = grp; j < nkeys; j += ngrp) {
        const float p = sc[j] / sum;
        const f32x4 vv = *reinterpret_cast<const f32x4*>(vb + (int64_t)j * a.ldv + 4 * dl);
        acc += vv * p;
    }
    for (int o = lpv; o < 64; o <<= 1) {
        acc[0] += __shfl_xor(acc[0], o, 64);
        acc[1] += __shfl_xor(acc[1], o, 64);
        acc[2] += __shfl_xor(acc[2], o, 64);
        acc[3] += __shfl_xor(acc[3], o, 64);
    }
    if (live && grp == 0)
        *reinterpret_cast<f32x4*>(a.out + qrow * a.ldo + (int64_t)h * dk + 4 * dl) = acc;
    if (live && lane == 0 && a.lse) a.lse[idx] = mx + logf(sum);
}

// Short sequences without a row list (teacher-forced forward, full-prefix decode): one workgroup per (b,h) stages
// that head's K and V once in LDS (rows padded to dk+16 floats: the 16-keys-by-4-parts sweep then touches 64
// distinct banks per 16 lanes) and its four wavefronts walk the query rows.  Same arithmetic, row by row, as
// attention_kernel.
template <int NM>
__global__ __launch_bounds__(256) void attention_short_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dk = a.dk, LD = dk + 16;
    float* Ks = lds;
    float* Vs = Ks + a.Tk * LD;
    float* scs = Vs + a.Tk * LD;           // [4][Tk] score rows
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int nv = dk / 4;
    int Tq = a.Tq, Tk = a.Tk;
    int64_t q0 = (int64_t)b * Tq, id0 = (int64_t)b * a.ldid;
    const float* kb = a.k + (int64_t)b * a.sk;
    const float* vb = a.v + (int64_t)b * a.sv;
    if (a.seq_off) {
        q0 = id0 = a.seq_off[b];
        Tq = min(a.Tq, a.seq_off[b + 1] - a.seq_off[b]);
        if (!a.ragged_q_only) {
            Tk = Tq;
            kb = a.k + q0 * a.ldk;
            vb = a.v + q0 * a.ldv;
        }
    }
    for (int idx = threadIdx.x; idx < Tk * nv; idx += 256) {
        const int r = idx / nv, c = idx - r * nv;
        *reinterpret_cast<f32x4*>(&Ks[r * LD + 4 * c]) =
            *reinterpret_cast<const f32x4*>(kb + (int64_t)r * a.ldk + (int64_t)h * dk + 4 * c);
        *reinterpret_cast<f32x4*>(&Vs[r * LD + 4 * c]) =
            *reinterpret_cast<const f32x4*>(vb + (int64_t)r * a.ldv + (int64_t)h * dk + 4 * c);
    }
    __syncthreads();
    float* sc = scs + wave * a.Tk;
    const float scale = sqrtf((float)dk);
    const int part = lane & 3, kl = lane >> 2;
    const int lpv = dk / 4, grp = lane / lpv, dl = lane - grp * lpv, ngrp = 64 / lpv;
    for (int i = wave; i < Tq; i += 4) {
        const int64_t qrow = q0 + i;
        const bool pad_row = a.ids && a.ids[id0 + i] == 0;
        int nkeys = Tk;
        if (a.causal && !pad_row) {
            nkeys = i + a.causal_off + 1;
            nkeys = nkeys < 1 ? 1 : (nkeys > Tk ? Tk : nkeys);
        }
        float mx = -1e9f, sum = (float)nkeys;
        if (!pad_row) {
            const float* qp = a.q + qrow * a.ldq + (int64_t)h * dk + 4 * part;
            f32x4 qreg[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) qreg[m] = *reinterpret_cast<const f32x4*>(qp + 16 * m);
            for (int j0 = 0; j0 < nkeys; j0 += 16) {
                const int j = j0 + kl;
                float d = 0.f;
                if (j < nkeys) {
                    const float* kp = Ks + j * LD + 4 * part;
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const f32x4 kk = *reinterpret_cast<const f32x4*>(kp + 16 * m);
                        d += qreg[m][0] * kk[0] + qreg[m][1] * kk[1] + qreg[m][2] * kk[2] + qreg[m][3] * kk[3];
                    }
                }
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                if (part == 0 && j < nkeys) sc[j] = d / scale;
            }
            // LDS serves one wavefront's accesses in order: the scores above are visible to its other lanes
            float m = -INFINITY;
            for (int j = lane; j < nkeys; j += 64) m = fmaxf(m, sc[j]);
            mx = wave_max(m);
            float s = 0.f;
            for (int j = lane; j < nkeys; j += 64) {
                const float e = expf(sc[j] - mx);
                sc[j] = e;
                s += e;
            }
            sum = wave_sum(s);
        } else {
            for (int j = lane; j < nkeys; j += 64) sc[j] = 1.f;      // uniform over all Tk keys
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = grp; j < nkeys; j += ngrp)
            acc += *reinterpret_cast<const f32x4*>(Vs + j * LD + 4 * dl) * (sc[j] / sum);
        for (int o = lpv; o < 64; o <<= 1) {
            acc[0] += __shfl_xor(acc[0], o, 64);
            acc[1] += __shfl_xor(acc[1], o, 64);
            acc[2] += __shfl_xor(acc[2], o, 64);
            acc[3] += __shfl_xor(acc[3], o, 64);
        }
        if (grp == 0) *reinterpret_cast<f32x4*>(a.out + qrow * a.ldo + (int64_t)h * dk + 4 * dl) = acc;
        if (lane == 0 && a.lse) a.lse[((int64_t)b * a.H + h) * a.Tq + i] = mx + logf(sum);
    }
    if (a.seq_off && b == a.B - 1) {
        // ragged batch: the slots between the last sample and the next multiple of 32 belong to no sample but are
        // read by the contractions that follow; this head's columns of those rows are zeroed here
        const int live = a.seq_off[a.B], upto = min(a.B * a.Tq, (live + 31) / 32 * 32);
        for (int idx = threadIdx.x; idx < (upto - live) * nv; idx += 256) {
            const int r = live + idx / nv, c = idx % nv;
            *reinterpret_cast<f32x4*>(a.out + (int64_t)r * a.ldo + (int64_t)h * dk + 4 * c) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// Causal self-attention of a ragged training batch at the caption decoder's own shape -- head width 128, at most 32
// positions -- on the matrix cores: S = Q K^T (32 x 32 x 128 with the sample's rows padded to 32) and O = P V
// (32 x 128 x 32) as fp32 MFMAs of 16 x 16 x 4 over LDS-resident operands; one workgroup of four wavefronts per
// (sample, head).  Stage 1: wavefront w owns the 16 x 16 score tile (w >> 1, w & 1); the row maxima and row sums of the
// two column tiles meet in LDS; stage 2: wavefront w owns the head columns [32w, 32w + 32) of the output, which leaves
// through LDS as 16-byte stores.  Operand layout: lane (c, g) = (lane & 15, lane >> 4) feeds A[row c][k = 4g' + m] and
// B[k][column c] of MFMA m; accumulator register e of lane (c, g) is row 4g + e, column c.  Same masks, same
// log-sum-exp as attention_short_kernel; the sums run in a different order.
constexpr int AFM_T = 32, AFM_DK = 128, AFM_LD = AFM_DK + 4, AFM_PL = AFM_T + 4;
static size_t afm_lds_bytes(int rows) {
    return ((size_t)3 * rows * AFM_LD + AFM_T * AFM_PL + 4 * AFM_T) * sizeof(float);
}

__global__ __launch_bounds__(256) void attention_mfma_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int R = a.Tq;                       // rows held per operand (the longest possible sample)
    float* Qs = lds;
    float* Ks = Qs + R * AFM_LD;
    float* Vs = Ks + R * AFM_LD;
    float* Ps = Vs + R * AFM_LD;              // [32][36] probabilities
    float* Mx = Ps + AFM_T * AFM_PL;          // [2][32] row maxima per column tile
    float* Sm = Mx + 2 * AFM_T;               // [2][32] row sums per column tile
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int64_t q0 = a.seq_off[b];
    const int T = min(a.Tq, a.seq_off[b + 1] - a.seq_off[b]);
    const int64_t hc = (int64_t)h * AFM_DK;
    const int ti = wave >> 1, tj = wave & 1;
    int pad_r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = ti * 16 + 4 * g + e;
        pad_r[e] = (i < T && a.ids) ? (a.ids[q0 + i] == 0) : 0;
    }
    for (int idx = threadIdx.x; idx < R * (AFM_DK / 4); idx += 256) {
        const int r = idx >> 5, v4 = (idx & 31) * 4;
        f32x4 xq = {0.f, 0.f, 0.f, 0.f}, xk = xq, xv = xq;
        if (r < T) {
            xq = *reinterpret_cast<const f32x4*>(a.q + (q0 + r) * a.ldq + hc + v4);
            xk = *reinterpret_cast<const f32x4*>(a.k + (q0 + r) * a.ldk + hc + v4);
            xv = *reinterpret_cast<const f32x4*>(a.v + (q0 + r) * a.ldv + hc + v4);
        }
        *reinterpret_cast<f32x4*>(Qs + r * AFM_LD + v4) = xq;
        *reinterpret_cast<f32x4*>(Ks + r * AFM_LD + v4) = xk;
        *reinterpret_cast<f32x4*>(Vs + r * AFM_LD + v4) = xv;
    }
    __syncthreads();
    // ---- stage 1: scores, masks, softmax ------------------------------------------------------------------------
    const float scale = sqrtf((float)AFM_DK);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        const float* qa = Qs + min(ti * 16 + c, R - 1) * AFM_LD + 4 * g;
        const float* kb = Ks + min(tj * 16 + c, R - 1) * AFM_LD + 4 * g;
#pragma unroll
        for (int kc = 0; kc < AFM_DK / 16; ++kc) {
            const f32x4 fq = *reinterpret_cast<const f32x4*>(qa + 16 * kc);
            const f32x4 fk = *reinterpret_cast<const f32x4*>(kb + 16 * kc);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[m], fk[m], acc, 0, 0, 0);
        }
    }
    const int j = tj * 16 + c;
    bool seen[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = ti * 16 + 4 * g + e;
        int nkeys = T;
        if (a.causal && !pad_r[e]) {
            nkeys = i + a.causal_off + 1;
            nkeys = nkeys < 1 ? 1 : (nkeys > T ? T : nkeys);
        }
        seen[e] = i < T && j < nkeys;
        acc[e] = acc[e] / scale;
        float m = (seen[e] && !pad_r[e]) ? acc[e] : -INFINITY;
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        m = fmaxf(m, __shfl_xor(m, 4, 64));
        m = fmaxf(m, __shfl_xor(m, 8, 64));
        if (c == 0) Mx[tj * AFM_T + i] = m;
    }
    __syncthreads();
    float mx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = ti * 16 + 4 * g + e;
        mx[e] = fmaxf(Mx[i], Mx[AFM_T + i]);
        float pe = 0.f;
        if (seen[e]) pe = pad_r[e] ? 1.f : expf(acc[e] - mx[e]);     // a <PAD> query row: the same value for every key
        acc[e] = pe;
        float sm = pe;
        sm += __shfl_xor(sm, 1, 64);
        sm += __shfl_xor(sm, 2, 64);
        sm += __shfl_xor(sm, 4, 64);
        sm += __shfl_xor(sm, 8, 64);
        if (c == 0) Sm[tj * AFM_T + i] = sm;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = ti * 16 + 4 * g + e;
        const float sum = Sm[i] + Sm[AFM_T + i];
        Ps[i * AFM_PL + j] = (i < T && acc[e] != 0.f) ? acc[e] / sum : 0.f;
        if (a.lse && tj == 0 && c == 0 && i < T)
            a.lse[((int64_t)b * a.H + h) * a.Tq + i] = (pad_r[e] ? -1e9f : mx[e]) + logf(sum);
    }
    __syncthreads();
    // ---- stage 2: O = P V, this wavefront's 32 head columns -------------------------------------------------------
    f32x4 ao[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) ao[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rt * 16 >= T) break;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            if (kc * 16 >= T) break;
            const f32x4 fa = *reinterpret_cast<const f32x4*>(Ps + (rt * 16 + c) * AFM_PL + kc * 16 + 4 * g);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int kr = min(kc * 16 + 4 * g + m, R - 1);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    ao[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m], Vs[kr * AFM_LD + wave * 32 + ct * 16 + c],
                                                                      ao[rt][ct], 0, 0, 0);
            }
        }
    }
    // the output through LDS (over Q, which nobody reads any more: the last reads were before the barriers above)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = rt * 16 + 4 * g + e;
                if (r < T) Qs[r * AFM_LD + wave * 32 + ct * 16 + c] = ao[rt][ct][e];
            }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * (AFM_DK / 4); idx += 256) {
        const int r = idx >> 5, v4 = (idx & 31) * 4;
        *reinterpret_cast<f32x4*>(a.out + (q0 + r) * a.ldo + hc + v4) = *reinterpret_cast<const f32x4*>(Qs + r * AFM_LD + v4);
    }
    if (b == a.B - 1) {
        // the slots between the last sample and the next multiple of 32 belong to no sample but are read by the
        // contractions that follow: this head's columns of those rows are zeroed here
        const int live = a.seq_off[a.B], upto = min(a.B * a.Tq, (live + 31) / 32 * 32);
        for (int idx = threadIdx.x; idx < (upto - live) * (AFM_DK / 4); idx += 256) {
            const int r = live + idx / (AFM_DK / 4), v4 = (idx % (AFM_DK / 4)) * 4;
            *reinterpret_cast<f32x4*>(a.out + (int64_t)r * a.ldo + hc + v4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// Long unmasked memories with several query rows per sample (cross-attention over a token or grid memory: 4 <= Tq <= 64,
// Tk > 64): one workgroup per (b,h) keeps that head's queries in LDS and streams the keys and values through LDS in
// chunks of AL_KC rows, each read from memory once; every wavefront carries the running maximum, the running sum and
// the un-normalised output of its query rows from chunk to chunk (the softmax is rescaled whenever the maximum grows).
// Rows may be ragged on the query side (seq_off, ragged_q_only) exactly as in attention_short_kernel.
constexpr int AL_KC = 32;

// Each wavefront owns the query rows i = wave + 4 * (g * RB + n), g < NG, n < RB.  The RB rows of a group are worked
// on together: a key fragment read from LDS is dotted with all RB queries (held in registers), a value fragment is
// scaled into all RB accumulators, so the chunk is read from LDS once per group instead of once per query row -- the
// one-row-at-a-time form of this kernel was bound by exactly that LDS traffic (5 x the bytes at 20 query rows).
template <int NM, int NG, int RB>
__global__ __launch_bounds__(256) void attention_long_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RPW = NG * RB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dk = a.dk, LD = dk + 16, Tk = a.Tk;
    float* Qs = lds;                       // [a.Tq][LD]
    float* Ks = Qs + a.Tq * LD;            // [AL_KC][LD]
    float* Vs = Ks + AL_KC * LD;           // [AL_KC][LD]
    float* sc = Vs + AL_KC * LD + wave * (RB * AL_KC);     // this wavefront's exp-scores of the chunk, [RB][AL_KC]
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int nv = dk / 4;
    int Tq = a.Tq;
    int64_t q0 = (int64_t)b * Tq;
    if (a.seq_off) {
        q0 = a.seq_off[b];
        Tq = min(a.Tq, a.seq_off[b + 1] - a.seq_off[b]);
    }
    const float* kb = a.k + (int64_t)b * a.sk + (int64_t)h * dk;
    const float* vb = a.v + (int64_t)b * a.sv + (int64_t)h * dk;
    for (int idx = threadIdx.x; idx < a.Tq * nv; idx += 256) {
        const int r = idx / nv, c = idx - r * nv;
        // rows past this sample's end are staged as zeros: their results are never stored
        *reinterpret_cast<f32x4*>(&Qs[r * LD + 4 * c]) =
            r < Tq ? *reinterpret_cast<const f32x4*>(a.q + (q0 + r) * a.ldq + (int64_t)h * dk + 4 * c)
                   : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_scale = 1.0f / sqrtf((float)dk);
    // score sweep: LPK lanes share a key (each holds NQ float4 of the query rows and of the key), 64 / LPK keys per
    // sweep.  Eight lanes per key halve the query registers of the four-lane form, which is what lets two
    // workgroups share a CU: with one, nothing hides the chunk loads.
    constexpr int LPK = NM >= 2 ? 8 : 4, NQ = 4 * NM / LPK, KPS = 64 / LPK, NSW = AL_KC / KPS;
    const int part = lane & (LPK - 1), kl = lane / LPK;
    constexpr int LPV = 4 * NM;            // lanes per key in the value sweep (one float4 of the head each)
    constexpr int NGRP = 64 / LPV;         // key groups of that sweep
    constexpr int KPG = AL_KC / NGRP;      // consecutive keys per group
    const int grp = lane / LPV, dl = lane - grp * LPV;
    float mrow[RPW], lrow[RPW];
    f32x4 acc[RPW];
#pragma unroll
    for (int n = 0; n < RPW; ++n) {
        mrow[n] = -INFINITY;
        lrow[n] = 0.f;
        acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 qreg[RB][NQ];
    if (NG == 1) {                         // one group: its queries stay in registers across all chunks
        __syncthreads();
#pragma unroll
        for (int n = 0; n < RB; ++n)
#pragma unroll
            for (int m = 0; m < NQ; ++m)
                qreg[n][m] = *reinterpret_cast<const f32x4*>(Qs + min(wave + 4 * n, a.Tq - 1) * LD + 4 * part +
                                                             4 * LPK * m);
    }
    for (int c0 = 0; c0 < Tk; c0 += AL_KC) {
        const int nk = min(AL_KC, Tk - c0);
        __syncthreads();                   // the previous chunk is consumed (and the queries are staged)
        for (int idx = threadIdx.x; idx < AL_KC * nv; idx += 256) {
            const int r = idx / nv, c = idx - r * nv;
            const int rr = min(r, nk - 1);                 // rows past the end repeat the last key: masked below
            *reinterpret_cast<f32x4*>(&Ks[r * LD + 4 * c]) =
                *reinterpret_cast<const f32x4*>(kb + (int64_t)(c0 + rr) * a.ldk + 4 * c);
            *reinterpret_cast<f32x4*>(&Vs[r * LD + 4 * c]) =
                *reinterpret_cast<const f32x4*>(vb + (int64_t)(c0 + rr) * a.ldv + 4 * c);
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (wave + 4 * g * RB >= Tq) break;
            if (NG > 1) {
#pragma unroll
                for (int n = 0; n < RB; ++n)
#pragma unroll
                    for (int m = 0; m < NQ; ++m)
                        qreg[n][m] = *reinterpret_cast<const f32x4*>(Qs + min(wave + 4 * (g * RB + n), a.Tq - 1) * LD +
                                                                     4 * part + 4 * LPK * m);
            }
            // scores of the chunk's 32 keys against the RB queries: KPS keys x LPK parts per sweep
            float d[NSW][RB];
#pragma unroll
            for (int sw = 0; sw < NSW; ++sw) {
                const int j = KPS * sw + kl;
                f32x4 kk[NQ];
#pragma unroll
                for (int m = 0; m < NQ; ++m)
                    kk[m] = *reinterpret_cast<const f32x4*>(Ks + j * LD + 4 * part + 4 * LPK * m);
#pragma unroll
                for (int n = 0; n < RB; ++n) {
                    float t = 0.f;
#pragma unroll
                    for (int m = 0; m < NQ; ++m)
                        t += qreg[n][m][0] * kk[m][0] + qreg[n][m][1] * kk[m][1] + qreg[n][m][2] * kk[m][2] +
                             qreg[n][m][3] * kk[m][3];
#pragma unroll
                    for (int o = 1; o < LPK; o <<= 1) t += __shfl_xor(t, o, 64);
                    d[sw][n] = j < nk ? t * inv_scale : -INFINITY;
                }
            }
            float alpha[RB];
#pragma unroll
            for (int n = 0; n < RB; ++n) {
                const int r = g * RB + n;
                float cm = d[0][n];
#pragma unroll
                for (int sw = 1; sw < NSW; ++sw) cm = fmaxf(cm, d[sw][n]);
#pragma unroll
                for (int o = LPK; o < 64; o <<= 1) cm = fmaxf(cm, __shfl_xor(cm, o, 64));
                const float mnew = fmaxf(mrow[r], cm);
                alpha[n] = expf(mrow[r] - mnew);               // 0 on the first chunk (exp(-inf))
                float ps = 0.f;
#pragma unroll
                for (int sw = 0; sw < NSW; ++sw) {
                    const float e = expf(d[sw][n] - mnew);      // 0 for the keys past the end
                    if (part == 0) sc[n * AL_KC + KPS * sw + kl] = e;
                    ps += e;
                }
                // every lane of a key's LPK holds the same values: the wave total counts each key LPK times
#pragma unroll
                for (int o = LPK; o < 64; o <<= 1) ps += __shfl_xor(ps, o, 64);
                lrow[r] = lrow[r] * alpha[n] + ps;
                mrow[r] = mnew;
            }
            // LDS serves one wavefront's accesses in order: the exp-scores above are visible to its other lanes
            f32x4 add[RB];
#pragma unroll
            for (int n = 0; n < RB; ++n) add[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (KPG % 4 == 0) {
#pragma unroll
                for (int jq = 0; jq < KPG / 4; ++jq) {
                    const int j4 = grp * KPG + 4 * jq;
                    f32x4 pf[RB];
#pragma unroll
                    for (int n = 0; n < RB; ++n) pf[n] = *reinterpret_cast<const f32x4*>(sc + n * AL_KC + j4);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const f32x4 vv = *reinterpret_cast<const f32x4*>(Vs + (j4 + jj) * LD + 4 * dl);
#pragma unroll
                        for (int n = 0; n < RB; ++n) add[n] += vv * pf[n][jj];
                    }
                }
            } else {
                for (int j = grp * KPG; j < (grp + 1) * KPG; ++j) {
                    const f32x4 vv = *reinterpret_cast<const f32x4*>(Vs + j * LD + 4 * dl);
#pragma unroll
                    for (int n = 0; n < RB; ++n) add[n] += vv * sc[n * AL_KC + j];
                }
            }
#pragma unroll
            for (int n = 0; n < RB; ++n) acc[g * RB + n] = acc[g * RB + n] * alpha[n] + add[n];
        }
    }
#pragma unroll
    for (int n = 0; n < RPW; ++n) {
        const int i = wave + 4 * n;
        if (i >= Tq) break;
        f32x4 o = acc[n];
        for (int s = LPV; s < 64; s <<= 1) {
            o[0] += __shfl_xor(o[0], s, 64);
            o[1] += __shfl_xor(o[1], s, 64);
            o[2] += __shfl_xor(o[2], s, 64);
            o[3] += __shfl_xor(o[3], s, 64);
        }
        if (grp == 0) *reinterpret_cast<f32x4*>(a.out + (q0 + i) * a.ldo + (int64_t)h * dk + 4 * dl) = o / lrow[n];
        if (lane == 0 && a.lse) a.lse[((int64_t)b * a.H + h) * a.Tq + i] = mrow[n] + logf(lrow[n]);
    }
    if (a.seq_off && b == a.B - 1) {
        // ragged batch: zero this head's columns of the rows between the last sample and the next multiple of 32
        const int live = a.seq_off[a.B], upto = min(a.B * a.Tq, (live + 31) / 32 * 32);
        for (int idx = threadIdx.x; idx < (upto - live) * nv; idx += 256) {
            const int r = live + idx / nv, c = idx % nv;
            *reinterpret_cast<f32x4*>(a.out + (int64_t)r * a.ldo + (int64_t)h * dk + 4 * c) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// ---- softmax over score rows held in memory (one wavefront per row) -------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* s, int64_t ld, int rows, int cols, int pad_to,
                                                           const int32_t* live, int live_idx, int mult) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = live ? min(rows, live[live_idx] * mult) : rows;
    if (row >= n) return;
    float* p = s + (int64_t)row * ld;
    float m = -INFINITY;
    for (int j = lane; j < cols; j += 64) m = fmaxf(m, p[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < cols; j += 64) {
        const float e = expf(p[j] - m);
        p[j] = e;                              // a lane re-reads only what it wrote itself
        sum += e;
    }
    sum = wave_sum(sum);
    for (int j = lane; j < cols; j += 64) p[j] = p[j] / sum;
    for (int j = cols + lane; j < pad_to; j += 64) p[j] = 0.f;
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, float* dp, int64_t ld, int rows,
                                                               int cols, int pad_to, float scale, const int32_t* live,
                                                               int live_idx, int mult) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = live ? min(rows, live[live_idx] * mult) : rows;
    if (row >= n) return;
    const float* pr = p + (int64_t)row * ld;
    float* d = dp + (int64_t)row * ld;
    float dot = 0.f;
    for (int j = lane; j < cols; j += 64) dot += pr[j] * d[j];
    dot = wave_sum(dot);
    for (int j = lane; j < cols; j += 64) d[j] = pr[j] * (d[j] - dot) * scale;
    for (int j = cols + lane; j < pad_to; j += 64) d[j] = 0.f;
}

// ---- decode-step attention over the raw memory (key / value projections folded away) ----------------------------
// One workgroup per decode row, four heads per pass.  The query of head h is q'_h = (q W_k)_h, E wide (computed by the
// caller with the folded weight W_k,h^T W_q,h), scores_j = q'_h . mem_j * scale, and the output is the context
// sum_j p_j mem_j, E wide per head (the caller applies the folded W_o,h W_v,h).  The sample's S memory rows are read
// from memory ONCE for all heads and for both products (the projected form reads a K row and a V row per memory row
// and head group: twice the bytes, plus the projection itself).
// No LDS staging and no barriers in the key loop: the workgroup's four wavefronts take the key groups w, w+4, ... (four
// keys each); a memory row goes straight into registers, a lane owning E/64 of its columns (16-byte slices lane,
// lane+64: every load instruction covers 1 KiB of one row); the 4 keys x 4 heads partial scores of a group are summed
// over the lanes with a reduce-scatter (8+4+2+1+2 shuffles: 4.25 per key instead of 24 for sixteen full reductions),
// after which lane L holds the score of (key, head) = ((L >> 4) & 3, (L >> 2) & 3) and keeps the running maximum and
// sum of that head; exp-scores and rescale factors go back to all lanes as constant-lane reads.  The next group's
// rows are in flight while the current one is worked on.  The four partial softmaxes meet once, at the end, in LDS.
// (Two earlier forms staged 32-row chunks through LDS: a head per wavefront -- LDS-bound, 290 us at 1024 rows x 261 keys --
//  and a column quarter per wavefront with the softmax through LDS -- four barriers per chunk, 166 us; this one 107 us.)
constexpr int AM_KG = 4, AM_HB = 4;

template <int E>
__global__ __launch_bounds__(256, 2) void attention_memory_kernel(const float* __restrict__ qp, int64_t ldq,
                                                                  const float* __restrict__ mem, int S, int H, float scale,
                                                                  float* __restrict__ ctx, int64_t ldc,
                                                                  const int32_t* __restrict__ row_map, int map_div,
                                                                  const int32_t* __restrict__ rows_dev) {
    constexpr int NS = E >= 256 ? E / 256 : 1;          // 16-byte slices of a row per lane
    constexpr int LIVE = E / 4 / NS;                    // lanes that own columns (all 64 from E = 256 on)
    __shared__ __attribute__((aligned(16))) float mrg[4][AM_HB][4 + NS * 4 * 64];   // per wavefront and head: max, sum, context
    const int row = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (rows_dev && row >= rows_dev[0]) return;
    const int b = (row_map ? row_map[row] : row) / map_div;
    const float* mb = mem + (int64_t)b * S * E;
    const bool own = lane < LIVE;
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int h0 = 0; h0 < H; h0 += AM_HB) {
        f32x4 q[AM_HB][NS];
#pragma unroll
        for (int hh = 0; hh < AM_HB; ++hh)
#pragma unroll
            for (int u = 0; u < NS; ++u)
                q[hh][u] = (own && h0 + hh < H)
                               ? *reinterpret_cast<const f32x4*>(qp + (int64_t)row * ldq + (int64_t)(h0 + hh) * E + 4 * (lane + 64 * u))
                               : zero;
        f32x4 acc[AM_HB][NS];
#pragma unroll
        for (int hh = 0; hh < AM_HB; ++hh)
#pragma unroll
            for (int u = 0; u < NS; ++u) acc[hh][u] = zero;
        float m = -INFINITY, l = 0.f;                   // of head (lane >> 2) & 3, over this wavefront's keys
        f32x4 kv[2][AM_KG][NS];
        auto fetch = [&](int set, int k0) {
#pragma unroll
            for (int g = 0; g < AM_KG; ++g) {
                const int k = min(k0 + g, S - 1);       // keys past the end repeat the last one: masked below
#pragma unroll
                for (int u = 0; u < NS; ++u)
                    kv[set][g][u] = own ? *reinterpret_cast<const f32x4*>(mb + (int64_t)k * E + 4 * (lane + 64 * u)) : zero;
            }
        };
        auto work = [&](int set, int k0) {
            float t[AM_KG * AM_HB];                     // index 4 * key + head
#pragma unroll
            for (int g = 0; g < AM_KG; ++g)
#pragma unroll
                for (int hh = 0; hh < AM_HB; ++hh) {
                    float s = 0.f;
#pragma unroll
                    for (int u = 0; u < NS; ++u)
                        s += q[hh][u][0] * kv[set][g][u][0] + q[hh][u][1] * kv[set][g][u][1] + q[hh][u][2] * kv[set][g][u][2] +
                             q[hh][u][3] * kv[set][g][u][3];
                    t[4 * g + hh] = s;
                }
            // reduce-scatter: after the stage with mask M a lane keeps the half of its values selected by its bit M
            float a8[8], a4[4], a2[2], a1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float send = b5 ? t[i] : t[i + 8];
                a8[i] = (b5 ? t[i + 8] : t[i]) + __shfl_xor(send, 32, 64);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float send = b4 ? a8[i] : a8[i + 4];
                a4[i] = (b4 ? a8[i + 4] : a8[i]) + __shfl_xor(send, 16, 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float send = b3 ? a4[i] : a4[i + 2];
                a2[i] = (b3 ? a4[i + 2] : a4[i]) + __shfl_xor(send, 8, 64);
            }
            {
                const float send = b2 ? a2[0] : a2[1];
                a1 = (b2 ? a2[1] : a2[0]) + __shfl_xor(send, 4, 64);
            }
            a1 += __shfl_xor(a1, 2, 64);
            a1 += __shfl_xor(a1, 1, 64);
            // this lane: value index 8 b5 + 4 b4 + 2 b3 + b2 = 4 key + head
            const int g_mine = (b5 ? 2 : 0) + (b4 ? 1 : 0);
            const float sc = k0 + g_mine < S ? a1 * scale : -INFINITY;
            float gm = sc;                              // maximum over the group's keys of this lane's head
            gm = fmaxf(gm, __shfl_xor(gm, 16, 64));
            gm = fmaxf(gm, __shfl_xor(gm, 32, 64));
            const float mnew = fmaxf(m, gm);            // finite: a group's first key exists
            const float e = expf(sc - mnew);
            const float alpha = expf(m - mnew);         // 0 on the first group
            float es = e;
            es += __shfl_xor(es, 16, 64);
            es += __shfl_xor(es, 32, 64);
            l = l * alpha + es;
            m = mnew;
#pragma unroll
            for (int hh = 0; hh < AM_HB; ++hh) {
                const float ah = __shfl(alpha, 4 * hh, 64);       // head hh: the lanes with (lane >> 2) & 3 == hh
#pragma unroll
                for (int u = 0; u < NS; ++u) acc[hh][u] *= ah;
            }
#pragma unroll
            for (int g = 0; g < AM_KG; ++g)
#pragma unroll
                for (int hh = 0; hh < AM_HB; ++hh) {
                    const float p = __shfl(e, 4 * (4 * g + hh), 64);
#pragma unroll
                    for (int u = 0; u < NS; ++u) acc[hh][u] += kv[set][g][u] * p;
                }
        };
        const int stride = 4 * AM_KG;
        int k0 = wave * AM_KG;
        if (k0 < S) fetch(0, k0);
        while (k0 < S) {
            if (k0 + stride < S) fetch(1, k0 + stride);
            work(0, k0);
            k0 += stride;
            if (k0 >= S) break;
            if (k0 + stride < S) fetch(0, k0 + stride);
            work(1, k0);
            k0 += stride;
        }
        // the four wavefronts' partial softmaxes meet in LDS; wavefront w finishes head h0 + w
#pragma unroll
        for (int hh = 0; hh < AM_HB; ++hh) {
            float* dst = &mrg[wave][hh][0];
            const float mh = __shfl(m, 4 * hh, 64), lh = __shfl(l, 4 * hh, 64);
            if (lane == 0) {
                dst[0] = mh;
                dst[1] = lh;
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) *reinterpret_cast<f32x4*>(dst + 4 + 4 * (lane + 64 * u)) = acc[hh][u];
        }
        __syncthreads();
        {
            const int hh = wave;
            float mm = -INFINITY;
#pragma unroll
            for (int w = 0; w < 4; ++w) mm = fmaxf(mm, mrg[w][hh][0]);
            float lt = 0.f;
            f32x4 o[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) o[u] = zero;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float mw = mrg[w][hh][0];
                const float f = mw == -INFINITY ? 0.f : expf(mw - mm);      // a wavefront without keys (S < 16)
                lt += mrg[w][hh][1] * f;
#pragma unroll
                for (int u = 0; u < NS; ++u) o[u] += *reinterpret_cast<const f32x4*>(&mrg[w][hh][4 + 4 * (lane + 64 * u)]) * f;
            }
            if (own && h0 + hh < H) {
                const float inv = 1.0f / lt;
#pragma unroll
                for (int u = 0; u < NS; ++u)
                    *reinterpret_cast<f32x4*>(ctx + (int64_t)row * ldc + (int64_t)(h0 + hh) * E + 4 * (lane + 64 * u)) = o[u] * inv;
            }
        }
        __syncthreads();                                // mrg is reused by the next pass
    }
}

// ---- BatchNorm1d on [B, C] ----------------------------------------------------------------------
// eval: fold the running statistics into a per-column scale/shift for a GEMM epilogue
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                               float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// train: batch statistics per column (two passes: mean, then centred variance), normalise, and update
// the running buffers (biased variance normalises, unbiased variance feeds the running average).
// The batch is small (a few hundred rows), so the kernel is latency-bound: 16 columns per workgroup,
// 16 threads per column striding the rows, four independent loads in flight per thread.
constexpr int BN_CW = 16, BN_RG = 16;

__device__ __forceinline__ float bn_col_reduce(float v, float (*red)[BN_CW], int rg, int cl) {
    __syncthreads();                 // previous use of red is over
    red[rg][cl] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < BN_RG; ++g) t += red[g][cl];
    return t;
}

__global__ __launch_bounds__(256) void bn_train_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float momentum,
                                                       float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, int B, int C,
                                                       float* __restrict__ y, float* __restrict__ save_mean,
                                                       float* __restrict__ save_rstd) {
    __shared__ float red[BN_RG][BN_CW];
    const int cl = threadIdx.x & (BN_CW - 1), rg = threadIdx.x / BN_CW;
    const int c = blockIdx.x * BN_CW + cl;
    const bool ok = c < C;
    const float* xc = x + (ok ? c : 0);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok) {
        int r = rg;
        for (; r + 3 * BN_RG < B; r += 4 * BN_RG) {
            s0 += xc[(int64_t)r * C];
            s1 += xc[(int64_t)(r + BN_RG) * C];
            s2 += xc[(int64_t)(r + 2 * BN_RG) * C];
            s3 += xc[(int64_t)(r + 3 * BN_RG) * C];
        }
        for (; r < B; r += BN_RG) s0 += xc[(int64_t)r * C];
    }
    const float mean = bn_col_reduce((s0 + s1) + (s2 + s3), red, rg, cl) / (float)B;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    if (ok) {
        int r = rg;
        for (; r + 3 * BN_RG < B; r += 4 * BN_RG) {
            const float d0 = xc[(int64_t)r * C] - mean, d1 = xc[(int64_t)(r + BN_RG) * C] - mean;
            const float d2 = xc[(int64_t)(r + 2 * BN_RG) * C] - mean, d3 = xc[(int64_t)(r + 3 * BN_RG) * C] - mean;
            q0 += d0 * d0; q1 += d1 * d1; q2 += d2 * d2; q3 += d3 * d3;
        }
        for (; r < B; r += BN_RG) {
            const float d = xc[(int64_t)r * C] - mean;
            q0 += d * d;
        }
    }
    const float var = bn_col_reduce((q0 + q1) + (q2 + q3), red, rg, cl) / (float)B;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (!ok) return;
    const float g = gamma[c], bt = beta[c];
    for (int r = rg; r < B; r += BN_RG) y[(int64_t)r * C + c] = (xc[(int64_t)r * C] - mean) * rstd * g + bt;
    if (rg == 0) {
        if (save_mean) save_mean[c] = mean;
        if (save_rstd) save_rstd[c] = rstd;
        if (running_mean) {
            const float unbiased = var * ((float)B / (float)(B - 1));
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
}

CAPDEC_HOOK_VAR(g_attention_no_long);   // tests: 1 keeps long memories on the one-wavefront-per-row kernel
CAPDEC_HOOK_VAR(g_attention_ragged_rows);   // tests: 1 keeps the ragged self-attention on the row-by-row kernel

}  // namespace

// ---- row-list forms used by the decode engine (device-side row count, explicit (sample, position) per row) ----
int capdec_launch_layernorm_rows(const float* x, const float* add, int add_div, const int32_t* add_map,
                                 const float* gamma, const float* beta, float eps, int Mcap, int E, float* y,
                                 const int32_t* rows_dev, hipStream_t s, const float* gamma2, const float* beta2,
                                 const float* add2, int add2_div, const int32_t* add2_map) {
    if (Mcap == 0) return 0;
#define CAPDEC_CALL(NV_)                                                                                          \
    hipLaunchKernelGGL(add_layernorm_kernel<NV_>, dim3((Mcap + 3) / 4), dim3(256), 0, s, x, add, add ? add_div : 1, \
                       gamma, beta, eps, Mcap, E, y, (float*)nullptr, (float*)nullptr, add_map, rows_dev, gamma2,   \
                       beta2, add2, add2_div, add2_map)
    CAPDEC_LN_DISPATCH(E, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_launch_layernorm_pair_rows(const float* x, const float* gamma1, const float* beta1, const float* gamma2,
                                      const float* beta2, float eps, int Mcap, int E, float* y, const int32_t* rows_dev,
                                      const int32_t* row_map, int map_div, const float* G, const float* D, const float* c1,
                                      const float* sb, int H, float scale, hipStream_t s) {
    if (Mcap == 0) return 0;
#define CAPDEC_CALL(NV_)                                                                                             \
    hipLaunchKernelGGL(ln_pair_kernel<NV_>, dim3((Mcap + 3) / 4), dim3(256), 0, s, x, gamma1, beta1, gamma2, beta2, eps, \
                       Mcap, E, y, rows_dev, row_map, map_div, G, D, c1, sb, H, scale)
    CAPDEC_LN_DISPATCH(E, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_launch_attention_memory_rows(const float* qp, int64_t ldq, const float* mem, int S, int E, int H, float scale,
                                        float* ctx, int64_t ldc, int Rcap, const int32_t* row_map, int map_div,
                                        const int32_t* rows_dev, hipStream_t s) {
    CAPDEC_REQUIRE(qp && mem && ctx && S >= 1 && H >= 1 && map_div >= 1 && capdec_attention_memory_supported(E));
    CAPDEC_REQUIRE(ldq >= (int64_t)H * E && ldc >= (int64_t)H * E && ldq % 4 == 0 && ldc % 4 == 0);
    if (Rcap == 0) return 0;
#define CAPDEC_CALL(E_)                                                                                              \
    hipLaunchKernelGGL(attention_memory_kernel<E_>, dim3(Rcap), dim3(256), 0, s, qp, ldq, mem, S, H, scale, ctx, ldc,    \
                       row_map, map_div, rows_dev)
    switch (E) {
        case 64: CAPDEC_CALL(64); break;
        case 128: CAPDEC_CALL(128); break;
        case 256: CAPDEC_CALL(256); break;
        default: CAPDEC_CALL(512); break;
    }
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_launch_attention_rows(const float* q, int64_t ldq, const float* k, int64_t ldk, int64_t sk, const float* v,
                                 int64_t ldv, int64_t sv, float* out, int64_t ldo, int Rcap, int H, int Tk, int dk,
                                 int causal, const int32_t* ids, int64_t ldid, const int32_t* row_map, int map_div,
                                 const int32_t* rows_dev, hipStream_t s) {
    if (Rcap == 0) return 0;
    AttnArgs a{q, ldq, k, ldk, sk, v, ldv, sv, out, ldo, Rcap, H, 1, Tk, dk, causal, 0, ids, ldid, nullptr,
               row_map, map_div, rows_dev};
    const long total = (long)Rcap * H;
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attention_kernel<NM_>, dim3((unsigned)((total + 3) / 4)), dim3(256), 4 * Tk * sizeof(float), s, a)
    CAPDEC_DK_DISPATCH(dk, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

// the chunked kernel for long unmasked memories; *used is false when the shape is outside its range
static int launch_attention_long(const AttnArgs& a, hipStream_t s, bool* used) {
    *used = false;
    const int Tq = a.Tq, dk = a.dk;
    if (a.causal || a.ids || a.row_map || Tq < 4 || Tq > 64 || a.Tk <= 64 || g_attention_no_long) return 0;
    // rows per wavefront = ceil(Tq / 4), worked on in NG groups of RB rows
    const int rpw = (Tq + 3) / 4;
    const int rb = (rpw == 5 || rpw == 9 || rpw == 10) ? 5 : 4;
    const size_t smem = ((size_t)(Tq + 2 * AL_KC) * (dk + 16) + 4 * rb * AL_KC) * sizeof(float);
    if (smem > 144 * 1024) return 0;
#define CAPDEC_FORMS(NM_, X)  X(NM_, 1, 4) X(NM_, 1, 5) X(NM_, 2, 4) X(NM_, 2, 5) X(NM_, 4, 4)
    static const bool ready = [] {
        bool ok = true;
#define CAPDEC_SET(NM_, NG_, RB_)                                                                                      \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(attention_long_kernel<NM_, NG_, RB_>),                \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) == hipSuccess;
        CAPDEC_FORMS(1, CAPDEC_SET) CAPDEC_FORMS(2, CAPDEC_SET) CAPDEC_FORMS(4, CAPDEC_SET) CAPDEC_FORMS(8, CAPDEC_SET)
        CAPDEC_FORMS(16, CAPDEC_SET)
#undef CAPDEC_SET
        return ok;
    }();
    if (!ready) return CAPDEC_EINVAL;
    const dim3 grid((unsigned)(a.B * a.H));
#define CAPDEC_CALL(NM_)                                                                                       \
    do {                                                                                                       \
        if (rpw <= 4) hipLaunchKernelGGL((attention_long_kernel<NM_, 1, 4>), grid, dim3(256), smem, s, a);     \
        else if (rpw == 5) hipLaunchKernelGGL((attention_long_kernel<NM_, 1, 5>), grid, dim3(256), smem, s, a); \
        else if (rpw <= 8) hipLaunchKernelGGL((attention_long_kernel<NM_, 2, 4>), grid, dim3(256), smem, s, a); \
        else if (rpw <= 10) hipLaunchKernelGGL((attention_long_kernel<NM_, 2, 5>), grid, dim3(256), smem, s, a); \
        else hipLaunchKernelGGL((attention_long_kernel<NM_, 4, 4>), grid, dim3(256), smem, s, a);              \
    } while (0)
    CAPDEC_DK_DISPATCH(dk, CAPDEC_CALL);
#undef CAPDEC_CALL
#undef CAPDEC_FORMS
    CAPDEC_LAUNCH_CHECK();
    *used = true;
    return 0;
}

extern "C" {

int capdec_abi_version(void) { return CAPDEC_ABI_VERSION; }

const char* capdec_strerror(int code) {
    switch (code) {
        case CAPDEC_OK: return "ok";
        case CAPDEC_EINVAL: return "capdec: invalid argument (shape, null pointer or unsupported size)";
        case CAPDEC_ENOMEM: return "capdec: workspace too small";
        case CAPDEC_EALIGN: return "capdec: pointer or stride not 16-byte aligned";
        case CAPDEC_E2BIG: return "capdec: problem exceeds the fixed capacity of this allocation-free entry point";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "capdec: unknown error";
    }
}

int capdec_embed_pe(const void* ids, int ids32, int64_t ldid, const float* emb, const float* pe, int pos0, int B, int T,
                    int E, int V, float* x, void* stream) {
    CAPDEC_REQUIRE(ids && emb && x && B >= 0 && T >= 1 && E >= 4 && E % 4 == 0 && V >= 1 && pos0 >= 0 && ldid >= T);
    if (!capdec_aligned16(emb) || !capdec_aligned16(x) || (pe && !capdec_aligned16(pe))) return CAPDEC_EALIGN;
    const int rows = B * T;
    if (rows == 0) return 0;
    dim3 grid((rows + 3) / 4), block(256);
    if (ids32)
        hipLaunchKernelGGL(embed_pe_kernel<int32_t>, grid, block, 0, (hipStream_t)stream, (const int32_t*)ids, ldid,
                           emb, pe, pos0, rows, T, E, V, x);
    else
        hipLaunchKernelGGL(embed_pe_kernel<int64_t>, grid, block, 0, (hipStream_t)stream, (const int64_t*)ids, ldid,
                           emb, pe, pos0, rows, T, E, V, x);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_add_layernorm(const float* x, const float* add, int add_div, const float* gamma, const float* beta,
                         float eps, int M, int E, float* y, float* mean, float* rstd, void* stream) {
    CAPDEC_REQUIRE(x && gamma && beta && y && M >= 0 && E >= 4 && E % 4 == 0 && E <= 256 * LN_MAXV);
    CAPDEC_REQUIRE(!add || add_div >= 1);
    if (!capdec_aligned16(x) || !capdec_aligned16(y) || !capdec_aligned16(gamma) || !capdec_aligned16(beta) ||
        (add && !capdec_aligned16(add)))
        return CAPDEC_EALIGN;
    if (M == 0) return 0;
#define CAPDEC_CALL(NV_)                                                                                         \
    hipLaunchKernelGGL(add_layernorm_kernel<NV_>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, add,    \
                       add ? add_div : 1, gamma, beta, eps, M, E, y, mean, rstd, (const int32_t*)nullptr,          \
                       (const int32_t*)nullptr)
    CAPDEC_LN_DISPATCH(E, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_attention(const float* q, int64_t ldq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                     int64_t sv, float* out, int64_t ldo, int B, int H, int Tq, int Tk, int dk, int causal,
                     int causal_off, const int32_t* ids, int64_t ldid, float* lse, void* stream) {
    CAPDEC_REQUIRE(q && k && v && out && B >= 0 && H >= 1 && Tq >= 1 && Tk >= 1 && Tk <= 2048);
    CAPDEC_REQUIRE(dk == 16 || dk == 32 || dk == 64 || dk == 128 || dk == 256);
    CAPDEC_REQUIRE(!ids || ldid >= Tq);
    if (!capdec_aligned16(q) || !capdec_aligned16(k) || !capdec_aligned16(v) || !capdec_aligned16(out) || ldq % 4 ||
        ldk % 4 || sk % 4 || ldv % 4 || sv % 4 || ldo % 4)
        return CAPDEC_EALIGN;
    if (B == 0) return 0;
    AttnArgs a{q, ldq, k, ldk, sk, v, ldv, sv, out, ldo, B, H, Tq, Tk, dk, causal, causal_off, ids, ldid, lse,
               nullptr, 1, nullptr};
    // several query rows per head and a short key list: stage K/V of each (b,h) once in LDS
    const size_t short_lds = ((size_t)2 * Tk * (dk + 16) + 4 * (size_t)Tk) * sizeof(float);
    if (Tq >= 4 && Tk <= 64 && short_lds <= 64 * 1024) {
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attention_short_kernel<NM_>, dim3((unsigned)(B * H)), dim3(256), short_lds, (hipStream_t)stream, a)
        CAPDEC_DK_DISPATCH(dk, CAPDEC_CALL);
#undef CAPDEC_CALL
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    bool used_long = false;
    CAPDEC_TRY(launch_attention_long(a, (hipStream_t)stream, &used_long));
    if (used_long) return 0;
    const long total = (long)B * H * Tq;
#define CAPDEC_CALL(NM_)                                                                                          \
    hipLaunchKernelGGL(attention_kernel<NM_>, dim3((unsigned)((total + 3) / 4)), dim3(256), 4 * Tk * sizeof(float), \
                       (hipStream_t)stream, a)
    CAPDEC_DK_DISPATCH(dk, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_attention_ragged(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            float* out, int64_t ldo, int B, int H, int T, int dk, const int32_t* ids,
                            const int32_t* seq_off, float* lse, void* stream) {
    CAPDEC_REQUIRE(q && k && v && out && seq_off && B >= 0 && H >= 1 && T >= 1 && T <= 64);
    CAPDEC_REQUIRE(dk == 16 || dk == 32 || dk == 64 || dk == 128 || dk == 256);
    if (!capdec_aligned16(q) || !capdec_aligned16(k) || !capdec_aligned16(v) || !capdec_aligned16(out) || ldq % 4 ||
        ldk % 4 || ldv % 4 || ldo % 4)
        return CAPDEC_EALIGN;
    if (B == 0) return 0;
    const size_t short_lds = ((size_t)2 * T * (dk + 16) + 4 * (size_t)T) * sizeof(float);
    CAPDEC_REQUIRE(short_lds <= 64 * 1024);
    AttnArgs a{q, ldq, k, ldk, 0, v, ldv, 0, out, ldo, B, H, T, T, dk, 1, 0, ids, 0, lse, nullptr, 1, nullptr, seq_off};
    if (dk == AFM_DK && T <= AFM_T && !g_attention_ragged_rows) {
        // the caption decoder's own shape: scores and context on the matrix cores
        static const bool ready_m = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)afm_lds_bytes(AFM_T)) == hipSuccess;
        if (!ready_m) return CAPDEC_EINVAL;
        hipLaunchKernelGGL(attention_mfma_kernel, dim3((unsigned)(B * H)), dim3(256), afm_lds_bytes(T), (hipStream_t)stream, a);
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attention_short_kernel<NM_>, dim3((unsigned)(B * H)), dim3(256), short_lds, (hipStream_t)stream, a)
    CAPDEC_DK_DISPATCH(dk, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_cross_attention_ragged(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                  float* out, int64_t ldo, int B, int H, int T, int S, int dk, const int32_t* seq_off,
                                  float* lse, void* stream) {
    CAPDEC_REQUIRE(q && k && v && out && seq_off && B >= 0 && H >= 1 && T >= 1 && T <= 64 && S >= 1 && S <= 2048);
    CAPDEC_REQUIRE(dk == 16 || dk == 32 || dk == 64 || dk == 128 || dk == 256);
    if (!capdec_aligned16(q) || !capdec_aligned16(k) || !capdec_aligned16(v) || !capdec_aligned16(out) || ldq % 4 ||
        ldk % 4 || ldv % 4 || ldo % 4)
        return CAPDEC_EALIGN;
    if (B == 0) return 0;
    AttnArgs a{q, ldq, k, ldk, (int64_t)S * ldk, v, ldv, (int64_t)S * ldv, out, ldo, B, H, T, S, dk, 0, 0, nullptr, 0, lse,
               nullptr, 1, nullptr, seq_off, 1};
    if (S > 64) {
        CAPDEC_REQUIRE(T >= 4 && !g_attention_no_long);       // long memories: the chunked kernel only
        bool used_long = false;
        CAPDEC_TRY(launch_attention_long(a, (hipStream_t)stream, &used_long));
        return used_long ? 0 : CAPDEC_EINVAL;
    }
    const size_t short_lds = ((size_t)2 * S * (dk + 16) + 4 * (size_t)S) * sizeof(float);
    CAPDEC_REQUIRE(short_lds <= 64 * 1024);
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attention_short_kernel<NM_>, dim3((unsigned)(B * H)), dim3(256), short_lds, (hipStream_t)stream, a)
    CAPDEC_DK_DISPATCH(dk, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

#ifdef CAPDEC_TUNING_HOOKS
int capdec_attention_force_row_kernel(int on) {
    g_attention_no_long = on ? 1 : 0;
    return 0;
}
int capdec_attention_ragged_force_row_form(int on) {
    g_attention_ragged_rows = on ? 1 : 0;
    return 0;
}
#endif

int capdec_softmax_rows(float* s, int64_t ld, int rows, int cols, int pad_to, const int32_t* live, int live_idx, int mult,
                        void* stream) {
    CAPDEC_REQUIRE(s && rows >= 0 && cols >= 1 && pad_to >= cols && ld >= pad_to && live_idx >= 0 && mult >= 1);
    if (rows == 0) return 0;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, s, ld, rows, cols,
                       pad_to, live, live_idx, mult);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_softmax_rows_bwd(const float* p, float* dp, int64_t ld, int rows, int cols, int pad_to, float scale,
                            const int32_t* live, int live_idx, int mult, void* stream) {
    CAPDEC_REQUIRE(p && dp && rows >= 0 && cols >= 1 && pad_to >= cols && ld >= pad_to && live_idx >= 0 && mult >= 1);
    if (rows == 0) return 0;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, p, dp, ld, rows,
                       cols, pad_to, scale, live, live_idx, mult);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                   float eps, int C, float* scale, float* shift, void* stream) {
    CAPDEC_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C >= 1);
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_bn_train_forward(const float* x, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, int B, int C, float* y, float* save_mean,
                            float* save_rstd, void* stream) {
    CAPDEC_REQUIRE(x && gamma && beta && y && B >= 2 && C >= 1);
    CAPDEC_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    hipLaunchKernelGGL(bn_train_kernel, dim3((C + BN_CW - 1) / BN_CW), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps,
                       momentum, running_mean, running_var, B, C, y, save_mean, save_rstd);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
