// Training-side row kernels for gfx950: dropout, LayerNorm (forward with saved statistics, backward),
// attention backward, embedding scatter-add, BatchNorm1d backward, column sums for bias gradients, Adam.
// All are HBM-bound: one 64-lane wavefront per row, 16-byte accesses, shuffle reductions; cross-row
// reductions (dgamma/dbeta/bias gradients) accumulate per workgroup and finish with float atomics.
#include "common.h"
#include <limits.h>
#include <math.h>

namespace {

// Counter-based dropout masks: one splitmix64 of (seed, element index / 4) yields four 16-bit uniforms, one per
// element of an aligned group of four; an element is kept when its 16 bits are >= round(p * 65536).  The same
// (seed, index) gives the same mask in the forward and the backward pass, so masks are never stored.
__device__ __forceinline__ uint64_t rng_bits(uint64_t seed, uint64_t q) {
    uint64_t z = seed + (q + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned drop_threshold(float p) { return (unsigned)(p * 65536.0f + 0.5f); }
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
    const unsigned u = (unsigned)(rng_bits(seed, idx >> 2) >> (16 * (idx & 3))) & 0xFFFFu;
    return u >= drop_threshold(p) ? inv_keep : 0.f;
}
// the four scales of the aligned element group starting at idx0 (idx0 % 4 == 0)
__device__ __forceinline__ f32x4 drop_scale4(uint64_t seed, uint64_t idx0, float p, float inv_keep) {
    const uint64_t bits = rng_bits(seed, idx0 >> 2);
    const unsigned thr = drop_threshold(p);
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = ((unsigned)(bits >> (16 * k)) & 0xFFFFu) >= thr ? inv_keep : 0.f;
    return r;
}

// seed_dev (nullable): a device-resident seed added to the by-value one, so that a step captured in a HIP graph draws
// new masks at every replay (the by-value part then only tells the dropout sites of one step apart)
__global__ void dropout_kernel(const float* __restrict__ x, int64_t n, float p, uint64_t seed,
                               const uint64_t* __restrict__ seed_dev, float* __restrict__ y) {
    if (seed_dev) seed += seed_dev[0];
    const float inv_keep = 1.0f / (1.0f - p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[i] * drop_scale(seed, (uint64_t)i, p, inv_keep);
}
__global__ void dropout4_kernel(const f32x4* __restrict__ x, int64_t n4, float p, uint64_t seed,
                                const uint64_t* __restrict__ seed_dev, f32x4* __restrict__ y) {
    if (seed_dev) seed += seed_dev[0];
    const float inv_keep = 1.0f / (1.0f - p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[i] * drop_scale4(seed, (uint64_t)i * 4, p, inv_keep);
}

// ---- LayerNorm forward for training: s = add + dropout(x); y = LN(s) ---------------------------------
constexpr int LN_MAXV = 8;

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
__global__ __launch_bounds__(256) void ln_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int M, int E,
                                                           int x_div, float p, uint64_t seed,
                                                           const uint64_t* __restrict__ seed_dev, float* __restrict__ y,
                                                           float* __restrict__ s_out, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out,
                                                           const int32_t* __restrict__ x_map,
                                                           const int32_t* __restrict__ rows_dev) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M || (rows_dev && row >= rows_dev[0])) return;
    if (seed_dev) seed += seed_dev[0];
    const int nv = E / 4;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (int64_t)(x_map ? x_map[row] : row / x_div) * E);
    const f32x4* ar = add ? reinterpret_cast<const f32x4*>(add + (int64_t)row * E) : nullptr;
    const float inv_keep = 1.0f / (1.0f - p);
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            v[u] = xr[c];
            if (p > 0.f) {
                v[u] *= drop_scale4(seed, (uint64_t)row * E + 4 * c, p, inv_keep);
            }
            if (ar) v[u] += ar[c];
            sum += v[u][0] + v[u][1] + v[u][2] + v[u][3];
        }
    }
    const float mean = wave_sum(sum) / (float)E;
    if (s_out) {
        f32x4* sr = reinterpret_cast<f32x4*>(s_out + (int64_t)row * E);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = lane + 64 * u;
            if (c < nv) sr[c] = v[u];
        }
    }
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
    f32x4* yr = reinterpret_cast<f32x4*>(y + (int64_t)row * E);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
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

// ---- LayerNorm backward -----------------------------------------------------------------------------
// ds = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma, xhat = (s - mean) * rstd
// d_add = ds ; d_x = ds * dropout mask ; dgamma += sum_r dy * xhat ; dbeta += sum_r dy
// Rows per workgroup (LNB_ROWS, a launch argument): 32 (8 per wavefront) for batches of thousands of rows (16 and 64 both
// measured slower there); fewer, down to 8, when that would leave most of the chip without a workgroup.

// The column sums (dgamma, dbeta, the bias gradient dxsum) of a workgroup's rows meet in 3*E floats of LDS (LDS atomic
// adds: 6 KiB at E = 512, so that the kernel finds room on a CU beside the 72 KiB tiles of a weight-gradient contraction
// running on the second stream).  From there they go to memory either as one float atomic per column and workgroup, or
// (`partials` given) as this workgroup's row of a [workgroups][3*E] slab that ln_bwd_reduce_kernel sums afterwards:
// 320 workgroups x 1536 columns of same-address atomics are a 16 us tail (about 50 ns each), the slab is not.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, int M, int E, float p,
                                                     uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                     float* __restrict__ d_add, float* __restrict__ d_x,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ dxsum, const int32_t* __restrict__ rows_dev,
                                                     int LNB_ROWS, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [3][E]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (rows_dev) M = min(M, rows_dev[0]);
    if (seed_dev) seed += seed_dev[0];
    float* prow = partials ? partials + (int64_t)blockIdx.x * 3 * E : nullptr;
    if ((int)blockIdx.x * LNB_ROWS >= M) {
        if (prow)                                   // no live row here: the reduction still reads this row of the slab
            for (int c = threadIdx.x; c < 3 * E; c += 256) prow[c] = 0.f;
        return;
    }
    for (int c = threadIdx.x; c < 3 * E; c += 256) lds[c] = 0.f;
    __syncthreads();
    const int nv = E / 4;
    const float inv_keep = 1.0f / (1.0f - p);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    f32x4 gam[NV], ag[NV], ab[NV], ax[NV];     // ax: column sums of d_x (the bias gradient of the layer that made x)
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        gam[u] = c < nv ? g4[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        ag[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        ax[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int r0 = blockIdx.x * LNB_ROWS;
    for (int rr = wave; rr < LNB_ROWS; rr += 4) {
        const int row = r0 + rr;
        if (row >= M) break;
        const f32x4* dyr = reinterpret_cast<const f32x4*>(dy + (int64_t)row * E);
        const f32x4* sr = reinterpret_cast<const f32x4*>(s + (int64_t)row * E);
        const float mu = mean[row], rs = rstd[row];
        f32x4 g[NV], xh[NV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = lane + 64 * u;
            g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            xh[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nv) {
                const f32x4 d = dyr[c];
                xh[u] = (sr[c] - mu) * rs;
                ag[u] += d * xh[u];
                ab[u] += d;
                g[u] = d * gam[u];
                c1 += g[u][0] + g[u][1] + g[u][2] + g[u][3];
                c2 += g[u][0] * xh[u][0] + g[u][1] * xh[u][1] + g[u][2] * xh[u][2] + g[u][3] * xh[u][3];
            }
        }
        c1 = wave_sum(c1) / (float)E;
        c2 = wave_sum(c2) / (float)E;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = lane + 64 * u;
            if (c < nv) {
                f32x4 ds = (g[u] - c1 - xh[u] * c2) * rs;
                if (d_add) reinterpret_cast<f32x4*>(d_add + (int64_t)row * E)[c] = ds;
                if (d_x || dxsum) {
                    if (p > 0.f) {
                        ds *= drop_scale4(seed, (uint64_t)row * E + 4 * c, p, inv_keep);
                    }
                    if (d_x) reinterpret_cast<f32x4*>(d_x + (int64_t)row * E)[c] = ds;
                    ax[u] += ds;
                }
            }
        }
    }
    // the 4 wavefronts' column partials meet in LDS (wavefront after wavefront in a fixed order: the sums, and with the
    // slab the gradients, are the same from run to run)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int c = lane + 64 * u;
                if (c < nv) {
                    f32x4* a0 = reinterpret_cast<f32x4*>(lds) + c;
                    a0[0] += ag[u];
                    a0[nv] += ab[u];
                    a0[2 * nv] += ax[u];
                }
            }
        }
        __syncthreads();
    }
    if (prow) {
        for (int c = threadIdx.x; c < 3 * nv; c += 256) reinterpret_cast<f32x4*>(prow)[c] = reinterpret_cast<const f32x4*>(lds)[c];
        return;
    }
    for (int c = threadIdx.x; c < E; c += 256) {
        atomicAdd(&dgamma[c], lds[c]);
        atomicAdd(&dbeta[c], lds[E + c]);
        if (dxsum) atomicAdd(&dxsum[c], lds[2 * E + c]);
    }
}

// Second stage: out_k[c] += sum over the nwg rows of the slab, k = 0, 1, 2 for dgamma, dbeta, dxsum (nullable).  A
// workgroup owns 64 columns (16 lanes x 16 B) of one of the three and its 16 row groups walk the slab rows, eight in
// flight per thread, in a fixed order.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ partials, int nwg, int E,
                                                            float* __restrict__ o0, float* __restrict__ o1,
                                                            float* __restrict__ o2) {
    __shared__ f32x4 red[16][16];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int strips = (E + 63) / 64;
    const int k = blockIdx.x / strips, j = (blockIdx.x - k * strips) * 64 + cl * 4;
    float* out = k == 0 ? o0 : (k == 1 ? o1 : o2);
    if (!out) return;
    f32x4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (j < E) {
        for (int r = rg; r < nwg; r += 128) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = r + 16 * u;
                if (rr < nwg) acc[u] += *reinterpret_cast<const f32x4*>(partials + ((int64_t)rr * 3 + k) * E + j);
            }
        }
    }
    red[rg][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (rg == 0 && j < E) {
        f32x4 t = red[0][cl];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += red[g][cl];
        f32x4* o = reinterpret_cast<f32x4*>(out + j);
        *o = *o + t;
    }
}

// ---- column sums: out[j] += sum_i x[i*ldx + j]  (bias gradients) -------------------------------------
// A workgroup sums CS_ROWS rows of a 64-column strip: 16 lanes x 16 B per row and 16 row groups, eight
// independent rows in flight per thread, one atomic per column and workgroup.  Same-address float atomics
// serialise (about 50 ns each), so the strip is kept narrow and tall: few partial sums per column.
constexpr int CS_ROWS = 256;

template <bool VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ldx, int M, int N,
                                                     float* __restrict__ out, const int32_t* __restrict__ rows_dev) {
    __shared__ f32x4 red[16][16];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j = blockIdx.x * 64 + cl * 4;
    const int r0 = blockIdx.y * CS_ROWS;
    if (rows_dev) M = min(M, rows_dev[0]);
    if (r0 >= M) return;
    const int r1 = min(M, r0 + CS_ROWS);
    f32x4 s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (j < N) {
        for (int r = r0 + rg; r < r1; r += 128) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rr = r + 16 * u;
                if (rr < r1) {
                    const float* p = x + (int64_t)rr * ldx + j;
                    if (VEC) s[u] += *reinterpret_cast<const f32x4*>(p);
                    else
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (j + c < N) s[u][c] += p[c];
                }
            }
        }
    }
    red[rg][cl] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (rg == 0 && j < N) {
        f32x4 t = red[0][cl];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += red[g][cl];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (j + c < N) atomicAdd(&out[j + c], t[c]);
    }
}

// ---- cross-attention over exactly two memory rows, training form ------------------------------------------------------
// The softmax over two keys is a logistic function of the score difference.  With dK = k0 - k1, dV = v0 - v1 (the
// projected memory rows of a sample) the block's output for a row x1 of sample b is
//     a2 = c1[b] + sum_h p_h D[b,h],   p_h = sigmoid((x1 . G[b,h] + sb[b,h]) * scale),
//     G[b,h] = W_q,h^T dK_h,  sb[b,h] = b_q,h . dK_h,  D[b,h] = W_o[:,h] dV_h,  c1[b] = W_o v1 + b_o
// (transformer.py:21-27,45-60,86 with Tk = 2): H dot products and H scaled vector additions per row instead of two
// E x E contractions over all rows and an attention launch.  Everything with a [b] is B rows of work.
// Forward: one wavefront per row.  sample: row -> sample (nullable: row / T).
template <int NV>
__global__ __launch_bounds__(256) void pair_gate_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ G,
                                                            const float* __restrict__ D, const float* __restrict__ c1,
                                                            const float* __restrict__ sb, const int32_t* __restrict__ sample,
                                                            int T, int M, int E, int H, float scale,
                                                            float* __restrict__ a2, float* __restrict__ p_out,
                                                            const int32_t* __restrict__ rows_dev) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M || (rows_dev && row >= rows_dev[0])) return;
    const int nv = E / 4;
    const int b = sample ? sample[row] : row / T;
    f32x4 x[NV], o[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        x[u] = c < nv ? reinterpret_cast<const f32x4*>(x1 + (int64_t)row * E)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        o[u] = c < nv ? reinterpret_cast<const f32x4*>(c1 + (int64_t)b * E)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int h = 0; h < H; ++h) {
        const f32x4* g = reinterpret_cast<const f32x4*>(G + ((int64_t)b * H + h) * E);
        const f32x4* d = reinterpret_cast<const f32x4*>(D + ((int64_t)b * H + h) * E);
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = lane + 64 * u;
            if (c < nv) {
                const f32x4 gv = g[c];
                part += x[u][0] * gv[0] + x[u][1] * gv[1] + x[u][2] * gv[2] + x[u][3] * gv[3];
            }
        }
        const float z = (wave_sum(part) + sb[(int64_t)b * H + h]) * scale;
        const float p = 1.0f / (1.0f + expf(-z));
        if (lane == 0) p_out[(int64_t)row * H + h] = p;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = lane + 64 * u;
            if (c < nv) o[u] += d[c] * p;
        }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = lane + 64 * u;
        if (c < nv) reinterpret_cast<f32x4*>(a2 + (int64_t)row * E)[c] = o[u];
    }
}

// Backward: one workgroup per sample (its rows [r0, r1): seg_off, or T consecutive rows), a wavefront per row in turn.
//   dp_h = d_a2 . D_h;  dz_h = dp_h p_h (1 - p_h) scale;  d_x1 += sum_h dz_h G_h;
//   dG_h = sum_rows dz_h x1;  dD_h = sum_rows p_h d_a2;  dc1 = sum_rows d_a2;  dsb_h = sum_rows dz_h.
// Heads in passes of four (register budget); the four wavefronts' partial sums meet in LDS one wavefront at a time.
template <int NV>
__global__ __launch_bounds__(256) void pair_gate_bwd_kernel(const float* __restrict__ d_a2, const float* __restrict__ x1,
                                                            const float* __restrict__ p_in, const float* __restrict__ G,
                                                            const float* __restrict__ D, const int32_t* __restrict__ seg_off,
                                                            int T, int E, int H, float scale, float* __restrict__ d_x1,
                                                            float* __restrict__ dG, float* __restrict__ dD,
                                                            float* __restrict__ dc1, float* __restrict__ dsb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [9][E] + 4: dG x4, dD x4, dc1, dsb x4
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nv = E / 4;
    const int r0 = seg_off ? seg_off[b] : b * T, r1 = seg_off ? seg_off[b + 1] : (b + 1) * T;
    for (int h0 = 0; h0 < H; h0 += 4) {
        f32x4 g[4][NV], d[4][NV], ag[4][NV], ad[4][NV], ac[NV];
        float as[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int c = lane + 64 * u;
                const bool ok = c < nv && h0 + hh < H;
                g[hh][u] = ok ? reinterpret_cast<const f32x4*>(G + ((int64_t)b * H + h0 + hh) * E)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
                d[hh][u] = ok ? reinterpret_cast<const f32x4*>(D + ((int64_t)b * H + h0 + hh) * E)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
                ag[hh][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                ad[hh][u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int u = 0; u < NV; ++u) ac[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int row = r0 + wave; row < r1; row += 4) {
            // everything a row needs in flight together: its x1 and d_a2 values, the four gates, and the d_x1 values the
            // result is added to (loaded where they are used, each gate and the read-modify-write were a round trip of
            // their own: five per row)
            f32x4 x[NV], da[NV], dx[NV];
            float pg4[4];
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) pg4[hh] = h0 + hh < H ? p_in[(int64_t)row * H + h0 + hh] : 0.f;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int c = lane + 64 * u;
                x[u] = c < nv ? reinterpret_cast<const f32x4*>(x1 + (int64_t)row * E)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
                da[u] = c < nv ? reinterpret_cast<const f32x4*>(d_a2 + (int64_t)row * E)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
                dx[u] = c < nv ? reinterpret_cast<const f32x4*>(d_x1 + (int64_t)row * E)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
                if (h0 == 0) ac[u] += da[u];
            }
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                if (h0 + hh >= H) break;
                float part = 0.f;
#pragma unroll
                for (int u = 0; u < NV; ++u)
                    part += da[u][0] * d[hh][u][0] + da[u][1] * d[hh][u][1] + da[u][2] * d[hh][u][2] + da[u][3] * d[hh][u][3];
                const float p = pg4[hh];
                const float dz = wave_sum(part) * p * (1.0f - p) * scale;
                as[hh] += dz;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    ag[hh][u] += x[u] * dz;
                    ad[hh][u] += da[u] * p;
                    dx[u] += g[hh][u] * dz;
                }
            }
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int c = lane + 64 * u;
                if (c < nv) reinterpret_cast<f32x4*>(d_x1 + (int64_t)row * E)[c] = dx[u];
            }
        }
        // partial sums of the four wavefronts -> LDS, one wavefront at a time
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int hh = 0; hh < 4; ++hh)
#pragma unroll
                    for (int u = 0; u < NV; ++u) {
                        const int c = lane + 64 * u;
                        if (c < nv) {
                            f32x4* pg = reinterpret_cast<f32x4*>(lds + (int64_t)hh * E) + c;
                            f32x4* pd = reinterpret_cast<f32x4*>(lds + (int64_t)(4 + hh) * E) + c;
                            *pg = w == 0 ? ag[hh][u] : *pg + ag[hh][u];
                            *pd = w == 0 ? ad[hh][u] : *pd + ad[hh][u];
                        }
                    }
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const int c = lane + 64 * u;
                    if (c < nv) {
                        f32x4* pc = reinterpret_cast<f32x4*>(lds + (int64_t)8 * E) + c;
                        *pc = w == 0 ? ac[u] : *pc + ac[u];
                    }
                }
                if (lane < 4) {
                    float v = as[0];
                    if (lane == 1) v = as[1];
                    if (lane == 2) v = as[2];
                    if (lane == 3) v = as[3];
                    float* ps = lds + (int64_t)9 * E + lane;
                    *ps = w == 0 ? v : *ps + v;
                }
            }
            __syncthreads();
        }
        for (int idx = threadIdx.x; idx < 4 * nv; idx += 256) {
            const int hh = idx / nv, c = idx - hh * nv;
            if (h0 + hh < H) {
                reinterpret_cast<f32x4*>(dG + ((int64_t)b * H + h0 + hh) * E)[c] = reinterpret_cast<const f32x4*>(lds + (int64_t)hh * E)[c];
                reinterpret_cast<f32x4*>(dD + ((int64_t)b * H + h0 + hh) * E)[c] = reinterpret_cast<const f32x4*>(lds + (int64_t)(4 + hh) * E)[c];
            }
        }
        if (h0 == 0)
            for (int c = threadIdx.x; c < nv; c += 256)
                reinterpret_cast<f32x4*>(dc1 + (int64_t)b * E)[c] = reinterpret_cast<const f32x4*>(lds + (int64_t)8 * E)[c];
        if (threadIdx.x < 4 && h0 + (int)threadIdx.x < H) dsb[(int64_t)b * H + h0 + threadIdx.x] = lds[(int64_t)9 * E + threadIdx.x];
        __syncthreads();
    }
}

// dK = k0 - k1, dV = v0 - v1 of each sample's projected memory rows ([B*2][E]) and sb[b,h] = b_q,h . dK_h
__global__ __launch_bounds__(256) void pair_diff_fwd_kernel(const float* __restrict__ k2, const float* __restrict__ v2,
                                                            const float* __restrict__ bq, int B, int E, int H,
                                                            float* __restrict__ dk_out, float* __restrict__ dv_out,
                                                            float* __restrict__ sb) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int nv = E / 4, hv = nv / H;
    const f32x4* k0 = reinterpret_cast<const f32x4*>(k2 + (int64_t)(2 * b) * E);
    const f32x4* v0 = reinterpret_cast<const f32x4*>(v2 + (int64_t)(2 * b) * E);
    const f32x4* bq4 = reinterpret_cast<const f32x4*>(bq);
    for (int c = lane; c < nv; c += 64) {
        reinterpret_cast<f32x4*>(dk_out + (int64_t)b * E)[c] = k0[c] - k0[nv + c];
        reinterpret_cast<f32x4*>(dv_out + (int64_t)b * E)[c] = v0[c] - v0[nv + c];
    }
    for (int h = 0; h < H; ++h) {
        float part = 0.f;
        for (int c = h * hv + lane; c < (h + 1) * hv; c += 64) {
            const f32x4 d = k0[c] - k0[nv + c], q = bq4[c];
            part += d[0] * q[0] + d[1] * q[1] + d[2] * q[2] + d[3] * q[3];
        }
        part = wave_sum(part);
        if (lane == 0) sb[(int64_t)b * H + h] = part;
    }
}

// Backward of pair_diff_fwd and of sb: given d(dK) (without the sb term), d(dV), d(sb), d(v1):
//   d(dK)[b, h*dk + d] += dsb[b,h] b_q[h*dk + d];   dk2[2b] = d(dK)[b], dk2[2b+1] = -d(dK)[b];
//   dv2[2b] = d(dV)[b], dv2[2b+1] = d(v1)[b] - d(dV)[b];   qb[b, h*dk + d] = dsb[b,h] dK[b, h*dk + d]  (its column sum is d b_q)
__global__ __launch_bounds__(256) void pair_diff_bwd_kernel(const float* __restrict__ ddk, const float* __restrict__ ddv,
                                                            const float* __restrict__ dsb, const float* __restrict__ dv1,
                                                            const float* __restrict__ bq, const float* __restrict__ dkf,
                                                            int B, int E, int H, float* __restrict__ dk2,
                                                            float* __restrict__ dv2, float* __restrict__ qb) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int nv = E / 4, hv = nv / H;
    for (int c = lane; c < nv; c += 64) {
        const float s = dsb[(int64_t)b * H + c / hv];
        const f32x4 a = reinterpret_cast<const f32x4*>(ddk + (int64_t)b * E)[c] + reinterpret_cast<const f32x4*>(bq)[c] * s;
        reinterpret_cast<f32x4*>(dk2 + (int64_t)(2 * b) * E)[c] = a;
        reinterpret_cast<f32x4*>(dk2 + (int64_t)(2 * b + 1) * E)[c] = -a;
        const f32x4 v = reinterpret_cast<const f32x4*>(ddv + (int64_t)b * E)[c];
        reinterpret_cast<f32x4*>(dv2 + (int64_t)(2 * b) * E)[c] = v;
        reinterpret_cast<f32x4*>(dv2 + (int64_t)(2 * b + 1) * E)[c] = reinterpret_cast<const f32x4*>(dv1 + (int64_t)b * E)[c] - v;
        reinterpret_cast<f32x4*>(qb + (int64_t)b * E)[c] = reinterpret_cast<const f32x4*>(dkf + (int64_t)b * E)[c] * s;
    }
}

// ---- sums of consecutive row groups: out[g, :] = sum_{t<T} x[g*T + t, :]  (gradient of a row broadcast) ----
__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ x, int G, int T, int E,
                                                        float* __restrict__ out, const int32_t* __restrict__ seg_off) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const int nv = E / 4;
    // seg_off: group g is the rows [seg_off[g], seg_off[g+1]) instead of [g*T, (g+1)*T)
    const int64_t r0 = seg_off ? seg_off[g] : (int64_t)g * T;
    if (seg_off) T = seg_off[g + 1] - seg_off[g];
    for (int c = lane; c < nv; c += 64) {
        // eight rows in flight at a time, added in row order (one row per round trip made 20 rows 25 us)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < T; t0 += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const f32x4*>(x + (r0 + min(t0 + u, T - 1)) * E)[c];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (t0 + u < T) acc += v[u];
        }
        reinterpret_cast<f32x4*>(out + (int64_t)g * E)[c] = acc;
    }
}

// y[r,:] = (accumulate ? y[r,:] : 0) + scale * x[r / x_div, :]   (a mean from a sum; a per-sample row added to
// each of its S memory rows)
__global__ void axpy_rows_kernel(const float* __restrict__ x, int x_div, float scale, int M, int E,
                                 float* __restrict__ y, int accumulate) {
    const int nv = E / 4;
    const int64_t n = (int64_t)M * nv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / nv), c = (int)(idx - (int64_t)r * nv);
        const f32x4 xv = reinterpret_cast<const f32x4*>(x + (int64_t)(r / x_div) * E)[c] * scale;
        f32x4* o = reinterpret_cast<f32x4*>(y + (int64_t)r * E) + c;
        *o = accumulate ? *o + xv : xv;
    }
}

// ---- embedding forward of a training step: x[s,:] = dropout(emb[id,:] + pe[i,:]) for slot s, whose dense row
// r = order[s] (or s) is position i = r % T of sample r / T and whose id sits at ids[(r / T)*ldid + i]; the mask is the
// one dropout4_kernel draws for element s*E + c of the output.  One wavefront per slot.
__global__ __launch_bounds__(256) void embed_train_fwd_kernel(const int64_t* __restrict__ ids, int64_t ldid,
                                                              const int64_t* __restrict__ order,
                                                              const float* __restrict__ emb, const float* __restrict__ pe,
                                                              int M, int T, int E, int V, float p, uint64_t seed,
                                                              const uint64_t* __restrict__ seed_dev, float* __restrict__ x,
                                                              int64_t* __restrict__ ids_out) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (s >= M) return;
    if (seed_dev) seed += seed_dev[0];
    const int64_t r = order ? order[s] : s;
    const int b = (int)(r / T), i = (int)(r - (int64_t)b * T);
    const int64_t raw = ids[(int64_t)b * ldid + i];
    if (ids_out && lane == 0) ids_out[s] = raw;
    const int64_t id = raw < 0 ? 0 : (raw >= V ? V - 1 : raw);
    const f32x4* src = reinterpret_cast<const f32x4*>(emb + id * (int64_t)E);
    const f32x4* pp = pe ? reinterpret_cast<const f32x4*>(pe + (int64_t)i * E) : nullptr;
    f32x4* dst = reinterpret_cast<f32x4*>(x + (int64_t)s * E);
    const float inv_keep = 1.0f / (1.0f - p);
    for (int c = lane; c < E / 4; c += 64) {
        f32x4 v = src[c];
        if (pp) v += pp[c];
        if (p > 0.f) v *= drop_scale4(seed, (uint64_t)s * E + 4 * c, p, inv_keep);
        dst[c] = v;
    }
}

// ---- embedding backward: demb[ids[r], :] += dx[r, :] ----------------------------------------------
// Same-address float atomics serialise, and a few ids are very frequent: the reference's vocabulary reserves
// ids 0..3 for <PAD>/<SOS>/<EOS>/<UNK> (get_loader.py:28-29), <SOS> opens every caption and <PAD> fills the
// tail of most.  Rows with an id below EB_HOT are therefore summed per workgroup in LDS first and leave as one
// atomic per column; every other row goes straight to memory.  (Purely a speed heuristic: any ids are correct.)
// A wavefront walks its rows one after the other (id, then the row, then the atomics: about 4 us a row), so a small
// batch takes fewer rows per workgroup (EB_ROWS down to 8) to spread them over the chip; a large one is bound by the
// atomics' throughput and keeps 64 rows per workgroup, whose hot rows leave as fewer atomics.
constexpr int EB_HOT = 4;

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dx,
                                                        int rows, int E, int V, float* __restrict__ demb,
                                                        const int32_t* __restrict__ rows_dev, int EB_ROWS, float p,
                                                        uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    // p > 0: dx is the gradient behind a dropout of the embedding rows (embed_train_fwd_kernel): element (row, c) is
    // scaled by the mask of element row*E + c first, exactly as a dropout launch in front of this one would
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [EB_HOT][E]
    __shared__ int used[EB_HOT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (seed_dev) seed += seed_dev[0];
    const float inv_keep = 1.0f / (1.0f - p);
    if (rows_dev) rows = min(rows, rows_dev[0]);
    if ((int)blockIdx.x * EB_ROWS >= rows) return;
    for (int c = threadIdx.x; c < EB_HOT * E; c += 256) lds[c] = 0.f;
    if (threadIdx.x < EB_HOT) used[threadIdx.x] = 0;
    __syncthreads();
    // the ids of this wavefront's rows up front (lane k holds the id of its k-th row; EB_ROWS <= 64: at most 16), and the
    // next row's values in flight while the current row's atomics are issued: otherwise a row is a chain of an id load,
    // a row load and the atomics, about 4 us
    int my_id = 0;
    {
        const int row = blockIdx.x * EB_ROWS + wave + 4 * lane;
        if (wave + 4 * lane < EB_ROWS && row < rows) {
            const long raw = ids[row];
            my_id = (int)(raw < 0 ? 0 : (raw >= V ? V - 1 : raw));
        }
    }
    constexpr int EB_PF = 8;                       // prefetched floats per lane: rows of up to 512 floats
    const bool pf = E <= 64 * EB_PF;
    float nxt[EB_PF];
    auto fetch = [&](int row) {
#pragma unroll
        for (int j = 0; j < EB_PF; ++j) {
            const int c = lane + 64 * j;
            nxt[j] = (row < rows && c < E) ? dx[(int64_t)row * E + c] : 0.f;
        }
    };
    if (pf) fetch(blockIdx.x * EB_ROWS + wave);
    int k = 0;
    for (int rr = wave; rr < EB_ROWS; rr += 4, ++k) {
        const int row = blockIdx.x * EB_ROWS + rr;
        if (row >= rows) break;
        const long id = __shfl(my_id, k, 64);
        const float* src = dx + (int64_t)row * E;
        const uint64_t e0 = (uint64_t)row * E;
        if (pf) {
            float cur[EB_PF];
#pragma unroll
            for (int j = 0; j < EB_PF; ++j) cur[j] = nxt[j];
            if (rr + 4 < EB_ROWS) fetch(row + 4);
#pragma unroll
            for (int j = 0; j < EB_PF; ++j) {
                const int c = lane + 64 * j;
                cur[j] = p > 0.f ? cur[j] * drop_scale(seed, e0 + c, p, inv_keep) : cur[j];
            }
            if (id < EB_HOT) {
                if (lane == 0) used[id] = 1;
#pragma unroll
                for (int j = 0; j < EB_PF; ++j)
                    if (lane + 64 * j < E) atomicAdd(&lds[id * E + lane + 64 * j], cur[j]);
            } else {
                float* dst = demb + id * (int64_t)E;
#pragma unroll
                for (int j = 0; j < EB_PF; ++j)
                    if (lane + 64 * j < E) atomicAdd(&dst[lane + 64 * j], cur[j]);
            }
            continue;
        }
        if (id < EB_HOT) {
            if (lane == 0) used[id] = 1;
            for (int c = lane; c < E; c += 64)
                atomicAdd(&lds[id * E + c], p > 0.f ? src[c] * drop_scale(seed, e0 + c, p, inv_keep) : src[c]);
        } else {
            float* dst = demb + id * (int64_t)E;
            for (int c = lane; c < E; c += 64)
                atomicAdd(&dst[c], p > 0.f ? src[c] * drop_scale(seed, e0 + c, p, inv_keep) : src[c]);
        }
    }
    __syncthreads();
    for (int hgt = 0; hgt < EB_HOT && hgt < V; ++hgt)
        if (used[hgt])
            for (int c = threadIdx.x; c < E; c += 256) atomicAdd(&demb[(int64_t)hgt * E + c], lds[hgt * E + c]);
}

// ---- BatchNorm1d backward (training statistics) ------------------------------------------------------
// Same shape as the forward: 16 columns per workgroup, 16 threads per column striding the rows.
constexpr int BN_CW = 16, BN_RG = 16;

__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, int B, int C,
                                                     float* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta) {
    __shared__ float red[2][BN_RG][BN_CW];
    const int cl = threadIdx.x & (BN_CW - 1), rg = threadIdx.x / BN_CW;
    const int c = blockIdx.x * BN_CW + cl;
    const bool ok = c < C;
    const float mu = ok ? mean[c] : 0.f, rs = ok ? rstd[c] : 0.f;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    if (ok) {
        int r = rg;
        for (; r + BN_RG < B; r += 2 * BN_RG) {
            const float d0 = dy[(int64_t)r * C + c], d1 = dy[(int64_t)(r + BN_RG) * C + c];
            const float x0 = x[(int64_t)r * C + c], x1 = x[(int64_t)(r + BN_RG) * C + c];
            a0 += d0; a1 += d1;
            b0 += d0 * (x0 - mu) * rs; b1 += d1 * (x1 - mu) * rs;
        }
        for (; r < B; r += BN_RG) {
            const float d = dy[(int64_t)r * C + c];
            a0 += d;
            b0 += d * (x[(int64_t)r * C + c] - mu) * rs;
        }
    }
    red[0][rg][cl] = a0 + a1;
    red[1][rg][cl] = b0 + b1;
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int g = 0; g < BN_RG; ++g) {
        s1 += red[0][g][cl];
        s2 += red[1][g][cl];
    }
    if (!ok) return;
    const float k = gamma[c] * rs;
    for (int r = rg; r < B; r += BN_RG) {
        const float xh = (x[(int64_t)r * C + c] - mu) * rs;
        dx[(int64_t)r * C + c] = k * (dy[(int64_t)r * C + c] - s1 / (float)B - xh * s2 / (float)B);
    }
    if (rg == 0) {
        dgamma[c] += s2;
        dbeta[c] += s1;
    }
}

// ---- attention backward --------------------------------------------------------------------------------
struct AttnBwdArgs {
    const float* q; int64_t ldq;
    const float* k; int64_t ldk, sk;
    const float* v; int64_t ldv, sv;
    const float* dout; int64_t ldo;
    const float* out; int64_t ldout;   // forward output (nullable): D_i = dO_i . O_i without a sweep over the keys
    const float* lse;            // [B,H,Tq]
    float* dsum;                 // [B,H,Tq]  D_i = sum_j p_ij dp_ij (written by the query pass)
    float* dq; int64_t lddq;
    float* dk; int64_t lddk, sdk;
    float* dv; int64_t lddv, sdv;
    int B, H, Tq, Tk, dk_;
    int causal, causal_off;
    const int32_t* ids; int64_t ldid;
    // ragged batch (fused kernel only): sample b is the rows [seq_off[b], seq_off[b+1]) of q, dout, dq and ids -- and of
    // k, v, dk, dv too (self-attention) unless ragged_q_only, where those stay Tk dense rows per sample (cross-attention)
    const int32_t* seq_off;
    int ragged_q_only;
};

template <int NM>
__device__ __forceinline__ float dot_parts(const f32x4 (&a)[NM], const float* __restrict__ p) {
    float d = 0.f;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + 16 * m);
        d += a[m][0] * b[0] + a[m][1] * b[1] + a[m][2] * b[2] + a[m][3] * b[3];
    }
    return d;
}
template <int NM>
__device__ __forceinline__ void load_parts(f32x4 (&a)[NM], const float* __restrict__ p) {
#pragma unroll
    for (int m = 0; m < NM; ++m) a[m] = *reinterpret_cast<const f32x4*>(p + 16 * m);
}
__device__ __forceinline__ float quad_sum(float d) {
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    return d;
}
__device__ __forceinline__ void group_reduce(f32x4& acc, int lpv) {
    for (int o = lpv; o < 64; o <<= 1) {
        acc[0] += __shfl_xor(acc[0], o, 64);
        acc[1] += __shfl_xor(acc[1], o, 64);
        acc[2] += __shfl_xor(acc[2], o, 64);
        acc[3] += __shfl_xor(acc[3], o, 64);
    }
}

// query pass: one wavefront per (b,h,i): D_i and dq_i
template <int NM>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long total = (long)a.B * a.H * a.Tq;
    long idx = (long)blockIdx.x * 4 + wave;
    const bool live = idx < total;
    if (!live) idx = total - 1;
    const int i = (int)(idx % a.Tq), h = (int)((idx / a.Tq) % a.H), b = (int)(idx / ((long)a.Tq * a.H));
    float* ps = lds + wave * 2 * a.Tk;      // p_ij
    float* dps = ps + a.Tk;                 // dp_ij = dO_i . v_j
    const int dk = a.dk_;
    const bool pad_row = a.ids && a.ids[(int64_t)b * a.ldid + i] == 0;
    int nkeys = a.Tk;
    if (a.causal && !pad_row) {
        nkeys = i + a.causal_off + 1;
        nkeys = nkeys < 1 ? 1 : (nkeys > a.Tk ? a.Tk : nkeys);
    }
    const float scale = sqrtf((float)dk);
    const int64_t qoff = ((int64_t)b * a.Tq + i);
    const float* kb = a.k + (int64_t)b * a.sk + (int64_t)h * dk;
    const float* vb = a.v + (int64_t)b * a.sv + (int64_t)h * dk;
    float D = 0.f;
    if (!pad_row) {
        // 4 lanes per key, 16 keys per sweep step; q_i and dO_i stay in registers
        const int part = lane & 3, kl = lane >> 2;
        f32x4 qreg[NM], oreg[NM];
        load_parts(qreg, a.q + qoff * a.ldq + (int64_t)h * dk + 4 * part);
        load_parts(oreg, a.dout + qoff * a.ldo + (int64_t)h * dk + 4 * part);
        const float l = a.lse[idx];
        float dacc = 0.f;
        for (int j0 = 0; j0 < nkeys; j0 += 16) {
            const int j = j0 + kl;
            float s = 0.f, dp = 0.f;
            if (j < nkeys) {
                s = dot_parts(qreg, kb + (int64_t)j * a.ldk + 4 * part);
                dp = dot_parts(oreg, vb + (int64_t)j * a.ldv + 4 * part);
            }
            s = quad_sum(s);
            dp = quad_sum(dp);
            if (j < nkeys && part == 0) {
                const float p = expf(s / scale - l);
                ps[j] = p;
                dps[j] = dp;
                dacc += p * dp;
            }
        }
        D = wave_sum(dacc);
    }
    __syncthreads();
    // dq_i = sum_j p_ij (dp_ij - D_i) k_j / sqrt(dk); a PAD row's softmax is constant, so its dq is zero
    const int lpv = dk / 4, grp = lane / lpv, dl = lane - grp * lpv, ngrp = 64 / lpv;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!pad_row)
        for (int j = grp; j < nkeys; j += ngrp)
            acc += *reinterpret_cast<const f32x4*>(kb + (int64_t)j * a.ldk + 4 * dl) * (ps[j] * (dps[j] - D) / scale);
    group_reduce(acc, lpv);
    if (live && grp == 0) *reinterpret_cast<f32x4*>(a.dq + qoff * a.lddq + (int64_t)h * dk + 4 * dl) = acc;
    if (live && lane == 0) a.dsum[idx] = D;
}

// key pass: one wavefront per (b,h,j): dk_j and dv_j
template <int NM>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long total = (long)a.B * a.H * a.Tk;
    long idx = (long)blockIdx.x * 4 + wave;
    const bool live = idx < total;
    if (!live) idx = total - 1;
    const int j = (int)(idx % a.Tk), h = (int)((idx / a.Tk) % a.H), b = (int)(idx / ((long)a.Tk * a.H));
    float* pp = lds + wave * 2 * a.Tq;     // p_ij for i in [0,Tq)
    float* dd = pp + a.Tq;                 // ds_ij
    const int dk = a.dk_;
    const float scale = sqrtf((float)dk);
    const int part = lane & 3, il = lane >> 2;
    const float* kj = a.k + (int64_t)b * a.sk + (int64_t)j * a.ldk + (int64_t)h * dk;
    const float* vj = a.v + (int64_t)b * a.sv + (int64_t)j * a.ldv + (int64_t)h * dk;
    f32x4 kreg[NM], vreg[NM];
    load_parts(kreg, kj + 4 * part);
    load_parts(vreg, vj + 4 * part);
    for (int i0 = 0; i0 < a.Tq; i0 += 16) {
        const int i = i0 + il;
        float s = 0.f, dp = 0.f;
        bool pad_row = false, seen = false;
        if (i < a.Tq) {
            pad_row = a.ids && a.ids[(int64_t)b * a.ldid + i] == 0;
            seen = pad_row || !a.causal || (j <= i + a.causal_off);
            if (seen && !pad_row) {
                const int64_t qoff = (int64_t)b * a.Tq + i;
                s = dot_parts(kreg, a.q + qoff * a.ldq + (int64_t)h * dk + 4 * part);
                dp = dot_parts(vreg, a.dout + qoff * a.ldo + (int64_t)h * dk + 4 * part);
            }
        }
        s = quad_sum(s);
        dp = quad_sum(dp);
        if (i < a.Tq && part == 0) {
            float p = 0.f, d = 0.f;
            if (pad_row) {
                p = 1.0f / (float)a.Tk;            // uniform row: no dependence on q or k
            } else if (seen) {
                const long ri = ((long)b * a.H + h) * a.Tq + i;
                p = expf(s / scale - a.lse[ri]);
                d = p * (dp - a.dsum[ri]);
            }
            pp[i] = p;
            dd[i] = d;
        }
    }
    __syncthreads();
    const int lpv = dk / 4, grp = lane / lpv, dl = lane - grp * lpv, ngrp = 64 / lpv;
    f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
    for (int i = grp; i < a.Tq; i += ngrp) {
        const int64_t qoff = (int64_t)b * a.Tq + i;
        const float p = pp[i], d = dd[i];
        if (p != 0.f) av += *reinterpret_cast<const f32x4*>(a.dout + qoff * a.ldo + (int64_t)h * dk + 4 * dl) * p;
        if (d != 0.f) ak += *reinterpret_cast<const f32x4*>(a.q + qoff * a.ldq + (int64_t)h * dk + 4 * dl) * (d / scale);
    }
    group_reduce(ak, lpv);
    group_reduce(av, lpv);
    if (live && grp == 0) {
        *reinterpret_cast<f32x4*>(a.dk + (int64_t)b * a.sdk + (int64_t)j * a.lddk + (int64_t)h * dk + 4 * dl) = ak;
        *reinterpret_cast<f32x4*>(a.dv + (int64_t)b * a.sdv + (int64_t)j * a.lddv + (int64_t)h * dk + 4 * dl) = av;
    }
}

// Short sequences (the caption decoder: T <= 50): one workgroup per (b,h) keeps Q, dO, K, V and the two T x T
// matrices P and dS in LDS, so every operand is read from memory once and the three gradients leave in one pass.
// Rows are padded to dk+16 floats: the 16-keys-by-4-parts score sweep then touches 64 distinct banks per 16 lanes.
template <int NM>
__global__ __launch_bounds__(256) void attn_bwd_fused_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dk = a.dk_, LD = dk + 16;
    float* Qs = lds;
    float* Os = Qs + a.Tq * LD;
    float* Ks = Os + a.Tq * LD;
    float* Vs = Ks + a.Tk * LD;
    float* Ps = Vs + a.Tk * LD;        // [Tq][Tk] softmax probabilities
    float* Ds = Ps + a.Tq * a.Tk;      // [Tq][Tk] dS = P * (dP - D)
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int nv = dk / 4;
    int Tq = a.Tq, Tk = a.Tk;
    int64_t q0 = (int64_t)b * Tq, id0 = (int64_t)b * a.ldid;
    const float* kb = a.k + (int64_t)b * a.sk;
    const float* vb = a.v + (int64_t)b * a.sv;
    float* dkb = a.dk + (int64_t)b * a.sdk;
    float* dvb = a.dv + (int64_t)b * a.sdv;
    if (a.seq_off) {
        q0 = id0 = a.seq_off[b];
        Tq = min(a.Tq, a.seq_off[b + 1] - a.seq_off[b]);
        if (!a.ragged_q_only) {
            Tk = Tq;
            kb = a.k + q0 * a.ldk;
            vb = a.v + q0 * a.ldv;
            dkb = a.dk + q0 * a.lddk;
            dvb = a.dv + q0 * a.lddv;
        }
    }
    for (int idx = threadIdx.x; idx < Tq * nv; idx += 256) {
        const int r = idx / nv, c = idx - r * nv;
        const int64_t row = q0 + r;
        *reinterpret_cast<f32x4*>(&Qs[r * LD + 4 * c]) =
            *reinterpret_cast<const f32x4*>(a.q + row * a.ldq + (int64_t)h * dk + 4 * c);
        *reinterpret_cast<f32x4*>(&Os[r * LD + 4 * c]) =
            *reinterpret_cast<const f32x4*>(a.dout + row * a.ldo + (int64_t)h * dk + 4 * c);
    }
    for (int idx = threadIdx.x; idx < Tk * nv; idx += 256) {
        const int r = idx / nv, c = idx - r * nv;
        *reinterpret_cast<f32x4*>(&Ks[r * LD + 4 * c]) =
            *reinterpret_cast<const f32x4*>(kb + (int64_t)r * a.ldk + (int64_t)h * dk + 4 * c);
        *reinterpret_cast<f32x4*>(&Vs[r * LD + 4 * c]) =
            *reinterpret_cast<const f32x4*>(vb + (int64_t)r * a.ldv + (int64_t)h * dk + 4 * c);
    }
    __syncthreads();
    const float scale = sqrtf((float)dk);
    const int part = lane & 3, kl = lane >> 2;
    const int lpv = dk / 4, grp = lane / lpv, dl = lane - grp * lpv, ngrp = 64 / lpv;
    // query rows: P, dS and dq
    for (int i = wave; i < Tq; i += 4) {
        const bool pad_row = a.ids && a.ids[id0 + i] == 0;
        int nkeys = Tk;
        if (a.causal && !pad_row) {
            nkeys = i + a.causal_off + 1;
            nkeys = nkeys < 1 ? 1 : (nkeys > Tk ? Tk : nkeys);
        }
        float* pi = Ps + i * a.Tk;
        float* di = Ds + i * a.Tk;
        if (pad_row) {
            // uniform softmax over all keys, independent of q and k: dS = 0
            for (int j = lane; j < Tk; j += 64) {
                pi[j] = 1.0f / (float)Tk;
                di[j] = 0.f;
            }
        } else {
            f32x4 qreg[NM], oreg[NM];
            load_parts(qreg, Qs + i * LD + 4 * part);
            load_parts(oreg, Os + i * LD + 4 * part);
            const float l = a.lse[((int64_t)b * a.H + h) * a.Tq + i];
            float dacc = 0.f;
            for (int j0 = 0; j0 < nkeys; j0 += 16) {
                const int j = j0 + kl;
                float sc = 0.f, dp = 0.f;
                if (j < nkeys) {
                    sc = dot_parts(qreg, Ks + j * LD + 4 * part);
                    dp = dot_parts(oreg, Vs + j * LD + 4 * part);
                }
                sc = quad_sum(sc);
                dp = quad_sum(dp);
                if (j < nkeys && part == 0) {
                    const float pr = expf(sc / scale - l);
                    pi[j] = pr;
                    di[j] = dp;
                    dacc += pr * dp;
                }
            }
            const float D = wave_sum(dacc);
            // LDS serves one wavefront's accesses in order: the values written above are visible to its other lanes
            for (int j = lane; j < Tk; j += 64) {
                if (j < nkeys) di[j] = pi[j] * (di[j] - D);
                else {
                    pi[j] = 0.f;
                    di[j] = 0.f;
                }
            }
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (!pad_row)
            for (int j = grp; j < nkeys; j += ngrp)
                acc += *reinterpret_cast<const f32x4*>(Ks + j * LD + 4 * dl) * (di[j] / scale);
        group_reduce(acc, lpv);
        if (grp == 0)
            *reinterpret_cast<f32x4*>(a.dq + (q0 + i) * a.lddq + (int64_t)h * dk + 4 * dl) = acc;
    }
    __syncthreads();
    // keys: dk_j = sum_i dS_ij q_i / sqrt(dk), dv_j = sum_i P_ij dO_i
    for (int j = wave; j < Tk; j += 4) {
        f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
        for (int i = grp; i < Tq; i += ngrp) {
            const float pr = Ps[i * a.Tk + j], d = Ds[i * a.Tk + j];
            av += *reinterpret_cast<const f32x4*>(Os + i * LD + 4 * dl) * pr;
            ak += *reinterpret_cast<const f32x4*>(Qs + i * LD + 4 * dl) * (d / scale);
        }
        group_reduce(ak, lpv);
        group_reduce(av, lpv);
        if (grp == 0) {
            *reinterpret_cast<f32x4*>(dkb + (int64_t)j * a.lddk + (int64_t)h * dk + 4 * dl) = ak;
            *reinterpret_cast<f32x4*>(dvb + (int64_t)j * a.lddv + (int64_t)h * dk + 4 * dl) = av;
        }
    }
    if (a.seq_off && b == a.B - 1) {
        // ragged batch: the slots between the last sample and the next multiple of 32 belong to no sample but are
        // read by the contractions that follow; this head's columns of those gradient rows are zeroed here
        const int live = a.seq_off[a.B], upto = min(a.B * a.Tq, (live + 31) / 32 * 32);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int idx = threadIdx.x; idx < (upto - live) * nv; idx += 256) {
            const int r = live + idx / nv, c = idx % nv;
            *reinterpret_cast<f32x4*>(a.dq + (int64_t)r * a.lddq + (int64_t)h * dk + 4 * c) = z;
            if (!a.ragged_q_only) {
                *reinterpret_cast<f32x4*>(a.dk + (int64_t)r * a.lddk + (int64_t)h * dk + 4 * c) = z;
                *reinterpret_cast<f32x4*>(a.dv + (int64_t)r * a.lddv + (int64_t)h * dk + 4 * c) = z;
            }
        }
    }
}

// The same for the caption decoder's own shape -- head width 128, at most 32 positions, a ragged batch -- on the matrix
// cores.  The five contractions of a (sample, head) are 32 x 32 x 128 (scores S = Q K^T, dP = dO V^T) and 32 x 128 x 32
// (dQ = dS K, dK = dS^T Q, dV = P^T dO) when the sample's T <= 32 rows are padded to 32 (the padding's probabilities
// are forced to zero): 640 fp32 MFMAs of 16 x 16 x 4 instead of row-by-row dot products and shuffles; every operand is
// read from memory once and the three gradients leave as 16-byte stores.  One workgroup of four wavefronts per
// (sample, head):
//   stage 1: wavefront w owns the 16 x 16 tile (w >> 1, w & 1) of S and of dP; probabilities from the saved
//            log-sum-exp, the row sums D_i = sum_j P_ij dP_ij of the two column tiles meet in LDS, dS = P (dP - D);
//   stage 2: wavefront w owns the head columns [32w, 32w + 32) of dQ, dK, dV.
// Operand layout (v_mfma_f32_16x16x4_f32): lane (c, g) = (lane & 15, lane >> 4) feeds A[row c][k = 4g' + m] and
// B[k][column c] of MFMA m; accumulator register e of lane (c, g) is row 4g + e, column c.  Q, dO, K, V rows are padded
// to 132 floats and P, dS rows to 36: every 16-byte or 4-byte fragment read of 64 lanes falls on distinct banks.
constexpr int ABM_T = 32, ABM_DK = 128, ABM_LD = ABM_DK + 4, ABM_PL = ABM_T + 4;
// rows = the launch's T (the longest possible sample): LDS holds that many rows of Q, dO, K, V; the fragment reads of the
// rows from there up to 32 are clamped to the last one (any finite values do: their probabilities are zero)
static size_t abm_lds_bytes(int rows) {
    return ((size_t)4 * rows * ABM_LD + 2 * ABM_T * ABM_PL + 2 * ABM_T) * sizeof(float);
}

__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int R = a.Tq;                       // rows held per operand
    float* Qs = lds;
    float* Os = Qs + R * ABM_LD;
    float* Ks = Os + R * ABM_LD;
    float* Vs = Ks + R * ABM_LD;
    float* Ps = Vs + R * ABM_LD;              // [32][36] probabilities
    float* Ds = Ps + ABM_T * ABM_PL;          // [32][36] dS
    float* Dp = Ds + ABM_T * ABM_PL;          // [2][32] row sums of P dP per column tile
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    // a ragged batch (seq_off): the sample's rows are slots q0 ... q0 + T - 1 of every operand; a dense one: T = Tq = Tk
    // rows per sample, keys / values / their gradients at their own sample strides, ids [B, ldid]
    const int64_t q0 = a.seq_off ? a.seq_off[b] : (int64_t)b * a.Tq;
    const int T = a.seq_off ? min(a.Tq, a.seq_off[b + 1] - a.seq_off[b]) : a.Tq;      // this sample's rows: queries and keys
    const int64_t id0 = a.seq_off ? q0 : (int64_t)b * a.ldid;
    const float* kp = a.seq_off ? a.k + q0 * a.ldk : a.k + (int64_t)b * a.sk;
    const float* vp = a.seq_off ? a.v + q0 * a.ldv : a.v + (int64_t)b * a.sv;
    float* dkp = a.seq_off ? a.dk + q0 * a.lddk : a.dk + (int64_t)b * a.sdk;
    float* dvp = a.seq_off ? a.dv + q0 * a.lddv : a.dv + (int64_t)b * a.sdv;
    const int64_t hc = (int64_t)h * ABM_DK;
    const int ti = wave >> 1, tj = wave & 1;
    // the log-sum-exp and the <PAD> flag of this lane's four query rows, in flight with the operand rows
    float lse_r[4];
    int pad_r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = ti * 16 + 4 * g + e;
        lse_r[e] = i < T ? a.lse[((int64_t)b * a.H + h) * a.Tq + i] : 0.f;
        pad_r[e] = (i < T && a.ids) ? (a.ids[id0 + i] == 0) : 0;
    }
    // ---- stage 0: this head's columns of the sample's rows; the rows from T on are zero ---------------------------
    for (int idx = threadIdx.x; idx < R * (ABM_DK / 4); idx += 256) {
        const int r = idx >> 5, v4 = (idx & 31) * 4;
        f32x4 xq = {0.f, 0.f, 0.f, 0.f}, xo = xq, xk = xq, xv = xq;
        if (r < T) {
            xq = *reinterpret_cast<const f32x4*>(a.q + (q0 + r) * a.ldq + hc + v4);
            xo = *reinterpret_cast<const f32x4*>(a.dout + (q0 + r) * a.ldo + hc + v4);
            xk = *reinterpret_cast<const f32x4*>(kp + (int64_t)r * a.ldk + hc + v4);
            xv = *reinterpret_cast<const f32x4*>(vp + (int64_t)r * a.ldv + hc + v4);
        }
        *reinterpret_cast<f32x4*>(Qs + r * ABM_LD + v4) = xq;
        *reinterpret_cast<f32x4*>(Os + r * ABM_LD + v4) = xo;
        *reinterpret_cast<f32x4*>(Ks + r * ABM_LD + v4) = xk;
        *reinterpret_cast<f32x4*>(Vs + r * ABM_LD + v4) = xv;
    }
    __syncthreads();
    // ---- stage 1: S and dP tiles, probabilities, dS -------------------------------------------------------------
    const float scale = sqrtf((float)ABM_DK);
    f32x4 accs = {0.f, 0.f, 0.f, 0.f}, accp = accs;
    {
        const int ra = min(ti * 16 + c, R - 1), rb = min(tj * 16 + c, R - 1);
        const float* qa = Qs + ra * ABM_LD + 4 * g;
        const float* oa = Os + ra * ABM_LD + 4 * g;
        const float* kb = Ks + rb * ABM_LD + 4 * g;
        const float* vb = Vs + rb * ABM_LD + 4 * g;
#pragma unroll
        for (int kc = 0; kc < ABM_DK / 16; ++kc) {
            const f32x4 fq = *reinterpret_cast<const f32x4*>(qa + 16 * kc);
            const f32x4 fk = *reinterpret_cast<const f32x4*>(kb + 16 * kc);
            const f32x4 fo = *reinterpret_cast<const f32x4*>(oa + 16 * kc);
            const f32x4 fv = *reinterpret_cast<const f32x4*>(vb + 16 * kc);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                accs = __builtin_amdgcn_mfma_f32_16x16x4f32(fq[m], fk[m], accs, 0, 0, 0);
                accp = __builtin_amdgcn_mfma_f32_16x16x4f32(fo[m], fv[m], accp, 0, 0, 0);
            }
        }
    }
    f32x4 pr;
    {
        const int j = tj * 16 + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = ti * 16 + 4 * g + e;
            float pe = 0.f, pd = 0.f;
            if (i < T) {
                if (pad_r[e]) {
                    // every score is the same fill value: uniform over all T keys, independent of q and k (dS = 0)
                    pe = j < T ? 1.0f / (float)T : 0.f;
                } else {
                    int nkeys = T;
                    if (a.causal) {
                        nkeys = i + a.causal_off + 1;
                        nkeys = nkeys < 1 ? 1 : (nkeys > T ? T : nkeys);
                    }
                    if (j < nkeys) {
                        pe = expf(accs[e] / scale - lse_r[e]);
                        pd = pe * accp[e];
                    }
                }
            }
            pr[e] = pe;
            // the row's sum over this tile's 16 columns: lanes with the same g
            pd += __shfl_xor(pd, 1, 64);
            pd += __shfl_xor(pd, 2, 64);
            pd += __shfl_xor(pd, 4, 64);
            pd += __shfl_xor(pd, 8, 64);
            if (c == 0) Dp[tj * ABM_T + i] = pd;
        }
    }
    __syncthreads();
    {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = ti * 16 + 4 * g + e;
            const float D = Dp[i] + Dp[ABM_T + i];
            const float ds = (pad_r[e] || pr[e] == 0.f) ? 0.f : pr[e] * (accp[e] - D);
            Ps[i * ABM_PL + tj * 16 + c] = pr[e];
            Ds[i * ABM_PL + tj * 16 + c] = ds;
        }
    }
    __syncthreads();
    // ---- stage 2: dQ = dS K / sqrt(dk), dK = dS^T Q / sqrt(dk), dV = P^T dO; this wavefront's 32 head columns ------
    const float inv = 1.0f / scale;
    f32x4 aq[2][2], ak[2][2], av[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {                   // the two 16-row tiles of the outputs
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) aq[rt][ct] = ak[rt][ct] = av[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rt * 16 >= T) break;                       // (uniform: rows that are not stored)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {               // contraction over the 32 keys (dQ) or queries (dK, dV)
            if (kc * 16 >= T) break;                   // (uniform: nothing but zeros from here on)
            // dQ: A = dS[rt*16 + c][kc*16 + 4g + m]
            const f32x4 fa = *reinterpret_cast<const f32x4*>(Ds + (rt * 16 + c) * ABM_PL + kc * 16 + 4 * g);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int kk = kc * 16 + 4 * g + m;    // the contraction index this lane feeds to MFMA m
                const int kr = min(kk, R - 1);
                // dK, dV: A = dS^T, P^T: [key rt*16 + c][query kk]
                const float fds = Ds[kk * ABM_PL + rt * 16 + c];
                const float fp = Ps[kk * ABM_PL + rt * 16 + c];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int d = wave * 32 + ct * 16 + c;
                    aq[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m], Ks[kr * ABM_LD + d], aq[rt][ct], 0, 0, 0);
                    ak[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(fds, Qs[kr * ABM_LD + d], ak[rt][ct], 0, 0, 0);
                    av[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(fp, Os[kr * ABM_LD + d], av[rt][ct], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                   // every wavefront is done reading Q, dO, K
    // the three gradients through LDS (over Q, K, dO: rows of 132 floats again), so that they leave as 16-byte stores
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = rt * 16 + 4 * g + e;
                if (r < T) {
                    const int col = wave * 32 + ct * 16 + c;
                    Qs[r * ABM_LD + col] = aq[rt][ct][e] * inv;
                    Ks[r * ABM_LD + col] = ak[rt][ct][e] * inv;
                    Os[r * ABM_LD + col] = av[rt][ct][e];
                }
            }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * (ABM_DK / 4); idx += 256) {
        const int r = idx >> 5, v4 = (idx & 31) * 4;
        *reinterpret_cast<f32x4*>(a.dq + (q0 + r) * a.lddq + hc + v4) = *reinterpret_cast<const f32x4*>(Qs + r * ABM_LD + v4);
        *reinterpret_cast<f32x4*>(dkp + (int64_t)r * a.lddk + hc + v4) = *reinterpret_cast<const f32x4*>(Ks + r * ABM_LD + v4);
        *reinterpret_cast<f32x4*>(dvp + (int64_t)r * a.lddv + hc + v4) = *reinterpret_cast<const f32x4*>(Os + r * ABM_LD + v4);
    }
    if (a.seq_off && b == a.B - 1) {
        // the slots between the last sample and the next multiple of 32 belong to no sample but are read by the
        // contractions that follow: this head's columns of those gradient rows are zeroed here
        const int live = a.seq_off[a.B], upto = min(a.B * a.Tq, (live + 31) / 32 * 32);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int idx = threadIdx.x; idx < (upto - live) * (ABM_DK / 4); idx += 256) {
            const int r = live + idx / (ABM_DK / 4), v4 = (idx % (ABM_DK / 4)) * 4;
            *reinterpret_cast<f32x4*>(a.dq + (int64_t)r * a.lddq + hc + v4) = z;
            *reinterpret_cast<f32x4*>(a.dk + (int64_t)r * a.lddk + hc + v4) = z;
            *reinterpret_cast<f32x4*>(a.dv + (int64_t)r * a.lddv + hc + v4) = z;
        }
    }
}

// Long memory, short query block (cross-attention over a token or grid memory: Tq <= 64, any Tk, no mask): one
// workgroup per (b,h) keeps Q, dO and a dQ accumulator in LDS and streams the keys through LDS in chunks of AB_KC.
// D_i = sum_j P_ij dP_ij equals dO_i . O_i, so with the forward output at hand every chunk is independent: its P and
// dS give the chunk's dK, dV rows at once (each key row is stored exactly once) and add into dQ.
constexpr int AB_KC = 32;

template <int NM>
__global__ __launch_bounds__(256) void attn_bwd_chunked_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dk = a.dk_, LD = dk + 16, Tk = a.Tk;
    float* Qs = lds;                       // [Tq][LD]
    float* Os = Qs + a.Tq * LD;            // dO [Tq][LD]
    float* dQs = Os + a.Tq * LD;           // [Tq][LD]
    float* Ks = dQs + a.Tq * LD;           // [AB_KC][LD]
    float* Vs = Ks + AB_KC * LD;           // [AB_KC][LD]
    float* Ps = Vs + AB_KC * LD;           // [Tq][AB_KC]
    float* Ds = Ps + a.Tq * AB_KC;         // [Tq][AB_KC]
    float* Dsum = Ds + a.Tq * AB_KC;       // [Tq]
    float* Lse = Dsum + a.Tq;              // [Tq]
    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int nv = dk / 4;
    // ragged on the query side: sample b's rows of q, dout, out and dq are [seq_off[b], seq_off[b+1]); k, v stay dense
    int Tq = a.Tq;
    int64_t q0 = (int64_t)b * Tq;
    if (a.seq_off) {
        q0 = a.seq_off[b];
        Tq = min(a.Tq, a.seq_off[b + 1] - a.seq_off[b]);
    }
    for (int idx = threadIdx.x; idx < Tq * nv; idx += 256) {
        const int r = idx / nv, c = idx - r * nv;
        const int64_t row = q0 + r;
        *reinterpret_cast<f32x4*>(&Qs[r * LD + 4 * c]) =
            *reinterpret_cast<const f32x4*>(a.q + row * a.ldq + (int64_t)h * dk + 4 * c);
        *reinterpret_cast<f32x4*>(&Os[r * LD + 4 * c]) =
            *reinterpret_cast<const f32x4*>(a.dout + row * a.ldo + (int64_t)h * dk + 4 * c);
        *reinterpret_cast<f32x4*>(&dQs[r * LD + 4 * c]) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // D_i = dO_i . O_i, one wavefront per row
    for (int i = wave; i < Tq; i += 4) {
        const int64_t row = q0 + i;
        float d = 0.f;
        for (int c = lane; c < dk; c += 64)
            d += a.dout[row * a.ldo + (int64_t)h * dk + c] * a.out[row * a.ldout + (int64_t)h * dk + c];
        d = wave_sum(d);
        if (lane == 0) {
            Dsum[i] = d;
            Lse[i] = a.lse[((int64_t)b * a.H + h) * a.Tq + i];
        }
    }
    const float scale = sqrtf((float)dk);
    const int part = lane & 3, kl = lane >> 2;
    const int lpv = dk / 4, grp = lane / lpv, dl = lane - grp * lpv, ngrp = 64 / lpv;
    for (int c0 = 0; c0 < Tk; c0 += AB_KC) {
        const int nk = min(AB_KC, Tk - c0);
        __syncthreads();                                   // previous chunk fully consumed (and the prologue done)
        for (int idx = threadIdx.x; idx < nk * nv; idx += 256) {
            const int r = idx / nv, c = idx - r * nv;
            *reinterpret_cast<f32x4*>(&Ks[r * LD + 4 * c]) = *reinterpret_cast<const f32x4*>(
                a.k + (int64_t)b * a.sk + (int64_t)(c0 + r) * a.ldk + (int64_t)h * dk + 4 * c);
            *reinterpret_cast<f32x4*>(&Vs[r * LD + 4 * c]) = *reinterpret_cast<const f32x4*>(
                a.v + (int64_t)b * a.sv + (int64_t)(c0 + r) * a.ldv + (int64_t)h * dk + 4 * c);
        }
        __syncthreads();
        // query rows: P and dS of this chunk, dQ += dS K / sqrt(dk)
        for (int i = wave; i < Tq; i += 4) {
            f32x4 qreg[NM], oreg[NM];
            load_parts(qreg, Qs + i * LD + 4 * part);
            load_parts(oreg, Os + i * LD + 4 * part);
            const float l = Lse[i], D = Dsum[i];
            for (int j0 = 0; j0 < nk; j0 += 16) {
                const int j = j0 + kl;
                float sc = 0.f, dp = 0.f;
                if (j < nk) {
                    sc = dot_parts(qreg, Ks + j * LD + 4 * part);
                    dp = dot_parts(oreg, Vs + j * LD + 4 * part);
                }
                sc = quad_sum(sc);
                dp = quad_sum(dp);
                if (j < nk && part == 0) {
                    const float pr = expf(sc / scale - l);
                    Ps[i * AB_KC + j] = pr;
                    Ds[i * AB_KC + j] = pr * (dp - D);
                }
            }
            // LDS serves one wavefront's accesses in order: the dS values above are visible to its other lanes
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int j = grp; j < nk; j += ngrp)
                acc += *reinterpret_cast<const f32x4*>(Ks + j * LD + 4 * dl) * (Ds[i * AB_KC + j] / scale);
            group_reduce(acc, lpv);
            if (grp == 0) *reinterpret_cast<f32x4*>(dQs + i * LD + 4 * dl) += acc;     // row i belongs to this wave
        }
        __syncthreads();
        // keys of the chunk: dK_j = sum_i dS_ij Q_i / sqrt(dk), dV_j = sum_i P_ij dO_i
        for (int j = wave; j < nk; j += 4) {
            f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
            for (int i = grp; i < Tq; i += ngrp) {
                const float pr = Ps[i * AB_KC + j], d = Ds[i * AB_KC + j];
                av += *reinterpret_cast<const f32x4*>(Os + i * LD + 4 * dl) * pr;
                ak += *reinterpret_cast<const f32x4*>(Qs + i * LD + 4 * dl) * (d / scale);
            }
            group_reduce(ak, lpv);
            group_reduce(av, lpv);
            if (grp == 0) {
                const int64_t kr = (int64_t)(c0 + j);
                *reinterpret_cast<f32x4*>(a.dk + (int64_t)b * a.sdk + kr * a.lddk + (int64_t)h * dk + 4 * dl) = ak;
                *reinterpret_cast<f32x4*>(a.dv + (int64_t)b * a.sdv + kr * a.lddv + (int64_t)h * dk + 4 * dl) = av;
            }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < Tq * nv; idx += 256) {
        const int r = idx / nv, c = idx - r * nv;
        *reinterpret_cast<f32x4*>(a.dq + (q0 + r) * a.lddq + (int64_t)h * dk + 4 * c) =
            *reinterpret_cast<const f32x4*>(&dQs[r * LD + 4 * c]);
    }
    if (a.seq_off && b == a.B - 1) {
        // ragged batch: zero this head's columns of the dq rows between the last sample and the next multiple of 32
        const int live = a.seq_off[a.B], upto = min(a.B * a.Tq, (live + 31) / 32 * 32);
        for (int idx = threadIdx.x; idx < (upto - live) * nv; idx += 256) {
            const int r = live + idx / nv, c = idx % nv;
            *reinterpret_cast<f32x4*>(a.dq + (int64_t)r * a.lddq + (int64_t)h * dk + 4 * c) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// ---- Adam (torch.optim.Adam defaults: no weight decay, no amsgrad) over one flat buffer -----------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// The same with step size and bias corrections read from the device-resident training state, so that the launch
// has no per-step arguments and can sit in a captured graph.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, int64_t n, float b1, float b2, float eps,
                                const capdec_train_state* __restrict__ st, float gscale) {
    const float lr = st->lr, bc1 = st->bias1, bc2_sqrt = st->bias2_sqrt;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// One thread: next optimizer step.  step += 1; a new dropout seed (splitmix64 of the old one); Adam's bias corrections
// for the new step count.
__global__ void train_state_advance_kernel(capdec_train_state* __restrict__ st, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int t = st->step + 1;
    uint64_t z = st->seed + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    st->seed = (z ^ (z >> 31)) >> 2;                 // 62 bits: the by-value site offsets cannot overflow the sum
    st->step = t;
    st->bias1 = (float)(1.0 - pow((double)b1, (double)t));
    st->bias2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)t));
}

// loss = sum nll / n valid ; gscale = 1 / n valid  (CrossEntropyLoss(ignore_index) mean reduction)
__global__ void loss_finish_kernel(const float* __restrict__ sums, float* __restrict__ loss, float* __restrict__ gscale) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        loss[0] = sums[0] / sums[1];
        gscale[0] = 1.0f / sums[1];
    }
}

inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

// ---- materialised vocabulary loss: logits [M,V] -> d(mean NLL)/d(logits) in place ---------------------
// count_targets: sums[1] += number of counted targets (sums zeroed by the caller; integer-valued float adds are exact)
__global__ __launch_bounds__(256) void count_targets_kernel(const int64_t* __restrict__ targets, int M,
                                                            int64_t ignore_index, float* __restrict__ sums) {
    __shared__ int red[4];
    int c = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) c += targets[i] != ignore_index ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int n = red[0] + red[1] + red[2] + red[3];
        if (n) atomicAdd(&sums[1], (float)n);
    }
}

// One wavefront per row (SM_ROWS rows per workgroup): pass 1 keeps a running (max, sum exp) per lane and merges
// them over the wave; pass 2 re-reads the row (cache-resident) and overwrites it with
// (softmax - onehot(target)) * gscale, or zeros for an ignored target.  The row's NLL joins the workgroup's
// partial, which leaves as one atomic.
constexpr int SM_ROWS = 8;

template <bool VEC>
__global__ __launch_bounds__(256) void softmax_nll_grad_kernel(float* __restrict__ logits, int64_t ld,
                                                               const int64_t* __restrict__ targets, int M, int V,
                                                               int64_t ignore_index, float* __restrict__ gscale,
                                                               float* __restrict__ sums, float* __restrict__ lse_out,
                                                               const int32_t* __restrict__ rows_dev) {
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float gs = 1.0f / sums[1];            // complete: the counting kernel ran before this one
    if (blockIdx.x == 0 && threadIdx.x == 0) gscale[0] = gs;
    // rows_dev: only the first rows_dev[0] rows hold logits; the rows up to the next multiple of 32 are zeroed (a
    // contraction over the rows reads whole k-tiles) and the rest is left alone
    const int live = rows_dev ? min(M, rows_dev[0]) : M;
    const int upto = rows_dev ? min(M, (live + 31) / 32 * 32) : M;
    float acc = 0.f;
    for (int rr = wave; rr < SM_ROWS; rr += 4) {
        const int row = blockIdx.x * SM_ROWS + rr;
        if (row >= upto) break;
        float* lr = logits + (int64_t)row * ld;
        if (row >= live || (!lse_out && targets[row] == ignore_index)) {
            // no gradient and no log-sum-exp wanted: the row is not read
            if (VEC) {
                f32x4* l4 = reinterpret_cast<f32x4*>(lr);
                for (int c = lane; c < V / 4; c += 64) l4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                for (int c = lane; c < V; c += 64) lr[c] = 0.f;
            }
            continue;
        }
        float mx = -INFINITY, sm = 0.f;
        if (VEC) {
            // four independent 16-byte loads in flight per lane
            const f32x4* l4 = reinterpret_cast<const f32x4*>(lr);
            const int n4 = V / 4;
            for (int c0 = lane; c0 < n4; c0 += 256) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + 64 * u;
                    v[u] = c < n4 ? l4[c] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
                float m4 = mx;
#pragma unroll
                for (int u = 0; u < 4; ++u) m4 = fmaxf(m4, fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3])));
                if (m4 > mx) {
                    sm *= __expf(mx - m4);
                    mx = m4;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    sm += __expf(v[u][0] - mx) + __expf(v[u][1] - mx) + __expf(v[u][2] - mx) + __expf(v[u][3] - mx);
            }
        } else {
            for (int c = lane; c < V; c += 64) {
                const float v = lr[c];
                if (v > mx) {
                    sm *= __expf(mx - v);
                    mx = v;
                }
                sm += __expf(v - mx);
            }
        }
        const float wm = wave_max(mx);
        sm = wave_sum(mx > -INFINITY ? sm * __expf(mx - wm) : 0.f);
        const float lse = wm + logf(sm);
        const int64_t tg = targets[row];
        const bool counted = tg != ignore_index;
        // a counted target outside [0, V) is a caller error (torch's CrossEntropyLoss raises): no host can be told
        // from here, so the row's loss term becomes NaN -- loud in the very next loss value -- instead of a read
        // outside the row
        const bool in_range = tg >= 0 && tg < (int64_t)V;
        const float tl = counted ? (in_range ? lr[tg] : NAN) : 0.f;
        // the target logit is in a register before any lane overwrites the row
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (counted && lane == 0) acc += lse - tl;
        if (lane == 0 && lse_out) lse_out[row] = lse;
        if (VEC) {
            f32x4* l4 = reinterpret_cast<f32x4*>(lr);
            const int n4 = V / 4;
            for (int c0 = lane; c0 < n4; c0 += 256) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + 64 * u;
                    if (c < n4) v[u] = l4[c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + 64 * u;
                    if (c >= n4) continue;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        v[u][k] = counted ? (__expf(v[u][k] - lse) - ((int64_t)(4 * c + k) == tg ? 1.f : 0.f)) * gs : 0.f;
                    l4[c] = v[u];
                }
            }
        } else {
            for (int c = lane; c < V; c += 64)
                lr[c] = counted ? (__expf(lr[c] - lse) - ((int64_t)c == tg ? 1.f : 0.f)) * gs : 0.f;
        }
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = red[0] + red[1] + red[2] + red[3];
        if (t != 0.f) atomicAdd(&sums[0], t);
    }
}


// The same for rows of at most 256 * NV4 logits: the whole row sits in the wavefront's registers (4 * NV4 per lane), so
// it is read from memory once -- re-read in a second pass, a 40 KB row has long left the L2 that 32 CUs share.  Buffer
// addressing (one lane offset in a VGPR, the vector's offset in an SGPR) keeps the NV4 loads and stores from each
// holding an address register pair; loads past the row's end return zeros and stores there are dropped.
typedef int capdec_i32x4 __attribute__((ext_vector_type(4)));

template <int NV4>
__global__ __launch_bounds__(256, 2) void softmax_nll_grad_reg_kernel(float* __restrict__ logits, int64_t ld,
                                                                      const int64_t* __restrict__ targets, int M, int V,
                                                                      int64_t ignore_index, float* __restrict__ gscale,
                                                                      float* __restrict__ sums,
                                                                      float* __restrict__ lse_out,
                                                                      const int32_t* __restrict__ rows_dev) {
    __shared__ float red[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const float gs = 1.0f / sums[1];            // complete: the counting kernel ran before this one
    if (blockIdx.x == 0 && threadIdx.x == 0) gscale[0] = gs;
    const int live = rows_dev ? min(M, rows_dev[0]) : M;
    const int upto = rows_dev ? min(M, (live + 31) / 32 * 32) : M;
    const int n4 = V / 4;
    float acc = 0.f;
#pragma unroll 1
    for (int rr = wave; rr < SM_ROWS; rr += 4) {
        const int row = blockIdx.x * SM_ROWS + rr;
        if (row >= upto) break;
        float* lr = logits + (int64_t)row * ld;
        const int64_t tg = row < live ? targets[row] : ignore_index;
        const bool counted = tg != ignore_index;
        if (row >= live || (!lse_out && !counted)) {
            f32x4* l4 = reinterpret_cast<f32x4*>(lr);
            for (int c = lane; c < n4; c += 64) l4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(lr, 0, V * 4, 0x00020000);
        f32x4 v[NV4];
#pragma unroll
        for (int u = 0; u < NV4; ++u) {
            v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, u * 1024, 0));
            if (lane + 64 * u >= n4) v[u] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
        const float tl = counted ? ((tg >= 0 && tg < (int64_t)V) ? lr[tg] : NAN) : 0.f;   // out-of-range target: NaN loss, no stray read
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < NV4; ++u) mx = fmaxf(mx, fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3])));
        mx = wave_max(mx);
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < NV4; ++u)
            sm += __expf(v[u][0] - mx) + __expf(v[u][1] - mx) + __expf(v[u][2] - mx) + __expf(v[u][3] - mx);
        sm = wave_sum(sm);
        const float lse = mx + logf(sm);
        // the target logit is in a register before any lane overwrites the row
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (counted && lane == 0) acc += lse - tl;
        if (lane == 0 && lse_out) lse_out[row] = lse;
        const int tgi = counted ? (int)tg : -1;
        const float gsr = counted ? gs : 0.f;
#pragma unroll
        for (int u = 0; u < NV4; ++u) {
            const int c = lane + 64 * u;
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (__expf(v[u][k] - lse) - (4 * c + k == tgi ? 1.f : 0.f)) * gsr;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(capdec_i32x4, o), rs, lane * 16, u * 1024, 0);
        }
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = red[0] + red[1] + red[2] + red[3];
        if (t != 0.f) atomicAdd(&sums[0], t);
    }
}


// ---- rows with a counted target first ------------------------------------------------------------------------
// One workgroup: thread t owns the rows [t*per, (t+1)*per); an exclusive scan over the threads' counts places
// them, so the order of the counted rows is kept.
__global__ __launch_bounds__(1024) void compact_targets_kernel(const int64_t* __restrict__ targets, int M,
                                                               int64_t ignore_index, int64_t* __restrict__ order,
                                                               int64_t* __restrict__ slot, int64_t* __restrict__ tgt_c,
                                                               int32_t* __restrict__ count) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (M + 1023) / 1024;
    const int lo = min(M, t * per), hi = min(M, lo + per);
    int n = 0;
    for (int r = lo; r < hi; ++r) n += targets[r] != ignore_index;
    part[t] = n;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {          // inclusive scan
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const int total = part[1023];
    int at = part[t] - n;
    for (int r = lo; r < hi; ++r) {
        const int64_t tg = targets[r];
        if (tg != ignore_index) {
            order[at] = r;
            tgt_c[at] = tg;
            slot[r] = at++;
        } else {
            slot[r] = M;                          // the caller's all-zero extra row
        }
    }
    for (int r = total + t; r < M; r += 1024) {   // unused slots: any valid row, never a counted target
        order[r] = 0;
        tgt_c[r] = ignore_index;
    }
    if (t == 0) count[0] = total;
}


// ---- live prefixes of a padded caption batch ---------------------------------------------------------------------
// Sample b's rows from its last counted target on are dead: they feed nothing but ignored loss terms (causal
// self-attention: a row only sees earlier rows).  len[b] = 1 + last position with a counted target, or T when an
// input id inside that prefix is <PAD> (the reference lets such a query row attend uniformly over all T keys,
// transformer.py:18-22, so the sample keeps its full length).  Slots are assigned sample by sample, so each sample's
// live rows stay contiguous and ordered: a ragged batch.  One workgroup of 1024 threads sweeps the [B,T] arrays
// three times with coalesced loads; the per-sample lengths and offsets live in LDS.
__global__ __launch_bounds__(1024) void compact_prefix_kernel(const int64_t* __restrict__ ids,
                                                              const int64_t* __restrict__ targets, int64_t ld, int B, int T,
                                                              int64_t pad, int64_t* __restrict__ order,
                                                              int64_t* __restrict__ slot, int32_t* __restrict__ seq_off,
                                                              int32_t* __restrict__ sample, int32_t* __restrict__ ids_c,
                                                              int64_t* __restrict__ tgt_c, int32_t* __restrict__ count) {
    extern __shared__ int sm[];            // [B] lengths, [B] offsets, [B] marks, [1024] scan
    int* len = sm;
    int* off = sm + B;
    int* mark = sm + 2 * B;
    int* part = sm + 3 * B;
    const int t = threadIdx.x;
    const int M = B * T;
    for (int b = t; b < B; b += 1024) {
        len[b] = 0;
        mark[b] = 0;
    }
    __syncthreads();
    // (element (b, p) of ids and targets sits at b*ld + p: ld = T for packed arrays, T + 1 for the two views of a
    //  [B, T+1] caption matrix)
    for (int r = t; r < M; r += 1024)
        if (targets[(int64_t)(r / T) * ld + r % T] != pad) atomicMax(&len[r / T], r % T + 1);
    __syncthreads();
    for (int r = t; r < M; r += 1024)
        if (ids[(int64_t)(r / T) * ld + r % T] == pad && r % T < len[r / T]) mark[r / T] = 1;      // a <PAD> input inside the prefix
    __syncthreads();
    for (int b = t; b < B; b += 1024)
        if (mark[b]) len[b] = T;
    __syncthreads();
    // exclusive scan of the lengths: thread t owns the samples [t*per, (t+1)*per)
    const int per = (B + 1023) / 1024;
    const int lo = min(B, t * per), hi = min(B, lo + per);
    int n = 0;
    for (int b = lo; b < hi; ++b) n += len[b];
    part[t] = n;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const int total = part[1023];
    const int total32 = min(M, (total + 31) / 32 * 32);
    int at = part[t] - n;
    for (int b = lo; b < hi; ++b) {
        off[b] = at;
        seq_off[b] = at;
        at += len[b];
    }
    if (t == 0) seq_off[B] = total;
    __syncthreads();
    for (int r = t; r < M; r += 1024) {
        const int b = r / T, p = r - b * T;
        if (p < len[b]) {
            const int s = off[b] + p;
            order[s] = r;
            sample[s] = b;
            ids_c[s] = (int32_t)ids[(int64_t)b * ld + p];
            tgt_c[s] = targets[(int64_t)b * ld + p];
            slot[r] = s;
        } else {
            slot[r] = M;                          // the caller's all-zero extra row
        }
    }
    // slots that belong to no sample: copies of row 0 with an ignored target.  Those below the next multiple of
    // 32 are computed like live rows (contractions over the rows read whole k-tiles); their gradients are zero.
    for (int r = total + t; r < M; r += 1024) {
        order[r] = 0;
        sample[r] = 0;
        ids_c[r] = (int32_t)ids[0];
        tgt_c[r] = pad;
    }
    if (t == 0) {
        count[0] = total32;
        count[1] = total;
    }
}

}  // namespace

extern "C" {

int capdec_dropout(const float* x, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, float* y, void* stream) {
    CAPDEC_REQUIRE(x && y && n >= 0 && p >= 0.f && p < 1.f);
    if (n == 0) return 0;
    if (n % 4 == 0 && capdec_aligned16(x) && capdec_aligned16(y))
        hipLaunchKernelGGL(dropout4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const f32x4*>(x), n / 4, p, seed, seed_dev, reinterpret_cast<f32x4*>(y));
    else
        hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, p, seed, seed_dev,
                           y);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_add_layernorm_train(const float* x, int x_div, const float* add, const float* gamma, const float* beta,
                               float eps, int M, int E, float p, uint64_t seed, float* y, float* s_out, float* mean,
                               float* rstd, void* stream) {
    return capdec_add_layernorm_train_rows(x, x_div, nullptr, add, gamma, beta, eps, M, E, p, seed, nullptr, y, s_out, mean,
                                           rstd, nullptr, stream);
}

int capdec_add_layernorm_train_rows(const float* x, int x_div, const int32_t* x_map, const float* add, const float* gamma,
                                    const float* beta, float eps, int M, int E, float p, uint64_t seed,
                                    const uint64_t* seed_dev, float* y, float* s_out, float* mean, float* rstd,
                                    const int32_t* rows_dev, void* stream) {
    CAPDEC_REQUIRE(x && gamma && beta && y && M >= 0 && E >= 4 && E % 4 == 0 && E <= 256 * LN_MAXV && p >= 0.f && p < 1.f);
    CAPDEC_REQUIRE(x_div >= 1);
    if (!capdec_aligned16(x) || !capdec_aligned16(y) || !capdec_aligned16(gamma) || !capdec_aligned16(beta) ||
        (add && !capdec_aligned16(add)) || (s_out && !capdec_aligned16(s_out)))
        return CAPDEC_EALIGN;
    if (M == 0) return 0;
#define CAPDEC_CALL(NV_)                                                                                          \
    hipLaunchKernelGGL(ln_train_fwd_kernel<NV_>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, add, gamma, \
                       beta, eps, M, E, x_div, p, seed, seed_dev, y, s_out, mean, rstd, x_map, rows_dev)
    CAPDEC_LN_DISPATCH(E, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_layernorm_bwd(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma,
                         int M, int E, float p, uint64_t seed, float* d_add, float* d_x, float* dgamma, float* dbeta,
                         float* dxsum, void* stream) {
    return capdec_layernorm_bwd_rows(dy, s, mean, rstd, gamma, M, E, p, seed, nullptr, d_add, d_x, dgamma, dbeta, dxsum,
                                     nullptr, stream);
}

static int ln_bwd_rows_per_block(int M) {
    const int per = (M + 255) / 256;
    return M >= 8192 ? 32 : (per < 8 ? 8 : (per > 32 ? 32 : (per + 3) / 4 * 4));
}

size_t capdec_layernorm_bwd_workspace_floats(int M, int E) {
    if (M <= 0 || E < 4) return 0;
    const int per = ln_bwd_rows_per_block(M);
    return (size_t)((M + per - 1) / per) * 3 * (size_t)E;
}

int capdec_layernorm_bwd_rows(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma,
                              int M, int E, float p, uint64_t seed, const uint64_t* seed_dev, float* d_add, float* d_x,
                              float* dgamma, float* dbeta, float* dxsum, const int32_t* rows_dev, void* stream) {
    return capdec_layernorm_bwd_staged(dy, s, mean, rstd, gamma, M, E, p, seed, seed_dev, d_add, d_x, dgamma, dbeta, dxsum,
                                       rows_dev, nullptr, 0, stream);
}

int capdec_layernorm_bwd_staged(const float* dy, const float* s, const float* mean, const float* rstd, const float* gamma,
                                int M, int E, float p, uint64_t seed, const uint64_t* seed_dev, float* d_add, float* d_x,
                                float* dgamma, float* dbeta, float* dxsum, const int32_t* rows_dev, float* workspace,
                                size_t workspace_floats, void* stream) {
    CAPDEC_REQUIRE(dy && s && mean && rstd && gamma && dgamma && dbeta && M >= 0 && E >= 4 && E % 4 == 0 &&
                   E <= 256 * LN_MAXV && p >= 0.f && p < 1.f);
    if (!capdec_aligned16(dy) || !capdec_aligned16(s) || !capdec_aligned16(gamma) || (d_add && !capdec_aligned16(d_add)) ||
        (d_x && !capdec_aligned16(d_x)))
        return CAPDEC_EALIGN;
    if (M == 0) return 0;
    const int per = ln_bwd_rows_per_block(M);
    const int nwg = (M + per - 1) / per;
    if (workspace) {
        if (workspace_floats < (size_t)nwg * 3 * E) return CAPDEC_ENOMEM;
        if (!capdec_aligned16(workspace) || !capdec_aligned16(dgamma) || !capdec_aligned16(dbeta) ||
            (dxsum && !capdec_aligned16(dxsum)))
            return CAPDEC_EALIGN;
    }
#define CAPDEC_CALL(NV_)                                                                                           \
    hipLaunchKernelGGL(ln_bwd_kernel<NV_>, dim3(nwg), dim3(256), 3 * E * sizeof(float), (hipStream_t)stream, dy, s, mean, \
                       rstd, gamma, M, E, p, seed, seed_dev, d_add, d_x, dgamma, dbeta, dxsum, rows_dev, per, workspace)
    CAPDEC_LN_DISPATCH(E, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    if (workspace) {
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(3 * ((E + 63) / 64)), dim3(256), 0, (hipStream_t)stream, workspace, nwg,
                           E, dgamma, dbeta, dxsum);
        CAPDEC_LAUNCH_CHECK();
    }
    return 0;
}

int capdec_colsum_rows(const float* x, int64_t ldx, int M, int N, float* out, const int32_t* rows_dev, void* stream) {
    CAPDEC_REQUIRE(x && out && M >= 0 && N >= 1 && ldx >= N);
    if (M == 0) return 0;
    const bool vec = N % 4 == 0 && ldx % 4 == 0 && capdec_aligned16(x);
    dim3 grid((N + 63) / 64, (M + CS_ROWS - 1) / CS_ROWS);
    if (vec) hipLaunchKernelGGL(colsum_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, M, N, out, rows_dev);
    else hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, M, N, out, rows_dev);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_colsum(const float* x, int64_t ldx, int M, int N, float* out, void* stream) {
    return capdec_colsum_rows(x, ldx, M, N, out, nullptr, stream);
}

int capdec_group_sum(const float* x, int G, int T, int E, float* out, void* stream) {
    CAPDEC_REQUIRE(x && out && G >= 0 && T >= 1 && E >= 4 && E % 4 == 0);
    if (!capdec_aligned16(x) || !capdec_aligned16(out)) return CAPDEC_EALIGN;
    if (G == 0) return 0;
    hipLaunchKernelGGL(group_sum_kernel, dim3((G + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, G, T, E, out, nullptr);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_axpy_rows(const float* x, int x_div, float scale, int M, int E, float* y, int accumulate, void* stream) {
    CAPDEC_REQUIRE(x && y && x_div >= 1 && M >= 0 && E >= 4 && E % 4 == 0);
    if (!capdec_aligned16(x) || !capdec_aligned16(y)) return CAPDEC_EALIGN;
    if (M == 0) return 0;
    hipLaunchKernelGGL(axpy_rows_kernel, dim3(grid_for((int64_t)M * E / 4)), dim3(256), 0, (hipStream_t)stream, x, x_div, scale,
                       M, E, y, accumulate);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_pair_diff(const float* k2, const float* v2, const float* bq, int B, int E, int H, float* dk, float* dv, float* sb,
                     void* stream) {
    CAPDEC_REQUIRE(k2 && v2 && bq && dk && dv && sb && B >= 0 && H >= 1 && E >= 4 && E % (4 * H) == 0);
    if (!capdec_aligned16(k2) || !capdec_aligned16(v2) || !capdec_aligned16(bq) || !capdec_aligned16(dk) || !capdec_aligned16(dv))
        return CAPDEC_EALIGN;
    if (B == 0) return 0;
    hipLaunchKernelGGL(pair_diff_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, k2, v2, bq, B, E, H, dk, dv, sb);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_pair_diff_bwd(const float* ddk, const float* ddv, const float* dsb, const float* dv1, const float* bq,
                         const float* dk, int B, int E, int H, float* dk2, float* dv2, float* qb, void* stream) {
    CAPDEC_REQUIRE(ddk && ddv && dsb && dv1 && bq && dk && dk2 && dv2 && qb && B >= 0 && H >= 1 && E >= 4 && E % (4 * H) == 0);
    if (!capdec_aligned16(ddk) || !capdec_aligned16(ddv) || !capdec_aligned16(dv1) || !capdec_aligned16(bq) ||
        !capdec_aligned16(dk) || !capdec_aligned16(dk2) || !capdec_aligned16(dv2) || !capdec_aligned16(qb))
        return CAPDEC_EALIGN;
    if (B == 0) return 0;
    hipLaunchKernelGGL(pair_diff_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, ddk, ddv, dsb, dv1, bq, dk, B,
                       E, H, dk2, dv2, qb);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_pair_gate_fwd(const float* x1, const float* G, const float* D, const float* c1, const float* sb,
                         const int32_t* sample, int T, int M, int E, int H, float scale, float* a2, float* p,
                         const int32_t* rows_dev, void* stream) {
    CAPDEC_REQUIRE(x1 && G && D && c1 && sb && a2 && p && M >= 0 && T >= 1 && H >= 1 && E >= 4 && E % 4 == 0 && E <= 2048);
    if (!capdec_aligned16(x1) || !capdec_aligned16(G) || !capdec_aligned16(D) || !capdec_aligned16(c1) || !capdec_aligned16(a2))
        return CAPDEC_EALIGN;
    if (M == 0) return 0;
#define CAPDEC_CALL(NV_)                                                                                            \
    hipLaunchKernelGGL(pair_gate_fwd_kernel<NV_>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x1, G, D, c1, sb, \
                       sample, T, M, E, H, scale, a2, p, rows_dev)
    CAPDEC_LN_DISPATCH(E, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_pair_gate_bwd(const float* d_a2, const float* x1, const float* p, const float* G, const float* D,
                         const int32_t* seg_off, int B, int T, int E, int H, float scale, float* d_x1, float* dG, float* dD,
                         float* dc1, float* dsb, void* stream) {
    CAPDEC_REQUIRE(d_a2 && x1 && p && G && D && d_x1 && dG && dD && dc1 && dsb && B >= 0 && T >= 1 && H >= 1 && E >= 4 &&
                   E % 4 == 0 && E <= 1024);
    if (!capdec_aligned16(d_a2) || !capdec_aligned16(x1) || !capdec_aligned16(G) || !capdec_aligned16(D) ||
        !capdec_aligned16(d_x1) || !capdec_aligned16(dG) || !capdec_aligned16(dD) || !capdec_aligned16(dc1))
        return CAPDEC_EALIGN;
    if (B == 0) return 0;
    const size_t smem = ((size_t)9 * E + 4) * sizeof(float);
#define CAPDEC_CALL(NV_)                                                                                            \
    hipLaunchKernelGGL(pair_gate_bwd_kernel<NV_>, dim3(B), dim3(256), smem, (hipStream_t)stream, d_a2, x1, p, G, D, seg_off, T, \
                       E, H, scale, d_x1, dG, dD, dc1, dsb)
    const int nvl = (E / 4 + 63) / 64;
    if (nvl <= 1) { CAPDEC_CALL(1); } else if (nvl <= 2) { CAPDEC_CALL(2); } else { CAPDEC_CALL(4); }
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_group_sum_segments(const float* x, int G, const int32_t* seg_off, int E, float* out, void* stream) {
    CAPDEC_REQUIRE(x && out && seg_off && G >= 0 && E >= 4 && E % 4 == 0);
    if (!capdec_aligned16(x) || !capdec_aligned16(out)) return CAPDEC_EALIGN;
    if (G == 0) return 0;
    hipLaunchKernelGGL(group_sum_kernel, dim3((G + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, G, 0, E, out, seg_off);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_embed_bwd(const int64_t* ids, const float* dx, int rows, int E, int V, float* demb, void* stream) {
    return capdec_embed_bwd_rows(ids, dx, rows, E, V, demb, nullptr, stream);
}

int capdec_embed_bwd_rows(const int64_t* ids, const float* dx, int rows, int E, int V, float* demb,
                          const int32_t* rows_dev, void* stream) {
    return capdec_embed_train_bwd(ids, dx, rows, E, V, 0.f, 0, nullptr, demb, rows_dev, stream);
}

int capdec_embed_train_fwd(const int64_t* ids, int64_t ldid, const int64_t* order, const float* emb, const float* pe, int M,
                           int T, int E, int V, float p, uint64_t seed, const uint64_t* seed_dev, float* x, int64_t* ids_out,
                           void* stream) {
    CAPDEC_REQUIRE(ids && emb && x && M >= 0 && T >= 1 && ldid >= T && E >= 4 && E % 4 == 0 && V >= 1 && p >= 0.f && p < 1.f);
    if (!capdec_aligned16(emb) || !capdec_aligned16(x) || (pe && !capdec_aligned16(pe))) return CAPDEC_EALIGN;
    if (M == 0) return 0;
    hipLaunchKernelGGL(embed_train_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, ldid, order, emb, pe,
                       M, T, E, V, p, seed, seed_dev, x, ids_out);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_embed_train_bwd(const int64_t* ids, const float* dx, int rows, int E, int V, float p, uint64_t seed,
                           const uint64_t* seed_dev, float* demb, const int32_t* rows_dev, void* stream) {
    CAPDEC_REQUIRE(ids && dx && demb && rows >= 0 && E >= 1 && V >= 1 && p >= 0.f && p < 1.f);
    if (rows == 0) return 0;
    CAPDEC_REQUIRE((size_t)EB_HOT * E * sizeof(float) <= 60 * 1024);
    int per = (rows + 255) / 256;                      // about one workgroup per CU before the rows per workgroup grow
    per = per < 8 ? 8 : (per > 64 ? 64 : (per + 3) / 4 * 4);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((rows + per - 1) / per), dim3(256), EB_HOT * E * sizeof(float),
                       (hipStream_t)stream, ids, dx, rows, E, V, demb, rows_dev, per, p, seed, seed_dev);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_bn_train_backward(const float* dy, const float* x, const float* gamma, const float* save_mean,
                             const float* save_rstd, int B, int C, float* dx, float* dgamma, float* dbeta,
                             void* stream) {
    CAPDEC_REQUIRE(dy && x && gamma && save_mean && save_rstd && dx && dgamma && dbeta && B >= 1 && C >= 1);
    hipLaunchKernelGGL(bn_bwd_kernel, dim3((C + BN_CW - 1) / BN_CW), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, save_mean,
                       save_rstd, B, C, dx, dgamma, dbeta);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

static CAPDEC_HOOK_VAR(g_attn_bwd_two_pass);
static CAPDEC_HOOK_VAR(g_attn_bwd_rows);
#ifdef CAPDEC_TUNING_HOOKS
int capdec_attention_bwd_force_two_pass(int on) {
    g_attn_bwd_two_pass = on ? 1 : 0;
    return 0;
}
int capdec_attention_bwd_force_row_form(int on) {
    g_attn_bwd_rows = on ? 1 : 0;
    return 0;
}
#endif

int capdec_attention_bwd_ragged(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                const float* dout, int64_t ldo, const float* lse, float* dq, int64_t lddq, float* dk,
                                int64_t lddk, float* dv, int64_t lddv, int B, int H, int T, int dkh, const int32_t* ids,
                                const int32_t* seq_off, void* stream) {
    CAPDEC_REQUIRE(q && k && v && dout && lse && dq && dk && dv && seq_off && B >= 0 && H >= 1 && T >= 1 && T <= 64);
    CAPDEC_REQUIRE(dkh == 16 || dkh == 32 || dkh == 64 || dkh == 128 || dkh == 256);
    if (!capdec_aligned16(q) || !capdec_aligned16(k) || !capdec_aligned16(v) || !capdec_aligned16(dout) ||
        !capdec_aligned16(dq) || !capdec_aligned16(dk) || !capdec_aligned16(dv) || ldq % 4 || ldk % 4 || ldv % 4 ||
        ldo % 4 || lddq % 4 || lddk % 4 || lddv % 4)
        return CAPDEC_EALIGN;
    if (B == 0) return 0;
    const size_t fused_lds = ((size_t)4 * T * (dkh + 16) + (size_t)2 * T * T) * sizeof(float);
    CAPDEC_REQUIRE(fused_lds <= 144 * 1024);
    AttnBwdArgs a{q, ldq, k, ldk, 0, v, ldv, 0, dout, ldo, nullptr, 0, lse, nullptr, dq, lddq, dk, lddk, 0, dv, lddv, 0,
                  B, H, T, T, dkh, 1, 0, ids, 0, seq_off, 0};
    if (dkh == ABM_DK && T <= ABM_T && !g_attn_bwd_two_pass && !g_attn_bwd_rows) {
        // the caption decoder's own shape: the five contractions of a (sample, head) on the matrix cores
        static const bool ready_m = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_mfma_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)abm_lds_bytes(ABM_T)) == hipSuccess;
        if (!ready_m) return CAPDEC_EINVAL;
        hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3((unsigned)(B * H)), dim3(256), abm_lds_bytes(T), (hipStream_t)stream, a);
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    static const bool ready = [] {
        bool ok = true;
#define CAPDEC_CALL(NM_)                                                                                    \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<NM_>),               \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) == hipSuccess
        CAPDEC_CALL(1); CAPDEC_CALL(2); CAPDEC_CALL(4); CAPDEC_CALL(8); CAPDEC_CALL(16);
#undef CAPDEC_CALL
        return ok;
    }();
    if (!ready) return CAPDEC_EINVAL;
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attn_bwd_fused_kernel<NM_>, dim3((unsigned)(B * H)), dim3(256), fused_lds, (hipStream_t)stream, a)
    CAPDEC_DK_DISPATCH(dkh, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_cross_attention_bwd_ragged(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                      int64_t ldv, const float* dout, int64_t ldo, const float* out, int64_t ldout,
                                      const float* lse, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                                      int64_t lddv, int B, int H, int T, int S, int dkh, const int32_t* seq_off,
                                      void* stream) {
    CAPDEC_REQUIRE(q && k && v && dout && lse && dq && dk && dv && seq_off && B >= 0 && H >= 1 && T >= 1 && T <= 64 &&
                   S >= 1 && S <= 2048);
    if (S > 64) {
        // long memories: keys streamed through LDS in chunks; needs the forward output
        CAPDEC_REQUIRE(out && ldout % 4 == 0 && capdec_aligned16(out));
        CAPDEC_REQUIRE(dkh == 16 || dkh == 32 || dkh == 64 || dkh == 128 || dkh == 256);
        if (!capdec_aligned16(q) || !capdec_aligned16(k) || !capdec_aligned16(v) || !capdec_aligned16(dout) ||
            !capdec_aligned16(dq) || !capdec_aligned16(dk) || !capdec_aligned16(dv) || ldq % 4 || ldk % 4 || ldv % 4 ||
            ldo % 4 || lddq % 4 || lddk % 4 || lddv % 4)
            return CAPDEC_EALIGN;
        if (B == 0) return 0;
        const size_t chunk_lds = ((size_t)(3 * T + 2 * AB_KC) * (dkh + 16) + (size_t)2 * T * AB_KC + 2 * T) * sizeof(float);
        CAPDEC_REQUIRE(chunk_lds <= 144 * 1024);
        AttnBwdArgs a{q, ldq, k, ldk, (int64_t)S * ldk, v, ldv, (int64_t)S * ldv, dout, ldo, out, ldout, lse, nullptr, dq,
                      lddq, dk, lddk, (int64_t)S * lddk, dv, lddv, (int64_t)S * lddv, B, H, T, S, dkh, 0, 0, nullptr, 0,
                      seq_off, 1};
        static const bool ready_c = [] {
            bool ok = true;
#define CAPDEC_CALL(NM_)                                                                                    \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_chunked_kernel<NM_>),             \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) == hipSuccess
            CAPDEC_CALL(1); CAPDEC_CALL(2); CAPDEC_CALL(4); CAPDEC_CALL(8); CAPDEC_CALL(16);
#undef CAPDEC_CALL
            return ok;
        }();
        if (!ready_c) return CAPDEC_EINVAL;
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attn_bwd_chunked_kernel<NM_>, dim3((unsigned)(B * H)), dim3(256), chunk_lds, (hipStream_t)stream, a)
        CAPDEC_DK_DISPATCH(dkh, CAPDEC_CALL);
#undef CAPDEC_CALL
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    CAPDEC_REQUIRE(dkh == 16 || dkh == 32 || dkh == 64 || dkh == 128 || dkh == 256);
    if (!capdec_aligned16(q) || !capdec_aligned16(k) || !capdec_aligned16(v) || !capdec_aligned16(dout) ||
        !capdec_aligned16(dq) || !capdec_aligned16(dk) || !capdec_aligned16(dv) || ldq % 4 || ldk % 4 || ldv % 4 ||
        ldo % 4 || lddq % 4 || lddk % 4 || lddv % 4)
        return CAPDEC_EALIGN;
    if (B == 0) return 0;
    const size_t fused_lds = ((size_t)2 * (T + S) * (dkh + 16) + (size_t)2 * T * S) * sizeof(float);
    CAPDEC_REQUIRE(fused_lds <= 144 * 1024);
    AttnBwdArgs a{q, ldq, k, ldk, (int64_t)S * ldk, v, ldv, (int64_t)S * ldv, dout, ldo, nullptr, 0, lse, nullptr, dq, lddq,
                  dk, lddk, (int64_t)S * lddk, dv, lddv, (int64_t)S * lddv, B, H, T, S, dkh, 0, 0, nullptr, 0, seq_off, 1};
    static const bool ready = [] {
        bool ok = true;
#define CAPDEC_CALL(NM_)                                                                                    \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<NM_>),               \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) == hipSuccess
        CAPDEC_CALL(1); CAPDEC_CALL(2); CAPDEC_CALL(4); CAPDEC_CALL(8); CAPDEC_CALL(16);
#undef CAPDEC_CALL
        return ok;
    }();
    if (!ready) return CAPDEC_EINVAL;
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attn_bwd_fused_kernel<NM_>, dim3((unsigned)(B * H)), dim3(256), fused_lds, (hipStream_t)stream, a)
    CAPDEC_DK_DISPATCH(dkh, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_attention_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, int64_t sk, const float* v,
                         int64_t ldv, int64_t sv, const float* dout, int64_t ldo, const float* out, int64_t ldout,
                         const float* lse, float* dsum,
                         float* dq, int64_t lddq, float* dk, int64_t lddk, int64_t sdk, float* dv, int64_t lddv,
                         int64_t sdv, int B, int H, int Tq, int Tk, int dkh, int causal, int causal_off,
                         const int32_t* ids, int64_t ldid, void* stream) {
    CAPDEC_REQUIRE(q && k && v && dout && lse && dsum && dq && dk && dv && B >= 0 && H >= 1 && Tq >= 1 && Tk >= 1 &&
                   Tk <= 2048 && Tq <= 2048);
    CAPDEC_REQUIRE(dkh == 16 || dkh == 32 || dkh == 64 || dkh == 128 || dkh == 256);
    CAPDEC_REQUIRE(!ids || ldid >= Tq);
    if (!capdec_aligned16(q) || !capdec_aligned16(k) || !capdec_aligned16(v) || !capdec_aligned16(dout) ||
        !capdec_aligned16(dq) || !capdec_aligned16(dk) || !capdec_aligned16(dv) || ldq % 4 || ldk % 4 || sk % 4 ||
        ldv % 4 || sv % 4 || ldo % 4 || lddq % 4 || lddk % 4 || sdk % 4 || lddv % 4 || sdv % 4)
        return CAPDEC_EALIGN;
    if (B == 0) return 0;
    if (out && (!capdec_aligned16(out) || ldout % 4)) return CAPDEC_EALIGN;
    AttnBwdArgs a{q, ldq, k, ldk, sk, v, ldv, sv, dout, ldo, out, ldout, lse, dsum, dq, lddq, dk, lddk, sdk, dv, lddv, sdv,
                  B, H, Tq, Tk, dkh, causal, causal_off, ids, ldid};
    hipStream_t s = (hipStream_t)stream;
    if (dkh == ABM_DK && Tq == Tk && Tq <= ABM_T && !g_attn_bwd_two_pass && !g_attn_bwd_rows) {
        // self-attention at the caption decoder's own shape: on the matrix cores (see attn_bwd_mfma_kernel)
        static const bool ready_m = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_mfma_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)abm_lds_bytes(ABM_T)) == hipSuccess;
        if (!ready_m) return CAPDEC_EINVAL;
        hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3((unsigned)(B * H)), dim3(256), abm_lds_bytes(Tq), s, a);
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    // short sequences: everything of one (b,h) fits in LDS
    const size_t fused_lds = ((size_t)2 * (Tq + Tk) * (dkh + 16) + (size_t)2 * Tq * Tk) * sizeof(float);
    if (Tq <= 64 && Tk <= 64 && fused_lds <= 144 * 1024 && !g_attn_bwd_two_pass) {
        static const bool ready = [] {
            bool ok = true;
#define CAPDEC_CALL(NM_)                                                                                    \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<NM_>),               \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) == hipSuccess
            CAPDEC_CALL(1); CAPDEC_CALL(2); CAPDEC_CALL(4); CAPDEC_CALL(8); CAPDEC_CALL(16);
#undef CAPDEC_CALL
            return ok;
        }();
        if (!ready) return CAPDEC_EINVAL;
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attn_bwd_fused_kernel<NM_>, dim3((unsigned)(B * H)), dim3(256), fused_lds, s, a)
        CAPDEC_DK_DISPATCH(dkh, CAPDEC_CALL);
#undef CAPDEC_CALL
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    // long unmasked memory with the forward output at hand: keys streamed through LDS in chunks
    const size_t chunk_lds = ((size_t)(3 * Tq + 2 * AB_KC) * (dkh + 16) + (size_t)2 * Tq * AB_KC + 2 * Tq) * sizeof(float);
    if (out && !causal && !ids && Tq <= 64 && chunk_lds <= 144 * 1024 && !g_attn_bwd_two_pass) {
        static const bool ready = [] {
            bool ok = true;
#define CAPDEC_CALL(NM_)                                                                                    \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_chunked_kernel<NM_>),             \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) == hipSuccess
            CAPDEC_CALL(1); CAPDEC_CALL(2); CAPDEC_CALL(4); CAPDEC_CALL(8); CAPDEC_CALL(16);
#undef CAPDEC_CALL
            return ok;
        }();
        if (!ready) return CAPDEC_EINVAL;
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attn_bwd_chunked_kernel<NM_>, dim3((unsigned)(B * H)), dim3(256), chunk_lds, s, a)
        CAPDEC_DK_DISPATCH(dkh, CAPDEC_CALL);
#undef CAPDEC_CALL
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    const long tq = (long)B * H * Tq, tk = (long)B * H * Tk;
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attn_bwd_q_kernel<NM_>, dim3((unsigned)((tq + 3) / 4)), dim3(256), 4 * 2 * Tk * sizeof(float), s, a)
    CAPDEC_DK_DISPATCH(dkh, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
#define CAPDEC_CALL(NM_) \
    hipLaunchKernelGGL(attn_bwd_kv_kernel<NM_>, dim3((unsigned)((tk + 3) / 4)), dim3(256), 4 * 2 * Tq * sizeof(float), s, a)
    CAPDEC_DK_DISPATCH(dkh, CAPDEC_CALL);
#undef CAPDEC_CALL
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_softmax_nll_grad(float* logits, int64_t ld, const int64_t* targets, int M, int V, int64_t ignore_index,
                            float* sums, float* gscale, float* lse_out, void* stream) {
    return capdec_softmax_nll_grad_rows(logits, ld, targets, M, V, ignore_index, sums, gscale, lse_out, nullptr, stream);
}

int capdec_softmax_nll_grad_rows(float* logits, int64_t ld, const int64_t* targets, int M, int V, int64_t ignore_index,
                                 float* sums, float* gscale, float* lse_out, const int32_t* rows_dev, void* stream) {
    CAPDEC_REQUIRE(logits && targets && sums && gscale && M >= 0 && V >= 1 && ld >= V);
    if (M == 0) return 0;
    const int count_grid = M >= 16 * 256 ? 16 : (M + 255) / 256;
    hipLaunchKernelGGL(count_targets_kernel, dim3(count_grid), dim3(256), 0, (hipStream_t)stream, targets, M, ignore_index,
                       sums);
    const bool vec = V % 4 == 0 && ld % 4 == 0 && capdec_aligned16(logits);
    const dim3 grid((M + SM_ROWS - 1) / SM_ROWS);
    if (vec && V > 256 * 20 && V <= 256 * 40)
        hipLaunchKernelGGL(softmax_nll_grad_reg_kernel<40>, grid, dim3(256), 0, (hipStream_t)stream, logits, ld, targets,
                           M, V, ignore_index, gscale, sums, lse_out, rows_dev);
    else if (vec)
        hipLaunchKernelGGL(softmax_nll_grad_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, logits, ld, targets, M,
                           V, ignore_index, gscale, sums, lse_out, rows_dev);
    else
        hipLaunchKernelGGL(softmax_nll_grad_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, logits, ld, targets, M,
                           V, ignore_index, gscale, sums, lse_out, rows_dev);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_compact_targets(const int64_t* targets, int M, int64_t ignore_index, int64_t* order, int64_t* slot,
                           int64_t* targets_c, int32_t* count, void* stream) {
    CAPDEC_REQUIRE(targets && order && slot && targets_c && count && M >= 0);
    if (M == 0) return (int)hipMemsetAsync(count, 0, sizeof(int32_t), (hipStream_t)stream);
    hipLaunchKernelGGL(compact_targets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, targets, M, ignore_index, order,
                       slot, targets_c, count);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_compact_prefix(const int64_t* ids, const int64_t* targets, int B, int T, int64_t pad, int64_t* order,
                          int64_t* slot, int32_t* seq_off, int32_t* sample, int32_t* ids_c, int64_t* targets_c,
                          int32_t* count, void* stream) {
    return capdec_compact_prefix_strided(ids, targets, T, B, T, pad, order, slot, seq_off, sample, ids_c, targets_c, count,
                                         stream);
}

int capdec_compact_prefix_strided(const int64_t* ids, const int64_t* targets, int64_t ld, int B, int T, int64_t pad,
                                  int64_t* order, int64_t* slot, int32_t* seq_off, int32_t* sample, int32_t* ids_c,
                                  int64_t* targets_c, int32_t* count, void* stream) {
    CAPDEC_REQUIRE(ids && targets && order && slot && seq_off && sample && ids_c && targets_c && count);
    CAPDEC_REQUIRE(B >= 1 && T >= 1 && ld >= T && (int64_t)B * T < INT_MAX);
    const size_t smem = ((size_t)3 * B + 1024) * sizeof(int);
    CAPDEC_REQUIRE(smem <= 144 * 1024);           // B <= 11946: the per-sample bookkeeping lives in LDS
    static const bool ready = hipFuncSetAttribute(reinterpret_cast<const void*>(compact_prefix_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) == hipSuccess;
    if (!ready) return CAPDEC_EINVAL;
    hipLaunchKernelGGL(compact_prefix_kernel, dim3(1), dim3(1024), smem, (hipStream_t)stream, ids, targets, ld, B, T, pad,
                       order, slot, seq_off, sample, ids_c, targets_c, count);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_loss_finish(const float* sums, float* loss, float* gscale, void* stream) {
    CAPDEC_REQUIRE(sums && loss && gscale);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, loss, gscale);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, int step, float grad_scale, void* stream) {
    CAPDEC_REQUIRE(p && g && m && v && n >= 0 && step >= 1);
    if (n == 0) return 0;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, bc1, bc2_sqrt, grad_scale);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_train_state_advance(capdec_train_state* st, float beta1, float beta2, void* stream) {
    CAPDEC_REQUIRE(st && beta1 > 0.f && beta1 < 1.f && beta2 > 0.f && beta2 < 1.f);
    hipLaunchKernelGGL(train_state_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, st, beta1, beta2);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                         const capdec_train_state* st, float grad_scale, void* stream) {
    CAPDEC_REQUIRE(p && g && m && v && st && n >= 0);
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2,
                       eps, st, grad_scale);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
