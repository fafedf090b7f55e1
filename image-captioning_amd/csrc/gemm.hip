// fp32 MFMA contraction kernels for gfx950 (v_mfma_f32_32x32x2_f32, exact fp32 fma chains).
//
//   D[i,j] = sum_k P[i,k] * Q[j,k]         P: [I,K] rows, Q: [J,K] rows (both K-contiguous)
//
// A 256-thread workgroup (4 wavefronts as 2x2) owns a (64*WI) x (64*WJ) tile; each wavefront owns
// WI x WJ accumulators of 32x32.  In the accumulator the j index (rows of Q) sits on the lane and
// the i index (rows of P) in the 16 registers, so
//   - a plain store with j contiguous is a 128-byte segment per register (coalesced), and
//   - any reduction over i (arg-max / log-sum-exp over the vocabulary when P is the vocabulary
//     matrix) is lane-local: no cross-lane traffic until one final exchange between lane halves.
// Operand tiles go global -> registers -> LDS (register staging, next tile's loads in flight while
// the current one is multiplied); LDS rows are padded to 36 floats so the ds_read_b128 fragment
// reads (32 rows x 16 B per half-wave) are bank-conflict free.
#include "common.h"
#include <limits.h>
#include <math.h>

namespace {

constexpr int BK = 32;    // k extent of one staged tile
constexpr int LDK = 36;   // padded LDS row stride in floats (144 B) of a k-contiguous operand tile

struct GemmKArgs {
    capdec_gemm_group g[4];
    int K;
    int flags;
    capdec_gemm_extra x;
};

template <int WI, int WJ>
constexpr int lds_floats() { return 2 * (64 * WI + 64 * WJ) * LDK; }

// One operand tile of NX = 64*W rows (the i or j axis) by BK columns (the contraction axis).
//  T == false: the operand is stored [row][k] (k contiguous).  LDS image [row][LDK]; a lane reads its four
//              k values with one ds_read_b128.
//  T == true : the operand is stored [k][row] (row contiguous; the contraction runs over its leading
//              axis, as in dX = dY W and dW = dY^T X).  LDS image [k][NX+4]; a lane reads its four k values
//              with four ds_read_b32 (32 consecutive floats per half-wave: conflict-free).
template <int W, bool VEC, bool T>
struct Operand {
    static constexpr int NX = 64 * W;
    static constexpr int NR = NX / 32;            // float4 per thread per tile
    static constexpr int LDT = NX + 4;
    static constexpr int NV = NX / 4;             // float4 per k-row when T
    static constexpr int KPP = 256 / NV;          // k-rows per pass when T
    const float* __restrict__ base;
    int64_t ld;
    int n, K, x0;
    float* lds;                                   // 2 buffers
    f32x4 reg[NR];

    __device__ __forceinline__ void gload(int kt) {
        const int tid = threadIdx.x;
        if (!T) {
            const int lrow = tid >> 3, k = kt * BK + (tid & 7) * 4;
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                const int row = x0 + lrow + 32 * p;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < n) {
                    const float* q = base + (int64_t)row * ld + k;
                    if (VEC) {
                        if (k < K) v = *reinterpret_cast<const f32x4*>(q);
                    } else {
                        if (k + 0 < K) v[0] = q[0];
                        if (k + 1 < K) v[1] = q[1];
                        if (k + 2 < K) v[2] = q[2];
                        if (k + 3 < K) v[3] = q[3];
                    }
                }
                reg[p] = v;
            }
        } else {
            const int iv = tid % NV, kr = tid / NV;
            const int row = x0 + 4 * iv;
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                const int k = kt * BK + kr + KPP * p;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < K) {
                    const float* q = base + (int64_t)k * ld + row;
                    if (VEC && row + 3 < n) {
                        v = *reinterpret_cast<const f32x4*>(q);
                    } else {
                        if (row + 0 < n) v[0] = q[0];
                        if (row + 1 < n) v[1] = q[1];
                        if (row + 2 < n) v[2] = q[2];
                        if (row + 3 < n) v[3] = q[3];
                    }
                }
                reg[p] = v;
            }
        }
    }
    __device__ __forceinline__ void lstore(int buf) {
        const int tid = threadIdx.x;
        if (!T) {
            const int lrow = tid >> 3, lk = (tid & 7) * 4;
#pragma unroll
            for (int p = 0; p < NR; ++p)
                *reinterpret_cast<f32x4*>(&lds[(buf * NX + lrow + 32 * p) * LDK + lk]) = reg[p];
        } else {
            const int iv = tid % NV, kr = tid / NV;
#pragma unroll
            for (int p = 0; p < NR; ++p)
                *reinterpret_cast<f32x4*>(&lds[(buf * BK + kr + KPP * p) * LDT + 4 * iv]) = reg[p];
        }
    }
    // the four k values {kc*8 + 4h + m} of row (w0 + r) for the lane (r, h)
    __device__ __forceinline__ f32x4 frag(int buf, int w0, int kc, int r, int h) const {
        if (!T) return *reinterpret_cast<const f32x4*>(&lds[(buf * NX + w0 + r) * LDK + kc * 8 + 4 * h]);
        const float* q = &lds[(buf * BK + kc * 8 + 4 * h) * LDT + w0 + r];
        return f32x4{q[0], q[LDT], q[2 * LDT], q[3 * LDT]};
    }
};

// Accumulate the (i0, j0) tile into acc.  Ends with every wave past the last barrier, so the caller
// may reuse `lds` right away.
template <int WI, int WJ, bool VEC, bool TP = false, bool TQ = false>
__device__ __forceinline__ void mainloop(const float* __restrict__ P, const float* __restrict__ Q, int64_t ldp,
                                         int64_t ldq, int I, int J, int K, int i0, int j0, float* lds,
                                         f32x16 (&acc)[WI][WJ]) {
    constexpr int BI = 64 * WI;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    Operand<WI, VEC, TP> op{P, ldp, I, K, i0, lds};
    Operand<WJ, VEC, TQ> oq{Q, ldq, J, K, j0, lds + 2 * BI * LDK};

#pragma unroll
    for (int ti = 0; ti < WI; ++ti)
#pragma unroll
        for (int tj = 0; tj < WJ; ++tj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ti][tj][e] = 0.f;

    const int nk = (K + BK - 1) / BK;
    op.gload(0);
    oq.gload(0);
    op.lstore(0);
    oq.lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
            op.gload(kt + 1);
            oq.gload(kt + 1);
        }
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            f32x4 a[WI], b[WJ];
#pragma unroll
            for (int ti = 0; ti < WI; ++ti) a[ti] = op.frag(cur, wi * 32 * WI + ti * 32, kc, r, h);
#pragma unroll
            for (int tj = 0; tj < WJ; ++tj) b[tj] = oq.frag(cur, wj * 32 * WJ + tj * 32, kc, r, h);
            // lane half h feeds k = kc*8 + 4h + m, so MFMA m sums k in {kc*8+m, kc*8+4+m}
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ti = 0; ti < WI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < WJ; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti][m], b[tj][m], acc[ti][tj], 0, 0, 0);
        }
        if (more) {
            op.lstore(cur ^ 1);
            oq.lstore(cur ^ 1);
        }
        __syncthreads();
        cur ^= 1;
    }
}

// accumulator register e of lane-half h is row (e&3) + 8*(e>>2) + 4*h of the 32x32 tile
__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

template <int WI, int WJ, bool VEC, bool TP, bool TQ>
__global__ __launch_bounds__(256) void gemm_kernel(GemmKArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const capdec_gemm_group& g = args.g[blockIdx.z];
    constexpr int BI = 64 * WI, BJ = 64 * WJ;
    const int i0 = blockIdx.y * BI, j0 = blockIdx.x * BJ;
    if (i0 >= g.I || j0 >= g.J) return;   // groups of different size share the grid
    f32x16 acc[WI][WJ];
    mainloop<WI, WJ, VEC, TP, TQ>(g.P, g.Q, g.ldp, g.ldq, g.I, g.J, args.K, i0, j0, lds, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1, r = lane & 31, h = lane >> 5;
    const bool relu = (args.flags & CAPDEC_GEMM_RELU) != 0;
    const capdec_gemm_extra& x = args.x;
#pragma unroll
    for (int tj = 0; tj < WJ; ++tj) {
        const int j = j0 + wj * 32 * WJ + tj * 32 + r;
        if (j >= g.J) continue;
        const float bj = g.bias_j ? g.bias_j[j] : 0.f;
        const float sc = g.scale_j ? g.scale_j[j] : 1.f;
        const float sh = g.shift_j ? g.shift_j[j] : 0.f;
#pragma unroll
        for (int ti = 0; ti < WI; ++ti) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + wi * 32 * WI + ti * 32 + acc_row(e, h);
                if (i >= g.I) continue;
                float v = acc[ti][tj][e] + bj;
                if (g.bias_i) v += g.bias_i[i];
                if (g.scale_j) v = v * sc + sh;
                if (relu) v = fmaxf(v, 0.f);
                if (x.lse) {
                    // d(mean NLL)/d(logit): (softmax - onehot) * gscale on rows with a counted target
                    const int64_t tg = x.tgt[i];
                    v = (tg == x.ignore_index) ? 0.f : (expf(v - x.lse[i]) - (tg == (int64_t)j ? 1.f : 0.f)) * x.gscale[0];
                }
                if (x.mask && !(x.mask[(int64_t)i * x.ldm + j] > 0.f)) v = 0.f;
                if (g.R) v += g.R[(int64_t)i * g.ldri + (int64_t)j * g.ldrj];
                g.C[(int64_t)i * g.ldci + (int64_t)j * g.ldcj] = v;
            }
        }
    }
}

// ---- vocabulary projection with lane-local reductions over the vocabulary axis -------------------
// P = W [V,K] (i = vocabulary id), Q = x [M,K] (j = row).  MODE 0: arg-max; MODE 1: (max, sum exp).
template <int WI, int WJ, bool VEC, int MODE>
__global__ __launch_bounds__(256) void vocab_reduce_kernel(const float* __restrict__ W, const float* __restrict__ x,
                                                           int64_t ldx, const float* __restrict__ bias, int V, int M,
                                                           int K, float* __restrict__ out_a, void* __restrict__ out_b,
                                                           int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BI = 64 * WI, BJ = 64 * WJ;
    const int i0 = blockIdx.y * BI, j0 = blockIdx.x * BJ;
    f32x16 acc[WI][WJ];
    mainloop<WI, WJ, VEC>(W, x, (int64_t)K, ldx, V, M, K, i0, j0, lds, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = wave >> 1, wj = wave & 1, r = lane & 31, h = lane >> 5;
    float* sm_a = lds;                                   // [2][BJ]
    float* sm_b = lds + 2 * BJ;                          // [2][BJ] (float sum or int index bits)

    float bi[WI][16];
#pragma unroll
    for (int ti = 0; ti < WI; ++ti)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wi * 32 * WI + ti * 32 + acc_row(e, h);
            bi[ti][e] = (i < V) ? bias[i] : 0.f;
        }

#pragma unroll
    for (int tj = 0; tj < WJ; ++tj) {
        float best = -INFINITY;
        int bidx = INT_MAX;
        // registers walk the vocabulary in increasing order for a fixed lane half, so a strict
        // compare keeps the lowest index on ties
#pragma unroll
        for (int ti = 0; ti < WI; ++ti)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + wi * 32 * WI + ti * 32 + acc_row(e, h);
                const float v = acc[ti][tj][e] + bi[ti][e];
                if (i < V && v > best) { best = v; bidx = i; }
            }
        float sum = 0.f;
        if (MODE == 1) {
#pragma unroll
            for (int ti = 0; ti < WI; ++ti)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = i0 + wi * 32 * WI + ti * 32 + acc_row(e, h);
                    if (i < V) sum += __expf(acc[ti][tj][e] + bi[ti][e] - best);
                }
        }
        // the other lane half holds the interleaved vocabulary rows of the same column
        const float v2 = __shfl_xor(best, 32, 64);
        if (MODE == 0) {
            const int i2 = __shfl_xor(bidx, 32, 64);
            argmax_merge(best, bidx, v2, i2);
        } else {
            const float s2 = __shfl_xor(sum, 32, 64);
            const float mx = fmaxf(best, v2);
            float s = 0.f;
            if (best > -INFINITY) s += sum * __expf(best - mx);
            if (v2 > -INFINITY) s += s2 * __expf(v2 - mx);
            best = mx;
            sum = s;
        }
        if (h == 0) {
            const int jl = wj * 32 * WJ + tj * 32 + r;
            sm_a[wi * BJ + jl] = best;
            if (MODE == 0) reinterpret_cast<int*>(sm_b)[wi * BJ + jl] = bidx;
            else sm_b[wi * BJ + jl] = sum;
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < BJ && j0 + t < M) {
        float a0 = sm_a[t], a1 = sm_a[BJ + t];
        const int64_t o = (int64_t)(j0 + t) * ntiles + blockIdx.y;
        if (MODE == 0) {
            int b0 = reinterpret_cast<int*>(sm_b)[t], b1 = reinterpret_cast<int*>(sm_b)[BJ + t];
            argmax_merge(a0, b0, a1, b1);
            out_a[o] = a0;
            reinterpret_cast<int*>(out_b)[o] = b0;
        } else {
            const float s0 = sm_b[t], s1 = sm_b[BJ + t];
            const float mx = fmaxf(a0, a1);
            float s = 0.f;
            if (a0 > -INFINITY) s += s0 * __expf(a0 - mx);
            if (a1 > -INFINITY) s += s1 * __expf(a1 - mx);
            out_a[o] = mx;
            reinterpret_cast<float*>(out_b)[o] = s;
        }
    }
}

// one wavefront per row: merge the per-tile winners
__global__ __launch_bounds__(256) void argmax_finalize_kernel(const float* __restrict__ pv, const int* __restrict__ pi,
                                                              int M, int ntiles, int* __restrict__ out_idx,
                                                              float* __restrict__ out_val) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float v = -INFINITY;
    int i = INT_MAX;
    for (int t = lane; t < ntiles; t += 64) argmax_merge(v, i, pv[(int64_t)row * ntiles + t], pi[(int64_t)row * ntiles + t]);
    wave_argmax(v, i);
    if (lane == 0) {
        out_idx[row] = i;
        if (out_val) out_val[row] = v;
    }
}

// one wavefront per row: lse from per-tile (max, sum), target logit by a K-long dot product
__global__ __launch_bounds__(256) void nll_finalize_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                           int M, int ntiles, const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           int K, const int64_t* __restrict__ targets,
                                                           int64_t ignore_index, float* __restrict__ lse_out,
                                                           float* __restrict__ nll_out, float* __restrict__ loss_sum) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float mx = -INFINITY;
    for (int t = lane; t < ntiles; t += 64) mx = fmaxf(mx, pm[(int64_t)row * ntiles + t]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int t = lane; t < ntiles; t += 64) {
        const float m = pm[(int64_t)row * ntiles + t];
        if (m > -INFINITY) s += ps[(int64_t)row * ntiles + t] * expf(m - mx);
    }
    s = wave_sum(s);
    const float lse = mx + logf(s);
    const int64_t tg = targets[row];
    float nll = 0.f;
    if (tg != ignore_index) {
        float d = 0.f;
        for (int k = lane; k < K; k += 64) d += x[(int64_t)row * ldx + k] * W[tg * (int64_t)K + k];
        d = wave_sum(d) + bias[tg];
        nll = lse - d;
    }
    if (lane == 0) {
        if (lse_out) lse_out[row] = lse;
        if (nll_out) nll_out[row] = nll;
        if (tg != ignore_index && loss_sum) {
            atomicAdd(&loss_sum[0], nll);
            atomicAdd(&loss_sum[1], 1.f);
        }
    }
}

// workgroups above 64 KiB of dynamic LDS need the limit raised once per kernel
template <typename F>
bool allow_lds(F* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes) == hipSuccess;
}

template <int WI, int WJ, bool TP, bool TQ>
int launch_gemm_cfg(const GemmKArgs& a, int ngroups, int maxI, int maxJ, bool vec, hipStream_t s) {
    dim3 grid((maxJ + 64 * WJ - 1) / (64 * WJ), (maxI + 64 * WI - 1) / (64 * WI), ngroups);
    constexpr size_t smem = lds_floats<WI, WJ>() * sizeof(float);
    static const bool ready = allow_lds(gemm_kernel<WI, WJ, true, TP, TQ>, smem) &&
                              allow_lds(gemm_kernel<WI, WJ, false, TP, TQ>, smem);
    if (!ready) return CAPDEC_EINVAL;
    if (vec) hipLaunchKernelGGL((gemm_kernel<WI, WJ, true, TP, TQ>), grid, dim3(256), smem, s, a);
    else hipLaunchKernelGGL((gemm_kernel<WI, WJ, false, TP, TQ>), grid, dim3(256), smem, s, a);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

template <bool TP, bool TQ>
int launch_gemm_layout(const GemmKArgs& a, int ngroups, int maxI, int maxJ, bool vec, hipStream_t s) {
    const long tiles128 = (long)((maxI + 127) / 128) * ((maxJ + 127) / 128) * ngroups;
    if (tiles128 >= 192) return launch_gemm_cfg<2, 2, TP, TQ>(a, ngroups, maxI, maxJ, vec, s);
    return launch_gemm_cfg<1, 1, TP, TQ>(a, ngroups, maxI, maxJ, vec, s);
}

constexpr int VOCAB_BI = 128;   // vocabulary ids per tile of the fused projections

}  // namespace

int capdec_launch_gemm(const capdec_gemm_group* groups, int ngroups, int K, int flags, hipStream_t s,
                       const capdec_gemm_extra* extra) {
    CAPDEC_REQUIRE(groups && ngroups >= 1 && ngroups <= 4 && K >= 1);
    const bool tp = (flags & CAPDEC_GEMM_TP) != 0, tq = (flags & CAPDEC_GEMM_TQ) != 0;
    CAPDEC_REQUIRE(!(tp && !tq));          // layouts built: (P,Q) = (n,n), (n,t), (t,t)
    GemmKArgs a;
    a.K = K;
    a.flags = flags;
    a.x = capdec_gemm_extra{};
    if (extra) {
        a.x = *extra;
        CAPDEC_REQUIRE(!a.x.lse || (a.x.tgt && a.x.gscale));
    }
    int maxI = 0, maxJ = 0;
    bool vec = true;
    for (int n = 0; n < ngroups; ++n) {
        const capdec_gemm_group& g = groups[n];
        CAPDEC_REQUIRE(g.P && g.Q && g.C && g.I >= 0 && g.J >= 0);
        CAPDEC_REQUIRE(g.ldp >= (tp ? g.I : K) && g.ldq >= (tq ? g.J : K));
        CAPDEC_REQUIRE((g.scale_j == nullptr) == (g.shift_j == nullptr));
        a.g[n] = g;
        if (g.I > maxI) maxI = g.I;
        if (g.J > maxJ) maxJ = g.J;
        vec = vec && capdec_aligned16(g.P) && capdec_aligned16(g.Q) && g.ldp % 4 == 0 && g.ldq % 4 == 0;
    }
    if (!tp || !tq) vec = vec && (K % 4 == 0);   // a k-contiguous operand is read four k at a time
    for (int n = ngroups; n < 4; ++n) a.g[n] = a.g[0];
    if (maxI == 0 || maxJ == 0) return 0;
    if (tp) return launch_gemm_layout<true, true>(a, ngroups, maxI, maxJ, vec, s);
    if (tq) return launch_gemm_layout<false, true>(a, ngroups, maxI, maxJ, vec, s);
    return launch_gemm_layout<false, false>(a, ngroups, maxI, maxJ, vec, s);
}

extern "C" {

int capdec_gemm(const capdec_gemm_group* groups, int ngroups, int K, int flags, void* stream) {
    return capdec_launch_gemm(groups, ngroups, K, flags, (hipStream_t)stream, nullptr);
}

int capdec_gemm_ex(const capdec_gemm_group* groups, int ngroups, int K, int flags, const capdec_gemm_extra* extra,
                   void* stream) {
    return capdec_launch_gemm(groups, ngroups, K, flags, (hipStream_t)stream, extra);
}

int capdec_vocab_tiles(int V) { return (V + VOCAB_BI - 1) / VOCAB_BI; }

static int vocab_reduce_launch(int mode, const float* x, int64_t ldx, const float* W, const float* bias, int M, int V,
                               int K, float* out_a, void* out_b, hipStream_t s) {
    CAPDEC_REQUIRE(x && W && bias && out_a && out_b && M >= 0 && V >= 1 && K >= 1 && ldx >= K);
    if (M == 0) return 0;
    const int nt = capdec_vocab_tiles(V);
    const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && capdec_aligned16(x) && capdec_aligned16(W);
    // samples sit on the j (lane) axis: 128-wide tiles when there are enough rows to fill the chip
    const bool wide = ((long)nt * ((M + 127) / 128) >= 192);
    const size_t smem = (wide ? lds_floats<2, 2>() : lds_floats<2, 1>()) * sizeof(float);
    static const bool ready = allow_lds(vocab_reduce_kernel<2, 2, true, 0>, lds_floats<2, 2>() * sizeof(float)) &&
                              allow_lds(vocab_reduce_kernel<2, 2, false, 0>, lds_floats<2, 2>() * sizeof(float)) &&
                              allow_lds(vocab_reduce_kernel<2, 2, true, 1>, lds_floats<2, 2>() * sizeof(float)) &&
                              allow_lds(vocab_reduce_kernel<2, 2, false, 1>, lds_floats<2, 2>() * sizeof(float));
    if (!ready) return CAPDEC_EINVAL;
    dim3 grid(wide ? (M + 127) / 128 : (M + 63) / 64, nt, 1);
#define CAPDEC_VR(WJ_, VEC_, MODE_)                                                                               \
    hipLaunchKernelGGL((vocab_reduce_kernel<2, WJ_, VEC_, MODE_>), grid, dim3(256), smem, s, W, x, ldx, bias, V, M, \
                       K, out_a, out_b, nt)
    if (mode == 0) {
        if (wide) { if (vec) CAPDEC_VR(2, true, 0); else CAPDEC_VR(2, false, 0); }
        else      { if (vec) CAPDEC_VR(1, true, 0); else CAPDEC_VR(1, false, 0); }
    } else {
        if (wide) { if (vec) CAPDEC_VR(2, true, 1); else CAPDEC_VR(2, false, 1); }
        else      { if (vec) CAPDEC_VR(1, true, 1); else CAPDEC_VR(1, false, 1); }
    }
#undef CAPDEC_VR
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_vocab_argmax_partial(const float* x, int64_t ldx, const float* W, const float* bias, int M, int V, int K,
                                float* part_val, int32_t* part_idx, void* stream) {
    return vocab_reduce_launch(0, x, ldx, W, bias, M, V, K, part_val, part_idx, (hipStream_t)stream);
}

int capdec_argmax_finalize(const float* part_val, const int32_t* part_idx, int M, int ntiles, int32_t* out_idx,
                           float* out_val, void* stream) {
    CAPDEC_REQUIRE(part_val && part_idx && out_idx && M >= 0 && ntiles >= 1);
    if (M == 0) return 0;
    hipLaunchKernelGGL(argmax_finalize_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, part_val,
                       part_idx, M, ntiles, out_idx, out_val);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_vocab_lse_partial(const float* x, int64_t ldx, const float* W, const float* bias, int M, int V, int K,
                             float* part_max, float* part_sum, void* stream) {
    return vocab_reduce_launch(1, x, ldx, W, bias, M, V, K, part_max, part_sum, (hipStream_t)stream);
}

int capdec_nll_finalize(const float* part_max, const float* part_sum, int M, int ntiles, const float* x, int64_t ldx,
                        const float* W, const float* bias, int K, const int64_t* targets, int64_t ignore_index,
                        float* lse, float* nll, float* loss_sum, void* stream) {
    CAPDEC_REQUIRE(part_max && part_sum && x && W && bias && targets && M >= 0 && ntiles >= 1 && K >= 1);
    if (M == 0) return 0;
    hipLaunchKernelGGL(nll_finalize_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, part_max, part_sum,
                       M, ntiles, x, ldx, W, bias, K, targets, ignore_index, lse, nll, loss_sum);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
