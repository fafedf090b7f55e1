// Fused LSTM step for gfx950: the gate contraction [B, E+Hd] x [E+Hd, 4Hd] on fp32 MFMA with the
// gate non-linearities and the cell update in the epilogue (nn.LSTM gate order i,f,g,o).
//
// A workgroup owns 32 samples x 64 hidden units; its four wavefronts split that tile by unit half (32 units) and by
// gate pair: two wavefronts accumulate the i and f gates, two the g and o gates, each with two 32x32 accumulators whose
// (sample, unit) -> (register, lane) maps are identical.  (One wavefront with all four gates would leave half of the
// chip's SIMDs idle at batch 1024: the MFMA pipe is per SIMD.)  After the k loop the g/o wavefronts pass their
// accumulators through LDS to their i/f partners, which compute c' and h' in registers, so the [B,4Hd] gate tensor
// is never written (unless the caller asks for it for a backward pass).
// The k loop runs over two operand pairs back to back: (x, W_ih) then (h, W_hh).
#include "common.h"
#include <math.h>

namespace {

constexpr int BK = 32;
constexpr int LDK = 36;
constexpr int TB = 32;                 // samples per workgroup
constexpr int TU = 64;                 // hidden units per workgroup
constexpr int LSTM_LDS_FLOATS = 2 * (TB + 4 * TU) * LDK;

template <bool VEC>
__device__ __forceinline__ f32x4 ld4(const float* __restrict__ base, int64_t ld, int row, int nrows, int k, int K) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nrows) {
        const float* p = base + (int64_t)row * ld + k;
        if (VEC) {
            if (k < K) v = *reinterpret_cast<const f32x4*>(p);
        } else {
            if (k + 0 < K) v[0] = p[0];
            if (k + 1 < K) v[1] = p[1];
            if (k + 2 < K) v[2] = p[2];
            if (k + 3 < K) v[3] = p[3];
        }
    }
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct LstmArgs {
    const float* x; int64_t ldx;
    const float* h0; int64_t ldh0; const float* c0;
    const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;
    int B, E, Hd;
    float* h1; int64_t ldh1; float* c1; float* gates;
    const int32_t* live;     // nullable device scalar: zero = a decode loop whose captions have all finished: nothing to do
};

template <bool VEC>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmArgs a) {
    if (a.live && a.live[0] <= 0) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ps = lds;                       // [2][TB][LDK]
    float* Qs = lds + 2 * TB * LDK;        // [2][4*TU][LDK], row = gate*TU + local unit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = wave >> 1, wj = wave & 1, r = lane & 31, h = lane >> 5;   // wg: gate pair (i,f) or (g,o)
    const int lrow = tid >> 3, lk = (tid & 7) * 4;
    const int i0 = blockIdx.y * TB, u0 = blockIdx.x * TU;
    const int nkx = (a.E + BK - 1) / BK;
    const int nkh = a.h0 ? (a.Hd + BK - 1) / BK : 0;
    const int nk = nkx + nkh;

    f32x16 acc[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;

    f32x4 pr[TB / 32], qr[4 * TU / 32];
    auto gload = [&](int kt) {
        const bool second = kt >= nkx;
        const float* P = second ? a.h0 : a.x;
        const float* W = second ? a.w_hh : a.w_ih;
        const int64_t ldp = second ? a.ldh0 : a.ldx;
        const int K = second ? a.Hd : a.E;
        const int k = (second ? kt - nkx : kt) * BK + lk;
#pragma unroll
        for (int p = 0; p < TB / 32; ++p) pr[p] = ld4<VEC>(P, ldp, i0 + lrow + 32 * p, a.B, k, K);
#pragma unroll
        for (int p = 0; p < 4 * TU / 32; ++p) {
            const int qrow = lrow + 32 * p;            // gate*TU + local unit
            const int g = qrow / TU, ul = qrow - g * TU;
            const int u = u0 + ul;
            qr[p] = (u < a.Hd) ? ld4<VEC>(W, (int64_t)K, g * a.Hd + u, 4 * a.Hd, k, K) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < TB / 32; ++p)
            *reinterpret_cast<f32x4*>(&Ps[(buf * TB + lrow + 32 * p) * LDK + lk]) = pr[p];
#pragma unroll
        for (int p = 0; p < 4 * TU / 32; ++p)
            *reinterpret_cast<f32x4*>(&Qs[(buf * 4 * TU + lrow + 32 * p) * LDK + lk]) = qr[p];
    };

    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1);
        const float* pa = &Ps[(cur * TB + r) * LDK + 4 * h];
        const float* qa = &Qs[(cur * 4 * TU + 2 * wg * TU + wj * 32 + r) * LDK + 4 * h];
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(pa + kc * 8);
            f32x4 bv[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) bv[g] = *reinterpret_cast<const f32x4*>(qa + g * TU * LDK + kc * 8);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[g][m], acc[g], 0, 0, 0);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // the loop ended on a barrier: the tile buffers are free.  g/o wavefronts hand their accumulators to the i/f
    // wavefront of the same unit half: [wj][gate][register][lane]
    float* xch = lds;
    if (wg == 1) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e) xch[((wj * 2 + g) * 16 + e) * 64 + lane] = acc[g][e];
    }
    __syncthreads();
    if (wg == 1) return;
    const int u = u0 + wj * 32 + r;
    if (u >= a.Hd) return;
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = a.b_ih[g * a.Hd + u] + a.b_hh[g * a.Hd + u];
    // the previous cell state of all 16 rows first: inside the loop each load would wait for the stores before it
    float cprev[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        cprev[e] = (a.c0 && i < a.B) ? a.c0[(int64_t)i * a.Hd + u] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i >= a.B) continue;
        const float gi = acc[0][e] + bias[0], gf = acc[1][e] + bias[1];
        const float gg = xch[((wj * 2 + 0) * 16 + e) * 64 + lane] + bias[2];
        const float go = xch[((wj * 2 + 1) * 16 + e) * 64 + lane] + bias[3];
        const float cp = cprev[e];
        const float cn = sigmoidf_(gf) * cp + sigmoidf_(gi) * tanhf(gg);
        const float hn = sigmoidf_(go) * tanhf(cn);
        a.c1[(int64_t)i * a.Hd + u] = cn;
        a.h1[(int64_t)i * a.ldh1 + u] = hn;
        if (a.gates) {
            float* gp = a.gates + (int64_t)i * 4 * a.Hd + u;
            gp[0] = gi; gp[a.Hd] = gf; gp[2 * a.Hd] = gg; gp[3 * a.Hd] = go;
        }
    }
}

// The same step with the operand tiles written to LDS by the loads themselves (global_load_lds_dwordx4: one 1 KiB
// piece = 8 rows of 32 floats per wavefront instruction, lane l at byte 16*l).  For E and Hd multiples of 32 and
// 16-byte aligned operands.  The tile image is linear, [288 rows][32 floats] (32 sample rows, then gate*64 + unit),
// with the eight 16-byte slots of a row XOR-swizzled by (row/2)%8 -- applied to the source address of a piece and to
// the LDS address of a fragment read -- so the fragment reads are bank-conflict free without padding.  Three
// buffers: a workgroup is alone on its CU here, so two tiles' loads stay in flight across each barrier.  The sums
// run in the same order as in the register-staged kernel.
constexpr int DROWS = TB + 4 * TU;                 // rows of one tile image
constexpr int DBUF = DROWS * BK;                   // floats per buffer
constexpr int DNPW = DROWS / 8 / 4;                // pieces per wavefront and tile
constexpr int DNBUF = 3;
constexpr int LSTM_DMA_LDS_FLOATS = DNBUF * DBUF;
static_assert(DROWS % 32 == 0 && 2 * DNPW <= 63, "tile does not cut into pieces");

__device__ __forceinline__ int lstm_swz(int row) { return (row >> 1) & 7; }

__global__ __launch_bounds__(256) void lstm_step_dma_kernel(LstmArgs a) {
    if (a.live && a.live[0] <= 0) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = wave >> 1, wj = wave & 1, r = lane & 31, h = lane >> 5;   // wg: gate pair (i,f) or (g,o)
    const int i0 = blockIdx.y * TB, u0 = blockIdx.x * TU;
    const int nkx = a.E / BK;
    const int nkh = a.h0 ? a.Hd / BK : 0;
    const int nk = nkx + nkh;

    f32x16 acc[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][e] = 0.f;

    // this lane's source inside each of its wavefront's pieces, for the (x, W_ih) and the (h, W_hh) part of the k loop
    const float* sx[DNPW];
    const float* sh[DNPW];
#pragma unroll
    for (int i = 0; i < DNPW; ++i) {
        const int piece = i * 4 + wave;                 // wave-uniform
        const int row = piece * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ lstm_swz(row);
        if (piece * 8 < TB) {
            const int smp = min(i0 + row, a.B - 1);
            sx[i] = a.x + (int64_t)smp * a.ldx + chunk * 4;
            sh[i] = a.h0 ? a.h0 + (int64_t)smp * a.ldh0 + chunk * 4 : a.x;
        } else {
            const int q = row - TB, g = q / TU, u = min(u0 + q % TU, a.Hd - 1);
            sx[i] = a.w_ih + ((int64_t)g * a.Hd + u) * a.E + chunk * 4;
            sh[i] = a.w_hh + ((int64_t)g * a.Hd + u) * a.Hd + chunk * 4;
        }
    }
    auto issue = [&](int kt, int buf) {
        const bool second = kt >= nkx;                  // uniform
        const int off = (second ? kt - nkx : kt) * BK;
#pragma unroll
        for (int i = 0; i < DNPW; ++i) {
            const float* src = (second ? sh[i] : sx[i]) + off;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + buf * DBUF + (i * 4 + wave) * 256),
                                             16, 0, 0);
        }
    };
    const int rq = TB + 2 * wg * TU + wj * 32 + r;
    const int sp = lstm_swz(r), sq = lstm_swz(rq);
    auto compute = [&](int buf) {
        const float* b0 = lds + buf * DBUF;
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(b0 + r * BK + (((kc * 2 + h) ^ sp) * 4));
            f32x4 bv[2];
#pragma unroll
            for (int g = 0; g < 2; ++g)
                bv[g] = *reinterpret_cast<const f32x4*>(b0 + (rq + g * TU) * BK + (((kc * 2 + h) ^ sq) * 4));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[g][m], acc[g], 0, 0, 0);
        }
    };
#pragma unroll
    for (int j = 0; j < DNBUF - 1; ++j)
        if (j < nk) issue(j, j);
    for (int kt0 = 0; kt0 < nk; kt0 += DNBUF) {
#pragma unroll
        for (int j = 0; j < DNBUF; ++j) {
            const int kt = kt0 + j;
            if (kt >= nk) break;
            // tile kt has landed (the loads of tile kt+1 may stay in flight: they retire in order) and every
            // wavefront is done reading the buffer that the next loads overwrite
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DNPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + DNBUF - 1 < nk) issue(kt + DNBUF - 1, (j + DNBUF - 1) % DNBUF);
            __builtin_amdgcn_sched_barrier(0);
            compute(j);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();   // the tile buffers are free

    // g/o wavefronts hand their accumulators to the i/f wavefront of the same unit half: [wj][gate][register][lane]
    float* xch = lds;
    if (wg == 1) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 16; ++e) xch[((wj * 2 + g) * 16 + e) * 64 + lane] = acc[g][e];
    }
    __syncthreads();
    if (wg == 1) return;
    const int u = u0 + wj * 32 + r;
    if (u >= a.Hd) return;
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = a.b_ih[g * a.Hd + u] + a.b_hh[g * a.Hd + u];
    // the previous cell state of all 16 rows first: inside the loop each load would wait for the stores before it
    float cprev[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        cprev[e] = (a.c0 && i < a.B) ? a.c0[(int64_t)i * a.Hd + u] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i >= a.B) continue;
        const float gi = acc[0][e] + bias[0], gf = acc[1][e] + bias[1];
        const float gg = xch[((wj * 2 + 0) * 16 + e) * 64 + lane] + bias[2];
        const float go = xch[((wj * 2 + 1) * 16 + e) * 64 + lane] + bias[3];
        const float cp = cprev[e];
        const float cn = sigmoidf_(gf) * cp + sigmoidf_(gi) * tanhf(gg);
        const float hn = sigmoidf_(go) * tanhf(cn);
        a.c1[(int64_t)i * a.Hd + u] = cn;
        a.h1[(int64_t)i * a.ldh1 + u] = hn;
        if (a.gates) {
            float* gp = a.gates + (int64_t)i * 4 * a.Hd + u;
            gp[0] = gi; gp[a.Hd] = gf; gp[2 * a.Hd] = gg; gp[3 * a.Hd] = go;
        }
    }
}

// ---- small batches (B <= 32: the reference's own CPU-sized decode, cnnrnn_model.py:57-110 at batch 32) ---------------
// At 32 samples the tiling above has Hd/64 workgroups (8 at Hd = 512) whose wavefronts each walk the whole 768-deep
// contraction: 768 dependent MFMAs, 20 us, on 32 of the chip's 1024 SIMDs.  Here a workgroup owns 8 hidden units: the 32
// columns of one accumulator are (gate, unit) = (j / 8, j % 8), so a single 32x32 tile holds all four gates of its
// units for all samples, and the workgroup's four wavefronts split the contraction four ways (each reads its own k
// range of the operands straight from memory into MFMA fragments: nothing is shared between them, so no LDS
// staging); the partial tiles meet in LDS, where one thread per (sample, unit) adds them and applies the cell update.
// Hd/8 workgroups (64 at Hd = 512), 192 MFMAs per wavefront.
__global__ __launch_bounds__(256) void lstm_step_small_kernel(LstmArgs a) {
    if (a.live && a.live[0] <= 0) return;
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int u0 = blockIdx.x * 8;
    const int nkx = a.E / 8, nkh = a.h0 ? a.Hd / 8 : 0, nk = nkx + nkh;    // k-chunks of 8
    const int per = (nk + 3) / 4;
    const int kc0 = wave * per, kc1 = min(nk, kc0 + per);
    const int arow = min(r, a.B - 1);                       // rows past the batch repeat the last one: never stored
    const int wrow = (r >> 3) * a.Hd + min(u0 + (r & 7), a.Hd - 1);
    const float* xa = a.x + (int64_t)arow * a.ldx + 4 * h;
    const float* ha = a.h0 ? a.h0 + (int64_t)arow * a.ldh0 + 4 * h : nullptr;
    const float* wx = a.w_ih + (int64_t)wrow * a.E + 4 * h;
    const float* wh = a.w_hh + (int64_t)wrow * a.Hd + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    constexpr int U = 4;                                     // k-chunks per batch of loads
    f32x4 av[2][U], bv[2][U];
    auto fetch = [&](int set, int kc) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int k = kc + i;
            av[set][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            bv[set][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < kc1) {
                if (k < nkx) {
                    av[set][i] = *reinterpret_cast<const f32x4*>(xa + 8 * k);
                    bv[set][i] = *reinterpret_cast<const f32x4*>(wx + 8 * k);
                } else {
                    av[set][i] = *reinterpret_cast<const f32x4*>(ha + 8 * (k - nkx));
                    bv[set][i] = *reinterpret_cast<const f32x4*>(wh + 8 * (k - nkx));
                }
            }
        }
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int i = 0; i < U; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][i][m], bv[set][i][m], acc, 0, 0, 0);
    };
    // the loads of the next batch are in flight while the current one is multiplied
    fetch(0, kc0);
    for (int kc = kc0; kc < kc1; kc += 2 * U) {
        fetch(1, kc + U);
        __builtin_amdgcn_sched_barrier(0);
        multiply(0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(0, kc + 2 * U);
        __builtin_amdgcn_sched_barrier(0);
        multiply(1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][e][lane] = acc[e];
    __syncthreads();
    // thread -> (sample i, unit ul); gate g of that pair sits in register e(i), lane 32 * h(i) + 8 * g + ul of every partial
    const int i = tid >> 3, ul = tid & 7, u = u0 + ul;
    if (i >= a.B || u >= a.Hd) return;
    const int e = (i & 3) + 4 * (i >> 3), hl = ((i >> 2) & 1) * 32 + ul;
    float gate[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int l = hl + 8 * g;
        gate[g] = ((red[0][e][l] + red[1][e][l]) + (red[2][e][l] + red[3][e][l])) + (a.b_ih[g * a.Hd + u] + a.b_hh[g * a.Hd + u]);
    }
    const float cp = a.c0 ? a.c0[(int64_t)i * a.Hd + u] : 0.f;
    const float cn = sigmoidf_(gate[1]) * cp + sigmoidf_(gate[0]) * tanhf(gate[2]);
    const float hn = sigmoidf_(gate[3]) * tanhf(cn);
    a.c1[(int64_t)i * a.Hd + u] = cn;
    a.h1[(int64_t)i * a.ldh1 + u] = hn;
    if (a.gates) {
        float* gp = a.gates + (int64_t)i * 4 * a.Hd + u;
        gp[0] = gate[0]; gp[a.Hd] = gate[1]; gp[2 * a.Hd] = gate[2]; gp[3 * a.Hd] = gate[3];
    }
}

// The same tiling with whole-line loads (E and Hd multiples of 32): a load instruction of the kernel above touches 32
// rows, 32 bytes of each, and the lines the wavefronts hold open compete for the 32 KiB vector cache.  Here a wavefront
// fetches its 32 x 32 blocks of the samples and of the weights eight lanes per 128-byte row segment and turns them into
// MFMA fragments through a private LDS region (rows padded to 36 floats: the 16-byte fragment reads of 16 lanes fall on
// distinct banks); still no workgroup barrier in the loop.
constexpr int LS_LD = 36;

__global__ __launch_bounds__(256) void lstm_step_small_lds_kernel(LstmArgs a) {
    if (a.live && a.live[0] <= 0) return;
    __shared__ __attribute__((aligned(16))) float stage[4][2 * 32 * LS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int u0 = blockIdx.x * 8;
    const int s0 = 32 * blockIdx.y;                           // first sample of this workgroup's block of 32
    const int nbx = a.E / 32, nbh = a.h0 ? a.Hd / 32 : 0, nb = nbx + nbh;  // 32-float blocks of the contraction
    const int per = (nb + 3) / 4;
    const int b0 = wave * per, b1 = min(nb, b0 + per);
    float* sa = stage[wave];                                 // [32][LS_LD] samples
    float* sw = sa + 32 * LS_LD;                             // [32][LS_LD] weight rows (gate, unit)
    const int lrow = lane >> 3, lslot = lane & 7;
    const float* xg[4];
    const float* hg[4];
    const float* wxg[4];
    const float* whg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + lrow;
        const int arow = min(s0 + row, a.B - 1);             // rows past the batch repeat the last one: never stored
        const int wrow = (row >> 3) * a.Hd + min(u0 + (row & 7), a.Hd - 1);
        xg[i] = a.x + (int64_t)arow * a.ldx + 4 * lslot;
        hg[i] = a.h0 ? a.h0 + (int64_t)arow * a.ldh0 + 4 * lslot : nullptr;
        wxg[i] = a.w_ih + (int64_t)wrow * a.E + 4 * lslot;
        whg[i] = a.w_hh + (int64_t)wrow * a.Hd + 4 * lslot;
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 ar[2][4], wr[2][4];
    auto fetch = [&](int set, int b) {
        if (b >= b1) return;
        if (b < nbx) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ar[set][i] = *reinterpret_cast<const f32x4*>(xg[i] + 32 * b);
                wr[set][i] = *reinterpret_cast<const f32x4*>(wxg[i] + 32 * b);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ar[set][i] = *reinterpret_cast<const f32x4*>(hg[i] + 32 * (b - nbx));
                wr[set][i] = *reinterpret_cast<const f32x4*>(whg[i] + 32 * (b - nbx));
            }
        }
    };
    auto consume = [&](int set, int b) {
        if (b >= b1) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(sa + (8 * i + lrow) * LS_LD + 4 * lslot) = ar[set][i];
            *reinterpret_cast<f32x4*>(sw + (8 * i + lrow) * LS_LD + 4 * lslot) = wr[set][i];
        }
        // LDS serves one wavefront's accesses in order: the block written above is what the reads below see
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(sa + r * LS_LD + 8 * c + 4 * h);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(sw + r * LS_LD + 8 * c + 4 * h);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[m], acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };
    fetch(0, b0);
    for (int b = b0; b < b1; b += 2) {
        fetch(1, b + 1);
        consume(0, b);
        fetch(0, b + 2);
        consume(1, b + 1);
    }
    // partial tiles: each wavefront's own staging region holds its 16 x 64 values
#pragma unroll
    for (int e = 0; e < 16; ++e) sa[e * 64 + lane] = acc[e];
    __syncthreads();
    // thread -> (sample i, unit ul); gate g of that pair sits in register e(i), lane 32 * h(i) + 8 * g + ul of every partial
    const int il = tid >> 3, ul = tid & 7, u = u0 + ul;      // il: sample within the block
    const int i = s0 + il;
    if (i >= a.B || u >= a.Hd) return;
    const int e = (il & 3) + 4 * (il >> 3), hl = ((il >> 2) & 1) * 32 + ul;
    float gate[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int o = e * 64 + hl + 8 * g;
        gate[g] = ((stage[0][o] + stage[1][o]) + (stage[2][o] + stage[3][o])) + (a.b_ih[g * a.Hd + u] + a.b_hh[g * a.Hd + u]);
    }
    const float cp = a.c0 ? a.c0[(int64_t)i * a.Hd + u] : 0.f;
    const float cn = sigmoidf_(gate[1]) * cp + sigmoidf_(gate[0]) * tanhf(gate[2]);
    const float hn = sigmoidf_(gate[3]) * tanhf(cn);
    a.c1[(int64_t)i * a.Hd + u] = cn;
    a.h1[(int64_t)i * a.ldh1 + u] = hn;
    if (a.gates) {
        float* gp = a.gates + (int64_t)i * 4 * a.Hd + u;
        gp[0] = gate[0]; gp[a.Hd] = gate[1]; gp[2 * a.Hd] = gate[2]; gp[3 * a.Hd] = gate[3];
    }
}

struct LstmBwdArgs {
    const float* gates; const float* c0; const float* c1;
    const float* dh_out; int64_t lddh; const float* dh_rec; const float* dc_in;
    int B, Hd;
    float* dgates; float* dc0;
};

// Pointwise backward of one step: one thread per (sample, unit); recomputes the activations from the
// saved pre-activation gates.
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(LstmBwdArgs a) {
    const int64_t n = (int64_t)a.B * a.Hd;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int i = (int)(idx / a.Hd), u = (int)(idx - (int64_t)i * a.Hd);
    const float* gp = a.gates + (int64_t)i * 4 * a.Hd + u;
    const float si = sigmoidf_(gp[0]), sf = sigmoidf_(gp[a.Hd]);
    const float tg = tanhf(gp[2 * a.Hd]), so = sigmoidf_(gp[3 * a.Hd]);
    const float cp = a.c0 ? a.c0[idx] : 0.f;
    const float tc = tanhf(a.c1[idx]);
    float dh = 0.f;
    if (a.dh_out) dh += a.dh_out[(int64_t)i * a.lddh + u];
    if (a.dh_rec) dh += a.dh_rec[idx];
    float dc = dh * so * (1.f - tc * tc);
    if (a.dc_in) dc += a.dc_in[idx];
    float* dg = a.dgates + (int64_t)i * 4 * a.Hd + u;
    dg[0] = dc * tg * si * (1.f - si);
    dg[a.Hd] = dc * cp * sf * (1.f - sf);
    dg[2 * a.Hd] = dc * si * (1.f - tg * tg);
    dg[3 * a.Hd] = dh * tc * so * (1.f - so);
    a.dc0[idx] = dc * sf;
}

// 0: automatic; 1: always the register-staged kernel; 2: small batches keep the fragments-from-memory kernel (tests
// compare the forms)
CAPDEC_HOOK_VAR(g_lstm_staging);

}  // namespace

#ifdef CAPDEC_TUNING_HOOKS
extern "C" int capdec_lstm_force_staging(int mode) {
    CAPDEC_REQUIRE(mode == 0 || mode == 1 || mode == 2);
    g_lstm_staging = mode;
    return 0;
}
#endif

extern "C" int capdec_lstm_step(const float* x, int64_t ldx, const float* h0, int64_t ldh0, const float* c0, const float* w_ih,
                                const float* w_hh, const float* b_ih, const float* b_hh, int B, int E, int Hd,
                                float* h1, int64_t ldh1, float* c1, float* gates_out, void* stream) {
    return capdec_launch_lstm_step(x, ldx, h0, ldh0, c0, w_ih, w_hh, b_ih, b_hh, B, E, Hd, h1, ldh1, c1, gates_out, nullptr,
                                   (hipStream_t)stream);
}

int capdec_launch_lstm_step(const float* x, int64_t ldx, const float* h0, int64_t ldh0, const float* c0, const float* w_ih,
                            const float* w_hh, const float* b_ih, const float* b_hh, int B, int E, int Hd, float* h1,
                            int64_t ldh1, float* c1, float* gates_out, const int32_t* live, hipStream_t stream) {
    CAPDEC_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && h1 && c1 && B >= 0 && E >= 1 && Hd >= 1);
    CAPDEC_REQUIRE(ldx >= E && ldh1 >= Hd && (!h0 || ldh0 >= Hd));
    CAPDEC_REQUIRE(h1 != h0 && c1 != c0);   // other workgroups still read the old state
    if (B == 0) return 0;
    LstmArgs a{x, ldx, h0, ldh0, c0, w_ih, w_hh, b_ih, b_hh, B, E, Hd, h1, ldh1, c1, gates_out, live};
    const bool vec = E % 4 == 0 && Hd % 4 == 0 && ldx % 4 == 0 && capdec_aligned16(x) && capdec_aligned16(w_ih) &&
                     capdec_aligned16(w_hh) && (!h0 || (capdec_aligned16(h0) && ldh0 % 4 == 0));
    constexpr size_t smem = LSTM_LDS_FLOATS * sizeof(float);
    static const bool ready =
        hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess;
    if (!ready) return CAPDEC_EINVAL;
    if (B <= 256 && vec && E % 32 == 0 && Hd % 32 == 0 && g_lstm_staging == 0) {
        // small batches (the reference's loaders default to 32 and 64 samples): blocks of 32
        hipLaunchKernelGGL(lstm_step_small_lds_kernel, dim3(Hd / 8, (B + 31) / 32), dim3(256), 0, (hipStream_t)stream, a);
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    if (B <= 32 && vec && E % 8 == 0 && Hd % 8 == 0 && g_lstm_staging != 1) {
        // widths that are multiples of 8 but not of 32 (or override 2): fragments read straight from memory
        hipLaunchKernelGGL(lstm_step_small_kernel, dim3(Hd / 8), dim3(256), 0, (hipStream_t)stream, a);
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((Hd + TU - 1) / TU, (B + TB - 1) / TB);
    if (vec && E % BK == 0 && Hd % BK == 0 && g_lstm_staging != 1) {
        constexpr size_t smem_dma = LSTM_DMA_LDS_FLOATS * sizeof(float);
        static const bool ready_dma =
            hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_dma_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dma) == hipSuccess;
        if (!ready_dma) return CAPDEC_EINVAL;
        hipLaunchKernelGGL(lstm_step_dma_kernel, grid, dim3(256), smem_dma, (hipStream_t)stream, a);
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    if (vec) hipLaunchKernelGGL(lstm_step_kernel<true>, grid, dim3(256), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(lstm_step_kernel<false>, grid, dim3(256), smem, (hipStream_t)stream, a);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

extern "C" int capdec_lstm_step_bwd(const float* gates, const float* c0, const float* c1, const float* dh_out,
                                    int64_t lddh, const float* dh_rec, const float* dc_in, int B, int Hd,
                                    float* dgates, float* dc0, void* stream) {
    CAPDEC_REQUIRE(gates && c1 && dgates && dc0 && B >= 0 && Hd >= 1);
    CAPDEC_REQUIRE(!dh_out || lddh >= Hd);
    if (B == 0) return 0;
    LstmBwdArgs a{gates, c0, c1, dh_out, lddh, dh_rec, dc_in, B, Hd, dgates, dc0};
    const int64_t n = (int64_t)B * Hd;
    hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}
