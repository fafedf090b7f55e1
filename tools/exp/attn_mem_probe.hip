// Standalone timing of the LDS-staged form of the decode attention over the raw memory (a column quarter per wavefront,
// all four heads; superseded in csrc/ops.hip by the register form of attn_mem_probe3.hip) with parts knocked out:
// KNOCK bit 0: one slice of the score sweep only; bit 1: one key of the context sweep only; bit 2: no global fetches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef KNOCK
#define KNOCK 0
#endif
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
#ifndef KC
#define KC 32
#endif
#ifndef MINB
#define MINB 1
#endif
#ifndef PF
#define PF 1
#endif
constexpr int AM_KC = KC, AM_HB = 4;

template <int E>
__global__ __launch_bounds__(256, MINB) void attention_memory_kernel(const float* __restrict__ qp, int64_t ldq,
                                                               const float* __restrict__ mem, int S, int H, float scale,
                                                               float* __restrict__ ctx, int64_t ldc,
                                                               const int32_t* __restrict__ row_map, int map_div,
                                                               const int32_t* __restrict__ rows_dev) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int row = blockIdx.x;
    if (rows_dev && row >= rows_dev[0]) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int LD = E + 4, NV = E / 4;
    constexpr int SQ = E / 16;                         // 16-byte slices of a row in one wavefront's column quarter
    constexpr int PARTS = SQ < 8 ? SQ : 8;             // lanes per key in the score sweep
    constexpr int NS = SQ / PARTS;                     // slices per lane
    constexpr int KPS = 64 / PARTS, NSW = AM_KC / KPS; // keys per sweep, sweeps per chunk
    constexpr int CW = E >= 256 ? E / 256 : 1;         // context columns per lane (of the quarter's E/4)
    constexpr int CL = E / 4 / CW;                     // lanes that own context columns
    constexpr int NPF = AM_KC * NV / 256;              // 16-byte slices of a chunk per thread
    float* Ms = lds;                                   // [AM_KC][LD]
    float* sp = Ms + AM_KC * LD;                       // [4 waves][AM_HB][AM_KC] partial scores
    float* pp = sp + 4 * AM_HB * AM_KC;                // [AM_KC][AM_HB] exp-scores
    float* al = pp + AM_KC * AM_HB;                    // [AM_HB] rescale factors, then 1 / sum
    const int b = (row_map ? row_map[row] : row) / map_div;
    const float* mb = mem + (int64_t)b * S * E;
    const int part = lane % PARTS, kl = lane / PARTS;
    for (int h0 = 0; h0 < H; h0 += AM_HB) {
        // this lane's slices of the four queries: columns wave * E/4 + 4 * (part + PARTS * i)
        f32x4 q[AM_HB][NS];
#pragma unroll
        for (int hh = 0; hh < AM_HB; ++hh)
#pragma unroll
            for (int i = 0; i < NS; ++i)
                q[hh][i] = h0 + hh < H ? *reinterpret_cast<const f32x4*>(qp + (int64_t)row * ldq + (int64_t)(h0 + hh) * E +
                                                                         wave * (E / 4) + 4 * (part + PARTS * i))
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
        float m = -INFINITY, l = 0.f;                  // running softmax of head h0 + wave (held by wavefront `wave`)
        float acc[AM_HB][CW];
#pragma unroll
        for (int hh = 0; hh < AM_HB; ++hh)
#pragma unroll
            for (int u = 0; u < CW; ++u) acc[hh][u] = 0.f;
        // a chunk is fetched into registers one chunk ahead (in flight while the current chunk is worked on) and
        // written to LDS once every wavefront is done with the previous one
        f32x4 pf[NPF];
        auto fetch = [&](int c0) {
            const int nk = min(AM_KC, S - c0);
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int idx = threadIdx.x + 256 * i;
                const int r = idx / NV, c = idx - r * NV;
                const int rr = min(r, nk - 1);         // rows past the end repeat the last one: masked below
                pf[i] = *reinterpret_cast<const f32x4*>(mb + (int64_t)(c0 + rr) * E + 4 * c);
            }
        };
        if (PF) fetch(0);
        for (int c0 = 0; c0 < S; c0 += AM_KC) {
            const int nk = min(AM_KC, S - c0);
            if (!PF) fetch(c0);
            __syncthreads();                           // the previous chunk (and its scores) are consumed
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int idx = threadIdx.x + 256 * i;
                const int r = idx / NV, c = idx - r * NV;
                *reinterpret_cast<f32x4*>(&Ms[r * LD + 4 * c]) = pf[i];
            }
            __syncthreads();
            if (PF && !(KNOCK & 4) && c0 + AM_KC < S) fetch(c0 + AM_KC);
            // partial scores over this wavefront's column quarter
#pragma unroll
            for (int sw = 0; sw < NSW; ++sw) {
                const int j = KPS * sw + kl;
                const float* kr = Ms + j * LD + wave * (E / 4) + 4 * part;
                float t[AM_HB];
#pragma unroll
                for (int hh = 0; hh < AM_HB; ++hh) t[hh] = 0.f;
#pragma unroll
                for (int i = 0; i < ((KNOCK & 1) ? 1 : NS); ++i) {
                    const f32x4 kk = *reinterpret_cast<const f32x4*>(kr + 4 * PARTS * i);
#pragma unroll
                    for (int hh = 0; hh < AM_HB; ++hh)
                        t[hh] += q[hh][i][0] * kk[0] + q[hh][i][1] * kk[1] + q[hh][i][2] * kk[2] + q[hh][i][3] * kk[3];
                }
#pragma unroll
                for (int hh = 0; hh < AM_HB; ++hh) {
#pragma unroll
                    for (int o = 1; o < PARTS; o <<= 1) t[hh] += __shfl_xor(t[hh], o, 64);
                }
                if (part < AM_HB) {
                    // lane `part` of the key's group stores head `part`
                    float v = t[0];
                    if (part == 1) v = t[1];
                    if (part == 2) v = t[2];
                    if (part == 3) v = t[3];
                    sp[(wave * AM_HB + part) * AM_KC + j] = v;
                }
            }
            __syncthreads();
            // wavefront `wave` finishes head h0 + wave: one key per lane
            {
                float sc = -INFINITY;
                if (lane < AM_KC && lane < nk)
                    sc = ((sp[(0 * AM_HB + wave) * AM_KC + lane] + sp[(1 * AM_HB + wave) * AM_KC + lane]) +
                          (sp[(2 * AM_HB + wave) * AM_KC + lane] + sp[(3 * AM_HB + wave) * AM_KC + lane])) * scale;
                const float mnew = fmaxf(m, wave_max(sc));
                const float e = expf(sc - mnew);           // 0 for the lanes without a key
                const float alpha = expf(m - mnew);        // 0 on the first chunk
                l = l * alpha + wave_sum(e);
                m = mnew;
                if (lane < AM_KC) pp[lane * AM_HB + wave] = e;
                if (lane == 0) al[wave] = alpha;
            }
            __syncthreads();
            // contexts: this lane's columns of the quarter, four heads
            if (lane < CL) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(al);
#pragma unroll
                for (int hh = 0; hh < AM_HB; ++hh)
#pragma unroll
                    for (int u = 0; u < CW; ++u) acc[hh][u] *= a4[hh];
                const float* col = Ms + wave * (E / 4) + CW * lane;
#pragma unroll 8
                for (int j = 0; j < ((KNOCK & 2) ? 1 : AM_KC); ++j) {          // all rows: the exp-scores past the end are zero
                    const f32x4 p4 = *reinterpret_cast<const f32x4*>(pp + j * AM_HB);
                    float v[CW];
                    if (CW == 2) {
                        const float2 t2 = *reinterpret_cast<const float2*>(col + j * LD);
                        v[0] = t2.x;
                        v[CW - 1] = t2.y;
                    } else {
                        v[0] = col[j * LD];
                    }
#pragma unroll
                    for (int hh = 0; hh < AM_HB; ++hh)
#pragma unroll
                        for (int u = 0; u < CW; ++u) acc[hh][u] += v[u] * p4[hh];
                }
            }
        }
        __syncthreads();
        if (lane == 0) al[wave] = 1.0f / l;
        __syncthreads();
        if (lane < CL) {
            const f32x4 inv = *reinterpret_cast<const f32x4*>(al);
#pragma unroll
            for (int hh = 0; hh < AM_HB; ++hh) {
                if (h0 + hh >= H) break;
                float* dst = ctx + (int64_t)row * ldc + (int64_t)(h0 + hh) * E + wave * (E / 4) + CW * lane;
#pragma unroll
                for (int u = 0; u < CW; ++u) dst[u] = acc[hh][u] * inv[hh];
            }
        }
        __syncthreads();                               // al is reused by the next pass
    }
}


#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int B = 1024, S = 261, E = 512, H = 4;
    float *qp, *mem, *ctx;
    CHECK(hipMalloc(&qp, (size_t)B * H * E * 4)); CHECK(hipMalloc(&ctx, (size_t)B * H * E * 4));
    CHECK(hipMalloc(&mem, (size_t)B * S * E * 4));
    CHECK(hipMemset(qp, 0, (size_t)B * H * E * 4)); CHECK(hipMemset(mem, 0, (size_t)B * S * E * 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_memory_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    const size_t smem = ((size_t)AM_KC * (E + 4) + 4 * AM_HB * AM_KC + AM_KC * AM_HB + 4) * sizeof(float);
    hipEvent_t a, e; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&e));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(a, 0));
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL(attention_memory_kernel<512>, dim3(B), dim3(256), smem, 0, qp, (int64_t)H * E, mem, S, H, 0.088f, ctx, (int64_t)H * E, nullptr, 1, nullptr);
        CHECK(hipEventRecord(e, 0));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, a, e));
        if (rep) printf("KNOCK=%d: %.1f us per launch (%.0f GB/s of memory rows)\n", KNOCK, ms * 100, (double)B * S * E * 4 / (ms * 1e-4) / 1e9);
    }
    return 0;
}
