// Decode-step attention over the raw memory WITHOUT LDS staging: a wavefront per decode row, all four heads; memory rows
// go straight into registers (a lane owns 8 of the 512 columns), the 8 keys x 4 heads partial scores of a group are
// reduced over the lanes with a 32-shuffle reduce-scatter, exp-scores are broadcast back with constant-lane reads.
// Standalone timing + check against a plain double-precision evaluation on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef G
#define G 8
#endif

template <int E>
__global__ __launch_bounds__(256) void attention_memory_rows_kernel(const float* __restrict__ qp, int64_t ldq,
                                                                    const float* __restrict__ mem, int S, int H, float scale,
                                                                    float* __restrict__ ctx, int64_t ldc,
                                                                    const int32_t* __restrict__ row_map, int map_div,
                                                                    const int32_t* __restrict__ rows_dev, int R) {
    static_assert(E == 512 && G == 8, "built for E = 512, groups of 8 keys");
    constexpr int HB = 4, NS = E / 256;                 // heads per pass, 16-byte slices per lane
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R || (rows_dev && row >= rows_dev[0])) return;
    const int b = (row_map ? row_map[row] : row) / map_div;
    const float* mb = mem + (int64_t)b * S * E;
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    for (int h0 = 0; h0 < H; h0 += HB) {
        f32x4 q[HB][NS];
#pragma unroll
        for (int hh = 0; hh < HB; ++hh)
#pragma unroll
            for (int u = 0; u < NS; ++u)
                q[hh][u] = h0 + hh < H ? *reinterpret_cast<const f32x4*>(qp + (int64_t)row * ldq + (int64_t)(h0 + hh) * E + 4 * (lane + 64 * u))
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 acc[HB][NS];
#pragma unroll
        for (int hh = 0; hh < HB; ++hh)
#pragma unroll
            for (int u = 0; u < NS; ++u) acc[hh][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        // this lane tracks the running maximum and sum of head (lane >> 1) & 3 (all lanes of that class agree)
        float m = -INFINITY, l = 0.f;
        f32x4 kv[2][G][NS];
        auto fetch = [&](int set, int k0) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int k = min(k0 + g, S - 1);
#pragma unroll
                for (int u = 0; u < NS; ++u)
                    kv[set][g][u] = *reinterpret_cast<const f32x4*>(mb + (int64_t)k * E + 4 * (lane + 64 * u));
            }
        };
        auto work = [&](int set, int k0) {
            float t[G * HB];                            // index 4 * g + h
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int hh = 0; hh < HB; ++hh) {
                    float s = 0.f;
#pragma unroll
                    for (int u = 0; u < NS; ++u)
                        s += q[hh][u][0] * kv[set][g][u][0] + q[hh][u][1] * kv[set][g][u][1] + q[hh][u][2] * kv[set][g][u][2] +
                             q[hh][u][3] * kv[set][g][u][3];
                    t[4 * g + hh] = s;
                }
            // reduce-scatter over the lanes: after the stage with mask M a lane keeps the half of its values selected by
            // its bit M; lane L ends with the total of value index (L >> 1) & 31 (both lanes of a pair hold it)
            float a16[16], a8[8], a4[4], a2[2], a1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float send = b5 ? t[i] : t[i + 16];
                a16[i] = (b5 ? t[i + 16] : t[i]) + __shfl_xor(send, 32, 64);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float send = b4 ? a16[i] : a16[i + 8];
                a8[i] = (b4 ? a16[i + 8] : a16[i]) + __shfl_xor(send, 16, 64);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float send = b3 ? a8[i] : a8[i + 4];
                a4[i] = (b3 ? a8[i + 4] : a8[i]) + __shfl_xor(send, 8, 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float send = b2 ? a4[i] : a4[i + 2];
                a2[i] = (b2 ? a4[i + 2] : a4[i]) + __shfl_xor(send, 4, 64);
            }
            {
                const float send = b1 ? a2[0] : a2[1];
                a1 = (b1 ? a2[1] : a2[0]) + __shfl_xor(send, 2, 64);
            }
            a1 += __shfl_xor(a1, 1, 64);
            // this lane: value index idx = 16 b5 + 8 b4 + 4 b3 + 2 b2 + b1 = 4 g + h  ->  key g = idx >> 2, head idx & 3
            const int idx = (b5 ? 16 : 0) + (b4 ? 8 : 0) + (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
            const int g_mine = idx >> 2;
            const float sc = k0 + g_mine < S ? a1 * scale : -INFINITY;
            // maximum over the 8 keys of this lane's head: the lanes that differ in the key bits (idx bits 2..4 = b3, b4, b5)
            float gm = sc;
            gm = fmaxf(gm, __shfl_xor(gm, 8, 64));
            gm = fmaxf(gm, __shfl_xor(gm, 16, 64));
            gm = fmaxf(gm, __shfl_xor(gm, 32, 64));
            const float mnew = fmaxf(m, gm);
            const float e = expf(sc - mnew);
            const float alpha = expf(m - mnew);
            float es = e;
            es += __shfl_xor(es, 8, 64);
            es += __shfl_xor(es, 16, 64);
            es += __shfl_xor(es, 32, 64);
            l = l * alpha + es;
            m = mnew;
            // rescale factors of the four heads (head h is tracked by the lanes with (lane >> 1) & 3 == h: lane 2 h)
#pragma unroll
            for (int hh = 0; hh < HB; ++hh) {
                const float ah = __shfl(alpha, 2 * hh, 64);
#pragma unroll
                for (int u = 0; u < NS; ++u) acc[hh][u] *= ah;
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int hh = 0; hh < HB; ++hh) {
                    // idx = 4 g + hh lives in the lanes with (b1, b2) = bits 0, 1 of idx and (b3, b4, b5) = bits 2..4
                    const int src = 2 * ((hh & 1) | ((hh >> 1) << 1) | (g << 2));
                    const float p = __shfl(e, src, 64);
#pragma unroll
                    for (int u = 0; u < NS; ++u) acc[hh][u] += kv[set][g][u] * p;
                }
        };
        fetch(0, 0);
        for (int k0 = 0; k0 < S; k0 += 2 * G) {
            if (k0 + G < S) fetch(1, k0 + G);
            work(0, k0);
            if (k0 + G >= S) break;
            if (k0 + 2 * G < S) fetch(0, k0 + 2 * G);
            work(1, k0 + G);
        }
#pragma unroll
        for (int hh = 0; hh < HB; ++hh) {
            if (h0 + hh >= H) break;
            const float inv = 1.0f / __shfl(l, 2 * hh, 64);
#pragma unroll
            for (int u = 0; u < NS; ++u)
                *reinterpret_cast<f32x4*>(ctx + (int64_t)row * ldc + (int64_t)(h0 + hh) * E + 4 * (lane + 64 * u)) = acc[hh][u] * inv;
        }
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int B = 1024, E = 512, H = 4;
    for (int S : {261, 100, 7}) {
        std::vector<float> hq((size_t)B * H * E), hm((size_t)B * S * E), hc((size_t)B * H * E);
        uint32_t seed = 12345u + S;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto& v : hq) v = rnd();
        for (auto& v : hm) v = rnd();
        float *qp, *mem, *ctx;
        CHECK(hipMalloc(&qp, hq.size() * 4)); CHECK(hipMalloc(&ctx, hc.size() * 4)); CHECK(hipMalloc(&mem, hm.size() * 4));
        CHECK(hipMemcpy(qp, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(mem, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
        hipEvent_t a, e; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&e));
        const float scale = 0.0884f;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(a, 0));
            for (int i = 0; i < 10; ++i)
                hipLaunchKernelGGL(attention_memory_rows_kernel<512>, dim3((B + 3) / 4), dim3(256), 0, 0, qp, (int64_t)H * E, mem, S, H, scale,
                                   ctx, (int64_t)H * E, nullptr, 1, nullptr, B);
            CHECK(hipEventRecord(e, 0));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, a, e));
            if (rep) printf("S=%d: %.1f us per launch (%.0f GB/s of memory rows)\n", S, ms * 100, (double)B * S * E * 4 / (ms * 1e-4) / 1e9);
        }
        CHECK(hipMemcpy(hc.data(), ctx, hc.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int r : {0, 1, 513, 1023})
            for (int h = 0; h < H; ++h) {
                std::vector<double> sc(S);
                double mx = -1e300;
                for (int j = 0; j < S; ++j) {
                    double d = 0;
                    for (int c = 0; c < E; ++c) d += (double)hq[((size_t)r * H + h) * E + c] * hm[((size_t)r * S + j) * E + c];
                    sc[j] = d * scale; mx = std::max(mx, sc[j]);
                }
                double sum = 0;
                for (int j = 0; j < S; ++j) { sc[j] = std::exp(sc[j] - mx); sum += sc[j]; }
                for (int c = 0; c < E; ++c) {
                    double o = 0;
                    for (int j = 0; j < S; ++j) o += sc[j] * hm[((size_t)r * S + j) * E + c];
                    worst = std::max(worst, std::fabs(o / sum - hc[((size_t)r * H + h) * E + c]));
                }
            }
        printf("S=%d: largest deviation from the fp64 evaluation %.3e\n", S, worst);
        CHECK(hipFree(qp)); CHECK(hipFree(ctx)); CHECK(hipFree(mem));
    }
    return 0;
}
