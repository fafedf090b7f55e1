// All decoder layers of one KV-cached greedy decode step in ONE launch (transformer.py:83-90 run once per step of
// the loop at transformer.py:151-171), for the single-memory-row form of the cross-attention block (S == 1: the block
// is one vector per sample and layer, see capdec_tf_greedy_decode), for two memory rows (S == 2: the logistic closed form,
// pair_block below) and at most two layers; also the layers of a KV-cached beam-search step over its B*W hypothesis rows.
//
// A decode step is a chain of small dependent operations per row: q/k/v projections -> self-attention over the row's
// own cache -> output projection + residual -> LayerNorm 1, + cross-attention constant, LayerNorm 2 -> fc1 + ReLU ->
// fc2 + residual -> LayerNorm 3, and again for the next layer.  No operation mixes rows of different samples.  As
// separate launches every link of that chain pays a kernel boundary and a cold start (7 launches per layer, each about
// 7 us of fixed cost for 5 us of matrix work at 1024 rows).  Here the rows are cut into blocks of 16 and each block is
// given to a TEAM of 8 workgroups that walks the whole chain for it (78 KB of LDS and at most 128 registers per lane:
// two workgroups share a CU from 513 rows on, one fills the wait of the other):
//   - a workgroup owns E/8 output columns of every projection (64 at E = 512) and the rows' full contraction depth:
//     its 8 wavefronts split the contraction (k-slabs of E/8), hold the activations as MFMA fragments in REGISTERS
//     (16 x 16 x 4 fp32 tiles), stream their weight slab through a two-slot LDS ring with global_load_lds in entries of
//     32 columns x 32 k per wavefront (the next operation's slab is already
//     on its way while the team waits for its members), sum the 8 partial tiles through LDS and apply bias / ReLU /
//     residual;
//   - between operations the 8 members exchange their 16 x E/8 slices through global memory in a fragment-major
//     layout ([k/4][row][4 floats]: the 16-byte fragment loads of a consumer are contiguous 1 KiB runs), written with
//     write-through (sc1) 16-byte stores, published by one write-through flag store per workgroup behind a drain of
//     every storing wavefront and a workgroup barrier, and read with sc1 loads behind a relaxed poll of the team's eight
//     flags by eight lanes and a workgroup barrier; teams never wait for one another;
//   - LayerNorm runs on the fragments in registers (row sums over the 8 k-slabs meet in LDS), redundantly in the 8
//     members, so it costs no hand-off of its own;
//   - attention: a member takes one head for 8 of the 16 rows (two (row, head) pairs per wavefront, four wavefronts).
//
// Generated <PAD> tokens.  The reference masks a <PAD> QUERY row, which then attends uniformly over the CURRENT length
// (transformer.py:117-123), so the layer-0 output of such a position changes at every later step.  Layer-0 keys and
// values depend on the embeddings only and the causal rows around it never see the difference in layer 0; what
// changes is the <PAD> position's INPUT TO LAYER 1, i.e. its layer-1 key and value, which the sample's newest row
// attends to.  (Later rows' layer-1 outputs change too, but with two layers nothing reads them.)  So a team carries
// the <PAD> positions of its own samples along as up to 16 SIDE rows (a second 16-row tile that follows the main one
// through each operation of layer 0, on the same weight slices streamed once more): mean of the layer-0 values -> output projection -> LayerNorms -> FFN -> LayerNorm 3 ->
// layer-1 key and value, written to the cache before the team's layer-1 attention reads them.  With one layer nothing
// has to be re-run at all.  A team whose samples hold more than 16 <PAD> positions flags the surplus samples for the
// exact path (out_dirty).
// The launch of step t >= 1 also finishes step t-1: every member merges the per-tile arg-max winners of the previous
// vocabulary projection for its team's 16 rows (the token ids it needs for the embedding rows anyway), member 0 does the
// reference's bookkeeping for them (transformer.py:155-171), and the input rows are built in registers: no separate
// token-pick launch, no round trip of the input rows through memory.
// Progress: a team only ever waits for its own members; workgroups are dispatched in index order, so the members of the
// oldest unfinished team are resident or next in line.  Every wait is a bounded spin: a workgroup that gives up raises
// `err` and returns, and the caller re-decodes the batch on the exact path.
#include "common.h"
#include <limits.h>
#include <math.h>
#include <mutex>
#ifdef CAPDEC_FUSED_TRACE
#include <stdlib.h>
#endif

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int FT_TEAM = CAPDEC_FUSED_TEAM;
constexpr int FT_ROWS = CAPDEC_FUSED_ROWS;
constexpr int FT_SIDE = CAPDEC_FUSED_SIDE_ROWS;
constexpr int FT_WAVES = 8;
constexpr int FT_THREADS = 512;
constexpr unsigned FT_SPIN_LIMIT = 1u << 20;
constexpr int FT_SC1 = 16;      // cache-policy bit of the buffer intrinsics: write-through store / L1-bypassing load

template <int E>
struct FT {
    static constexpr int NC = E / FT_TEAM;            // output columns of a workgroup
    static constexpr int KW = E / FT_WAVES;           // contraction range of a wavefront
    static constexpr int NF = KW / 16;                // 16-byte activation fragments per lane (16-row tile)
    static constexpr int NCT = NC / 16;               // 16x16 accumulators per wavefront and projection
    static constexpr int NE = 4;                      // ring entries per projection: (column half, k half)
    static constexpr int PIECES = 4;                  // global_load_lds instructions per entry (8 rows x 128 B each)
    static constexpr int WSLAB = 32 * 32;             // floats of one wavefront's entry image [32 columns][32 k]
    static constexpr int WSLOT = FT_WAVES * WSLAB;
    static constexpr int DK = E / 4;                  // head width (H == 4)
    static constexpr int OFF_W0 = 0;
    static constexpr int OFF_W1 = WSLOT;              // second ring slot; the k-split partial tiles between operations
    static constexpr int OFF_R = 2 * WSLOT;           // [NC/4][16][4] own columns of the residual operand (main rows)
    static constexpr int OFF_RS = OFF_R + 16 * NC;    // [NC/4][16][4] the same of the side rows (fc2 only) ...
    static constexpr int OFF_CST = OFF_RS;            // ... and, in the attention stage, [8][DK] outputs before they are stored
    static constexpr int OFF_PART = OFF_RS + 16 * NC; // [2][8][16] LayerNorm partial (mean, squared deviations) pairs
    static constexpr int OFF_SC = OFF_PART + 4 * FT_WAVES * 16;   // [8][64] probability rows (one (row, head) pair per wavefront)
    static constexpr int OFF_META = OFF_SC + FT_WAVES * 64;   // ints: row descriptors, wait flag
    static constexpr int LDS_FLOATS = OFF_META + 160;
    static_assert(NF == 4 && NCT == 4 && KW == NC && NC == 64 && 8 * DK == 16 * NC, "team kernel built for E = 512");
    static_assert(LDS_FLOATS * 4 <= 80 * 1024, "two workgroups per CU");
};

// The lane id again, opaque to the optimiser: index arithmetic that starts from it is redone where it is used instead of
// being kept in a register from the top of the kernel (at 128 registers per lane such values get spilled, and a reload
// from scratch memory waits for every store and weight load the wavefront has in flight).
__device__ __forceinline__ int fresh(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ f32x4 load_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, FT_SC1));
}
__device__ __forceinline__ void store_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs, (int)byte_off, 0, FT_SC1);
}

// One ring entry of a weight slice, straight into this wavefront's part of a ring slot: 32 output columns x 32 k, LDS
// image [32][32] floats with the 16-byte slots of a row XOR-swizzled by (row >> 1) & 7 (conflict-free 16-byte fragment
// reads of 16 rows).  Piece p (8 rows of 128 bytes) of the image comes from rows p*8 + lane/8; the swizzle of a lane's
// slot depends on p through p & 1 only, so two per-lane offsets serve all pieces and everything else is uniform.
struct LaneOff { int even, odd; };
template <int E>
__device__ __forceinline__ LaneOff lane_offsets(int lane) {
    const int r8 = lane >> 3, c = lane & 7;
    LaneOff o;
    o.even = (r8 * E + ((c ^ ((r8 >> 1) & 7)) << 2)) * 4;
    o.odd = (r8 * E + ((c ^ ((4 + (r8 >> 1)) & 7)) << 2)) * 4;
    return o;
}
// entry e of a projection: column half e >> 1, k half e & 1
template <int E>
__device__ __forceinline__ void issue_entry(const float* __restrict__ W, int member, int wave, LaneOff lo, int e, float* slot) {
    using C = FT<E>;
    float* dst = slot + wave * C::WSLAB;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, E * E * 4, 0x00020000);
    const int base = ((member * C::NC + (e >> 1) * 32) * E + wave * C::KW + (e & 1) * 32) * 4;      // uniform
#ifndef FT_EXPERIMENT_NO_DMA      // timing experiment only (tools/exp): the contractions then read whatever LDS holds
#pragma unroll
    for (int p = 0; p < C::PIECES; ++p)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + p * 256), 16,
                                                 (p & 1) ? lo.odd : lo.even, base + p * 8 * E * 4, 0, 0);
#endif
}

// 16 rows: lane (c, g) holds a[kq] = X[c][k0 + kq*16 + 4g + {0..3}]; MFMA m of chunk kq sums k in {kq*16 + 4g' + m};
// accumulator register e of lane (c, g) is row 4g + e, column ct*16 + c
template <int E>
__device__ __forceinline__ void mma_entry(const float* slot, int wave, int lane, const f32x4* a2, f32x4* acc2) {
    using C = FT<E>;
    const int c = lane & 15, g = lane >> 4;
    const float* img = slot + wave * C::WSLAB;
    // (the two column tiles in turn inside the m loop: consecutive MFMAs never depend on one another)
    f32x4 b[2][2];
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int row = ct * 16 + c;
            b[kq][ct] = *reinterpret_cast<const f32x4*>(img + row * 32 + ((((kq * 4 + g) ^ ((row >> 1) & 7))) << 2));
        }
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                acc2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[kq][m], b[kq][ct][m], acc2[ct], 0, 0, 0);
}

// NPROJ projections of the 16-row tile streamed through the ring.  The stream of entries W[0] e0..e3, W[1] e0.., ...
// alternates between the two ring slots; its first two entries were issued before the call, entry n + 2 is issued as
// soon as this wavefront has read entry n, and once slot 0 is free for good the first entry of `next` (the next
// operation's weights; nullable) follows.  Nothing else of this wavefront is in flight inside the loop, so the counted
// waits see exactly the ring's loads (they retire in order).
template <int E, int NPROJ>
__device__ __forceinline__ void gemm_stream(float* lds, const float* const (&W)[NPROJ], const float* next, int member,
                                            int wave, int lane, LaneOff lo, const f32x4 (&a)[FT<E>::NF],
                                            f32x4 (&acc)[NPROJ][FT<E>::NCT]) {
    using C = FT<E>;
    constexpr int N = NPROJ * C::NE;
    float* slot0 = lds + C::OFF_W0;
    float* slot1 = lds + C::OFF_W1;
#pragma unroll
    for (int p = 0; p < NPROJ; ++p)
#pragma unroll
        for (int t = 0; t < C::NCT; ++t) acc[p][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < N; ++n) {
        if (n == 0 || (n == N - 1 && !next)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PIECES) : "memory");
        const int p = n / C::NE, e = n % C::NE;
        float* slot = (n & 1) ? slot1 : slot0;
        mma_entry<E>(slot, wave, lane, &a[(e & 1) * 2], &acc[p][(e >> 1) * 2]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wavefront's fragment reads of the slot are done
        if (n + 2 < N) issue_entry<E>(W[(n + 2) / C::NE], member, wave, lo, (n + 2) % C::NE, slot);
        else if (n + 2 == N && next) issue_entry<E>(next, member, wave, lo, 0, slot0);
    }
}

// this wavefront's 16 x NC partial tile into its own part of slot 1 (exactly the bytes of its entry image: no other
// wavefront's weights are overwritten); 16-byte slots of a row XOR-swizzled by the row
template <int E>
__device__ __forceinline__ void write_partial(float* lds, int wave, int lane, const f32x4 (&acc)[FT<E>::NCT],
                                              int area = FT<E>::OFF_W1) {
    using C = FT<E>;
    lane = fresh(lane);
    const int c = lane & 15, g = lane >> 4;
    float* d = lds + area + wave * C::WSLAB;
#pragma unroll
    for (int ct = 0; ct < C::NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e, j = ct * 16 + c;
            d[i * C::NC + ((((j >> 2) ^ i) << 2) | (j & 3))] = acc[ct][e];
        }
}
template <int E>
__device__ __forceinline__ f32x4 sum_partials(const float* lds, int i, int sl, int area = FT<E>::OFF_W1) {
    using C = FT<E>;
    const float* p = lds + area + i * C::NC + ((sl ^ i) << 2);
    f32x4 t[FT_WAVES];
#pragma unroll
    for (int w = 0; w < FT_WAVES; ++w) t[w] = *reinterpret_cast<const f32x4*>(p + w * C::WSLAB);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (all eight reads in flight together, one wait)
    return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
}

// Workgroup barrier that orders LDS traffic only.  (__syncthreads() also waits for every global access of the
// wavefront - weight loads in flight, the acknowledgement of a flag store - which nothing here needs.)
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Hand-off flags of a team: one word per member in one 128-byte line, holding the number of the member's last
// finished operation.  Publishing: every storing wavefront drains its stores, the workgroup meets, one lane stores the
// flag write-through.  (An atomic add on a shared counter costs its issuer about 3.5 us until it is acknowledged.)
__device__ __forceinline__ void team_signal(unsigned* flags, int member, unsigned epoch) {
#ifndef FT_EXPERIMENT_NO_WAIT     // timing experiment only (tools/exp): results are garbage
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    wg_barrier();
    if (threadIdx.x == 0) __hip_atomic_store(flags + member, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Lanes 0..7 of the first wavefront poll the eight flags (relaxed, L1-bypassing), the workgroup meets behind them;
// false (for every thread) when the spin ran out.
__device__ __forceinline__ bool team_wait(unsigned* flags, unsigned epoch, unsigned* err, int* flag) {
#ifdef FT_EXPERIMENT_NO_WAIT
    wg_barrier();
    return true;
#endif
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned spins = 0;
        int ok = 1;
        for (;;) {
            const unsigned v = lane < FT_TEAM ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
            if (__all(v >= epoch)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > FT_SPIN_LIMIT) {
                if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        if (lane == 0) flag[epoch & 1] = ok;
    }
    wg_barrier();
    return flag[epoch & 1] != 0;     // (two words in turn: the next wait writes the other one)
}

// Fragments of a [k/4][16][4] exchange block: lane (row, g) of wavefront w loads the 16-byte slots w*KW/4 + kf*4 + g
template <int E>
__device__ __forceinline__ void load_frags(__amdgpu_buffer_rsrc_t rs, int wave, int lane, f32x4 (&a)[FT<E>::NF]) {
    using C = FT<E>;
    lane = fresh(lane);
    const int row = lane & 15, g = lane >> 4;
#pragma unroll
    for (int kf = 0; kf < C::NF; ++kf) a[kf] = load_sc1(rs, (unsigned)(((wave * (C::KW / 4) + kf * 4 + g) * 16 + row) * 16));
}

// LayerNorm (eps 1e-5, affine) of 16 rows held as fragments: lane (row, g) of wavefront w holds the columns
// w*KW + kf*16 + 4g + {0..3} of its row.  Every wavefront takes the mean and the sum of squared deviations of its own
// KW columns (two passes over registers), the 8 pairs of a row meet in LDS once and are combined exactly
// (sum of squares about the common mean = sum of the parts' + KW * (part mean - mean)^2): one barrier per LayerNorm.
template <int E>
__device__ __forceinline__ void ln_frags(f32x4 (&a)[FT<E>::NF], const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* part, int& pb, int wave, int lane) {
    using C = FT<E>;
    lane = fresh(lane);
    const int row = lane & 15, g = lane >> 4;
    float s = 0.f;
#pragma unroll
    for (int kf = 0; kf < C::NF; ++kf) s += (a[kf][0] + a[kf][1]) + (a[kf][2] + a[kf][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mw = s * (1.0f / (float)C::KW);
    float q = 0.f;
#pragma unroll
    for (int kf = 0; kf < C::NF; ++kf) {
        const f32x4 d = a[kf] - mw;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    float2* p = reinterpret_cast<float2*>(part) + pb * (FT_WAVES * 16);
    pb ^= 1;
    if (g == 0) p[wave * 16 + row] = make_float2(mw, q);
    wg_barrier();
    float2 t[FT_WAVES];
#pragma unroll
    for (int w = 0; w < FT_WAVES; ++w) t[w] = p[w * 16 + row];
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < FT_WAVES; ++w) mean += t[w].x;
    mean *= 1.0f / (float)FT_WAVES;
    float m2 = 0.f;
#pragma unroll
    for (int w = 0; w < FT_WAVES; ++w) m2 += t[w].y + (float)C::KW * (t[w].x - mean) * (t[w].x - mean);
    const float rstd = 1.0f / sqrtf(m2 / (float)E + 1e-5f);
    const int c0 = wave * C::KW + 4 * g;
#pragma unroll
    for (int kf = 0; kf < C::NF; ++kf) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(gamma + c0 + kf * 16);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + c0 + kf * 16);
        a[kf] = (a[kf] - mean) * rstd * gg + bb;
    }
}

// The cross-attention block over TWO memory rows in closed form, on the LayerNorm-1 output held as fragments (see
// ln_pair_kernel in ops.hip, same arithmetic): a2 = c1 + sum_h sigmoid((x1 . G_h + sb_h) / sqrt(d_k)) D_h, added to x1.
// G, D, c1, sb: this lane's row's sample.  The H partial dot products of a wavefront's KW columns meet in LDS once.
template <int E>
__device__ __forceinline__ void pair_block(f32x4 (&a)[FT<E>::NF], const float* __restrict__ G, const float* __restrict__ D,
                                           const float* __restrict__ c1, const float* __restrict__ sb, f32x4* gp, int wave,
                                           int lane) {
    using C = FT<E>;
    lane = fresh(lane);
    const int row = lane & 15, g = lane >> 4;
    const int c0 = wave * C::KW + 4 * g;
    f32x4 d;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        float part = 0.f;
#pragma unroll
        for (int kf = 0; kf < C::NF; ++kf) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(G + h * E + c0 + kf * 16);
            part += (a[kf][0] * gg[0] + a[kf][1] * gg[1]) + (a[kf][2] * gg[2] + a[kf][3] * gg[3]);
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        d[h] = part;
    }
    if (g == 0) gp[wave * 16 + row] = d;
    wg_barrier();
    f32x4 t[FT_WAVES];
#pragma unroll
    for (int w = 0; w < FT_WAVES; ++w) t[w] = gp[w * 16 + row];
    const f32x4 tot = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    const float scale = 1.0f / sqrtf((float)C::DK);
    float pr[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) pr[h] = 1.0f / (1.0f + expf(-(tot[h] + sb[h]) * scale));
#pragma unroll
    for (int kf = 0; kf < C::NF; ++kf) {
        f32x4 a2 = *reinterpret_cast<const f32x4*>(c1 + c0 + kf * 16);
#pragma unroll
        for (int h = 0; h < 4; ++h) a2 += *reinterpret_cast<const f32x4*>(D + h * E + c0 + kf * 16) * pr[h];
        a[kf] += a2;
    }
}

// buffer descriptor of `bytes` bytes at p, made where it is used (four scalar registers; none stay live across stages)
#define FT_RS(p, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(p), 0, (bytes), 0x00020000)

// Stage timestamps of one team (100 MHz wall clock), compiled in only for the probe under tools/exp.
#ifdef CAPDEC_FUSED_TRACE
#define FT_MARK()                                                                                   \
    do {                                                                                            \
        if (A.trace && tid == 0 && team == A.trace_team && nmark < 64) A.trace[member * 64 + nmark] = wall_clock64(); \
        ++nmark;                                                                                    \
    } while (0)
#else
#define FT_MARK() do { } while (0)
#endif

template <int E>
__global__ __launch_bounds__(FT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void decode_layers_team_kernel(capdec_fused_args A) {
    using C = FT<E>;
    int nmark = 0;
    (void)nmark;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c16 = lane & 15, g4 = lane >> 4;              // fragment maps: lane = (row, k group)
    // consecutive workgroup ids go round-robin to the XCDs: the members of a team share id & 7 (speed only)
    const int xg = blockIdx.x & 7, gslot = blockIdx.x >> 3;
    const int team = (gslot >> 3) * 8 + xg, member = gslot & 7;
    if (team >= A.teams) return;
#ifdef CAPDEC_FUSED_TRACE
    if (A.hwid && tid == 0)      // HW_ID (register 4) and XCC_ID (register 20), 32 bits each
        A.hwid[blockIdx.x] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffff);
#endif
    const int pos_new = A.Tk - 1;
    const bool pick = A.pval != nullptr && pos_new >= 1;     // this launch also picks the previous step's tokens
    if (pick) {
        // The reference leaves its loop once every caption holds an <EOS> (transformer.py:166-168): fin[s] counts the
        // captions that finished at step s; the counts of the steps before the previous one are final (no workgroup of
        // this launch changes them), so every workgroup decides alike.  Nothing is recorded then, and the vocabulary
        // projection of this step finds a row count of zero.
        int fin_before = 0;
        for (int q = lane; q < pos_new - 1; q += 64) fin_before += A.fin[q];
        fin_before = (int)wave_sum((float)fin_before);
        if (fin_before >= A.B) {
            if (blockIdx.x == 0 && tid == 0) A.nrows_w[0] = 0;
            return;
        }
    } else if (A.nrows[0] <= 0) {
        return;                                             // every caption has finished
    }

    float* slot0 = lds + C::OFF_W0;
    float* slot1 = lds + C::OFF_W1;
    const LaneOff lo = lane_offsets<E>(lane);
    issue_entry<E>(A.ly[0].wq, member, wave, lo, 0, slot0);
    issue_entry<E>(A.ly[0].wq, member, wave, lo, 1, slot1);

    int* meta = reinterpret_cast<int*>(lds + C::OFF_META);
    int *s_b = meta, *s_flags = meta + 16;                  // main rows: sample (clamped), bit 0 live, bit 1 <PAD> query
    int *ss_map = meta + 32, *ss_b = meta + 48, *ss_pos = meta + 64, *ss_act = meta + 80;   // side rows
    int* s_ok = meta + 96;
    int *s_tokn = meta + 112, *s_done = meta + 128;         // newest token and finished flag of the main rows
    if (pick) {
        // the arg-max of the previous step's vocabulary projection (per-tile winners merged, lowest index on ties), as
        // cached_advance_kernel does: two rows per wavefront, every member for all 16 rows of its team
        // (the loads of both rows are issued before the first merge: one round trip, not two)
        float pv[2][3];
        int pi[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int b = min(team * FT_ROWS + wave * 2 + q, A.B - 1);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int n = lane + 64 * u;
                const int64_t off = capdec_part_off(b, n < A.ntiles ? n : 0, A.ntiles);
                pv[q][u] = A.pval[off];
                pi[q][u] = A.pidx[off];
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = wave * 2 + q;
            const int b = min(team * FT_ROWS + row, A.B - 1);
            float v = -INFINITY;
            int i = INT_MAX;
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (lane + 64 * u < A.ntiles) argmax_merge(v, i, pv[q][u], pi[q][u]);
            for (int n = lane + 192; n < A.ntiles; n += 64)          // vocabularies beyond 192 tiles of 64 ids
                argmax_merge(v, i, A.pval[capdec_part_off(b, n, A.ntiles)], A.pidx[capdec_part_off(b, n, A.ntiles)]);
            wave_argmax(v, i);
            if (i < 0 || i >= A.V) i = 0;                   // all-NaN row: no finite winner
            if (lane == 0) {
                s_tokn[row] = i;
                s_done[row] = (A.done_w[b] != 0 || i == A.eos) ? 1 : 0;   // (member 0 may be setting the flag: either way 1)
            }
        }
        wg_barrier();
        if (member == 0 && tid < FT_ROWS && team * FT_ROWS + tid < A.B) {
            // the reference's bookkeeping of the token (transformer.py:155-171): a finished row records nothing
            const int b = team * FT_ROWS + tid, i = s_tokn[tid];
            A.tok_w[(int64_t)b * A.ldtok + pos_new] = i;
            if (!A.done_w[b]) {
                if (i == A.eos) {
                    A.done_w[b] = 1;
                    atomicAdd(&A.fin[pos_new - 1], 1);
                } else {
                    const int n = A.out_len[b];
                    A.out_ids[(int64_t)b * A.max_len + n] = i;
                    A.out_len[b] = n + 1;
                    if (i == 0 && A.L >= 2) {
                        // a <PAD> position: a side row of this team from the next step on
                        const int slot = atomicAdd(&A.pcount_w[team], 1);
                        if (slot < FT_SIDE) A.plist_w[team * FT_SIDE + slot] = b * A.max_len + pos_new;
                        else if (A.dirty[b] == 0) A.dirty[b] = 1;
                    }
                }
            }
        }
    }
    if (tid < FT_ROWS) {
        const int b0 = team * FT_ROWS + tid;
        const bool act = b0 < A.B;
        const int b = act ? b0 : team * FT_ROWS;
        const int pad = (pick ? s_tokn[tid] : A.tok[(int64_t)b * A.ldtok + pos_new]) == 0;
        s_b[tid] = b;
        s_flags[tid] = (act ? 1 : 0) | (pad << 1);
        if (!pick) s_done[tid] = 0;
    } else if (tid < FT_ROWS + FT_SIDE) {
        // <PAD> positions of the team's samples before the newest one, still decoding.  (List slots start at -1 and are
        // written once, by one 4-byte store: a reader sees -1 or the entry, whatever the writer of this launch does.)
        const int sr = tid - FT_ROWS;
        const int m0 = A.L >= 2 ? A.plist[team * FT_SIDE + sr] : -1;
        const int m = m0 >= 0 ? m0 : team * FT_ROWS * A.max_len;
        const int b = m / A.max_len, pos = m - b * A.max_len;
        ss_map[sr] = m;
        ss_b[sr] = b;
        ss_pos[sr] = pos;
        const int fin_row = pick ? s_done[b - team * FT_ROWS] : A.done[b];
        ss_act[sr] = (m0 >= 0 && pos < pos_new && !fin_row) ? 1 : 0;
    }
    wg_barrier();
    bool side = false;
#pragma unroll
    for (int i = 0; i < FT_SIDE; ++i) side |= ss_act[i] != 0;

    unsigned* tctr = A.ctr + (size_t)team * 32;
    unsigned nsig = 0;
    const size_t tb = (size_t)team * FT_ROWS * E;           // this team's 16 x E block in the exchange buffers
    const size_t tsb = (size_t)team * FT_SIDE * E;          // and its 16 x E block in the side rows' buffers
    constexpr int blk = 16 * E * 4;
    float* part = lds + C::OFF_PART;
    int pb = 0;
    f32x4* R = reinterpret_cast<f32x4*>(lds + C::OFF_R);
    f32x4* RS = reinterpret_cast<f32x4*>(lds + C::OFF_RS);
    // epilogue maps of the first 256 threads: (ja, ta) for fragment-major stores (rows fastest), (jb, tb_) for row-major
    // stores (columns fastest)
    // (recomputed from the thread id where they are used - FT_EPI_IDS - instead of living in registers across the
    //  whole kernel: at 128 registers per lane they would be spilled, and a reload from scratch memory waits for every
    //  store and weight load in flight)
#define FT_EPI_IDS()                                                                                          \
    int t_ = threadIdx.x;                                                                                     \
    asm volatile("" : "+v"(t_));                                                                              \
    const int ja = t_ & 15, ta = (t_ >> 4) & 15, jb = ta, tb_ = ja;                                           \
    const unsigned frag_off = (unsigned)(((member * (C::NC / 4) + ta) * 16 + ja) * 16);                       \
    const bool ethr = t_ < 256;                                                                               \
    (void)jb; (void)tb_; (void)frag_off; (void)ja; (void)ta
    const int cache_bytes = A.cache_bytes;

    FT_MARK();                                   // 0: set-up done
    for (int l = 0; l < A.L; ++l) {
        const capdec_fused_layer& ly = A.ly[l];
        const bool side_kv = side && l == 1;     // the side rows' layer-1 keys and values
        const bool side_l0 = side && l == 0;     // the side rows walk layer 0 behind the attention

        // ---- q, k, v projections of the layer input ---------------------------------------------------------
        {
            f32x4 a[C::NF];
            if (l == 0 && pick) {
                // the input row of the newest position: embedding of the token just picked + its position (transformer.py:126)
                const float* er = A.emb + (int64_t)s_tokn[c16] * E + wave * C::KW + 4 * g4;
                const float* pr = A.pe + (int64_t)pos_new * E + wave * C::KW + 4 * g4;
#pragma unroll
                for (int kf = 0; kf < C::NF; ++kf)
                    a[kf] = *reinterpret_cast<const f32x4*>(er + kf * 16) + *reinterpret_cast<const f32x4*>(pr + kf * 16);
            } else if (l == 0) {
                const float* xr = A.x + (int64_t)s_b[c16] * E + wave * C::KW + 4 * g4;
#pragma unroll
                for (int kf = 0; kf < C::NF; ++kf) a[kf] = *reinterpret_cast<const f32x4*>(xr + kf * 16);
            } else {
                if (!team_wait(tctr, nsig, A.err, s_ok)) return;
                if (wave == 0) issue_entry<E>(ly.wq, member, wave, lo, 1, slot1);
                load_frags<E>(FT_RS(A.f_z + tb, blk), wave, lane, a);
                ln_frags<E>(a, A.ly[l - 1].n3g, A.ly[l - 1].n3b, part, pb, wave, lane);
            }
            if (wave == member) {
#pragma unroll
                for (int kf = 0; kf < C::NF; ++kf) R[(kf * 4 + g4) * 16 + c16] = a[kf];
            }
            FT_MARK();                               // layer input in registers
            f32x4 acc[3][C::NCT];
            const float* const W3[3] = {ly.wq, ly.wk, ly.wv};
            // (no prefetch of the next operation's weights behind this stream: both ring slots take partial tiles)
            gemm_stream<E, 3>(lds, W3, nullptr, member, wave, lane, lo, a, acc);
            FT_MARK();                           // q, k, v contractions done
            // q and k partial tiles side by side (slot 1, slot 0), then v: four barriers instead of six
            write_partial<E>(lds, wave, lane, acc[0]);
            write_partial<E>(lds, wave, lane, acc[1], C::OFF_W0);
            wg_barrier();
            {
                FT_EPI_IDS();
                if (ethr) {
                    const int col = member * C::NC + tb_ * 4;
                    const f32x4 vq = sum_partials<E>(lds, jb, tb_) + *reinterpret_cast<const f32x4*>(ly.bq + col);
                    store_sc1(FT_RS(A.q + tb, blk), (unsigned)((jb * E + col) * 4), vq);
                    const f32x4 vk = sum_partials<E>(lds, jb, tb_, C::OFF_W0) + *reinterpret_cast<const f32x4*>(ly.bk + col);
                    if (s_flags[jb] & 1)
                        store_sc1(FT_RS(ly.kc, cache_bytes), ((unsigned)(s_b[jb] * A.max_len + pos_new) * (unsigned)E + col) * 4u, vk);
                }
            }
            wg_barrier();
            issue_entry<E>(side_kv ? ly.wk : ly.wo, member, wave, lo, 0, slot0);
            write_partial<E>(lds, wave, lane, acc[2]);
            wg_barrier();
            {
                FT_EPI_IDS();
                if (ethr && (s_flags[jb] & 1)) {
                    const int col = member * C::NC + tb_ * 4;
                    const f32x4 vv = sum_partials<E>(lds, jb, tb_) + *reinterpret_cast<const f32x4*>(ly.bv + col);
                    store_sc1(FT_RS(ly.vc, cache_bytes), ((unsigned)(s_b[jb] * A.max_len + pos_new) * (unsigned)E + col) * 4u, vv);
                }
            }
            wg_barrier();
        }
        if (side_kv) {
            // the side rows' layer-1 keys and values: the two weight slices once more, for their tile
            issue_entry<E>(ly.wk, member, wave, lo, 1, slot1);
            f32x4 as[C::NF];
            load_frags<E>(FT_RS(A.s_z + tsb, blk), wave, lane, as);
            ln_frags<E>(as, A.ly[l - 1].n3g, A.ly[l - 1].n3b, part, pb, wave, lane);
            f32x4 accs[2][C::NCT];
            const float* const W2[2] = {ly.wk, ly.wv};
            gemm_stream<E, 2>(lds, W2, ly.wo, member, wave, lane, lo, as, accs);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                write_partial<E>(lds, wave, lane, accs[p]);
                wg_barrier();
                FT_EPI_IDS();
                if (ethr && ss_act[jb]) {
                    const int col = member * C::NC + tb_ * 4;
                    const f32x4 v = sum_partials<E>(lds, jb, tb_) + *reinterpret_cast<const f32x4*>((p == 0 ? ly.bk : ly.bv) + col);
                    store_sc1(FT_RS(p == 0 ? ly.kc : ly.vc, cache_bytes), ((unsigned)ss_map[jb] * (unsigned)E + col) * 4u, v);
                }
                wg_barrier();
            }
        }
        FT_MARK();                               // q, k, v stored
        team_signal(tctr, member, nsig + 1);
        ++nsig;
        // (the wavefront that polls issues its part after the poll: its loads return in order, so a load in flight
        //  would hold the poll's answer back)
        if (wave != 0) issue_entry<E>(ly.wo, member, wave, lo, 1, slot1);
        FT_MARK();                               // signalled

        // ---- self-attention: member = (head, half of the rows), one (row, head) pair per wavefront ------------
        if (!team_wait(tctr, nsig, A.err, s_ok)) return;
        if (wave == 0) issue_entry<E>(ly.wo, member, wave, lo, 1, slot1);
        FT_MARK();                               // team has q, k, v
        {
            // One (row, head) pair per wavefront: row w of the member's 8 rows, head hd.  Scores: 4 lanes per key, 16 keys
            // per pass, kept in registers; softmax over the key lanes; context: the two half-wavefronts take the even and
            // the odd keys, one lane per 16 bytes of the head, and add up at the end.  Queries, the keys of the first pass
            // and the values of the first 16 keys are in flight together.
            constexpr int NM = C::DK / 16, LPV = C::DK / 4;
            static_assert(LPV == 32, "one half-wavefront per head row");
            const int hd = member >> 1, rbase = (member & 1) * 8;
            float* cst = lds + C::OFF_CST;
            const __amdgpu_buffer_rsrc_t rs_v = FT_RS(ly.vc, cache_bytes);
            {
                const int lf = fresh(lane);
                const int hf = lf >> 5, l32 = lf & 31;
                const int li = wave, i = rbase + li;
                const float scale = sqrtf((float)C::DK);
                const __amdgpu_buffer_rsrc_t rs_q = FT_RS(A.q + tb, blk);
                const __amdgpu_buffer_rsrc_t rs_k = FT_RS(ly.kc, cache_bytes);
                const int b = s_b[i];
                const bool pad_row = (s_flags[i] & 2) != 0;
                const int nkeys = A.Tk;              // the newest position sees every key, <PAD> query or not
                const unsigned kb = ((unsigned)(b * A.max_len) * (unsigned)E + (unsigned)(hd * C::DK)) * 4u;
                const int prt = lf & 3, kl = lf >> 2;
                f32x4 qreg[NM];
#pragma unroll
                for (int m = 0; m < NM; ++m) qreg[m] = load_sc1(rs_q, (unsigned)((i * E + hd * C::DK + 4 * prt + 16 * m) * 4));
                f32x4 vv0[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int jj = 2 * u + hf < nkeys ? 2 * u + hf : nkeys - 1;
                    vv0[u] = load_sc1(rs_v, kb + (unsigned)((jj * E + 4 * l32) * 4));
                }
                // lane kl of pass ps holds the score of key 16*ps + kl (max_len <= 64: 4 passes)
                float mx = -INFINITY;
                float sreg[4];
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    sreg[ps] = -INFINITY;
                    const int j = 16 * ps + kl;
                    if (16 * ps < nkeys) {           // uniform
                        float d = 0.f;
                        const int jj = j < nkeys ? j : nkeys - 1;
#pragma unroll
                        for (int m = 0; m < NM; ++m) {
                            const f32x4 kk = load_sc1(rs_k, kb + (unsigned)((jj * E + 4 * prt + 16 * m) * 4));
                            d += qreg[m][0] * kk[0] + qreg[m][1] * kk[1] + qreg[m][2] * kk[2] + qreg[m][3] * kk[3];
                        }
                        d += __shfl_xor(d, 1, 64);
                        d += __shfl_xor(d, 2, 64);
                        if (j < nkeys) sreg[ps] = d / scale;
                        mx = fmaxf(mx, sreg[ps]);
                    }
                }
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                float sum = 0.f;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const float e = (16 * ps + kl < nkeys) ? expf(sreg[ps] - mx) : 0.f;
                    sreg[ps] = e;
                    sum += e;
                }
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
                if (pad_row) {
                    // every score of a <PAD> query row is the same fill value: uniform over all Tk keys
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) sreg[ps] = 1.f;
                    sum = (float)nkeys;
                }
                // probabilities through LDS: a wavefront-private row of 64 floats
                float* prow = lds + C::OFF_SC + wave * 64;
                if (prt == 0) {
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps)
                        if (16 * ps + kl < nkeys) prow[16 * ps + kl] = sreg[ps] / sum;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (2 * u + hf < nkeys) o += vv0[u] * prow[2 * u + hf];
                for (int j0 = 16; j0 < nkeys; j0 += 16) {
                    f32x4 vv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int jj = j0 + 2 * u + hf < nkeys ? j0 + 2 * u + hf : nkeys - 1;
                        vv[u] = load_sc1(rs_v, kb + (unsigned)((jj * E + 4 * l32) * 4));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + 2 * u + hf < nkeys) o += vv[u] * prow[j0 + 2 * u + hf];
                }
                o[0] += __shfl_xor(o[0], 32, 64);
                o[1] += __shfl_xor(o[1], 32, 64);
                o[2] += __shfl_xor(o[2], 32, 64);
                o[3] += __shfl_xor(o[3], 32, 64);
                if (hf == 0) *reinterpret_cast<f32x4*>(cst + li * C::DK + 4 * l32) = o;
            }
            wg_barrier();
            FT_EPI_IDS();
            if (ethr) {
                const int sl = t_ & 31, i8 = t_ >> 5;
                const f32x4 v = *reinterpret_cast<const f32x4*>(cst + i8 * C::DK + sl * 4);
                store_sc1(FT_RS(A.f_ctx + tb, blk), (unsigned)(((hd * LPV + sl) * 16 + rbase + i8) * 16), v);
            }
            if (side_l0 && ethr) {
                // a <PAD> query row attends uniformly over all Tk positions: the mean of the layer's values, the newest
                // (written by this team a moment ago) included; this member's columns of every side row
                const int col = member * C::NC + tb_ * 4;
                const unsigned vb = ((unsigned)(ss_b[jb] * A.max_len) * (unsigned)E + col) * 4u;
                const float p = 1.0f / (float)A.Tk;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                for (int j = 0; j < A.Tk; ++j) o += load_sc1(rs_v, vb + (unsigned)(j * E * 4)) * p;
                store_sc1(FT_RS(A.s_ctx + tsb, blk), (unsigned)(((member * (C::NC / 4) + tb_) * 16 + jb) * 16), o);
            }
        }
        FT_MARK();                               // attention stored
        team_signal(tctr, member, nsig + 1);
        ++nsig;
        FT_MARK();

        // ---- output projection + residual ---------------------------------------------------------------------
        if (!team_wait(tctr, nsig, A.err, s_ok)) return;
        FT_MARK();                               // team has the attention output
        {
            f32x4 a[C::NF];
            f32x4 acc[1][C::NCT];
            const float* const W1[1] = {ly.wo};
            load_frags<E>(FT_RS(A.f_ctx + tb, blk), wave, lane, a);
            gemm_stream<E, 1>(lds, W1, side_l0 ? ly.wo : ly.w1, member, wave, lane, lo, a, acc);
            FT_MARK();                           // output projection contracted
            write_partial<E>(lds, wave, lane, acc[0]);
            wg_barrier();
            FT_EPI_IDS();
            if (ethr) {
                const f32x4 v = sum_partials<E>(lds, ja, ta) + *reinterpret_cast<const f32x4*>(ly.bo + member * C::NC + ta * 4) +
                                R[ta * 16 + ja];
                store_sc1(FT_RS(A.f_y + tb, blk), frag_off, v);
            }
            if (side_l0) {
                wg_barrier();
                issue_entry<E>(ly.wo, member, wave, lo, 1, slot1);
                load_frags<E>(FT_RS(A.s_ctx + tsb, blk), wave, lane, a);
                gemm_stream<E, 1>(lds, W1, ly.w1, member, wave, lane, lo, a, acc);
                write_partial<E>(lds, wave, lane, acc[0]);
                wg_barrier();
                FT_EPI_IDS();
                if (ethr) {
                    // residual: the side row's own input, embedding + position
                    const int col = member * C::NC + ta * 4;
                    const int id = A.tok[(int64_t)ss_b[ja] * A.ldtok + ss_pos[ja]];
                    const f32x4 xin = *reinterpret_cast<const f32x4*>(A.emb + (int64_t)id * E + col) +
                                      *reinterpret_cast<const f32x4*>(A.pe + (int64_t)ss_pos[ja] * E + col);
                    const f32x4 w = sum_partials<E>(lds, ja, ta) + *reinterpret_cast<const f32x4*>(ly.bo + col) + xin;
                    store_sc1(FT_RS(A.s_y + tsb, blk), frag_off, w);
                }
            }
        }
        FT_MARK();
        team_signal(tctr, member, nsig + 1);
        ++nsig;
        if (wave != 0) issue_entry<E>(ly.w1, member, wave, lo, 1, slot1);
        FT_MARK();

        // ---- LayerNorm 1, + the cross-attention block (one vector per sample), LayerNorm 2, fc1 + ReLU ---------
        {
            f32x4 a[C::NF];
            f32x4 cadd[C::NF];
            {
                const float* cr = ly.cc + (int64_t)(s_b[c16] / A.cc_div) * E + wave * C::KW + 4 * g4;
#pragma unroll
                for (int kf = 0; kf < C::NF; ++kf) cadd[kf] = *reinterpret_cast<const f32x4*>(cr + kf * 16);
            }
            if (!team_wait(tctr, nsig, A.err, s_ok)) return;
            if (wave == 0) issue_entry<E>(ly.w1, member, wave, lo, 1, slot1);
            FT_MARK();                               // team has y
            load_frags<E>(FT_RS(A.f_y + tb, blk), wave, lane, a);
            ln_frags<E>(a, ly.n1g, ly.n1b, part, pb, wave, lane);
            if (ly.pg) {
                // two memory rows (the gates' partial sums use the side rows' residual area: written later in this stage)
                const int64_t bs = s_b[c16] / A.cc_div, bh = bs * 4;
                pair_block<E>(a, ly.pg + bh * E, ly.pd + bh * E, ly.cc + bs * E, ly.psb + bh, RS, wave, lane);
            } else {
#pragma unroll
                for (int kf = 0; kf < C::NF; ++kf) a[kf] += cadd[kf];
            }
            ln_frags<E>(a, ly.n2g, ly.n2b, part, pb, wave, lane);
            if (wave == member) {
#pragma unroll
                for (int kf = 0; kf < C::NF; ++kf) R[(kf * 4 + g4) * 16 + c16] = a[kf];
            }
            FT_MARK();                               // both LayerNorms done
            f32x4 acc[1][C::NCT];
            const float* const W1[1] = {ly.w1};
            gemm_stream<E, 1>(lds, W1, side_l0 ? ly.w1 : ly.w2, member, wave, lane, lo, a, acc);
            FT_MARK();
            write_partial<E>(lds, wave, lane, acc[0]);
            wg_barrier();
            FT_EPI_IDS();
            if (ethr) {
                f32x4 v = sum_partials<E>(lds, ja, ta) + *reinterpret_cast<const f32x4*>(ly.b1 + member * C::NC + ta * 4);
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                store_sc1(FT_RS(A.f_h + tb, blk), frag_off, v);
            }
            if (side_l0) {
                wg_barrier();
                issue_entry<E>(ly.w1, member, wave, lo, 1, slot1);
                {
                    const float* cs = ly.cc + (int64_t)(ss_b[c16] / A.cc_div) * E + wave * C::KW + 4 * g4;
#pragma unroll
                    for (int kf = 0; kf < C::NF; ++kf) cadd[kf] = *reinterpret_cast<const f32x4*>(cs + kf * 16);
                }
                load_frags<E>(FT_RS(A.s_y + tsb, blk), wave, lane, a);
                ln_frags<E>(a, ly.n1g, ly.n1b, part, pb, wave, lane);
                if (ly.pg) {
                    const int64_t bs = ss_b[c16] / A.cc_div, bh = bs * 4;
                    pair_block<E>(a, ly.pg + bh * E, ly.pd + bh * E, ly.cc + bs * E, ly.psb + bh, RS, wave, lane);
                } else {
#pragma unroll
                    for (int kf = 0; kf < C::NF; ++kf) a[kf] += cadd[kf];
                }
                ln_frags<E>(a, ly.n2g, ly.n2b, part, pb, wave, lane);
                if (wave == member) {
#pragma unroll
                    for (int kf = 0; kf < C::NF; ++kf) RS[(kf * 4 + g4) * 16 + c16] = a[kf];
                }
                gemm_stream<E, 1>(lds, W1, ly.w2, member, wave, lane, lo, a, acc);
                write_partial<E>(lds, wave, lane, acc[0]);
                wg_barrier();
                FT_EPI_IDS();
                if (ethr) {
                    f32x4 w = sum_partials<E>(lds, ja, ta) + *reinterpret_cast<const f32x4*>(ly.b1 + member * C::NC + ta * 4);
                    w[0] = fmaxf(w[0], 0.f); w[1] = fmaxf(w[1], 0.f); w[2] = fmaxf(w[2], 0.f); w[3] = fmaxf(w[3], 0.f);
                    store_sc1(FT_RS(A.s_h + tsb, blk), frag_off, w);
                }
            }
        }
        FT_MARK();
        team_signal(tctr, member, nsig + 1);
        ++nsig;
        if (wave != 0) issue_entry<E>(ly.w2, member, wave, lo, 1, slot1);
        FT_MARK();

        // ---- fc2 + residual -------------------------------------------------------------------------------------
        if (!team_wait(tctr, nsig, A.err, s_ok)) return;
        if (wave == 0) issue_entry<E>(ly.w2, member, wave, lo, 1, slot1);
        FT_MARK();                               // team has h
        {
            const float* nxt = l + 1 < A.L ? A.ly[l + 1].wq : nullptr;
            f32x4 a[C::NF];
            f32x4 acc[1][C::NCT];
            const float* const W1[1] = {ly.w2};
            load_frags<E>(FT_RS(A.f_h + tb, blk), wave, lane, a);
            gemm_stream<E, 1>(lds, W1, side_l0 ? ly.w2 : nxt, member, wave, lane, lo, a, acc);
            FT_MARK();
            write_partial<E>(lds, wave, lane, acc[0]);
            wg_barrier();
            FT_EPI_IDS();
            if (ethr) {
                const f32x4 v = sum_partials<E>(lds, ja, ta) + *reinterpret_cast<const f32x4*>(ly.b2 + member * C::NC + ta * 4) +
                                R[ta * 16 + ja];
                store_sc1(FT_RS(A.f_z + tb, blk), frag_off, v);
            }
            if (side_l0) {
                wg_barrier();
                issue_entry<E>(ly.w2, member, wave, lo, 1, slot1);
                load_frags<E>(FT_RS(A.s_h + tsb, blk), wave, lane, a);
                gemm_stream<E, 1>(lds, W1, nxt, member, wave, lane, lo, a, acc);
                write_partial<E>(lds, wave, lane, acc[0]);
                wg_barrier();
                FT_EPI_IDS();
                if (ethr) {
                    const f32x4 w = sum_partials<E>(lds, ja, ta) + *reinterpret_cast<const f32x4*>(ly.b2 + member * C::NC + ta * 4) +
                                    RS[ta * 16 + ja];
                    store_sc1(FT_RS(A.s_z + tsb, blk), frag_off, w);
                }
            }
            FT_MARK();
            team_signal(tctr, member, nsig + 1);
            ++nsig;
            if (nxt && wave != 0) issue_entry<E>(nxt, member, wave, lo, 1, slot1);
            FT_MARK();
        }
    }

    // ---- LayerNorm 3 of the last layer: the rows that feed the vocabulary projection -------------------------------
    if (!team_wait(tctr, nsig, A.err, s_ok)) return;
    FT_MARK();                                   // team has the last layer's z
    {
        f32x4 a[C::NF];
        load_frags<E>(FT_RS(A.f_z + tb, blk), wave, lane, a);
        ln_frags<E>(a, A.ly[A.L - 1].n3g, A.ly[A.L - 1].n3b, part, pb, wave, lane);
        FT_MARK();
        if (wave == member && (s_flags[c16] & 1)) {
            float* dst = A.x + (int64_t)s_b[c16] * E + wave * C::KW + 4 * g4;
#pragma unroll
            for (int kf = 0; kf < C::NF; ++kf) *reinterpret_cast<f32x4*>(dst + kf * 16) = a[kf];
        }
    }
}

std::mutex g_attr_mutex;
bool g_attr_done[64] = {};

}  // namespace

bool capdec_fused_layers_supported(int E, int H, int L, int64_t B, int max_len) {
    return E == 512 && H == 4 && L >= 1 && L <= 2 && max_len <= 64 && B >= 1 &&
           B * (int64_t)max_len * E * 4 < (int64_t)INT_MAX;
}

int capdec_launch_fused_layers(const capdec_fused_args& a, int E, hipStream_t s) {
    CAPDEC_REQUIRE(E == 512 && a.L >= 1 && a.L <= 2 && a.cc_div >= 1);
    using C = FT<512>;
#ifdef CAPDEC_FUSED_TRACE
    const size_t lds_bytes = getenv("FUSED_PROBE_LDS") ? (size_t)atoi(getenv("FUSED_PROBE_LDS")) : (size_t)C::LDS_FLOATS * sizeof(float);
#else
    constexpr size_t lds_bytes = (size_t)C::LDS_FLOATS * sizeof(float);
#endif
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) return (int)rc;
    CAPDEC_REQUIRE(dev >= 0 && dev < 64);
    {
        std::lock_guard<std::mutex> lock(g_attr_mutex);
        if (!g_attr_done[dev]) {
            rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_layers_team_kernel<512>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (rc != hipSuccess) return (int)rc;
            g_attr_done[dev] = true;
        }
    }
    const unsigned grid = (unsigned)((a.teams + 7) / 8) * 64u;
    hipLaunchKernelGGL(decode_layers_team_kernel<512>, dim3(grid), dim3(FT_THREADS), lds_bytes, s, a);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}
