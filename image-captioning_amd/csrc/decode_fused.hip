// All decoder layers of one KV-cached greedy decode step in ONE launch (transformer.py:83-90 run once per step of
// the loop at transformer.py:151-171), for the single-memory-row form of the cross-attention block (S == 1: the block
// is one vector per sample and layer, see capdec_tf_greedy_decode).
//
// A decode step is a chain of small dependent operations per row: q/k/v projections -> self-attention over the row's
// own cache -> output projection + residual -> LayerNorm 1, + cross-attention constant, LayerNorm 2 -> fc1 + ReLU ->
// fc2 + residual -> LayerNorm 3, and again for the next layer.  No operation mixes rows of different samples.  As
// separate launches every link of that chain pays a kernel boundary and a cold start (7 launches per layer, each about
// 7 us of fixed cost for 5 us of matrix work at 1024 rows).  Here the rows are cut into blocks of 32 and each block is
// given to a TEAM of 8 workgroups that walks the whole chain for it:
//   - a workgroup owns E/8 output columns of every projection (64 at E = 512) and the rows' full contraction depth:
//     its 8 wavefronts split the contraction (k-slabs of E/8), hold the activations as MFMA fragments in REGISTERS,
//     stream their weight slab through a two-slot LDS ring with global_load_lds (the next operation's slab is already
//     on its way while the team waits for its members), sum the 8 partial tiles through LDS and apply bias / ReLU /
//     residual;
//   - between operations the 8 members exchange their 32 x E/8 slices through global memory in a fragment-major
//     layout ([k/4][row][4 floats]: the 16-byte fragment loads of a consumer are contiguous 1 KiB runs), written with
//     write-through (sc1) 16-byte stores, published by one agent-scope atomic add per workgroup behind a drain of
//     every storing wavefront and a workgroup barrier, and read with sc1 loads behind a relaxed poll of that counter
//     by one lane and a workgroup barrier; teams never wait for one another;
//   - LayerNorm runs on the fragments in registers (row sums over the 8 k-slabs meet in LDS), redundantly in the 8
//     members, so it costs no hand-off of its own;
//   - attention: a member takes one head for 16 of the 32 rows (two (row, head) pairs per wavefront).
// Rows that re-run earlier positions of a sample that generated <PAD> (see capdec_tf_greedy_decode) form teams of their
// own at the FRONT of the grid; such a sample's rows attend to cache rows that other re-run rows rewrite in the same
// step, so those teams meet on one more counter between the projections and the attention of every layer.  They hold
// every row of such a sample, its newest position included (the main team's copy of that row stores nothing).
// Progress: a team only ever waits for its own members (the re-run teams: for one another); workgroups are dispatched
// in index order, so the members of the oldest unfinished team are resident or next in line.  Every wait is a bounded
// spin: a workgroup that gives up raises `err` and returns, and the caller re-decodes the batch on the exact path.
#include "common.h"
#include <limits.h>
#include <math.h>
#include <mutex>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int FT_TEAM = CAPDEC_FUSED_TEAM;
constexpr int FT_ROWS = CAPDEC_FUSED_ROWS;
constexpr int FT_WAVES = 8;
constexpr int FT_THREADS = 512;
constexpr unsigned FT_SPIN_LIMIT = 1u << 20;
constexpr int FT_SC1 = 16;      // cache-policy bit of the buffer intrinsics: write-through store / L1-bypassing load

template <int E>
struct FT {
    static constexpr int NC = E / FT_TEAM;            // output columns of a workgroup
    static constexpr int KW = E / FT_WAVES;           // contraction range of a wavefront
    static constexpr int NH = KW / 32;                // 32-deep half-slabs per wavefront and projection
    static constexpr int NT = NC / 32;                // 32x32 accumulators per wavefront
    static constexpr int NA = KW / 8;                 // 16-byte activation fragments per lane
    static constexpr int PIECES = NC / 8;             // global_load_lds instructions per half-slab (8 rows x 128 B each)
    static constexpr int WSLAB = NC * 32;             // floats of one wavefront's half-slab [NC][32]
    static constexpr int WSLOT = FT_WAVES * WSLAB;
    static constexpr int DK = E / 4;                  // head width (H == 4)
    static constexpr int CST_LD = DK + 4;
    static constexpr int OFF_W0 = 0;
    static constexpr int OFF_W1 = WSLOT;              // second ring slot; the k-split partial tiles between operations
    static constexpr int OFF_R1 = 2 * WSLOT;          // [NC/4][32][4] own columns of the layer input (residual)
    static constexpr int OFF_R2 = OFF_R1 + 32 * NC;   // the same of LayerNorm 2's output
    static constexpr int OFF_CST = OFF_R2 + 32 * NC;  // [16][DK+4] attention outputs before they are stored
    static constexpr int OFF_PART = OFF_CST + 16 * CST_LD;      // [2][8][32] LayerNorm partial sums
    static constexpr int OFF_SC = OFF_PART + 2 * FT_WAVES * 32; // [8][64] score rows
    static constexpr int OFF_META = OFF_SC + FT_WAVES * 64;     // ints: 6 x [32] row descriptors, wait flag
    static constexpr int LDS_FLOATS = OFF_META + 6 * 32 + 16;
    static_assert(NH == 2 && NT >= 1 && KW == NC, "team kernel built for E = 512");
};

__device__ __forceinline__ int acc_row32(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

__device__ __forceinline__ f32x4 load_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, FT_SC1));
}
__device__ __forceinline__ void store_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs, (int)byte_off, 0, FT_SC1);
}

// one half-slab of a weight slice, straight into this wavefront's part of a ring slot: LDS image [NC][32] floats with
// the 16-byte slots of a row XOR-swizzled by (row >> 1) & 7 (conflict-free 16-byte fragment reads of 32 rows)
template <int E>
__device__ __forceinline__ void issue_half(const float* __restrict__ W, int member, int wave, int lane, int half,
                                           float* slot) {
    using C = FT<E>;
    float* dst = slot + wave * C::WSLAB;
#pragma unroll
    for (int p = 0; p < C::PIECES; ++p) {
        const int row = p * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const float* src = W + (int64_t)(member * C::NC + row) * E + wave * C::KW + half * 32 + chunk * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + p * 256), 16, 0, 0);
    }
}

template <int E>
__device__ __forceinline__ void mma_half(const float* slot, int wave, int r, int h, const f32x4* a,
                                         f32x16 (&acc)[FT<E>::NT]) {
    using C = FT<E>;
    const float* img = slot + wave * C::WSLAB;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        f32x4 b[C::NT];
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const int row = t * 32 + r;
            b[t] = *reinterpret_cast<const f32x4*>(img + row * 32 + ((((kc * 2 + h) ^ ((row >> 1) & 7))) << 2));
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < C::NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kc][m], b[t][m], acc[t], 0, 0, 0);
    }
}

// NPROJ projections of the same activation fragments.  The stream of half-slabs W[0] half 0, W[0] half 1, W[1] half 0,
// ... alternates between the two ring slots; its first two entries were issued before the call, entry n + 2 is issued
// as soon as this wavefront has read entry n, and once slot 0 is free for good the first half-slab of `next` (the next
// operation's weights; nullable) follows.  Nothing else of this wavefront is in flight inside the loop, so the counted
// waits see exactly the ring's loads (they retire in order).
template <int E, int NPROJ>
__device__ __forceinline__ void gemm_stream(float* lds, const float* const (&W)[NPROJ], const float* next, int member,
                                            int wave, int lane, const f32x4 (&a)[FT<E>::NA],
                                            f32x16 (&acc)[NPROJ][FT<E>::NT]) {
    using C = FT<E>;
    constexpr int N = NPROJ * C::NH;
    const int r = lane & 31, h = lane >> 5;
    float* slot0 = lds + C::OFF_W0;
    float* slot1 = lds + C::OFF_W1;
#pragma unroll
    for (int p = 0; p < NPROJ; ++p)
#pragma unroll
        for (int t = 0; t < C::NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[p][t][e] = 0.f;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        if (n == 0 || (n == N - 1 && !next)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PIECES) : "memory");
        mma_half<E>((n & 1) ? slot1 : slot0, wave, r, h, &a[(n % C::NH) * 4], acc[n / C::NH]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wavefront's fragment reads of the slot are done
        if (n + 2 < N) issue_half<E>(W[(n + 2) / C::NH], member, wave, lane, (n + 2) % C::NH, (n & 1) ? slot1 : slot0);
        else if (n + 2 == N && next) issue_half<E>(next, member, wave, lane, 0, slot0);
    }
}

// this wavefront's 32 x NC partial tile into its own part of slot 1 (exactly the bytes of its half-slab: no other
// wavefront's weights are overwritten); 16-byte slots of a row XOR-swizzled by row & 15
template <int E>
__device__ __forceinline__ void write_partial(float* lds, int wave, int lane, const f32x16 (&acc)[FT<E>::NT]) {
    using C = FT<E>;
    const int r = lane & 31, h = lane >> 5;
    float* d = lds + C::OFF_W1 + wave * C::WSLAB;
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = acc_row32(e, h), j = t * 32 + r;
            d[i * C::NC + ((((j >> 2) ^ (i & 15)) << 2) | (j & 3))] = acc[t][e];
        }
}
template <int E>
__device__ __forceinline__ f32x4 sum_partials(const float* lds, int i, int sl) {
    using C = FT<E>;
    const float* p = lds + C::OFF_W1 + i * C::NC + ((sl ^ (i & 15)) << 2);
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int w = 1; w < FT_WAVES; ++w) v += *reinterpret_cast<const f32x4*>(p + w * C::WSLAB);
    return v;
}

// Every storing wavefront drains its stores, the workgroup meets, one lane publishes.
__device__ __forceinline__ void team_signal(unsigned* c0, unsigned* c1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(c0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c1) __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// One lane polls (relaxed, L1-bypassing), the workgroup meets behind it; false (for every thread) when the spin ran out.
__device__ __forceinline__ bool team_wait(unsigned* ctr, unsigned target, unsigned* err, int* flag) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > FT_SPIN_LIMIT) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *flag = ok;
    }
    __syncthreads();
    const bool ok = *flag != 0;
    __syncthreads();             // the flag may be rewritten by the next wait only after everyone has read it
    return ok;
}

template <int E>
__device__ __forceinline__ void load_frags(__amdgpu_buffer_rsrc_t rs, int wave, int lane, f32x4 (&a)[FT<E>::NA]) {
    using C = FT<E>;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int kc = 0; kc < C::NA; ++kc) a[kc] = load_sc1(rs, (unsigned)(((wave * (C::KW / 4) + kc * 2 + h) * 32 + r) * 16));
}

// LayerNorm (eps 1e-5, affine) of the 32 rows held as fragments: lane (r, h) of wavefront w holds the columns
// w*KW + kc*8 + 4h + {0..3} of row r.  Two-pass mean / variance; the 8 wavefronts' row sums meet in LDS.
template <int E>
__device__ __forceinline__ void ln_frags(f32x4 (&a)[FT<E>::NA], const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float* part, int& pb, int wave, int lane) {
    using C = FT<E>;
    const int r = lane & 31, h = lane >> 5;
    float s = 0.f;
#pragma unroll
    for (int kc = 0; kc < C::NA; ++kc) s += (a[kc][0] + a[kc][1]) + (a[kc][2] + a[kc][3]);
    s += __shfl_xor(s, 32, 64);
    float* p = part + pb * (FT_WAVES * 32);
    pb ^= 1;
    if (h == 0) p[wave * 32 + r] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < FT_WAVES; ++w) tot += p[w * 32 + r];
    const float mean = tot / (float)E;
    float q = 0.f;
#pragma unroll
    for (int kc = 0; kc < C::NA; ++kc) {
        a[kc] -= mean;
        q += (a[kc][0] * a[kc][0] + a[kc][1] * a[kc][1]) + (a[kc][2] * a[kc][2] + a[kc][3] * a[kc][3]);
    }
    q += __shfl_xor(q, 32, 64);
    p = part + pb * (FT_WAVES * 32);
    pb ^= 1;
    if (h == 0) p[wave * 32 + r] = q;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int w = 0; w < FT_WAVES; ++w) tot += p[w * 32 + r];
    const float rstd = 1.0f / sqrtf(tot / (float)E + 1e-5f);
    const int c0 = wave * C::KW + 4 * h;
#pragma unroll
    for (int kc = 0; kc < C::NA; ++kc) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c0 + kc * 8);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + c0 + kc * 8);
        a[kc] = a[kc] * rstd * g + bb;
    }
}

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
__global__ __launch_bounds__(FT_THREADS) void decode_layers_team_kernel(capdec_fused_args A) {
    using C = FT<E>;
    int nmark = 0;
    (void)nmark;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    // consecutive workgroup ids go round-robin to the XCDs: the members of a team share id & 7 (speed only)
    const int xg = blockIdx.x & 7, gslot = blockIdx.x >> 3;
    const int team = (gslot >> 3) * 8 + xg, member = gslot & 7;
    const int nteams = A.extra_teams + A.main_teams;
    if (team >= nteams) return;
    const int nrows = A.nrows[0];
    if (nrows <= 0) return;                                 // every caption has finished
    const bool extra = team < A.extra_teams;
    const int n_act_extra = (max(nrows - A.B, 0) + FT_ROWS - 1) / FT_ROWS;
    if (extra && team >= n_act_extra) return;

    float* slot0 = lds + C::OFF_W0;
    float* slot1 = lds + C::OFF_W1;
    issue_half<E>(A.ly[0].wq, member, wave, lane, 0, slot0);
    issue_half<E>(A.ly[0].wq, member, wave, lane, 1, slot1);

    int* meta = reinterpret_cast<int*>(lds + C::OFF_META);
    int *s_map = meta, *s_b = meta + 32, *s_pos = meta + 64, *s_flags = meta + 96, *s_xrow = meta + 128, *s_src = meta + 160;
    int* s_ok = meta + 192;
    if (tid < FT_ROWS) {
        int srow, store, xrow;
        if (extra) {
            srow = A.B + team * FT_ROWS + tid;
            const bool act = srow < nrows;
            if (!act) srow = 0;
            const int m = A.rowmap[srow];
            const int b = m / A.max_len, pos = m - b * A.max_len;
            store = act;
            xrow = (act && pos == A.Tk - 1) ? b : -1;        // the sample's newest position feeds the vocabulary projection
            s_map[tid] = m; s_b[tid] = b; s_pos[tid] = pos;
        } else {
            srow = (team - A.extra_teams) * FT_ROWS + tid;
            const bool act = srow < A.B;
            if (!act) srow = 0;
            const int m = A.rowmap[srow];
            const int b = m / A.max_len, pos = m - b * A.max_len;
            store = act && !A.infected[b];                   // a sample with re-run rows is computed by their teams
            xrow = store ? srow : -1;
            s_map[tid] = m; s_b[tid] = b; s_pos[tid] = pos;
        }
        const int pad = A.tok[(int64_t)s_b[tid] * A.ldtok + s_pos[tid]] == 0;
        s_flags[tid] = store | (pad << 1);
        s_xrow[tid] = xrow;
        s_src[tid] = srow;
    }
    __syncthreads();

    unsigned* tctr = A.ctr + (size_t)team * 32;
    unsigned* sctr = extra ? A.ctr + (size_t)nteams * 32 : nullptr;
    unsigned nsig = 0;
    const size_t tb = (size_t)team * FT_ROWS * E;            // this team's 32 x E block in every exchange buffer
    const int blk_bytes = FT_ROWS * E * 4;
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(A.q + tb, 0, blk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ctx = __builtin_amdgcn_make_buffer_rsrc(A.f_ctx + tb, 0, blk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(A.f_y + tb, 0, blk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(A.f_h + tb, 0, blk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(A.f_z + tb, 0, blk_bytes, 0x00020000);
    float* part = lds + C::OFF_PART;
    int pb = 0;
    f32x4* R1 = reinterpret_cast<f32x4*>(lds + C::OFF_R1);
    f32x4* R2 = reinterpret_cast<f32x4*>(lds + C::OFF_R2);
    // epilogue maps: (ia, sa) for fragment-major stores (rows fastest), (ib, sb) for row-major stores (columns fastest)
    const int ia = tid & 31, sa = tid >> 5, ib = tid >> 4, sb = tid & 15;
    const unsigned frag_off = (unsigned)(((member * (C::NC / 4) + sa) * 32 + ia) * 16);
    const int cache_bytes = A.cache_bytes;

    f32x4 a[C::NA];
    FT_MARK();                                   // 0: set-up done
    for (int l = 0; l < A.L; ++l) {
        const capdec_fused_layer& ly = A.ly[l];
        const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(ly.kc, 0, cache_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(ly.vc, 0, cache_bytes, 0x00020000);

        // ---- q, k, v projections of the layer input ---------------------------------------------------------
        if (l == 0) {
            const float* xr = A.x + (int64_t)s_src[r] * E + wave * C::KW + 4 * h;
#pragma unroll
            for (int kc = 0; kc < C::NA; ++kc) a[kc] = *reinterpret_cast<const f32x4*>(xr + kc * 8);
        } else {
            if (!team_wait(tctr, FT_TEAM * nsig, A.err, s_ok)) return;
            load_frags<E>(rs_z, wave, lane, a);
            ln_frags<E>(a, A.ly[l - 1].n3g, A.ly[l - 1].n3b, part, pb, wave, lane);
        }
        if (wave == member) {
#pragma unroll
            for (int kc = 0; kc < C::NA; ++kc) R1[(kc * 2 + h) * 32 + r] = a[kc];
        }
        FT_MARK();                               // layer input in registers
        {
            f32x16 acc[3][C::NT];
            const float* const W3[3] = {ly.wq, ly.wk, ly.wv};
            gemm_stream<E, 3>(lds, W3, ly.wo, member, wave, lane, a, acc);
            FT_MARK();                           // q, k, v contractions done
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                write_partial<E>(lds, wave, lane, acc[p]);
                __syncthreads();
                const float* bias = p == 0 ? ly.bq : (p == 1 ? ly.bk : ly.bv);
                const int col = member * C::NC + sb * 4;
                const f32x4 v = sum_partials<E>(lds, ib, sb) + *reinterpret_cast<const f32x4*>(bias + col);
                if (p == 0) store_sc1(rs_q, (unsigned)((ib * E + col) * 4), v);
                else if (s_flags[ib] & 1) store_sc1(p == 1 ? rs_k : rs_v, ((unsigned)s_map[ib] * (unsigned)E + col) * 4u, v);
                __syncthreads();
            }
        }
        FT_MARK();                               // q, k, v stored
        team_signal(tctr, sctr);
        ++nsig;
        issue_half<E>(ly.wo, member, wave, lane, 1, slot1);
        FT_MARK();                               // signalled

        // ---- self-attention: member = (head, half of the rows), two (row, head) pairs per wavefront -----------
        if (!team_wait(tctr, FT_TEAM * nsig, A.err, s_ok)) return;
        if (extra && !team_wait(sctr, (unsigned)(n_act_extra * FT_TEAM * (l + 1)), A.err, s_ok)) return;
        FT_MARK();                               // team has q, k, v
        {
            constexpr int NM = C::DK / 16, LPV = C::DK / 4, NGRP = 64 / LPV;
            const int hd = member >> 1, rbase = (member & 1) * 16;
            float* sc = lds + C::OFF_SC + wave * 64;
            float* cst = lds + C::OFF_CST;
            const float scale = sqrtf((float)C::DK);
#pragma unroll 1
            for (int it = 0; it < 2; ++it) {
                const int li = 2 * wave + it, i = rbase + li;
                const int b = s_b[i], pos = s_pos[i];
                const bool pad_row = (s_flags[i] & 2) != 0;
                const int nkeys = pad_row ? A.Tk : min(pos + 1, A.Tk);
                const unsigned kb = ((unsigned)(b * A.max_len) * (unsigned)E + (unsigned)(hd * C::DK)) * 4u;
                float sum = 1.f;
                if (!pad_row) {
                    const int prt = lane & 3, kl = lane >> 2;
                    f32x4 qreg[NM];
#pragma unroll
                    for (int m = 0; m < NM; ++m) qreg[m] = load_sc1(rs_q, (unsigned)((i * E + hd * C::DK + 4 * prt + 16 * m) * 4));
                    for (int j0 = 0; j0 < nkeys; j0 += 16) {
                        const int j = j0 + kl;
                        float d = 0.f;
                        if (j < nkeys) {
#pragma unroll
                            for (int m = 0; m < NM; ++m) {
                                const f32x4 kk = load_sc1(rs_k, kb + (unsigned)((j * E + 4 * prt + 16 * m) * 4));
                                d += qreg[m][0] * kk[0] + qreg[m][1] * kk[1] + qreg[m][2] * kk[2] + qreg[m][3] * kk[3];
                            }
                        }
                        d += __shfl_xor(d, 1, 64);
                        d += __shfl_xor(d, 2, 64);
                        if (prt == 0 && j < nkeys) sc[j] = d / scale;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    float mx = -INFINITY;
                    for (int j = lane; j < nkeys; j += 64) mx = fmaxf(mx, sc[j]);
                    mx = wave_max(mx);
                    float s = 0.f;
                    for (int j = lane; j < nkeys; j += 64) {
                        const float e = expf(sc[j] - mx);
                        sc[j] = e;
                        s += e;
                    }
                    sum = wave_sum(s);
                } else {
                    // every score of a <PAD> query row is the same fill value: uniform over all Tk keys
                    for (int j = lane; j < nkeys; j += 64) sc[j] = 1.f;
                    sum = (float)nkeys;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int grp = lane / LPV, dl = lane - grp * LPV;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                for (int j = grp; j < nkeys; j += NGRP) {
                    const float p = sc[j] / sum;
                    const f32x4 vv = load_sc1(rs_v, kb + (unsigned)((j * E + 4 * dl) * 4));
                    o += vv * p;
                }
#pragma unroll
                for (int off = LPV; off < 64; off <<= 1) {
                    o[0] += __shfl_xor(o[0], off, 64);
                    o[1] += __shfl_xor(o[1], off, 64);
                    o[2] += __shfl_xor(o[2], off, 64);
                    o[3] += __shfl_xor(o[3], off, 64);
                }
                if (grp == 0) *reinterpret_cast<f32x4*>(cst + li * C::CST_LD + 4 * dl) = o;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the score row is rewritten by the next pair
            }
            __syncthreads();
            for (int idx = tid; idx < 16 * LPV; idx += FT_THREADS) {
                const int i16 = idx & 15, sl = idx >> 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(cst + i16 * C::CST_LD + sl * 4);
                store_sc1(rs_ctx, (unsigned)(((hd * LPV + sl) * 32 + rbase + i16) * 16), v);
            }
        }
        FT_MARK();                               // attention stored
        team_signal(tctr, nullptr);
        ++nsig;
        FT_MARK();

        // ---- output projection + residual ---------------------------------------------------------------------
        if (!team_wait(tctr, FT_TEAM * nsig, A.err, s_ok)) return;
        FT_MARK();                               // team has the attention output
        load_frags<E>(rs_ctx, wave, lane, a);
        {
            f32x16 acc[1][C::NT];
            const float* const W1[1] = {ly.wo};
            gemm_stream<E, 1>(lds, W1, ly.w1, member, wave, lane, a, acc);
            FT_MARK();                           // output projection contracted
            write_partial<E>(lds, wave, lane, acc[0]);
            __syncthreads();
            const f32x4 v = sum_partials<E>(lds, ia, sa) + *reinterpret_cast<const f32x4*>(ly.bo + member * C::NC + sa * 4) +
                            R1[sa * 32 + ia];
            store_sc1(rs_y, frag_off, v);
        }
        FT_MARK();
        team_signal(tctr, nullptr);
        ++nsig;
        issue_half<E>(ly.w1, member, wave, lane, 1, slot1);
        FT_MARK();

        // ---- LayerNorm 1, + the cross-attention block (one vector per sample), LayerNorm 2, fc1 + ReLU ---------
        f32x4 cadd[C::NA];
        {
            const float* cr = ly.cc + (int64_t)s_b[r] * E + wave * C::KW + 4 * h;
#pragma unroll
            for (int kc = 0; kc < C::NA; ++kc) cadd[kc] = *reinterpret_cast<const f32x4*>(cr + kc * 8);
        }
        if (!team_wait(tctr, FT_TEAM * nsig, A.err, s_ok)) return;
        FT_MARK();                               // team has y
        load_frags<E>(rs_y, wave, lane, a);
        ln_frags<E>(a, ly.n1g, ly.n1b, part, pb, wave, lane);
#pragma unroll
        for (int kc = 0; kc < C::NA; ++kc) a[kc] += cadd[kc];
        ln_frags<E>(a, ly.n2g, ly.n2b, part, pb, wave, lane);
        if (wave == member) {
#pragma unroll
            for (int kc = 0; kc < C::NA; ++kc) R2[(kc * 2 + h) * 32 + r] = a[kc];
        }
        FT_MARK();                               // both LayerNorms done
        {
            f32x16 acc[1][C::NT];
            const float* const W1[1] = {ly.w1};
            gemm_stream<E, 1>(lds, W1, ly.w2, member, wave, lane, a, acc);
            FT_MARK();
            write_partial<E>(lds, wave, lane, acc[0]);
            __syncthreads();
            f32x4 v = sum_partials<E>(lds, ia, sa) + *reinterpret_cast<const f32x4*>(ly.b1 + member * C::NC + sa * 4);
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            store_sc1(rs_h, frag_off, v);
        }
        FT_MARK();
        team_signal(tctr, nullptr);
        ++nsig;
        issue_half<E>(ly.w2, member, wave, lane, 1, slot1);
        FT_MARK();

        // ---- fc2 + residual -------------------------------------------------------------------------------------
        if (!team_wait(tctr, FT_TEAM * nsig, A.err, s_ok)) return;
        FT_MARK();                               // team has h
        load_frags<E>(rs_h, wave, lane, a);
        {
            const float* nxt = l + 1 < A.L ? A.ly[l + 1].wq : nullptr;
            f32x16 acc[1][C::NT];
            const float* const W1[1] = {ly.w2};
            gemm_stream<E, 1>(lds, W1, nxt, member, wave, lane, a, acc);
            FT_MARK();
            write_partial<E>(lds, wave, lane, acc[0]);
            __syncthreads();
            const f32x4 v = sum_partials<E>(lds, ia, sa) + *reinterpret_cast<const f32x4*>(ly.b2 + member * C::NC + sa * 4) +
                            R2[sa * 32 + ia];
            store_sc1(rs_z, frag_off, v);
            FT_MARK();
            team_signal(tctr, nullptr);
            ++nsig;
            if (nxt) issue_half<E>(nxt, member, wave, lane, 1, slot1);
            FT_MARK();
        }
    }

    // ---- LayerNorm 3 of the last layer: the rows that feed the vocabulary projection -------------------------------
    if (!team_wait(tctr, FT_TEAM * nsig, A.err, s_ok)) return;
    FT_MARK();                                   // team has the last layer's z
    load_frags<E>(rs_z, wave, lane, a);
    ln_frags<E>(a, A.ly[A.L - 1].n3g, A.ly[A.L - 1].n3b, part, pb, wave, lane);
    FT_MARK();
    if (wave == member) {
        const int xr = s_xrow[r];
        if (xr >= 0) {
            float* dst = A.x + (int64_t)xr * E + wave * C::KW + 4 * h;
#pragma unroll
            for (int kc = 0; kc < C::NA; ++kc) *reinterpret_cast<f32x4*>(dst + kc * 8) = a[kc];
        }
    }
}

std::mutex g_attr_mutex;
bool g_attr_done[64] = {};

}  // namespace

bool capdec_fused_layers_supported(int E, int H, int L, int64_t B, int max_len) {
    return E == 512 && H == 4 && L >= 1 && L <= CAPDEC_FUSED_MAX_LAYERS && max_len <= 64 &&
           B * (int64_t)max_len * E * 4 < (int64_t)INT_MAX;
}

int capdec_launch_fused_layers(const capdec_fused_args& a, int E, hipStream_t s) {
    CAPDEC_REQUIRE(E == 512);
    using C = FT<512>;
    constexpr size_t lds_bytes = (size_t)C::LDS_FLOATS * sizeof(float);
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
    const int nteams = a.extra_teams + a.main_teams;
    const unsigned grid = (unsigned)((nteams + 7) / 8) * 64u;
    hipLaunchKernelGGL(decode_layers_team_kernel<512>, dim3(grid), dim3(FT_THREADS), lds_bytes, s, a);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}
