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
// Operand tiles reach LDS in one of two ways: written there by the loads themselves (global_load_lds into a
// linear, XOR-swizzled image: mainloop_dma, taken for aligned operands made of whole k-tiles) or through
// registers (mainloop: rows padded to BK+4 floats, zero fill for ragged K).  Either way the ds_read_b128
// fragment reads (32 rows x 16 B per half-wave) are bank-conflict free and the sums run in the same order.
#include "common.h"
#include <mutex>
#include <limits.h>
#include <math.h>

namespace {

// Workgroup -> tile map.  Consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2, so a
// plain (x,y) grid spreads the tiles that share an operand slice over all eight L2s and every one of them
// fetches that slice.  Here workgroup L belongs to XCD L & 7 and takes the (L >> 3)-th tile of that XCD's own
// sequence; the sequences are cut so that tiles sharing operand rows run on one XCD, back to back:
//   by_z == 0: the (ni x nj) tile grid of every z is cut into super-blocks of si x sj tiles dealt round-robin to
//              the XCDs; a super-block reads si + sj operand slices for si*sj tiles;
//   by_z == 1: (split-K) XCD k takes the z planes k, k+8, ...: a plane's operand k-range is read by one L2 only.
// The XCD assignment is a performance assumption only; any placement computes the same tiles.
struct TileMap {
    int ni, nj, nz, si, sj, nsj, per_z, by_z;
};

__device__ __forceinline__ bool tile_of_block(const TileMap& m, unsigned L, int& z, int& ti, int& tj) {
    if (m.by_z) {
        const int xcd = L & 7, slot = L >> 3;
        const int nt = m.ni * m.nj;
        const int zq = slot / nt, t = slot - zq * nt;
        z = zq * 8 + xcd;
        ti = t / m.nj;
        tj = t - ti * m.nj;
        return z < m.nz;
    }
    z = L / m.per_z;
    L -= z * m.per_z;
    const int xcd = L & 7, slot = L >> 3;
    const int per = m.si * m.sj;
    const int sb = slot / per, w = slot - sb * per;
    const int G = sb * 8 + xcd;
    const int sbi = G / m.nsj, sbj = G - sbi * m.nsj;
    const int wi = w / m.sj;
    ti = sbi * m.si + wi;
    tj = sbj * m.sj + (w - wi * m.sj);
    return ti < m.ni && tj < m.nj;
}

// host side: fills the map, returns the 1-D grid size
unsigned make_tile_map(TileMap& m, int ni, int nj, int nz, bool by_z) {
    m.ni = ni; m.nj = nj; m.nz = nz;
    m.by_z = by_z ? 1 : 0;
    m.sj = nj < 8 ? nj : 8;
    m.nsj = (nj + m.sj - 1) / m.sj;
    m.si = 1;
    while (m.si < 8 && (long)((ni + 2 * m.si - 1) / (2 * m.si)) * m.nsj >= 128) m.si *= 2;
    const int nsb = ((ni + m.si - 1) / m.si) * m.nsj;
    m.per_z = 8 * ((nsb + 7) / 8) * m.si * m.sj;
    if (by_z) return (unsigned)(8 * ((nz + 7) / 8)) * (unsigned)(ni * nj);
    return (unsigned)m.per_z * (unsigned)nz;
}

struct GemmKArgs {
    capdec_gemm_group g[4];
    int K;
    int flags;
    int nsplit;          // k-range splits per group (z = group * nsplit + split); > 1 only with ATOMIC
    TileMap map;
    capdec_gemm_extra x;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// One operand tile of NX rows (the i or j axis) by BK columns (the contraction axis), staged by THREADS threads.
//  T == false: the operand is stored [row][k] (k contiguous).  LDS image [row][BK+4]: the padded row stride
//              (144 B at BK=32, 80 B at BK=16) makes the 16-B fragment reads of 32 rows bank-conflict free;
//              a lane reads its four k values with one ds_read_b128.
//  T == true : the operand is stored [k][row] (row contiguous; the contraction runs over its leading
//              axis, as in dX = dY W and dW = dY^T X).  LDS image [k][NX+4]; a lane reads its four k values
//              with four ds_read_b32 (32 consecutive floats per half-wave: conflict-free).
template <int NX_, int BK, bool VEC, bool T, int THREADS>
struct Operand {
    static constexpr int NX = NX_;
    static constexpr int LDK = BK + 4;
    static constexpr int LDT = NX + 4;
    static constexpr int TPR = BK / 4;            // threads per row (normal layout)
    static constexpr int RPP = THREADS / TPR;     // rows per pass (normal layout)
    static constexpr int NV = NX / 4;             // float4 per k-row (transposed layout)
    static constexpr int KPP = THREADS / NV;      // k-rows per pass (transposed layout)
    static constexpr int NR = NX * BK / 4 / THREADS;   // float4 per thread per tile
    static constexpr int BUF = cmax(NX * LDK, BK * LDT);
    static_assert(NX % RPP == 0 && BK % KPP == 0 && NR >= 1, "tile does not divide over the threads");
    const float* __restrict__ base;
    int64_t ld;
    int n, K, x0;
    float* lds;                                   // 2 buffers of BUF floats
    int tid;                                      // thread index inside its k-group

    // Loads are branch-free (addresses clamped into the operand) so that the compiler can keep them in
    // flight across the MFMAs with counted waits.  Rows past the operand's end receive copies of its
    // last row: they only feed output rows/columns that are never stored.  Columns past K are zeroed
    // when the tile is written to LDS (lstore), since they would otherwise enter every dot product.
    __device__ __forceinline__ void gload(int kt, f32x4 (&reg)[NR]) const {
        if (!T) {
            const int lrow = tid / TPR, k = kt * BK + (tid % TPR) * 4;
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                const int row = min(x0 + lrow + RPP * p, n - 1);
                const float* q = base + (int64_t)row * ld;
                if (VEC) {
                    reg[p] = *reinterpret_cast<const f32x4*>(q + min(k, K - 4));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) reg[p][e] = q[min(k + e, K - 1)];
                }
            }
        } else {
            const int iv = tid % NV, kr = tid / NV;
            const int row = x0 + 4 * iv;
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                const int k = min(kt * BK + kr + KPP * p, K - 1);
                const float* q = base + (int64_t)k * ld;
                if (VEC) {   // n % 4 == 0 here: a vector never straddles the operand's end
                    reg[p] = *reinterpret_cast<const f32x4*>(q + min(row, n - 4));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) reg[p][e] = q[min(row + e, n - 1)];
                }
            }
        }
    }
    __device__ __forceinline__ void lstore(int buf, int kt, const f32x4 (&reg)[NR]) const {
        if (!T) {
            const int lrow = tid / TPR, lk = (tid % TPR) * 4;
            const int k = kt * BK + lk;
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                f32x4 v = reg[p];
                if (VEC) {
                    if (k >= K) v = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e >= K) v[e] = 0.f;
                }
                *reinterpret_cast<f32x4*>(&lds[buf * BUF + (lrow + RPP * p) * LDK + lk]) = v;
            }
        } else {
            const int iv = tid % NV, kr = tid / NV;
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                f32x4 v = reg[p];
                if (kt * BK + kr + KPP * p >= K) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&lds[buf * BUF + (kr + KPP * p) * LDT + 4 * iv]) = v;
            }
        }
    }
    // the four k values {kc*8 + 4h + m} of row (w0 + r) for the lane (r, h)
    __device__ __forceinline__ f32x4 frag(int buf, int w0, int kc, int r, int h) const {
        if (!T) return *reinterpret_cast<const f32x4*>(&lds[buf * BUF + (w0 + r) * LDK + kc * 8 + 4 * h]);
        const float* q = &lds[buf * BUF + (kc * 8 + 4 * h) * LDT + w0 + r];
        return f32x4{q[0], q[LDT], q[2 * LDT], q[3 * LDT]};
    }
};

// workgroup = KG k-groups of (WM x 2) wavefronts; tile = (32*WM*WI) x (64*WJ)
template <int WM, int WI, int WJ, int BK, bool TP, bool TQ, int KG>
constexpr int lds_floats() {
    return KG * 2 * (Operand<32 * WM * WI, BK, true, TP, 128 * WM>::BUF + Operand<64 * WJ, BK, true, TQ, 128 * WM>::BUF);
}

// Accumulate k-tiles [kt0, kt1) of the (i0, j0) tile into acc.  Global loads run TWO tiles ahead of the
// MFMAs (two register sets, statically named), LDS is double buffered, one barrier per tile.
// A k-group is WM x 2 wavefronts (each owning WI x WJ accumulators of 32x32) on a (32*WM*WI) x (64*WJ) tile.
// KG > 1: the workgroup holds KG such groups; group g takes the k-tiles kt0+g, kt0+g+KG, ... into its own LDS
// buffers and the partial tiles are summed through LDS into group 0 at the end.  With WM = 1, KG = 2 a
// 256-thread workgroup covers a 32x64 tile: twice the workgroups of the 64x64 form, for problems whose grid
// would otherwise leave CUs idle (the MFMA pipe is per CU, so only more workgroups shorten such a launch).
// Ends with every wave past the last barrier, so the caller may reuse `lds` right away.
template <int WM, int WI, int WJ, int BK, bool VEC, bool TP = false, bool TQ = false, int KG = 1>
__device__ __forceinline__ void mainloop(const float* __restrict__ P, const float* __restrict__ Q, int64_t ldp,
                                         int64_t ldq, int I, int J, int K, int i0, int j0, int kt0, int kt1,
                                         float* lds, f32x16 (&acc)[WI][WJ], bool keep_acc = false, int KQ = -1) {
    // keep_acc: add onto the caller's accumulators instead of starting from zero (a second product into the same tile)
    // KQ: contraction length of Q when it differs from P's (P padded with zero columns up to a multiple of four)
    constexpr int GT = 128 * WM;                  // threads per k-group
    using OpP = Operand<32 * WM * WI, BK, VEC, TP, GT>;
    using OpQ = Operand<64 * WJ, BK, VEC, TQ, GT>;
    constexpr int GROUP_LDS = 2 * (OpP::BUF + OpQ::BUF);
    const int kg = KG > 1 ? (int)(threadIdx.x / GT) : 0;
    const int tid = threadIdx.x % GT;
    const int lane = tid & 63, wave = tid >> 6;
    const int wi = WM == 2 ? wave >> 1 : 0, wj = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    float* glds = lds + kg * GROUP_LDS;
    const OpP op{P, ldp, I, K, i0, glds, tid};
    const OpQ oq{Q, ldq, J, KQ >= 0 ? KQ : K, j0, glds + 2 * OpP::BUF, tid};

    if (!keep_acc) {
#pragma unroll
        for (int ti = 0; ti < WI; ++ti)
#pragma unroll
            for (int tj = 0; tj < WJ; ++tj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ti][tj][e] = 0.f;
    }
    if (kt0 >= kt1) return;

    auto compute = [&](int buf) {
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            f32x4 a[WI], b[WJ];
#pragma unroll
            for (int ti = 0; ti < WI; ++ti) a[ti] = op.frag(buf, wi * 32 * WI + ti * 32, kc, r, h);
#pragma unroll
            for (int tj = 0; tj < WJ; ++tj) b[tj] = oq.frag(buf, wj * 32 * WJ + tj * 32, kc, r, h);
            // lane half h feeds k = kc*8 + 4h + m, so MFMA m sums k in {kc*8+m, kc*8+4+m}
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ti = 0; ti < WI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < WJ; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti][m], b[tj][m], acc[ti][tj], 0, 0, 0);
        }
    };

    f32x4 pa[OpP::NR], qa[OpQ::NR], pb[OpP::NR], qb[OpQ::NR];   // register sets A and B
    const int last = kt1 - 1;
    const int nit = (kt1 - kt0 + KG - 1) / KG;   // the same trip count in every k-group (shared barriers)
    // this group's it-th tile; indices past the range are clamped to the last tile for the loads (redundant,
    // never multiplied), which keeps the loop body free of branches around memory operations
    auto tile = [&](int it) { return kt0 + kg + it * KG; };
    op.gload(min(tile(0), last), pa);
    oq.gload(min(tile(0), last), qa);
    op.gload(min(tile(1), last), pb);
    oq.gload(min(tile(1), last), qb);
    op.lstore(0, min(tile(0), last), pa);
    oq.lstore(0, min(tile(0), last), qa);
    __syncthreads();
    for (int it = 0; it < nit; it += 2) {
        // tile it is in LDS[0]; set B holds tile it+1; fetch tile it+2 into set A
        // (the scheduling fences keep the compiler from sinking the loads below the MFMAs)
        op.gload(min(tile(it + 2), last), pa);
        oq.gload(min(tile(it + 2), last), qa);
        __builtin_amdgcn_sched_barrier(0);
        if (KG == 1 || tile(it) <= last) compute(0);
        __builtin_amdgcn_sched_barrier(0);
        op.lstore(1, min(tile(it + 1), last), pb);
        oq.lstore(1, min(tile(it + 1), last), qb);
        __syncthreads();
        if (it + 1 >= nit) break;
        // tile it+1 is in LDS[1]; set A holds tile it+2; fetch tile it+3 into set B
        op.gload(min(tile(it + 3), last), pb);
        oq.gload(min(tile(it + 3), last), qb);
        __builtin_amdgcn_sched_barrier(0);
        if (KG == 1 || tile(it + 1) <= last) compute(1);
        __builtin_amdgcn_sched_barrier(0);
        op.lstore(0, min(tile(it + 2), last), pa);
        oq.lstore(0, min(tile(it + 2), last), qa);
        __syncthreads();
    }
    if (KG > 1) {
        // sum the k-groups' partial tiles into group 0: [group-1][wave][register][lane] floats
        float* red = lds;
        constexpr int PER_WAVE = WI * WJ * 16 * 64;
        if (kg > 0) {
            float* dst = red + ((kg - 1) * 2 * WM + wave) * PER_WAVE + lane;
#pragma unroll
            for (int ti = 0; ti < WI; ++ti)
#pragma unroll
                for (int tj = 0; tj < WJ; ++tj)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[((ti * WJ + tj) * 16 + e) * 64] = acc[ti][tj][e];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int g = 1; g < KG; ++g) {
                const float* src = red + ((g - 1) * 2 * WM + wave) * PER_WAVE + lane;
#pragma unroll
                for (int ti = 0; ti < WI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < WJ; ++tj)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[ti][tj][e] += src[((ti * WJ + tj) * 16 + e) * 64];
            }
        }
    }
}

// ---- the same contraction with the operand tiles written to LDS by the load itself ------------------------------
// For operands made of whole k-tiles of 16-byte aligned vectors (VEC, K % BK == 0).  Each global_load_lds_dwordx4
// writes one *piece* -- 1 KiB of LDS, lane l at byte 16*l, whatever address the lane loads from -- so the LDS image
// of a tile is linear and unpadded:
//   k-contiguous operand   [row][BK]: a piece is 256/BK rows.  Bank conflicts of the 16-byte fragment reads are
//                          avoided by an XOR swizzle of a row's 16-byte slots (slot = chunk ^ swz(row)), applied to
//                          the per-lane SOURCE address when a piece is loaded and to the LDS address when a fragment
//                          is read.
//   row-contiguous operand [k][NX]:   a piece is 256/NX k-rows; the four 4-byte fragment reads of a half-wave hit 32
//                          consecutive floats, conflict-free as they are.
// No staging registers and no ds_write: two or three LDS buffers of (BI+BJ)*BK floats per k-group, one barrier per k-tile, the
// load of a group's next tile issued part-way through the current tile's multiply.
template <int BK>
__device__ __forceinline__ int dma_swz(int row) {
    // 16-byte slots per 256-byte bank row: 16.  BK = 16: a row is 4 slots, the rows of one 16-lane read group that
    // share row % 4 have (row / 4) % 8 in {0,3,5,6} or {1,2,4,7}, which this maps to four distinct values.
    // BK = 32: a row is 8 slots and the rows of a group that share row % 2 have distinct (row / 2) % 8.
    if (BK == 16) {
        const int q = (row >> 2) & 7;
        return (q & 3) ^ ((q >> 2) * 3);
    }
    return (row >> 1) & 7;
}

template <int NX_, int BK, bool T>
struct DmaOperand {
    static constexpr int NX = NX_;
    static constexpr int PIECES = NX * BK / 256;
    static constexpr int SLOTS = BK / 4, PR = 64 / SLOTS;     // k-contiguous: 16-byte slots per row, rows per piece
    static constexpr int CPR = NX / 4, KPP = 64 / CPR;        // row-contiguous: vectors per k-row, k-rows per piece
    static_assert(PIECES >= 1 && (T ? (CPR <= 64 && BK % KPP == 0) : (NX % PR == 0)), "tile does not cut into pieces");
    // where this lane reads from for piece `pl` of k-tile `kt`
    __device__ static const float* source(const float* base, int64_t ld, int n, int x0, int kt, int pl, int lane) {
        if (!T) {
            const int row = pl * PR + lane / SLOTS;
            const int chunk = (lane % SLOTS) ^ dma_swz<BK>(row);
            return base + (int64_t)min(x0 + row, n - 1) * ld + (int64_t)kt * BK + chunk * 4;
        }
        const int kr = pl * KPP + lane / CPR, chunk = lane % CPR;
        return base + ((int64_t)kt * BK + kr) * ld + min(x0 + 4 * chunk, n - 4);
    }
    __device__ static int64_t step(int64_t ld) { return T ? (int64_t)BK * ld : (int64_t)BK; }
    // the four k values {kc*8 + 4h + m} of row (w0 + r) for the lane (r, h)
    __device__ static f32x4 frag(const float* img, int w0, int kc, int r, int h) {
        if (!T) return *reinterpret_cast<const f32x4*>(img + (w0 + r) * BK + (((kc * 2 + h) ^ dma_swz<BK>(w0 + r)) * 4));
        const float* q = img + (kc * 8 + 4 * h) * NX + w0 + r;
        return f32x4{q[0], q[NX], q[2 * NX], q[3 * NX]};
    }
};

// k-groups > 1 mark the tiles picked for launches too small to fill the chip: those get the deeper ring and issue
// their loads first
template <int KG>
constexpr int dma_nbuf() { return KG > 1 ? 3 : 2; }
template <int BK, int KG>
constexpr int dma_cut() { return KG > 1 ? 0 : BK / 16; }

template <int WM, int WI, int WJ, int BK, int KG>
constexpr int dma_lds_floats() { return KG * dma_nbuf<KG>() * (32 * WM * WI + 64 * WJ) * BK; }

template <int WM, int WI, int WJ, int BK, int CUT, int NBUF = 2, bool TP = false, bool TQ = false, int KG = 1>
__device__ __forceinline__ void mainloop_dma(const float* __restrict__ P, const float* __restrict__ Q, int64_t ldp,
                                             int64_t ldq, int I, int J, int i0, int j0, int kt0, int kt1, float* lds,
                                             f32x16 (&acc)[WI][WJ]) {
    static_assert(BK == 16 || BK == 32, "k-tile depths built: 16 and 32");
    constexpr int GT = 128 * WM, WAVES = 2 * WM;
    constexpr int BI = 32 * WM * WI, BJ = 64 * WJ;
    using OpP = DmaOperand<BI, BK, TP>;
    using OpQ = DmaOperand<BJ, BK, TQ>;
    constexpr int NP = OpP::PIECES + OpQ::PIECES, NPW = NP / WAVES;
    constexpr int BUF = (BI + BJ) * BK, QOFF = BI * BK;
    static_assert(NP % WAVES == 0, "pieces do not divide over the wavefronts");
    const int kg = KG > 1 ? (int)(threadIdx.x / GT) : 0;
    const int tid = threadIdx.x % GT;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = WM == 2 ? wave >> 1 : 0, wj = wave & 1, r = lane & 31, h = lane >> 5;
    float* glds = lds + kg * NBUF * BUF;
#pragma unroll
    for (int ti = 0; ti < WI; ++ti)
#pragma unroll
        for (int tj = 0; tj < WJ; ++tj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ti][tj][e] = 0.f;
    if (kt0 >= kt1) return;

    // this lane's source address inside each of its wavefront's pieces; a k-group walks the tiles kt0+kg, kt0+kg+KG, ...
    const float* src[NPW];
    int dst[NPW];
    int64_t stp[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int piece = i * WAVES + wave;                       // wave-uniform
        if (piece < OpP::PIECES) {
            src[i] = OpP::source(P, ldp, I, i0, kt0 + kg, piece, lane);
            dst[i] = piece * 256;
            stp[i] = OpP::step(ldp) * KG;
        } else {
            src[i] = OpQ::source(Q, ldq, J, j0, kt0 + kg, piece - OpP::PIECES, lane);
            dst[i] = QOFF + (piece - OpP::PIECES) * 256;
            stp[i] = OpQ::step(ldq) * KG;
        }
    }
    auto issue = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                             (__attribute__((address_space(3))) void*)(glds + buf * BUF + dst[i]), 16, 0, 0);
            src[i] += stp[i];
        }
    };
    auto compute = [&](int buf, int kc_lo, int kc_hi) {
        const float* b0 = glds + buf * BUF;
#pragma unroll
        for (int kc = kc_lo; kc < kc_hi; ++kc) {
            f32x4 a[WI], b[WJ];
#pragma unroll
            for (int ti = 0; ti < WI; ++ti) a[ti] = OpP::frag(b0, wi * 32 * WI + ti * 32, kc, r, h);
#pragma unroll
            for (int tj = 0; tj < WJ; ++tj) b[tj] = OpQ::frag(b0 + QOFF, wj * 32 * WJ + tj * 32, kc, r, h);
            // lane half h feeds k = kc*8 + 4h + m, so MFMA m sums k in {kc*8+m, kc*8+4+m}
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int ti = 0; ti < WI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < WJ; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti][m], b[tj][m], acc[ti][tj], 0, 0, 0);
        }
    };
    const int nit = (kt1 - kt0 + KG - 1) / KG;    // the same trip count in every k-group (shared barriers)
    auto live = [&](int it) { return KG == 1 ? it < nit : kt0 + kg + it * KG < kt1; };
    // CUT = the k-chunks (of 8) multiplied before the loads of a later tile are issued.  Issuing them first costs the
    // wavefront's own MFMA stream the most (the loads and the first fragment reads queue together); issuing them
    // last exposes their whole latency to this wavefront, which only pays when enough other wavefronts share the SIMD.
    // NBUF = LDS buffers per k-group: NBUF - 1 tiles are in flight or being read at any time.  Three buffers let a
    // workgroup that is alone on its CU keep two tiles' worth of loads outstanding across each barrier.
    constexpr int NKC = BK / 8;
    static_assert(CUT >= 0 && CUT <= NKC && (NBUF == 2 || NBUF == 3), "load issue point / ring depth not built");
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j)
        if (live(j)) issue(j);
    for (int it0 = 0; it0 < nit; it0 += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            const int it = it0 + j;
            if (it >= nit) break;
            // Every piece of tile `it` has landed (the loads of a later tile may stay in flight: they retire in
            // order) and every wavefront is done reading the buffer that the next loads overwrite.
            if (NBUF == 3 && live(it + 1)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (CUT > 0 && live(it)) compute(j, 0, CUT);
            __builtin_amdgcn_sched_barrier(0);
            if (live(it + NBUF - 1)) issue((j + NBUF - 1) % NBUF);
            __builtin_amdgcn_sched_barrier(0);
            if (CUT < NKC && live(it)) compute(j, CUT, NKC);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();   // past here the caller (or the k-group sum below) may reuse `lds`
    if (KG > 1) {
        // sum the k-groups' partial tiles into group 0: [group-1][wave][register][lane] floats
        float* red = lds;
        constexpr int PER_WAVE = WI * WJ * 16 * 64;
        if (kg > 0) {
            float* d = red + ((kg - 1) * 2 * WM + wave) * PER_WAVE + lane;
#pragma unroll
            for (int ti = 0; ti < WI; ++ti)
#pragma unroll
                for (int tj = 0; tj < WJ; ++tj)
#pragma unroll
                    for (int e = 0; e < 16; ++e) d[((ti * WJ + tj) * 16 + e) * 64] = acc[ti][tj][e];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int g = 1; g < KG; ++g) {
                const float* sp = red + ((g - 1) * 2 * WM + wave) * PER_WAVE + lane;
#pragma unroll
                for (int ti = 0; ti < WI; ++ti)
#pragma unroll
                    for (int tj = 0; tj < WJ; ++tj)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[ti][tj][e] += sp[((ti * WJ + tj) * 16 + e) * 64];
            }
        }
    }
}

// accumulator register e of lane-half h is row (e&3) + 8*(e>>2) + 4*h of the 32x32 tile
__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

template <int WM, int WI, int WJ, int BK, bool VEC, bool TP, bool TQ, int KG, bool DMA = false>
__global__ __launch_bounds__(128 * WM * KG, (WI * WJ >= 4 && BK == 16 && TP) ? 3 : 1) void gemm_kernel(GemmKArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int z, tile_i, tile_j;
    if (!tile_of_block(args.map, blockIdx.x, z, tile_i, tile_j)) return;
    const int grp = z / args.nsplit, split = z - grp * args.nsplit;
    const capdec_gemm_group& g = args.g[grp];
    constexpr int BI = 32 * WM * WI, BJ = 64 * WJ;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    // rows_dev: the number of live rows is only known on the device (decode steps that re-run a few rows)
    const int nI = args.x.rows_dev ? min(g.I, args.x.rows_dev[0]) : g.I;
    if (i0 >= nI || j0 >= g.J) return;   // groups of different size share the grid
    // k_dev: the contraction length is only known on the device (rows compacted by a kernel); whole k-tiles are
    // read, so the operands hold zeros (or anything finite against zeros) up to the next multiple of BK
    const int K = args.x.k_dev ? min(args.K, args.x.k_dev[0]) : args.K;
    const int nk = (K + BK - 1) / BK;
    const int per = (nk + args.nsplit - 1) / args.nsplit;
    const int kt0 = split * per, kt1 = min(nk, kt0 + per);
    f32x16 acc[WI][WJ];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = WM == 2 ? (wave >> 1) & 1 : 0, wj = wave & 1, r = lane & 31, h = lane >> 5;
    const bool relu = (args.flags & CAPDEC_GEMM_RELU) != 0;
    const bool atomic = (args.flags & CAPDEC_GEMM_ATOMIC) != 0;
    // One-tile wavefronts fetch their residual values before the contraction, so that the loads' latency is not paid
    // after it (short contractions: the decode step's projections).  Each element is read by the thread that stores
    // it, so R may be C itself.
    constexpr bool RPRE = (WI * WJ == 1);
    float rpre[RPRE ? 16 : 1];
    int cpre[RPRE ? 16 : 1];
    if (RPRE && threadIdx.x < 128 * WM) {
        const int j = j0 + wj * 32 + r;
        if (g.R && !atomic) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + wi * 32 + acc_row(e, h);
                rpre[e] = (i < nI && j < g.J) ? g.R[(int64_t)i * g.ldri + (int64_t)j * g.ldrj] : 0.f;
            }
        }
        if (g.c_rows) {      // the same for the row map of scattered outputs
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + wi * 32 + acc_row(e, h);
                cpre[e] = (i < nI) ? g.c_rows[i] : 0;
            }
        }
    }
    if (DMA) mainloop_dma<WM, WI, WJ, BK, dma_cut<BK, KG>(), dma_nbuf<KG>(), TP, TQ, KG>(g.P, g.Q, g.ldp, g.ldq, nI, g.J, i0, j0, kt0, kt1, lds, acc);
    else mainloop<WM, WI, WJ, BK, VEC, TP, TQ, KG>(g.P, g.Q, g.ldp, g.ldq, nI, g.J, K, i0, j0, kt0, kt1, lds, acc);
    if (KG > 1 && threadIdx.x >= 128 * WM) return;   // group 0 holds the summed tile

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
            // Large tiles make two passes over the 16 registers: every load of the epilogue first, then the stores.
            // Interleaved, each load has to wait for the store before it (the compiler cannot know that C and the
            // epilogue's operands do not overlap), one memory round trip per register.  One-tile wavefronts keep the
            // interleaved form, which needs fewer registers (their occupancy matters more), with the residual and the
            // row map already in registers.
            if constexpr (WI * WJ < 4) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = i0 + wi * 32 * WI + ti * 32 + acc_row(e, h);
                    if (i >= nI) continue;
                    // c_rows: output row i is stored at row c_rows[i] (k/v of a decode step go straight to their cache rows)
                    const int64_t ci = g.c_rows ? (int64_t)(RPRE ? cpre[RPRE ? e : 0] : g.c_rows[i]) : (int64_t)i;
                    float* c = &g.C[ci * g.ldci + (int64_t)j * g.ldcj];
                    if (atomic && x.splitk_ws) {
                        // fixed-order split-K: the raw partial sum of this k-range goes to its own slab
                        x.splitk_ws[((int64_t)split * g.I + i) * g.J + j] = acc[ti][tj][e];
                        continue;
                    }
                    if (atomic) {
                        // C += partial; the affine terms (bias, BatchNorm shift) enter once, with the first k-range
                        float v = acc[ti][tj][e];
                        if (split == 0) {
                            v += bj;
                            if (g.bias_i) v += g.bias_i[i];
                        }
                        if (g.scale_j) v = v * sc + (split == 0 ? sh : 0.f);
                        atomicAdd(c, v);
                        continue;
                    }
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
                    if (g.R) v += RPRE ? rpre[RPRE ? e : 0] : g.R[(int64_t)i * g.ldri + (int64_t)j * g.ldrj];
                    *c = v;
                }
            } else {
                float val[16];
                int64_t off[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = i0 + wi * 32 * WI + ti * 32 + acc_row(e, h);
                    off[e] = -1;
                    if (i >= nI) continue;
                    const int64_t ci = g.c_rows ? (int64_t)g.c_rows[i] : (int64_t)i;
                    off[e] = ci * g.ldci + (int64_t)j * g.ldcj;
                    float v;
                    if (atomic && x.splitk_ws) {
                        off[e] = ((int64_t)split * g.I + i) * g.J + j;
                        v = acc[ti][tj][e];
                    } else if (atomic) {
                        v = acc[ti][tj][e];
                        if (split == 0) {
                            v += bj;
                            if (g.bias_i) v += g.bias_i[i];
                        }
                        if (g.scale_j) v = v * sc + (split == 0 ? sh : 0.f);
                    } else {
                        v = acc[ti][tj][e] + bj;
                        if (g.bias_i) v += g.bias_i[i];
                        if (g.scale_j) v = v * sc + sh;
                        if (relu) v = fmaxf(v, 0.f);
                        if (x.lse) {
                            const int64_t tg = x.tgt[i];
                            v = (tg == x.ignore_index)
                                    ? 0.f
                                    : (expf(v - x.lse[i]) - (tg == (int64_t)j ? 1.f : 0.f)) * x.gscale[0];
                        }
                        if (x.mask && !(x.mask[(int64_t)i * x.ldm + j] > 0.f)) v = 0.f;
                        if (g.R) v += g.R[(int64_t)i * g.ldri + (int64_t)j * g.ldrj];
                    }
                    val[e] = v;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (off[e] < 0) continue;
                    if (atomic && x.splitk_ws) x.splitk_ws[off[e]] = val[e];
                    else if (atomic) atomicAdd(&g.C[off[e]], val[e]);
                    else g.C[off[e]] = val[e];
                }
            }
        }
    }
}

// ---- per-sample contractions over a ragged batch ----------------------------------------------------------------
// Sample b owns n_b consecutive rows of the "ragged" operands (n_b from seg_off, or a fixed count) and one slab of
// the "dense" operand.  A workgroup computes one 64x64 tile of one sample's product with the same k-loop as above.
//   FORM 0: C_b[n_b,J] = scale * P_b[n_b,K] Q_b[J,K]^T      P, C ragged; Q dense
//   FORM 1: C_b[n_b,J] = scale * P_b[n_b,K] Q_b[K,J]        P, C ragged; Q dense; P's rows hold zeros from column K
//                                                           up to the next multiple of four
//   FORM 2: C_b[I,J] (+)= scale * (P_b[n_b,I]^T Q_b[n_b,J] + P2_b^T Q2_b)     P, Q ragged; C dense; P's rows hold
//                                                           zeros from column I up to the next multiple of four
struct BgemmKArgs {
    capdec_bgemm_desc a;
    int ni, nj, per_xcd;
};

// WM = 2: 64x64 tiles (256 threads); WM = 1: 32x64 tiles (128 threads), for the forms whose row count is the ragged
// one (at most H*T = 80 rows per sample: three 32-row tiles instead of two 64-row ones, a quarter fewer MFMAs).
template <int FORM, int WM>
__global__ __launch_bounds__(128 * WM) void bgemm_kernel(BgemmKArgs k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const capdec_bgemm_desc& a = k.a;
    constexpr int BK = 16;
    // the tiles of a sample read the same dense slab: they run on one XCD, back to back (workgroup L belongs to XCD
    // L & 7); a performance assumption only
    const unsigned L = blockIdx.x;
    const int logical = (int)(L & 7) * k.per_xcd + (int)(L >> 3);
    const int per = k.ni * k.nj;
    const int b = logical / per;
    if (b >= a.B) return;
    const int t = logical - b * per, ti = t / k.nj, tj = t - ti * k.nj;
    int r0, n;
    if (a.seg_off) {
        r0 = a.seg_off[b] * a.mult;
        n = min(a.seg_off[b + 1] * a.mult - r0, a.seg * a.mult);
    } else {
        r0 = b * a.seg * a.mult;
        n = a.seg * a.mult;
    }
    const int i0 = ti * 32 * WM, j0 = tj * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = WM == 2 ? wave >> 1 : 0, wj = wave & 1, r = lane & 31, h = lane >> 5;
    f32x16 acc[1][1];
    const int j = j0 + wj * 32 + r;
    if (FORM == 0 || FORM == 1) {
        if (i0 >= n) return;
        const float* P = a.P + (int64_t)r0 * a.ldp;
        const float* Q = a.Q + (int64_t)b * a.dense_stride;
        if (FORM == 0) {
            mainloop<WM, 1, 1, BK, true, false, false, 1>(P, Q, a.ldp, a.ldq, n, a.J, a.K, i0, j0, 0, (a.K + BK - 1) / BK, lds,
                                                         acc);
        } else {
            const int KP = (a.K + 3) & ~3;
            mainloop<WM, 1, 1, BK, true, false, true, 1>(P, Q, a.ldp, a.ldq, n, a.J, KP, i0, j0, 0, (KP + BK - 1) / BK, lds, acc,
                                                        false, a.K);
        }
        if (j >= a.J) return;
        float* C = a.C + (int64_t)r0 * a.ldc + j;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wi * 32 + acc_row(e, h);
            if (i < n) C[(int64_t)i * a.ldc] = acc[0][0][e] * a.scale;
        }
    } else {
        if (n <= 0 && a.accumulate) return;
        const int nI = (a.I + 3) & ~3;
        const int nk = (max(n, 0) + BK - 1) / BK;
        mainloop<WM, 1, 1, BK, true, true, true, 1>(a.P + (int64_t)r0 * a.ldp, a.Q + (int64_t)r0 * a.ldq, a.ldp, a.ldq, nI, a.J,
                                                   max(n, 1), i0, j0, 0, nk, lds, acc);
        if (a.P2)
            mainloop<WM, 1, 1, BK, true, true, true, 1>(a.P2 + (int64_t)r0 * a.ldp, a.Q2 + (int64_t)r0 * a.ldq, a.ldp, a.ldq, nI,
                                                       a.J, max(n, 1), i0, j0, 0, nk, lds, acc, true);
        if (j >= a.J) return;
        float* C = a.C + (int64_t)b * a.dense_stride + j;
        float old[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wi * 32 + acc_row(e, h);
            old[e] = (a.accumulate && i < a.I) ? C[(int64_t)i * a.ldc] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wi * 32 + acc_row(e, h);
            if (i < a.I) C[(int64_t)i * a.ldc] = old[e] + acc[0][0][e] * a.scale;
        }
    }
}

// Second stage of the fixed-order split-K: C[i,j] = epi(ws[0][i][j] + ws[1][i][j] + ... ) with the slabs added in
// index order by one thread per element, so the sum does not depend on which workgroup finished first.
// [k_tail0, k_tail1): the last few k of a contraction whose length is not a whole number of k-tiles (both operands
// k-contiguous), left out of the slabs so that the first stage reads whole tiles only; added here, last.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int nsplit, capdec_gemm_group g,
                                                            int relu, int k_tail0, int k_tail1) {
    const int64_t n = (int64_t)g.I * g.J;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / g.J), j = (int)(idx - (int64_t)i * g.J);
        float v = ws[idx];
        for (int sp = 1; sp < nsplit; ++sp) v += ws[(int64_t)sp * n + idx];
        for (int k = k_tail0; k < k_tail1; ++k) v += g.P[(int64_t)i * g.ldp + k] * g.Q[(int64_t)j * g.ldq + k];
        if (g.bias_j) v += g.bias_j[j];
        if (g.bias_i) v += g.bias_i[i];
        if (g.scale_j) v = v * g.scale_j[j] + g.shift_j[j];
        if (relu) v = fmaxf(v, 0.f);
        if (g.R) v += g.R[(int64_t)i * g.ldri + (int64_t)j * g.ldrj];
        g.C[(int64_t)i * g.ldci + (int64_t)j * g.ldcj] = v;
    }
}

// ---- vocabulary projection with lane-local reductions over the vocabulary axis -------------------
// P = W [V,K] (i = vocabulary id), Q = x [M,K] (j = row).  MODE 0: arg-max; MODE 1: (max, sum exp).
template <int WI, int WJ, int BK, bool VEC, int MODE, bool DMA = false>
__global__ __launch_bounds__(256, 5) void vocab_reduce_kernel(const float* __restrict__ W, const float* __restrict__ x,
                                                           int64_t ldx, const float* __restrict__ bias, int V, int M,
                                                           int K, float* __restrict__ out_a, void* __restrict__ out_b,
                                                           int ntiles, TileMap map,
                                                           const int32_t* __restrict__ rows_dev) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BI = 64 * WI, BJ = 64 * WJ;
    int z, tile_i, tile_j;
    if (!tile_of_block(map, blockIdx.x, z, tile_i, tile_j)) return;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    // rows_dev: only the first rows_dev[0] rows are wanted (a decode step after every caption has finished: none);
    // the partials of the other rows are left as they are
    if (rows_dev && j0 >= rows_dev[0]) return;
    f32x16 acc[WI][WJ];
    if (DMA) mainloop_dma<2, WI, WJ, BK, BK / 8>(W, x, (int64_t)K, ldx, V, M, i0, j0, 0, K / BK, lds, acc);
    else mainloop<2, WI, WJ, BK, VEC>(W, x, (int64_t)K, ldx, V, M, K, i0, j0, 0, (K + BK - 1) / BK, lds, acc);

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
        const int64_t o = capdec_part_off(j0 + t, tile_i, ntiles);
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

// ---- vocabulary arg-max for small batches: blocks of 32 rows, at most eight (the reference's loaders default to
// batches of 32 and 64) -------------------------------------------------------------------------------------------
// The tiling above gives such a batch V/64 workgroups whose wavefronts each walk the whole contraction for a 64-row
// tile that is half empty: 18.6 us at 32 x 5000 x 512, a tenth of it arithmetic.  Here a workgroup of eight wavefronts
// owns one 64-id vocabulary tile (the same tile numbering, so the partial-result layout is unchanged): two wavefronts
// side by side on the vocabulary axis, four deep on the contraction.  Nothing is shared between wavefronts, so there is
// no workgroup barrier in the loop: a wavefront fetches its own 32 x 32 blocks of W and x with whole-line loads (eight
// lanes per 128-byte row segment), turns them into MFMA fragments through a private LDS region (rows padded to 36
// floats: the 16-byte fragment reads of 16 lanes fall on distinct banks), and the four partial tiles of a vocabulary
// half meet in LDS at the end, added in a fixed order.  (Fragments read straight from memory - 16 bytes per lane, 32
// rows per load instruction - ran at 13.3 us: the lines eight wavefronts hold open overflow the 32 KiB vector cache.)
// Vocabulary ids on the accumulator registers, rows on the lanes, as above.  K must be a multiple of 32.
constexpr int VS_LD = 36;                                   // floats per staged row
constexpr int VS_MAX_ROWS = 256;                            // larger batches take the general tiling
constexpr int VS_WAVE_FLOATS = 2 * 32 * VS_LD;              // W block + x block of one wavefront
constexpr size_t VS_LDS_BYTES = (size_t)8 * VS_WAVE_FLOATS * sizeof(float);

__global__ __launch_bounds__(512) void vocab_argmax_small_kernel(const float* __restrict__ W, const float* __restrict__ x,
                                                                 int64_t ldx, const float* __restrict__ bias, int V,
                                                                 int M, int K, float* __restrict__ out_val,
                                                                 int* __restrict__ out_idx, int ntiles,
                                                                 const int32_t* __restrict__ rows_dev) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float sm_v[2][32];
    __shared__ int sm_i[2][32];
    const int j0 = 32 * blockIdx.y;                          // first row of this workgroup's block of 32
    if (rows_dev && j0 >= rows_dev[0]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wv = wave & 1, wk = wave >> 1;
    const int tile = blockIdx.x, i0 = tile * 64 + 32 * wv;
    const int nb = K / 32;                                   // 32-float blocks of the contraction
    const int per = (nb + 3) / 4;
    const int b0 = wk * per, b1 = min(nb, b0 + per);
    float* sw = lds + wave * VS_WAVE_FLOATS;                 // [32][VS_LD] block of W
    float* sx = sw + 32 * VS_LD;                             // [32][VS_LD] block of x
    // loads: instruction i covers rows 8i .. 8i+7, eight lanes per row
    const int lrow = lane >> 3, lslot = lane & 7;
    const float* wg[4];
    const float* xg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wg[i] = W + (int64_t)min(i0 + 8 * i + lrow, V - 1) * K + 4 * lslot;    // ids past the vocabulary repeat the last one
        xg[i] = x + (int64_t)min(j0 + 8 * i + lrow, M - 1) * ldx + 4 * lslot;  // rows past the batch likewise: never stored
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 wr[2][4], xr[2][4];
    auto fetch = [&](int set, int b) {
        if (b < b1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wr[set][i] = *reinterpret_cast<const f32x4*>(wg[i] + 32 * b);
                xr[set][i] = *reinterpret_cast<const f32x4*>(xg[i] + 32 * b);
            }
        }
    };
    auto consume = [&](int set, int b) {
        if (b >= b1) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(sw + (8 * i + lrow) * VS_LD + 4 * lslot) = wr[set][i];
            *reinterpret_cast<f32x4*>(sx + (8 * i + lrow) * VS_LD + 4 * lslot) = xr[set][i];
        }
        // LDS serves one wavefront's accesses in order: the block written above is what the reads below see
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sw + r * VS_LD + 8 * c + 4 * h);
            const f32x4 bq = *reinterpret_cast<const f32x4*>(sx + r * VS_LD + 8 * c + 4 * h);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bq[m], acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };
    // the loads of the next block are in flight while the current one is multiplied
    fetch(0, b0);
    for (int b = b0; b < b1; b += 2) {
        fetch(1, b + 1);
        consume(0, b);
        fetch(0, b + 2);
        consume(1, b + 1);
    }
    // partial tiles: each wavefront's own staging region holds its 16 x 64 values
    float* red = lds + wave * VS_WAVE_FLOATS;
#pragma unroll
    for (int e = 0; e < 16; ++e) red[e * 64 + lane] = acc[e];
    __syncthreads();
    if (wk == 0) {
        // this wavefront's vocabulary half: the four k-partials in a fixed order, bias, lane-local winner (registers walk
        // the ids in increasing order for a fixed lane half, so a strict compare keeps the lowest id on ties)
        const float* p0 = lds + (size_t)(0 + wv) * VS_WAVE_FLOATS;
        const float* p1 = lds + (size_t)(2 + wv) * VS_WAVE_FLOATS;
        const float* p2 = lds + (size_t)(4 + wv) * VS_WAVE_FLOATS;
        const float* p3 = lds + (size_t)(6 + wv) * VS_WAVE_FLOATS;
        float best = -INFINITY;
        int bidx = INT_MAX;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + acc_row(e, h);
            const int o = e * 64 + lane;
            const float v = ((p0[o] + p1[o]) + (p2[o] + p3[o])) + (i < V ? bias[i] : 0.f);
            if (i < V && v > best) { best = v; bidx = i; }
        }
        const float v2 = __shfl_xor(best, 32, 64);
        const int i2 = __shfl_xor(bidx, 32, 64);
        argmax_merge(best, bidx, v2, i2);
        if (h == 0) {
            sm_v[wv][r] = best;
            sm_i[wv][r] = bidx;
        }
    }
    __syncthreads();
    if (tid < 32 && j0 + tid < M) {
        float a0 = sm_v[0][tid];
        int bb = sm_i[0][tid];
        argmax_merge(a0, bb, sm_v[1][tid], sm_i[1][tid]);
        const int64_t o = capdec_part_off(j0 + tid, tile, ntiles);
        out_val[o] = a0;
        out_idx[o] = bb;
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
    for (int t = lane; t < ntiles; t += 64) argmax_merge(v, i, pv[capdec_part_off(row, t, ntiles)], pi[capdec_part_off(row, t, ntiles)]);
    wave_argmax(v, i);
    if (lane == 0) {
        out_idx[row] = i;
        if (out_val) out_val[row] = v;
    }
}

// one wavefront per row, NLL_ROWS rows per workgroup: lse from per-tile (max, sum), target logit by a K-long dot
// product; the loss sum and the count leave the workgroup as ONE pair of atomics (same-address atomics serialise)
constexpr int NLL_ROWS = 32;

__global__ __launch_bounds__(256) void nll_finalize_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                           int M, int ntiles, const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           int K, const int64_t* __restrict__ targets,
                                                           int64_t ignore_index, float* __restrict__ lse_out,
                                                           float* __restrict__ nll_out, float* __restrict__ loss_sum,
                                                           int V) {
    __shared__ float red[2][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc_nll = 0.f, acc_cnt = 0.f;
    for (int rr = wave; rr < NLL_ROWS; rr += 4) {
        const int row = blockIdx.x * NLL_ROWS + rr;
        if (row >= M) break;
        float mx = -INFINITY;
        for (int t = lane; t < ntiles; t += 64) mx = fmaxf(mx, pm[capdec_part_off(row, t, ntiles)]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int t = lane; t < ntiles; t += 64) {
            const float m = pm[capdec_part_off(row, t, ntiles)];
            if (m > -INFINITY) s += ps[capdec_part_off(row, t, ntiles)] * expf(m - mx);
        }
        s = wave_sum(s);
        const float lse = mx + logf(s);
        const int64_t tg = targets[row];
        float nll = 0.f;
        if (tg != ignore_index) {
            float d = 0.f;
            // (a counted target outside the vocabulary is a caller error: its loss term is NaN, nothing is read there)
            const bool in_range = tg >= 0 && tg < (int64_t)V;
            if (in_range)
                for (int k = lane; k < K; k += 64) d += x[(int64_t)row * ldx + k] * W[tg * (int64_t)K + k];
            d = in_range ? wave_sum(d) + bias[tg] : NAN;
            nll = lse - d;
            acc_nll += nll;
            acc_cnt += 1.f;
        }
        if (lane == 0) {
            if (lse_out) lse_out[row] = lse;
            if (nll_out) nll_out[row] = nll;
        }
    }
    if (lane == 0) {
        red[0][wave] = acc_nll;
        red[1][wave] = acc_cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0 && loss_sum) {
        const float c = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (c > 0.f) {
            atomicAdd(&loss_sum[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
            atomicAdd(&loss_sum[1], c);
        }
    }
}

// 0: automatic; 1: operand tiles always staged through registers
CAPDEC_HOOK_VAR(g_staging);

// workgroups above 64 KiB of dynamic LDS need the limit raised once per kernel
template <typename F>
bool allow_lds(F* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes) == hipSuccess;
}

// split the contraction until about four workgroups per CU exist, at least 4 k-tiles per split, at most 64 splits
inline int splitk_count(long tiles, int nk) {
    long want = (1024 + tiles - 1) / tiles;
    const long cap = nk / 4 > 0 ? nk / 4 : 1;
    int n = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    return n > 64 ? 64 : n;
}
// tile form and split count of a split contraction of one I x J x K problem (the rule of launch_gemm_layout)
inline int splitk_plan(int I, int J, int K) {
    const long tiles64 = (long)((I + 63) / 64) * ((J + 63) / 64);
    if (tiles64 >= 1024) return splitk_count((long)((I + 127) / 128) * ((J + 127) / 128), (K + 15) / 16);
    return splitk_count(tiles64, (K + 31) / 32);
}

template <int WM, int WI, int WJ, int BK, bool TP, bool TQ, int KG>
int launch_gemm_cfg(GemmKArgs& a, int ngroups, int maxI, int maxJ, bool vec, hipStream_t s) {
    constexpr int BI = 32 * WM * WI, BJ = 64 * WJ, NT = 128 * WM * KG;
    const int ti = (maxI + BI - 1) / BI, tj = (maxJ + BJ - 1) / BJ;
    const int nk = (a.K + BK - 1) / BK;
    a.nsplit = 1;
    if (a.flags & CAPDEC_GEMM_ATOMIC) a.nsplit = splitk_count((long)ti * tj * ngroups, nk);
    if (a.x.splitk_ws) {
        // fixed-order split-K: one slab of partial sums per k-range, no more of them than the workspace holds
        const long fit = (long)(a.x.splitk_ws_elems / ((int64_t)maxI * maxJ));
        if (a.nsplit > fit) a.nsplit = fit < 1 ? 1 : (int)fit;
    }
    dim3 grid(make_tile_map(a.map, ti, tj, ngroups * a.nsplit, a.nsplit >= 8), 1, 1);
    constexpr size_t smem = lds_floats<WM, WI, WJ, BK, TP, TQ, KG>() * sizeof(float);
    static const bool ready = allow_lds(gemm_kernel<WM, WI, WJ, BK, true, TP, TQ, KG>, smem) &&
                              allow_lds(gemm_kernel<WM, WI, WJ, BK, false, TP, TQ, KG>, smem);
    if (!ready) return CAPDEC_EINVAL;
    if (vec && a.K % BK == 0 && g_staging != 1) {
        // whole k-tiles of aligned vectors: loaded straight into LDS
        constexpr size_t smem_dma = dma_lds_floats<WM, WI, WJ, BK, KG>() * sizeof(float);
        static const bool ready_dma = allow_lds(gemm_kernel<WM, WI, WJ, BK, true, TP, TQ, KG, true>, smem_dma);
        if (!ready_dma) return CAPDEC_EINVAL;
        hipLaunchKernelGGL((gemm_kernel<WM, WI, WJ, BK, true, TP, TQ, KG, true>), grid, dim3(NT), smem_dma, s, a);
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    if (vec) hipLaunchKernelGGL((gemm_kernel<WM, WI, WJ, BK, true, TP, TQ, KG>), grid, dim3(NT), smem, s, a);
    else hipLaunchKernelGGL((gemm_kernel<WM, WI, WJ, BK, false, TP, TQ, KG>), grid, dim3(NT), smem, s, a);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

// Tile choice.  cfg = 0: automatic; otherwise force: 11 / 116 = 64x64 (BK 32 / 16), 22 / 216 = 128x128 (BK 32 / 16),
// 32 / 316 = 32x64 tile with the k-loop split over the workgroup's two wave pairs (BK 32 / 16), 2132 / 2116 = 128x64
// (64x32 per wavefront; BK 32 / 16).
CAPDEC_HOOK_VAR(g_force_cfg);

template <bool TP, bool TQ>
int launch_gemm_layout(GemmKArgs& a, int ngroups, int maxI, int maxJ, bool vec, hipStream_t s) {
    int cfg = g_force_cfg;
    if (cfg == 0) {
        // Measured on MI355X at the decoder's shapes (dependent chains of 512-deep contractions).  From about four
        // 64x64 tiles per CU on, those tiles at BK 16 win (20 KiB of LDS: every tile of a 1280-tile grid is resident at
        // once and nothing runs as a lone tail); between two and four per CU the deeper k-tile does.  Below that the
        // chip is not full: a 32x64 tile whose k-loop is split over two wave pairs spreads the MFMA work over twice as
        // many SIMDs, at BK 32 when every workgroup has a CU to itself, at BK 16 (37 KiB of LDS, four workgroups per
        // CU) when some CUs hold several.  Split-K weight gradients keep BK 32 (their k loop is long), and 128x128
        // tiles pay off only for very large outputs.
        const long tiles64 = (long)((maxI + 63) / 64) * ((maxJ + 63) / 64) * ngroups;
        const bool atomic = (a.flags & CAPDEC_GEMM_ATOMIC) != 0;
        // Contractions split over workgroups and summed with float atomics (weight gradients, the vocabulary input
        // gradient of a small batch), measured over 704 ... 7488 contracted rows: the 32x64 two-k-group tile at BK 16
        // beats the 64x64 one by 10-20 % on 512- and 1536-row outputs (12.8 vs 14.7 us at 704 rows, 44.9 vs 52.1 us
        // at 7488) and by 30 % on the 10000-deep input gradient (92 vs 132 us at 704 rows; BK 16 divides 10000, so
        // the tiles go straight into LDS).  The 10000 x 512 vocabulary weight gradient takes 128x128 tiles only from
        // about 6000 contracted rows on (673 vs 703 us at 7488): they need a four-way split to fill the chip, and at
        // 704 rows those four times 5 M atomics are the whole launch (149 vs 85 us).  The fixed-order split (partial
        // slabs in a workspace sized by splitk_plan) keeps the rule its workspace was sized for.
        if (atomic && a.x.splitk_ws) cfg = tiles64 >= 1024 ? 216 : 11;
        else if (atomic) cfg = (tiles64 >= 1024 && a.K >= 8192) ? 216 : 316;
        // (512 ... 1023 tiles, 512-deep, 7488 rows: forward 44.4 us on 128x64 tiles at BK 32 against 47.7 on 64x64 ones,
        //  input gradient 44.6 us on 64x64 tiles at BK 16 against 45.5 at BK 32; at 10240 rows the 64x64 tiles stay:
        //  tools/exp/tile_probe.py.  Giving a wavefront 64x32 instead of 32x32 does not lift the 53-56 % of the MFMA peak
        //  these 512-deep launches reach: their LDS reads are not what holds them.)
        // (the training logits, 7488 ... 10240 x 10000 x 512, some 18000 tiles: launched back to back on their own, BK 32
        //  takes 675 / 903 us against 693-715 / 918-924 at BK 16, 128x64 tiles 700 / 925, 128x128 ones 742-760 / 979-998
        //  (tools/exp/vocab_tile_probe.py); inside the training step the two k-tile depths give the same 5.44-5.46 ms,
        //  so the rule stays as it is)
        else cfg = tiles64 <= 160 ? 32 : (tiles64 < 512 ? 316 : (tiles64 < 1024 ? (TQ ? 116 : 2132) : 116));
    }
    if (cfg == 2116) return launch_gemm_cfg<2, 2, 1, 16, TP, TQ, 1>(a, ngroups, maxI, maxJ, vec, s);
    if (cfg == 2132) return launch_gemm_cfg<2, 2, 1, 32, TP, TQ, 1>(a, ngroups, maxI, maxJ, vec, s);
    if (cfg == 22) return launch_gemm_cfg<2, 2, 2, 32, TP, TQ, 1>(a, ngroups, maxI, maxJ, vec, s);
    if (cfg == 216) return launch_gemm_cfg<2, 2, 2, 16, TP, TQ, 1>(a, ngroups, maxI, maxJ, vec, s);
    if (cfg == 32) return launch_gemm_cfg<1, 1, 1, 32, TP, TQ, 2>(a, ngroups, maxI, maxJ, vec, s);
    if (cfg == 316) return launch_gemm_cfg<1, 1, 1, 16, TP, TQ, 2>(a, ngroups, maxI, maxJ, vec, s);
    if (cfg == 116) return launch_gemm_cfg<2, 1, 1, 16, TP, TQ, 1>(a, ngroups, maxI, maxJ, vec, s);
    return launch_gemm_cfg<2, 1, 1, 32, TP, TQ, 1>(a, ngroups, maxI, maxJ, vec, s);
}

constexpr int VOCAB_WI = 1;                 // 32-row accumulator blocks per wavefront along the vocabulary axis
constexpr int VOCAB_BI = 64 * VOCAB_WI;     // vocabulary ids per tile of the fused projections

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
        if (tp) vec = vec && g.I % 4 == 0;     // a 16-byte load along a row-contiguous operand must not straddle its end
        if (tq) vec = vec && g.J % 4 == 0;
    }
    if (!tp || !tq) vec = vec && (K % 4 == 0) && K >= 4;   // a k-contiguous operand is read four k at a time
    for (int n = ngroups; n < 4; ++n) a.g[n] = a.g[0];
    if (maxI == 0 || maxJ == 0) return 0;
    if (flags & CAPDEC_GEMM_SPLITK) {
        // fixed-order split-K: partial sums into the caller's workspace, then one pass that adds them in index order
        const capdec_gemm_group& g = groups[0];
        CAPDEC_REQUIRE(ngroups == 1 && !(flags & CAPDEC_GEMM_ATOMIC) && !g.c_rows && !a.x.mask && !a.x.lse && !a.x.rows_dev &&
                       !a.x.k_dev && a.x.splitk_ws);
        if ((size_t)a.x.splitk_ws_elems < (size_t)g.I * g.J) return CAPDEC_ENOMEM;      // room for one slab at least
        a.flags = (flags & ~(CAPDEC_GEMM_SPLITK | CAPDEC_GEMM_RELU)) | CAPDEC_GEMM_ATOMIC;
        // A long contraction that is not a whole number of 32-deep k-tiles (the YOLO feature is 86436 wide) would send
        // every tile through the register-staged loads for the sake of its last one (930 against 800 us at
        // 1024 x 512 x 86436): the first stage takes the whole tiles, the second adds the remaining k.
        int k_main = K;
        if (!tp && !tq && vec && K % 32 != 0 && K >= 4096) k_main = K / 32 * 32;
        a.K = k_main;
        int rc;
        if (tp) rc = launch_gemm_layout<true, true>(a, 1, maxI, maxJ, vec, s);
        else if (tq) rc = launch_gemm_layout<false, true>(a, 1, maxI, maxJ, vec, s);
        else rc = launch_gemm_layout<false, false>(a, 1, maxI, maxJ, vec, s);
        if (rc != 0) return rc;
        const int nsplit = a.nsplit;                       // as many k-ranges as the launcher chose and the workspace holds
        const int64_t n = (int64_t)g.I * g.J;
        const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, a.x.splitk_ws, nsplit, g,
                           (flags & CAPDEC_GEMM_RELU) ? 1 : 0, k_main, K);
        CAPDEC_LAUNCH_CHECK();
        return 0;
    }
    a.x.splitk_ws = nullptr;
    if ((flags & CAPDEC_GEMM_ATOMIC) && ((flags & CAPDEC_GEMM_RELU) || a.x.mask || a.x.lse)) return CAPDEC_EINVAL;
    for (int n = 0; n < ngroups; ++n)
        if ((flags & CAPDEC_GEMM_ATOMIC) && groups[n].R) return CAPDEC_EINVAL;
    if (tp) return launch_gemm_layout<true, true>(a, ngroups, maxI, maxJ, vec, s);
    if (tq) return launch_gemm_layout<false, true>(a, ngroups, maxI, maxJ, vec, s);
    return launch_gemm_layout<false, false>(a, ngroups, maxI, maxJ, vec, s);
}

extern "C" {

#ifdef CAPDEC_TUNING_HOOKS
int capdec_gemm_force_config(int cfg) {
    CAPDEC_REQUIRE(cfg == 0 || cfg == 11 || cfg == 22 || cfg == 216 || cfg == 32 || cfg == 116 || cfg == 316 || cfg == 2116 ||
                   cfg == 2132);
    g_force_cfg = cfg;
    return 0;
}

int capdec_gemm_force_staging(int mode) {
    CAPDEC_REQUIRE(mode == 0 || mode == 1);
    g_staging = mode;
    return 0;
}
#endif

size_t capdec_gemm_splitk_workspace_elems(int I, int J, int K) {
    if (I < 1 || J < 1 || K < 1) return 0;
    return (size_t)splitk_plan(I, J, K) * (size_t)I * (size_t)J;
}

int capdec_gemm(const capdec_gemm_group* groups, int ngroups, int K, int flags, void* stream) {
    return capdec_launch_gemm(groups, ngroups, K, flags, (hipStream_t)stream, nullptr);
}

int capdec_gemm_ex(const capdec_gemm_group* groups, int ngroups, int K, int flags, const capdec_gemm_extra* extra,
                   void* stream) {
    return capdec_launch_gemm(groups, ngroups, K, flags, (hipStream_t)stream, extra);
}

int capdec_bgemm(const capdec_bgemm_desc* a, int form, void* stream) {
    CAPDEC_REQUIRE(a && a->P && a->Q && a->C && form >= 0 && form <= 2);
    CAPDEC_REQUIRE(a->B >= 0 && a->I >= 1 && a->J >= 1 && a->K >= 1 && a->seg >= 1 && a->mult >= 1);
    CAPDEC_REQUIRE((int64_t)a->seg * a->mult <= (1 << 20) && (a->P2 == nullptr) == (a->Q2 == nullptr));
    CAPDEC_REQUIRE(form == 2 || !a->P2);
    if (!(capdec_aligned16(a->P) && capdec_aligned16(a->Q) && capdec_aligned16(a->C) && a->ldp % 4 == 0 && a->ldq % 4 == 0 &&
          a->dense_stride % 4 == 0 && (!a->P2 || (capdec_aligned16(a->P2) && capdec_aligned16(a->Q2)))))
        return CAPDEC_EALIGN;
    if (form == 0) CAPDEC_REQUIRE(a->K % 4 == 0 && a->ldp >= a->K && a->ldq >= a->K && a->ldc >= a->J);
    if (form == 1) CAPDEC_REQUIRE(a->ldp >= ((a->K + 3) & ~3) && a->J % 4 == 0 && a->ldq >= a->J && a->ldc >= a->J);
    if (form == 2) CAPDEC_REQUIRE(a->ldp >= ((a->I + 3) & ~3) && a->J % 4 == 0 && a->ldq >= a->J && a->ldc >= a->J);
    if (a->B == 0) return 0;
    BgemmKArgs k;
    k.a = *a;
    const int maxI = form == 2 ? a->I : a->seg * a->mult;
    k.ni = form == 2 ? (maxI + 63) / 64 : (maxI + 31) / 32;
    k.nj = (a->J + 63) / 64;
    const int64_t total = (int64_t)a->B * k.ni * k.nj;
    CAPDEC_REQUIRE(total <= (1 << 28));
    k.per_xcd = (int)((total + 7) / 8);
    const dim3 grid(8u * (unsigned)k.per_xcd);
    hipStream_t s = (hipStream_t)stream;
    if (form == 0)
        hipLaunchKernelGGL((bgemm_kernel<0, 1>), grid, dim3(128), (lds_floats<1, 1, 1, 16, false, false, 1>() * sizeof(float)), s, k);
    else if (form == 1)
        hipLaunchKernelGGL((bgemm_kernel<1, 1>), grid, dim3(128), (lds_floats<1, 1, 1, 16, false, true, 1>() * sizeof(float)), s, k);
    else
        hipLaunchKernelGGL((bgemm_kernel<2, 2>), grid, dim3(256), (lds_floats<2, 1, 1, 16, true, true, 1>() * sizeof(float)), s, k);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_vocab_tiles(int V) { return (V + VOCAB_BI - 1) / VOCAB_BI; }
size_t capdec_vocab_partial_elems(int M, int V) { return (M < 0 || V < 1) ? 0 : capdec_part_elems(M, capdec_vocab_tiles(V)); }

// Readiness of vocab_argmax_small_kernel per device: 0 unknown, 1 ready, -1 the attribute call failed there.  An
// attribute call is not made while `s` is being captured (it is not a stream operation and would end the capture with
// an error of its own): the caller then takes the general tiling for this call.
static std::mutex g_vs_mutex;
static signed char g_vs_state[64];
static bool vocab_small_ready(hipStream_t s) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(g_vs_mutex);
    if (g_vs_state[dev] == 0) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
        g_vs_state[dev] = allow_lds(vocab_argmax_small_kernel, VS_LDS_BYTES) ? 1 : -1;
    }
    return g_vs_state[dev] == 1;
}

static int vocab_reduce_launch(int mode, const float* x, int64_t ldx, const float* W, const float* bias, int M, int V,
                               int K, float* out_a, void* out_b, hipStream_t s, const int32_t* rows_dev = nullptr) {
    CAPDEC_REQUIRE(x && W && bias && out_a && out_b && M >= 0 && V >= 1 && K >= 1 && ldx >= K);
    if (M == 0) return 0;
    const int nt = capdec_vocab_tiles(V);
    const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && capdec_aligned16(x) && capdec_aligned16(W);
    if (mode == 0 && M <= VS_MAX_ROWS && vec && K % 32 == 0 && g_staging != 1) {
        // a small batch (the reference's loaders default to 32 and 64 rows): one workgroup of eight wavefronts per
        // vocabulary tile and block of 32 rows
        // (its LDS limit is raised once per device, with that device current; where that fails, and while a stream is
        //  being captured before it has happened, the general tiling below takes the same shapes)
        if (vocab_small_ready(s)) {
            hipLaunchKernelGGL(vocab_argmax_small_kernel, dim3(nt, (M + 31) / 32), dim3(512), VS_LDS_BYTES, s, W, x, ldx, bias, V, M, K, out_a,
                               static_cast<int*>(out_b), nt, rows_dev);
            CAPDEC_LAUNCH_CHECK();
            return 0;
        }
    }
    // samples sit on the j (lane) axis: 128-wide tiles when there are enough rows to fill the chip
    const bool wide = ((long)nt * ((M + 127) / 128) >= 192);
    constexpr int VBK = 16;   // 64x128 tile at BK 16: 30 KiB of LDS, five workgroups per CU (1256 tiles at V=10000, M=1024: 4.9 per CU)
    constexpr size_t smem_w = lds_floats<2, VOCAB_WI, 2, VBK, false, false, 1>() * sizeof(float);
    constexpr size_t smem_n = lds_floats<2, VOCAB_WI, 1, VBK, false, false, 1>() * sizeof(float);
    // operand tiles go to LDS straight from the load when every 16-float k-tile is whole and 16-byte aligned
    const bool dma = vec && K % VBK == 0 && g_staging != 1;
    const size_t smem = dma ? (wide ? dma_lds_floats<2, VOCAB_WI, 2, VBK, 1>() : dma_lds_floats<2, VOCAB_WI, 1, VBK, 1>()) * sizeof(float)
                            : (wide ? smem_w : smem_n);
    TileMap map;
    dim3 grid(make_tile_map(map, nt, wide ? (M + 127) / 128 : (M + 63) / 64, 1, false), 1, 1);
#define CAPDEC_VR(WJ_, VEC_, MODE_)                                                                                    \
    hipLaunchKernelGGL((vocab_reduce_kernel<VOCAB_WI, WJ_, VBK, VEC_, MODE_>), grid, dim3(256), smem, s, W, x, ldx, bias, V, \
                       M, K, out_a, out_b, nt, map, rows_dev)
#define CAPDEC_VD(WJ_, MODE_)                                                                                          \
    hipLaunchKernelGGL((vocab_reduce_kernel<VOCAB_WI, WJ_, VBK, true, MODE_, true>), grid, dim3(256), smem, s, W, x, ldx, bias, \
                       V, M, K, out_a, out_b, nt, map, rows_dev)
    if (dma) {
        if (mode == 0) { if (wide) CAPDEC_VD(2, 0); else CAPDEC_VD(1, 0); }
        else           { if (wide) CAPDEC_VD(2, 1); else CAPDEC_VD(1, 1); }
    } else if (mode == 0) {
        if (wide) { if (vec) CAPDEC_VR(2, true, 0); else CAPDEC_VR(2, false, 0); }
        else      { if (vec) CAPDEC_VR(1, true, 0); else CAPDEC_VR(1, false, 0); }
    } else {
        if (wide) { if (vec) CAPDEC_VR(2, true, 1); else CAPDEC_VR(2, false, 1); }
        else      { if (vec) CAPDEC_VR(1, true, 1); else CAPDEC_VR(1, false, 1); }
    }
#undef CAPDEC_VR
#undef CAPDEC_VD
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_vocab_argmax_partial(const float* x, int64_t ldx, const float* W, const float* bias, int M, int V, int K,
                                float* part_val, int32_t* part_idx, void* stream) {
    return vocab_reduce_launch(0, x, ldx, W, bias, M, V, K, part_val, part_idx, (hipStream_t)stream);
}

}  // extern "C"

int capdec_launch_vocab_argmax_rows(const float* x, int64_t ldx, const float* W, const float* bias, int M, int V, int K,
                                    float* part_val, int32_t* part_idx, const int32_t* rows_dev, hipStream_t s) {
    return vocab_reduce_launch(0, x, ldx, W, bias, M, V, K, part_val, part_idx, s, rows_dev);
}

extern "C" {

int capdec_argmax_finalize(const float* part_val, const int32_t* part_idx, int M, int ntiles, int32_t* out_idx,
                           float* out_val, void* stream) {
    CAPDEC_REQUIRE(part_val && part_idx && out_idx && M >= 0 && ntiles >= 1);
    if (M == 0) return 0;
    hipLaunchKernelGGL(argmax_finalize_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, part_val,
                       part_idx, M, ntiles, out_idx, out_val);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

// ---- allocation-free vocabulary arg-max -------------------------------------------------------------------------
// The per-tile winners live in a device array that is part of the library's image instead of a caller workspace.
// Every GPU the library runs on holds its own copy of the array at its own address for the life of the process; in
// host code the array's NAME is only a handle for hipGetSymbolAddress, never a device address.
constexpr size_t VOCAB_STATIC_ELEMS = (size_t)CAPDEC_VOCAB_STATIC_MAX_ELEMS;
static __device__ __align__(256) unsigned char scratch[VOCAB_STATIC_ELEMS * (sizeof(float) + sizeof(int32_t))];

constexpr int STATIC_MAX_DEVICES = 64;
static std::mutex g_scratch_mutex;
static unsigned char* g_scratch_ptr[STATIC_MAX_DEVICES];     // per device; nullptr until resolved on that device

// Device address of `scratch` on the current device (resolved once per device, with that device current).
static int scratch_address(unsigned char** out, hipStream_t stream) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= STATIC_MAX_DEVICES) return CAPDEC_EINVAL;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        // the first call on a device resolves the address: not under stream capture (make one call before capturing)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return CAPDEC_EINVAL;
        void* ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!ptr) return CAPDEC_EINVAL;
        g_scratch_ptr[dev] = static_cast<unsigned char*>(ptr);
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

int capdec_vocab_argmax_static(const float* x, int64_t ldx, const float* W, const float* bias, int M, int V, int K,
                               int32_t* out_idx, float* out_val, void* stream) {
    CAPDEC_REQUIRE(x && W && bias && out_idx && M >= 0 && V >= 1 && K >= 1 && ldx >= K);
    const size_t elems = capdec_vocab_partial_elems(M, V);
    if (elems > VOCAB_STATIC_ELEMS) return CAPDEC_E2BIG;
    if (M == 0) return 0;
    unsigned char* base = nullptr;
    CAPDEC_TRY(scratch_address(&base, (hipStream_t)stream));
    // values first, ids behind them; every (row < M, tile) slot that the merge reads is written by the projection
    // kernel of the same call, so nothing has to be cleared between calls
    float* part_val = reinterpret_cast<float*>(base);
    int32_t* part_idx = reinterpret_cast<int32_t*>(base + VOCAB_STATIC_ELEMS * sizeof(float));
    CAPDEC_TRY(vocab_reduce_launch(0, x, ldx, W, bias, M, V, K, part_val, part_idx, (hipStream_t)stream));
    hipLaunchKernelGGL(argmax_finalize_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, part_val, part_idx, M,
                       capdec_vocab_tiles(V), out_idx, out_val);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

int capdec_vocab_lse_partial(const float* x, int64_t ldx, const float* W, const float* bias, int M, int V, int K,
                             float* part_max, float* part_sum, void* stream) {
    return vocab_reduce_launch(1, x, ldx, W, bias, M, V, K, part_max, part_sum, (hipStream_t)stream);
}

int capdec_nll_finalize(const float* part_max, const float* part_sum, int M, int V, const float* x, int64_t ldx,
                        const float* W, const float* bias, int K, const int64_t* targets, int64_t ignore_index,
                        float* lse, float* nll, float* loss_sum, void* stream) {
    CAPDEC_REQUIRE(part_max && part_sum && x && W && bias && targets && M >= 0 && V >= 1 && K >= 1);
    if (M == 0) return 0;
    hipLaunchKernelGGL(nll_finalize_kernel, dim3((M + NLL_ROWS - 1) / NLL_ROWS), dim3(256), 0, (hipStream_t)stream, part_max, part_sum,
                       M, capdec_vocab_tiles(V), x, ldx, W, bias, K, targets, ignore_index, lse, nll, loss_sum, V);
    CAPDEC_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
