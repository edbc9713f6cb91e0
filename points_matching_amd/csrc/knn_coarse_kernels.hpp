// knn_coarse_kernels.hpp — kernel bodies and launch templates of the two MFMA coarse passes of the L2 matcher and of
// the Hamming matcher (see knn_l2.hip for the overall scheme and docs/SPEC.md S1b for the error bounds).  Replaces,
// together with knn_l2.hip, `matcher.match(imageDesc1, imageDesc2, matchePoints, Mat())` (main.cpp:46).
//
// Everything is templated on an ablation policy.  Exactly two translation units include this header:
// csrc/knn_coarse.hip (the product: policy AblNone, nothing else is instantiated) and
// tools/ablation/knn_coarse_ablation.hip (timing-only variants for profiles/; never linked into libpm_hip.so).
//
// Built with -ffinite-math-only (see knn_shared.hpp).  Nothing in this file decides a result bit:
// the coarse values only nominate candidate rows for the canonical refinement.
#pragma once
#include <type_traits>

#include "knn_shared.hpp"

namespace pm_knn {
namespace {

// Compile-time switches of the timing-only ablation builds ("what does a tile cost without ...": outputs are wrong).
// The product instantiates every kernel with AblNone; the variants live in tools/ablation/knn_coarse_ablation.hip.
// No preprocessor hooks.
template <bool NO_EPI, bool NO_STAGE, bool NO_BARRIER, bool NO_LDSREAD>
struct Abl {
    static constexpr bool no_epi = NO_EPI;            // drop the in-chain selection (accumulators kept live)
    static constexpr bool no_stage = NO_STAGE;        // do not stage the next tile
    static constexpr bool no_barrier = NO_BARRIER;    // no workgroup barrier per tile
    static constexpr bool no_ldsread = NO_LDSREAD;    // A fragments read once, not per chunk
    static __device__ __forceinline__ void stamp(int) {}   // in-kernel clock stamp i (diagnostic builds: tools/ablation/knn_coarse_stamps.hip)
};
typedef Abl<false, false, false, false> AblNone;

// Workgroup -> (query block, train split).  Workgroups are dispatched round-robin over the 8 XCDs in launch order (x
// fastest), and every XCD has its own L2: in launch order each XCD ends up pulling ALL train splits (or all query
// blocks) through its L2 — 8 x the train copy per launch.  When the grid allows it, XCD x owns a 2-D tile of the
// grid instead: a quarter of the query blocks x half of the splits, i.e. Q/4 + T/2 through each L2 (the minimum of
// 8 * (Q/a + T/b) over a*b = 8 for Q = T).  Measured (profiles/r02_pmc_*): C3 f16 fetch 21.3 -> 14.4 MB, C4 i8 76.4 ->
// 50.8 MB; same-session A/B of the two orders (PM_OPT_KNN_XCD_TILE 1 / 2): tiled is 0-4 % faster at every shape tried
// (C3 f16 19.85 -> 19.61 us, C3 f32 route 147.1 -> 141.7, C4 i8 198.9 -> 196.2, 32k x 32k f16 211.4 -> 209.0).
// (The mapping is a bijection of the grid whatever the number of XCDs: on a partitioned device it is merely not useful.)
struct WgTile { int qb, split; };
__device__ __forceinline__ WgTile wg_tile(bool tiled)
{
    const int nx = gridDim.x, ny = gridDim.y;
    if (!tiled || (nx & 3) || (ny & 1)) return WgTile{static_cast<int>(blockIdx.x), static_cast<int>(blockIdx.y)};
    const int L = blockIdx.y * nx + blockIdx.x;
    const int x = L & 7, j = L >> 3;                    // XCD, index inside the XCD
    const int tw = nx >> 2, th = ny >> 1;               // the XCD's tile: tw query blocks x th splits
    return WgTile{(x & 3) * tw + j % tw, (x >> 2) * th + j / tw};
}

template <typename ABL>
__device__ __forceinline__ void tile_barrier()
{
    if constexpr (ABL::no_barrier) asm volatile("" ::: "memory");
    else __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// coarse pass on the matrix cores
//
// Per (query, train row) the MFMA chain accumulates w = q.t - ||t||^2/2 = -(d2a - ||q||^2)/2 (the
// seed -||t||^2/2 is one more k-step of the chain): the LARGEST w are the nearest rows.  The row's position in this lane's stream
// (lid = tile_in_split*32 + block*16 + reg) is written into the low `bits` mantissa bits of w, so
// a candidate is ONE float and keeping the 4 largest is branch-free:
//     n0 = max(x,w0); n1 = med3(x,w0,w1); n2 = med3(x,w1,w2); n3 = med3(x,w2,w3)
// (5 VALU per pair incl. the bit insert).  The truncation error 2^(bits-23)*|w| is part of the
// refinement's window (SPEC S1b).  Two accumulator sets: the epilogue of tile t-1 is issued
// between the MFMAs of tile t, so the matrix pipe does not wait for the selection.
// ---------------------------------------------------------------------------------------------
// (keep & w) | (~keep & id): the id lands in the low mantissa bits (hipcc emits v_bfi / v_and_or)
__device__ __forceinline__ float embed_lid(float w, unsigned keep_mask, unsigned lid)
{
    return __uint_as_float((__float_as_uint(w) & keep_mask) | (lid & ~keep_mask));
}

// This translation unit is built with -ffinite-math-only: fmaxf / fmed3 then need no canonicalising
// v_max in front of every operand (the operands ARE finite here: non-finite inputs divert to the
// exact re-scan), and because these are compiler-visible instructions hipcc's hazard recogniser
// keeps the required distance between an MFMA and the first VALU read of its result.  (An
// earlier version used inline-asm v_max3/v_med3 directly on the accumulators: hipcc pads nothing
// inside or in front of asm, and the f16 route then read accumulators one k-chunk early.)
__device__ __forceinline__ void top4_insert(f32x4& c, float x)
{
    const float n0 = fmaxf(x, c[0]);
    const float n1 = __builtin_amdgcn_fmed3f(x, c[0], c[1]);
    const float n2 = __builtin_amdgcn_fmed3f(x, c[1], c[2]);
    const float n3 = __builtin_amdgcn_fmed3f(x, c[2], c[3]);
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// The two lane halves of a query column (lanes r and r + 32: rows 4h..4h+3 of every 8-row stripe) keep
// separate lists while the sweep runs; at the end the half bit goes into bit 0 of every id and the two
// sorted lists are merged into ONE list of 4 per (query, split) — half the candidate volume for the
// refinement.  Once per kernel: four cross-half shuffles and four inserts.
__device__ __forceinline__ void merge_halves(f32x4& cl, int h)
{
    f32x4 own, other;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        own[i] = __uint_as_float(__float_as_uint(cl[i]) | static_cast<unsigned>(h));
        other[i] = __shfl_xor(own[i], 32, 64);
    }
    cl = own;
#pragma unroll
    for (int i = 0; i < 4; ++i) top4_insert(cl, other[i]);
}

// GROUPED selection.  Registers 4g..4g+3 of an accumulator are four CONSECUTIVE train rows
// (8g + 4*(lane>>5) + {0,1,2,3} of the 32-row block).  Only the group's largest w competes for the
// list (2 VALU per 4 values: v_max3 + v_max), tagged with the GROUP id; the refinement
// re-evaluates all four rows of a candidate group.  That is sound: a row inside the window has a
// group maximum inside the window, and the k largest group maxima are attained by k distinct rows.
// VALU cost per descriptor pair: (2 + 1 + 4) / 4 = 1.75 instead of 5 — the selection, not the
// matrix pipe, is what bounds the f16 route.
__device__ __forceinline__ float group_max(const f32x16& acc, int g)
{
    return fmaxf(fmaxf(fmaxf(acc[4 * g], acc[4 * g + 1]), acc[4 * g + 2]), acc[4 * g + 3]);
}

// NCH = padded dim / 8; FULL = (dim == 8*NCH), which drops the column guards; TT = train rows
// per LDS tile (64: two workgroups per CU, 128: one workgroup per CU with twice the MFMA work
// between barriers).  grid = (ceil(nq/QB), splits).  Dynamic LDS: 2 tiles of TT x (8*NCH + 4)
// floats (row stride padded by one 16-B slot: conflict-free ds_read_b128 for the 16 rows of a lane
// group) + 2 x TT seeds (-||t||^2/2, or -KNN_BIG/2 past the last row).
template <int NCH, bool FULL, int TT>
struct KnnTile {
    static constexpr int DP = NCH * 8;
    static constexpr int LDT = DP + 4;
    static constexpr int F4_PER_ROW = DP / 4;
    static constexpr int NSTG = TT * F4_PER_ROW / 256;
    static_assert(TT * F4_PER_ROW % 256 == 0, "tile must split evenly over the workgroup");

    f32x4 stg[NSTG];
    float stg_n;

    // global -> registers (unconditional loads: clamped addresses, zero-select afterwards)
    __device__ __forceinline__ void load(const float* __restrict__ T, const float* __restrict__ tnorm, int tile,
                                         int nt, int dim, int tid)
    {
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            const int f = tid + 256 * i;
            const int row = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
            int g = tile * TT + row;
            g = g < nt ? g : nt - 1;
            int col = 4 * c4;
            if (!FULL) col = col < dim ? col : dim - 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(T + static_cast<size_t>(g) * dim + col);
            if (!FULL && 4 * c4 >= dim) v = f32x4{0.f, 0.f, 0.f, 0.f};
            stg[i] = v;
        }
        const int gn = tile * TT + (tid & (TT - 1));
        const float nrm = tnorm[gn < nt ? gn : nt - 1];
        stg_n = gn < nt ? -0.5f * nrm : -0.5f * KNN_BIG;
    }
    // registers -> LDS buffer
    // the row seed -||t||^2/2 travels in the row's 16-byte pad slot: (seed, 0, 0, 0)
    __device__ __forceinline__ void store(float* __restrict__ Ts, int buf, int tid) const
    {
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            const int f = tid + 256 * i;
            const int row = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
            *reinterpret_cast<f32x4*>(Ts + (buf * TT + row) * LDT + 4 * c4) = stg[i];
        }
        if (tid < TT) *reinterpret_cast<f32x4*>(Ts + (buf * TT + tid) * LDT + DP) = f32x4{stg_n, 0.f, 0.f, 0.f};
    }
};

// One tile of the sweep.  Everything that is not an MFMA is issued INSIDE the chain, in the shadow
// of a 64-cycle MFMA: the global loads of the next tile after chunk 0, the selection of the
// previous tile's accumulators p[] spread over all chunks, and the LDS writes of the next tile
// after the last-but-one chunk.  The row seed -||t||^2/2 enters through the matrix pipe as well:
// one extra k-step per block multiplies the pad column (seed, 0) of the A tile by (1, 0), starting
// from the inline constant C = 0, so no accumulator is initialised and the selection needs no
// add.  The next tile is always staged (clamped addresses; past the end the data is unused),
// which keeps the chain free of branches.
// C[i][j] of lane (j = lane&31), register reg is train row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
template <int NCH, bool FULL, int NB, bool EPI, typename ABL>
__device__ __forceinline__ void knn_tile_compute(float* __restrict__ Ts, int buf, int r, int h,
                                                 const f32x4 (&qf)[NCH], f32x16 (&a)[NB], const f32x16 (&p)[NB],
                                                 unsigned pbase, unsigned keep_mask, f32x4& cl,
                                                 KnnTile<NCH, FULL, NB * 32>& st, const float* __restrict__ T,
                                                 const float* __restrict__ tnorm, int next_tile, int nt, int dim,
                                                 int tid)
{
    constexpr int DP = NCH * 8;
    constexpr int LDT = DP + 4;
    constexpr int TT = NB * 32;
    static_assert(NCH >= 4, "the in-chain schedule needs at least 4 chunks");
    const float* tb = Ts + buf * TT * LDT + r * LDT + 4 * h;
    const float one_or_zero = h == 0 ? 1.f : 0.f;          // B side of the seed step: k = h
    float sda[NB];                                          // A side: pad[h] = (seed, 0)[h]
    f32x4 xn[NB];                                           // A fragments, one chunk ahead of their MFMAs
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        sda[blk] = Ts[buf * TT * LDT + (32 * blk + r) * LDT + DP + h];
        xn[blk] = *reinterpret_cast<const f32x4*>(tb + 32 * blk * LDT);
    }
    {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
            a[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(sda[blk], one_or_zero, zero, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        f32x4 x[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) x[blk] = xn[blk];
        if constexpr (!ABL::no_ldsread) {
            if (c + 1 < NCH) {
#pragma unroll
                for (int blk = 0; blk < NB; ++blk)
                    xn[blk] = *reinterpret_cast<const f32x4*>(tb + 32 * blk * LDT + 8 * (c + 1));
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk)
                a[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[blk][t], qf[c][t], a[blk], 0, 0, 0);
        if constexpr (!ABL::no_stage) {
            if (c == 0) st.load(T, tnorm, next_tile, nt, dim, tid);                   // global -> registers
            if (c == NCH - 2) st.store(Ts, buf ^ 1, tid);                             // registers -> LDS
        }
        if (EPI && !ABL::no_epi) {
            // row groups of the previous tile, spread over chunks 1..NCH-1 (chunk 0 is left alone so
            // that a full round of this tile's MFMAs separates the previous tile's last MFMA from
            // the first read of its accumulators)
            constexpr int NG = 4 * NB;
#pragma unroll
            for (int g = (c == 0 ? 0 : (c - 1) * NG / (NCH - 1)); g < (c == 0 ? 0 : c * NG / (NCH - 1)); ++g)      // block g>>2, group g&3
                top4_insert(cl, embed_lid(group_max(p[g >> 2], g & 3), keep_mask, (pbase + static_cast<unsigned>(g)) << 1));
            // pin the selection to this chunk: without a use here hipcc sinks all of it below the
            // MFMA chain (in front of the barrier), where nothing hides it.
            asm volatile("" : "+v"(cl[0]), "+v"(cl[1]), "+v"(cl[2]), "+v"(cl[3]));
        }
    }
}

template <int NB>
__device__ __forceinline__ void knn_select_all(const f32x16 (&p)[NB], unsigned pbase, unsigned keep_mask, f32x4& cl)
{
#pragma unroll
    for (int g = 0; g < 4 * NB; ++g)
        top4_insert(cl, embed_lid(group_max(p[g >> 2], g & 3), keep_mask, (pbase + static_cast<unsigned>(g)) << 1));
}

template <int NCH, bool FULL, int TT, typename ABL>
__global__ __launch_bounds__(256, (TT == 64 ? 2 : 1)) void knn_l2_mfma(
    const float* __restrict__ Q, const float* __restrict__ T, const float* __restrict__ tnorm, int nq,
    int nt, int dim, int tiles_per_split, unsigned keep_mask, float* __restrict__ cand_val, int slots,
    const unsigned long long* __restrict__ stats, unsigned epoch, int only_if_ineligible)
{
    if (only_if_ineligible & 1) {              // auto mode (bit 1: XCD-tiled grid): the f16 kernel handles integer data and,
        const unsigned long long s1 = stats[1], s3 = stats[3];       // on rounded copies, general floats — unless that route withdrew
        if (!(static_cast<unsigned>(s1 >> 32) == epoch && (s1 & 2ull) && static_cast<unsigned>(s3 >> 32) == epoch)) return;
    }
    using Tile = KnnTile<NCH, FULL, TT>;
    constexpr int LDT = Tile::LDT;
    constexpr int NB = TT / 32;
    constexpr unsigned IDS = 4 * NB;           // row-group ids a lane sees per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ts = smem;                          // [2][TT][LDT], seed in each row's pad slot

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const WgTile wg = wg_tile((only_if_ineligible & 2) != 0);
    const int qrow = wg.qb * QB + wave * 32 + r;
    const int qld = qrow < nq ? qrow : nq - 1;

    // B operand: this lane's query row, k = 8c + 4h + {0..3} for chunk c (the k permutation is
    // shared with the A operand, and a dot product does not care about k order).
    f32x4 qf[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k0 = 8 * c + 4 * h;
        int col = k0;
        if (!FULL) col = col < dim ? col : dim - 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(Q + static_cast<size_t>(qld) * dim + col);
        if (!FULL && k0 >= dim) v = f32x4{0.f, 0.f, 0.f, 0.f};
        qf[c] = v;
    }

    const int ntiles = (nt + TT - 1) / TT;
    const int tile0 = wg.split * tiles_per_split;
    int tile1 = tile0 + tiles_per_split;
    if (tile1 > ntiles) tile1 = ntiles;

    f32x4 cl = {-KNN_BIG, -KNN_BIG, -KNN_BIG, -KNN_BIG};
    if (tile0 < tile1) {                       // block-uniform
        Tile st;
        st.load(T, tnorm, tile0, nt, dim, tid);
        st.store(Ts, 0, tid);
        __syncthreads();

        // every accumulator has a compile-time name: tiles alternate A, B, A, ...  (tix = tile - tile0)
        f32x16 accA[NB], accB[NB];
        int tix = 0;
        const int ntl = tile1 - tile0;
        knn_tile_compute<NCH, FULL, NB, false, ABL>(Ts, 0, r, h, qf, accA, accA, 0u, keep_mask, cl, st, T, tnorm,
                                               tile0 + 1, nt, dim, tid);
        tile_barrier<ABL>();
        ++tix;
        for (;;) {
            if (tix >= ntl) { knn_select_all<NB>(accA, static_cast<unsigned>(tix - 1) * IDS, keep_mask, cl); break; }
            // tile -> B while selecting A (tile-1)
            knn_tile_compute<NCH, FULL, NB, true, ABL>(Ts, tix & 1, r, h, qf, accB, accA, static_cast<unsigned>(tix - 1) * IDS,
                                                  keep_mask, cl, st, T, tnorm, tile0 + tix + 1, nt, dim, tid);
            tile_barrier<ABL>();
            ++tix;
            if (tix >= ntl) { knn_select_all<NB>(accB, static_cast<unsigned>(tix - 1) * IDS, keep_mask, cl); break; }
            // tile -> A while selecting B (tile-1)
            knn_tile_compute<NCH, FULL, NB, true, ABL>(Ts, tix & 1, r, h, qf, accA, accB, static_cast<unsigned>(tix - 1) * IDS,
                                                  keep_mask, cl, st, T, tnorm, tile0 + tix + 1, nt, dim, tid);
            tile_barrier<ABL>();
            ++tix;
        }
    }

    merge_halves(cl, h);
    if (qrow < nq && h == 0) {
        const size_t o = static_cast<size_t>(qrow) * slots + wg.split * KNN_C;
        *reinterpret_cast<f32x4*>(cand_val + o) = cl;
    }
}

// ---------------------------------------------------------------------------------------------
// f16 route: integer-valued descriptors (what OpenCV's SIFT emits: 0..255 stored as float).
// knn_l2_prep16 (knn_l2.hip) writes the padded f16 copies and verifies eligibility; see there.
//
// At this matrix rate the selection is as expensive as the MFMAs, so the kernel is organised
// around it: a wave owns 64 queries (two B blocks: every A fragment read from LDS feeds two
// MFMAs), walks the 128-row tile one 32-row block at a time and selects block b-1 between the
// MFMAs of block b, with two alternating accumulator sets.
// ---------------------------------------------------------------------------------------------
// The kernel is shared by two routes with the same structure (rows of NCH k-chunks of 32 B, 16 B per
// lane and MFMA; 288-byte rows with a seed chunk for f16, 256-byte rows for i8):
//   RouteF16  v_mfma_f32_32x32x16_f16, values float, group id in the low mantissa bits;
//   RouteI8   v_mfma_i32_32x32x32_i8 on +-1 bytes (binary descriptors: dot = bits - 2*hamming),
//             values int, candidate = (dot << I8_SHIFT) | group id.
// `par` is the f16 route's keep mask (unused by the i8 route: its shift is a constant).
// DP: padded data columns, 128 (9 k-chunks, 288-byte rows) or 256 (17 k-chunks, 544-byte rows: descriptors of up to 256
// dimensions, round 3).  The LDS row is one 16-byte slot longer: an ODD number of slots, so the 16 rows of a
// ds_read_b128 lane group hit 16 different slots.
template <int DP>
struct RouteF16T {
    static constexpr int NCH = DP / 16 + 1;           // data chunks + the seed chunk
    static constexpr int ROW16 = (DP + 16) / 8;       // 16-byte units per global row
    static constexpr int LDS_ROW16 = ROW16 + 1;       // ... per LDS row (one pad slot)
    static_assert(LDS_ROW16 % 2 == 1 && (H_TT * LDS_ROW16) % 64 == 0, "conflict-free rows, whole DMA pieces");
    static constexpr int GPB = 4;                     // row groups per 32-row block and lane: groups of 4 rows
    static constexpr bool MERGE = true;               // one list per (query, split): float ties are rare
    static constexpr bool SEEDED = false;             // the seed rides its own k-chunk
    typedef f16x8 frag;
    typedef f32x16 acc;
    typedef f32x4 list;
    static __device__ __forceinline__ acc zero()
    {
        return acc{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
    static __device__ __forceinline__ acc mfma(frag a, frag b, acc c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ list empty() { return list{-KNN_BIG, -KNN_BIG, -KNN_BIG, -KNN_BIG}; }
    static __device__ __forceinline__ void select(const acc& a, unsigned par, unsigned gid, list& cl, int g)
    {
        top4_insert(cl, embed_lid(group_max(a, g), par, gid));
    }
    static __device__ __forceinline__ void merge(list& cl, int h) { merge_halves(cl, h); }
    static __device__ __forceinline__ void put(list& cl, float x) { top4_insert(cl, x); }
};
typedef RouteF16T<128> RouteF16;
static_assert(RouteF16::NCH == H_NCH && RouteF16::ROW16 == H_ROW16 && RouteF16::LDS_ROW16 == H_LDS_ROW16, "knn_shared.hpp constants");

struct RouteI8 {
    static constexpr int NCH = I8_NCH;                // 8 data chunks, no seed: pad rows are all-zero (dot = 0)
    static constexpr int ROW16 = I8_ROW16;
    static constexpr int LDS_ROW16 = I8_LDS_ROW16;
    static constexpr int GPB = 2;                     // groups of 8 rows: the popcount refinement is cheap
    static constexpr bool MERGE = false;              // integer distances tie all the time: a 4-deep merged list would
                                                      // overflow (and force split re-scans) for most queries
    static constexpr bool SEEDED = false;             // Hamming: nothing to seed (dot = bits - 2*hamming)
    typedef i32x4 frag;
    typedef i32x16 acc;
    typedef i32x4 list;
    static __device__ __forceinline__ acc zero() { return acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
    static __device__ __forceinline__ acc mfma(frag a, frag b, acc c)
    {
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ list empty() { return list{I8_EMPTY, I8_EMPTY, I8_EMPTY, I8_EMPTY}; }
    static __device__ __forceinline__ int med3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }
    static __device__ __forceinline__ void select(const acc& a, unsigned par, unsigned gid, list& cl, int g)
    {
        const int m0 = max(max(a[8 * g], a[8 * g + 1]), a[8 * g + 2]);              // v_max3_i32 x3 + v_max_i32
        const int m1 = max(max(a[8 * g + 3], a[8 * g + 4]), a[8 * g + 5]);
        const int m = max(max(max(m0, m1), a[8 * g + 6]), a[8 * g + 7]);
        const int x = static_cast<int>((static_cast<unsigned>(m) << I8_SHIFT) | gid);   // v_lshl_or_b32
        (void)par;
        insert(cl, x);
    }
    static __device__ __forceinline__ void put(list& cl, int x) { insert(cl, x); }
    static __device__ __forceinline__ void insert(list& cl, int x)
    {
        const int n0 = max(x, cl[0]);
        const int n1 = med3(x, cl[0], cl[1]);
        const int n2 = med3(x, cl[1], cl[2]);
        const int n3 = med3(x, cl[2], cl[3]);
        cl[0] = n0; cl[1] = n1; cl[2] = n2; cl[3] = n3;
    }
    static __device__ __forceinline__ void merge(list& cl, int h)     // MERGE == false: only tags the ids with the half
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) cl[i] = cl[i] == I8_EMPTY ? I8_EMPTY : (cl[i] | h);
    }
};

// Seeded routes (knn_shared.hpp): the accumulators START from the per-row term, read from a per-tile seed array in LDS.
//   RouteF16S  RouteF16 without the seed chunk: 256-byte rows = 8 k-chunks, every MFMA is algorithmic work;
//   RouteU8    u8-valued descriptors centred to x - 128: v_mfma_i32_32x32x32_i8 on 128-byte rows = 4 k-chunks, exact
//              integers, candidate = (w << U8_SHIFT) | id (no mantissa truncation, no error term but the seed's half unit).
struct RouteF16S : RouteF16 {
    static constexpr int NCH = F16S_NCH;
    static constexpr int ROW16 = F16S_ROW16;
    static constexpr int LDS_ROW16 = F16S_LDS_ROW16;
    static constexpr bool SEEDED = true;
};

// GPB_ row groups per 32-row block and lane half: 4 = groups of 4 consecutive rows (the f16 route's geometry), 2 = groups
// of 8 rows (two runs of 4, as on the Hamming route), 1 = the lane's 16 rows of the block.  Larger groups = fewer
// selection instructions per descriptor pair (28 / 18 / 13 per block and wave) and more rows per candidate for the
// integer refinement (knn_l2_refine8), which evaluates one row per lane.
template <int GPB_>
struct RouteU8T {
    static constexpr int NCH = U8_NCH;
    static constexpr int ROW16 = U8_ROW16;
    static constexpr int LDS_ROW16 = U8_LDS_ROW16;
    static constexpr int GPB = GPB_;
    static constexpr int GROUP = 16 / GPB_;           // accumulator registers (rows) per group
    static constexpr bool MERGE = true;               // squared distances of SIFT rows rarely tie
    static constexpr bool SEEDED = true;
    typedef i32x4 frag;
    typedef i32x16 acc;
    typedef i32x4 list;
    static __device__ __forceinline__ acc zero() { return acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
    static __device__ __forceinline__ acc mfma(frag a, frag b, acc c)
    {
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ list empty() { return list{I8_EMPTY, I8_EMPTY, I8_EMPTY, I8_EMPTY}; }
    static __device__ __forceinline__ int max3(int a, int b, int c) { return max(max(a, b), c); }     // v_max3_i32
    static __device__ __forceinline__ void select(const acc& a, unsigned par, unsigned gid, list& cl, int g)
    {
        int m;
        if constexpr (GROUP == 4) {
            m = max(max3(a[4 * g], a[4 * g + 1], a[4 * g + 2]), a[4 * g + 3]);
        } else if constexpr (GROUP == 8) {
            const int m0 = max3(a[8 * g], a[8 * g + 1], a[8 * g + 2]);
            const int m1 = max3(m0, a[8 * g + 3], a[8 * g + 4]);
            m = max(max3(m1, a[8 * g + 5], a[8 * g + 6]), a[8 * g + 7]);
        } else {
            const int m0 = max3(a[0], a[1], a[2]), m1 = max3(a[3], a[4], a[5]), m2 = max3(a[6], a[7], a[8]);
            const int m3 = max3(a[9], a[10], a[11]), m4 = max3(a[12], a[13], a[14]);
            m = max(max3(m0, m1, m2), max3(m3, m4, a[15]));
        }
        (void)par;
        RouteI8::insert(cl, static_cast<int>((static_cast<unsigned>(m) << U8_SHIFT) | gid));   // v_lshl_or_b32
    }
    static __device__ __forceinline__ void put(list& cl, int x) { RouteI8::insert(cl, x); }
    static __device__ __forceinline__ void merge(list& cl, int h)
    {
        list own, other;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            own[i] = cl[i] == I8_EMPTY ? I8_EMPTY : (cl[i] | h);
            other[i] = __shfl_xor(own[i], 32, 64);
        }
        cl = own;
#pragma unroll
        for (int i = 0; i < 4; ++i) RouteI8::insert(cl, other[i]);
    }
};
typedef RouteU8T<4> RouteU8;

// a tile = 128 rows x R::ROW16 16-byte units, staged through registers by THREADS threads
template <typename R, int THREADS>
struct HTile {
    static constexpr int TOTAL = H_TT * R::ROW16;
    static constexpr int PIECES = (TOTAL + THREADS - 1) / THREADS;
    static constexpr bool EVEN = TOTAL % THREADS == 0;
    uint4 stg[PIECES];
    __device__ __forceinline__ void load(const uint4* __restrict__ Th, int tile, int tid)
    {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            int f = tid + THREADS * i;
            if (!EVEN) f = f < TOTAL ? f : TOTAL - 1;          // clamped: the load stays unconditional
            stg[i] = Th[static_cast<size_t>(tile) * TOTAL + f];  // rows are contiguous: piece f of the tile
        }
    }
    __device__ __forceinline__ void store(uint4* __restrict__ hsm, int buf, int tid) const
    {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int f = tid + THREADS * i;
            const int row = f / R::ROW16, c8 = f % R::ROW16;
            if (EVEN || f < TOTAL) hsm[(buf * H_TT + row) * R::LDS_ROW16 + c8] = stg[i];
        }
    }
};

// The same tile staged by LDS-DMA (global_load_lds_dwordx4: global -> LDS with no VGPR stop and no ds_write): the padded
// LDS image of a tile is 128 x LDS_ROW16 sixteen-byte slots = a whole number of 1-KiB pieces (38 for the f16 route,
// 34 for i8); one wave-instruction fills piece p, lane l writing slot 64p + l.  The destination is lane-linear, so the
// row padding lives in the SOURCE address: slot s belongs to row s / LDS_ROW16, column s % LDS_ROW16, and the lane that
// lands in a pad slot simply re-reads the row's last column (the pad is never read).  Per-lane source offsets are fixed
// for the whole kernel (wave w owns pieces w, w + NW, ...).
template <typename R, int THREADS>
struct HTileDma {
    static constexpr int NW = THREADS / 64;
    static constexpr int SLOTS = H_TT * R::LDS_ROW16;
    static_assert(SLOTS % 64 == 0, "the padded tile must be a whole number of 1-KiB pieces");
    static constexpr int NPIECES = SLOTS / 64;
    static constexpr int PER_WAVE = (NPIECES + NW - 1) / NW;
    static constexpr int TOTAL = H_TT * R::ROW16;
    unsigned src[PER_WAVE];                       // 16-byte units from the start of a tile in global memory
    __device__ __forceinline__ void init(int lane, int wave)
    {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int p = wave + NW * i;
            const int s = 64 * (p < NPIECES ? p : NPIECES - 1) + lane;
            const int row = s / R::LDS_ROW16, c = s % R::LDS_ROW16;
            src[i] = static_cast<unsigned>(row * R::ROW16 + (c < R::ROW16 ? c : R::ROW16 - 1));
        }
    }
    // MUBUF form (buffer_load_dwordx4 ... offen lds), not global_load_lds: hipcc files the FLAT-encoded LDS-DMA as a
    // pending FLAT access and from then on waits lgkmcnt(0) before every LDS operand use — the A-fragment ring (three
    // ds_read_b128 ahead) collapsed to one exposed LDS round trip per MFMA.  A buffer load only counts on vmcnt.
    // rsrc: the padded f16/i8 train copy (whole tiles), built from wave-uniform values; per-lane part = voffset.
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rsrc, int tile, uint4* __restrict__ hsm, int buf, int wave) const
    {
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int soff = tile * (TOTAL * 16);                // wave-uniform byte offset of the tile (< 2^31: checked on the host)
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int p = wave + NW * i;
            if (p < NPIECES) {                     // wave-uniform
                uint4* l = hsm + buf * SLOTS + 64 * p;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)l, 16, static_cast<int>(src[i]) * 16, soff, 0, 0);
            }
        }
    }
};

// one 32-row block of the tile: 9 k-chunks x NQB query blocks of MFMAs, selecting the previous
// block's accumulators p[] (group-id base pbase) in between, from chunk 1 on (see knn_tile_compute).
// tb: this lane's row in the LDS tile, in 16-byte units (chunk c = tb[2*c])
// sb (seeded routes): this lane's 64 bytes of the block's seeds in the LDS seed array (16-byte units): its 16 C-in
// registers, read once per block and shared by the NQB query blocks
struct Seed64 { uint4 v[4]; };
template <typename R, int NQB, bool EPI, typename ABL>
__device__ __forceinline__ void h_block(const uint4* __restrict__ tb, const uint4* __restrict__ sb,
                                        const typename R::frag (&qf)[NQB][R::NCH],
                                        typename R::acc (&a)[NQB], const typename R::acc (&p)[NQB], unsigned pbase,
                                        unsigned par, typename R::list (&cl)[NQB])
{
    typedef typename R::frag frag;
    typename R::acc seed = R::zero();
    if constexpr (R::SEEDED) seed = __builtin_bit_cast(typename R::acc, Seed64{{sb[0], sb[1], sb[2], sb[3]}});
    // A fragments run three chunks ahead of their use: one chunk is only 32*NQB pipe cycles, less
    // than an LDS round trip
    frag ring[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) ring[c] = *reinterpret_cast<const frag*>(tb + 2 * c);
#pragma unroll
    for (int c = 0; c < R::NCH; ++c) {
        const frag x = ring[c % 3];
        if constexpr (!ABL::no_ldsread) {
            if (c + 3 < R::NCH) ring[c % 3] = *reinterpret_cast<const frag*>(tb + 2 * (c + 3));
        }
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) a[qb] = R::mfma(x, qf[qb][c], c == 0 ? seed : a[qb]);
        if constexpr (ABL::no_epi) {
            if (EPI && c == 1) {                  // timing-only build: the accumulators stay live, nothing is selected
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(p[qb][e]));
            }
        }
        if (EPI && c >= 1 && !ABL::no_epi) {      // GPB*NQB row groups spread over chunks 1..NCH-1
            constexpr int NGB = R::GPB * NQB;
#pragma unroll
            for (int e = (c - 1) * NGB / (R::NCH - 1); e < c * NGB / (R::NCH - 1); ++e)   // query column e % NQB, group e / NQB
                R::select(p[e % NQB], par, (pbase + static_cast<unsigned>(e / NQB)) << 1, cl[e % NQB], e / NQB);
            if (NQB == 2)
                asm volatile("" : "+v"(cl[0][0]), "+v"(cl[0][1]), "+v"(cl[0][2]), "+v"(cl[0][3]), "+v"(cl[NQB - 1][0]),
                             "+v"(cl[NQB - 1][1]), "+v"(cl[NQB - 1][2]), "+v"(cl[NQB - 1][3]));
            else
                asm volatile("" : "+v"(cl[0][0]), "+v"(cl[0][1]), "+v"(cl[0][2]), "+v"(cl[0][3]));
        }
    }
}

// One 128-row tile = 4 blocks as ONE stream of 4 * NCH (block, chunk) steps (ring kernel).  h_block primes its
// A-fragment ring and reads its 16 seed registers at every block start: with NCH = 4 a block is only 128 matrix-pipe
// cycles per wave, and the eight waves of a workgroup reach the block boundary together, so each boundary exposed an
// LDS round trip behind a burst of 56 reads (stamps: 2100 cycles per tile against 1024 of MFMA).  Here the fragment
// of step s + 3 is requested at step s ACROSS block boundaries and the seeds of block b + 1 are requested right after
// block b's first MFMA has consumed block b's; only the tile boundary (a barrier: the next tile may not have landed)
// starts cold.  Selection of the previous block's accumulators as in h_block.
//   A, B  accumulator sets: blocks 0, 2 -> A, blocks 1, 3 -> B;  FIRST: no previous tile (block 0 selects nothing)
template <typename R, int NQB, bool FIRST, typename ABL>
__device__ __forceinline__ void h_tile(const uint4* __restrict__ tb, const uint4* __restrict__ sb,
                                       const typename R::frag (&qf)[NQB][R::NCH], typename R::acc (&A)[NQB],
                                       typename R::acc (&B)[NQB], unsigned lb, unsigned par, typename R::list (&cl)[NQB])
{
    typedef typename R::frag frag;
    typedef typename R::acc acc;
    constexpr int NCH = R::NCH, STEPS = 4 * NCH, PF = 3;
    constexpr unsigned G = R::GPB;
    auto a_ptr = [&](int step) { return reinterpret_cast<const frag*>(tb + (step / NCH) * 32 * R::LDS_ROW16 + 2 * (step % NCH)); };
    auto seed_of = [&](int blk) {
        if constexpr (R::SEEDED) {
            const uint4* p = sb + 8 * blk;
            return __builtin_bit_cast(acc, Seed64{{p[0], p[1], p[2], p[3]}});
        } else {
            return R::zero();
        }
    };
    frag ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = *a_ptr(i);
    acc seed = seed_of(0);
#pragma unroll
    for (int s_ = 0; s_ < STEPS; ++s_) {
        const int blk = s_ / NCH, c = s_ % NCH;
        const frag x = ring[s_ % PF];
        if constexpr (!ABL::no_ldsread) {
            if (s_ + PF < STEPS) ring[s_ % PF] = *a_ptr(s_ + PF);
        }
        acc(&cur)[NQB] = (blk & 1) ? B : A;
        const acc(&prev)[NQB] = (blk & 1) ? A : B;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) cur[qb] = R::mfma(x, qf[qb][c], c == 0 ? seed : cur[qb]);
        if (c == 0 && blk < 3) seed = seed_of(blk + 1);
        const bool epi = !(FIRST && blk == 0);
        if constexpr (ABL::no_epi) {
            if (epi && c == 1) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(prev[qb][e]));
            }
        }
        if (epi && c >= 1 && !ABL::no_epi) {
            constexpr int NGB = R::GPB * NQB;
            const unsigned pbase = lb + G * static_cast<unsigned>(blk) - G;       // the previous block's group ids
#pragma unroll
            for (int e = (c - 1) * NGB / (NCH - 1); e < c * NGB / (NCH - 1); ++e)
                R::select(prev[e % NQB], par, (pbase + static_cast<unsigned>(e / NQB)) << 1, cl[e % NQB], e / NQB);
            if (NQB == 2)
                asm volatile("" : "+v"(cl[0][0]), "+v"(cl[0][1]), "+v"(cl[0][2]), "+v"(cl[0][3]), "+v"(cl[NQB - 1][0]),
                             "+v"(cl[NQB - 1][1]), "+v"(cl[NQB - 1][2]), "+v"(cl[NQB - 1][3]));
            else
                asm volatile("" : "+v"(cl[0][0]), "+v"(cl[0][1]), "+v"(cl[0][2]), "+v"(cl[0][3]));
        }
    }
}

// NQB query blocks (of 32) per wave: 2 -> 4 waves per workgroup, 2 waves per SIMD (each A fragment
// feeds two MFMAs); 1 -> 8 waves per workgroup, 4 waves per SIMD at <= 128 VGPRs (more waves to
// cover LDS / barrier / MFMA-dependency latency).  Either way a workgroup owns H_QB = 256 queries.
// GR = 2 (with NQB = 2): TWO groups of four waves own the same 256 queries and split every tile's rows between them
// (group g takes the 32-row blocks 2g, 2g+1): 8 waves = 2 per SIMD in ONE workgroup per CU, every A fragment still
// feeds two MFMAs (half the LDS operand reads of the NQB = 1 form, which are what bounds it: 1 KiB per MFMA at
// 128 B/clk is exactly the matrix pipe's time), and the two groups' lists meet in LDS once, at the end — the split
// count (and with it the refinement's input) stays that of one workgroup per CU.
// mode bit 0: 0 = run always (hint / i8), 1 = run only if prep16 found the data eligible (auto); bit 1: XCD-tiled grid
template <typename R, int NQB, bool DMA, int GR, typename ABL>
__global__ __launch_bounds__(H_QB / (32 * NQB) * 64 * GR, (R::NCH > 9 ? (NQB == 2 ? 1 : 2) : (NQB == 2 ? 2 : 4))) void knn_mfma_rows288(
    const uint4* __restrict__ Qh, const uint4* __restrict__ Th, const uint4* __restrict__ seeds_g, int nq, int nt,
    int tiles_per_split, unsigned par, typename R::list* __restrict__ cand_val, int slots,
    const unsigned long long* __restrict__ stats, unsigned epoch, int mode)
{
    typedef typename R::frag frag;
    typedef typename R::acc acc;
    typedef typename R::list list;
    constexpr int WPG = H_QB / (32 * NQB);                 // waves per row group
    constexpr int THREADS = WPG * 64 * GR;
    static_assert(GR == 1 || (GR == 2 && NQB == 2 && R::MERGE), "row groups: the 64-query form of a merged-list route");
    if (mode & 1) {
        const unsigned long long s1 = stats[1], s3 = stats[3];
        if (static_cast<unsigned>(s1 >> 32) == epoch && (s1 & 2ull) && static_cast<unsigned>(s3 >> 32) == epoch)
            return;                    // not integer-valued AND the rounded-copy route withdrew: the f32 kernel takes over
    }
    extern __shared__ __attribute__((aligned(16))) uint4 hsm[];                   // [2][H_TT][R::LDS_ROW16] (+ [2][64] seeds)
    static_assert(!R::SEEDED || DMA, "the seeded routes stage by LDS-DMA only");
    constexpr int TILE_SLOTS = H_TT * R::LDS_ROW16;
    uint4* const ssm = hsm + 2 * TILE_SLOTS;                                      // seeded routes: [2][64] (1 KiB per DMA piece, 512 B used)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int gw = wave % WPG, grp = wave / WPG;           // wave inside its row group, row group
    const WgTile wg = wg_tile((mode & 2) != 0);
    const int qbase = wg.qb * H_QB + gw * 32 * NQB;

    // both global streams are requested before anything waits: the first train tile, then the query fragments
    HTile<R, THREADS> st;
    HTileDma<R, THREADS> dma;
    const int ntiles = (nt + H_TT - 1) / H_TT;
    const int tile0 = wg.split * tiles_per_split;
    const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(Th), 0, ntiles * (H_TT * R::ROW16 * 16), 0x00020000);
    // seeds: one more 1-KiB piece per tile, issued by the last wave (the source array carries one tile of slack)
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(R::SEEDED ? seeds_g : Th), 0, (ntiles + 1) * SEED_TILE_BYTES, 0x00020000);
    auto seed_issue = [&](int tile, int buf) {
        typedef __attribute__((address_space(3))) void* lptr_t;
        if (R::SEEDED && wave == THREADS / 64 - 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lptr_t)(ssm + 64 * buf), 16, lane * 16, tile * SEED_TILE_BYTES, 0, 0);
    };
    if (DMA) {
        dma.init(lane, wave);
        dma.issue(t_rsrc, tile0 < ntiles ? tile0 : ntiles - 1, hsm, 0, wave);
        seed_issue(tile0 < ntiles ? tile0 : ntiles - 1, 0);
    } else {
        st.load(Th, tile0 < ntiles ? tile0 : ntiles - 1, tid);      // unconditional (clamped): nt >= 1
    }
    frag qf[NQB][R::NCH];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int c = 0; c < R::NCH; ++c)
            qf[qb][c] = *reinterpret_cast<const frag*>(Qh + static_cast<size_t>(qbase + 32 * qb + r) * R::ROW16 + 2 * c + h);
    // make the fragments opaque: hipcc otherwise treats the loads as rematerialisable and re-reads
    // some of them from global memory inside the tile loop
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int c = 0; c < R::NCH; ++c) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 t = __builtin_bit_cast(u32x4, qf[qb][c]);
            asm volatile("" : "+v"(t));
            qf[qb][c] = __builtin_bit_cast(frag, t);
        }

    int tile1 = tile0 + tiles_per_split;
    if (tile1 > ntiles) tile1 = ntiles;
    list cl[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) cl[qb] = R::empty();

    if (tile0 < tile1) {
        if (!DMA) st.store(hsm, 0, tid);
        __syncthreads();                                        // (with DMA in flight the barrier's fence waits vmcnt(0))
        acc A[NQB], B[NQB];
        for (int tix = 0; tix < tile1 - tile0; ++tix) {
            const int buf = tix & 1;
            const uint4* tb = hsm + (buf * H_TT + r) * R::LDS_ROW16 + h;
            constexpr unsigned G = R::GPB;
            const unsigned lb = static_cast<unsigned>(tix) * (4u * G);     // group ids of this tile: lb + GPB*blk + g
            // the last tile is simply staged again: past the end nothing reads the other buffer
            if constexpr (!ABL::no_stage) {
                // the other buffer was last read in tile tix - 1, and every wave has passed that tile's barrier
                if (DMA) {
                    dma.issue(t_rsrc, tile0 + tix + 1 < ntiles ? tile0 + tix + 1 : tile0 + tix, hsm, buf ^ 1, wave);
                    seed_issue(tile0 + tix + 1 < ntiles ? tile0 + tix + 1 : tile0 + tix, buf ^ 1);
                } else st.load(Th, tile0 + tix + 1 < ntiles ? tile0 + tix + 1 : tile0 + tix, tid);
            }
            const uint4* sb = ssm + 64 * buf + 4 * h;             // block b of the tile: sb + 8 * b
            if constexpr (GR == 1) {
                if (tix == 0) h_block<R, NQB, false, ABL>(tb, sb, qf, A, A, 0u, par, cl);
                else h_block<R, NQB, true, ABL>(tb, sb, qf, A, B, lb - G, par, cl);                 // B = block 3 of tile-1
                h_block<R, NQB, true, ABL>(tb + 32 * R::LDS_ROW16, sb + 8, qf, B, A, lb, par, cl);
                h_block<R, NQB, true, ABL>(tb + 64 * R::LDS_ROW16, sb + 16, qf, A, B, lb + G, par, cl);
                if constexpr (!ABL::no_stage) {
                    if (!DMA) st.store(hsm, buf ^ 1, tid);
                }
                h_block<R, NQB, true, ABL>(tb + 96 * R::LDS_ROW16, sb + 24, qf, B, A, lb + 2u * G, par, cl);
            } else {
                // this group's blocks 2*grp and 2*grp + 1 of the tile (group ids lb + G*block + g, as above)
                const uint4* tg = tb + grp * 64 * R::LDS_ROW16;
                const uint4* sg = sb + 16 * grp;
                const unsigned gb = lb + 2u * static_cast<unsigned>(grp) * G;
                if (tix == 0) h_block<R, NQB, false, ABL>(tg, sg, qf, A, A, 0u, par, cl);
                else h_block<R, NQB, true, ABL>(tg, sg, qf, A, B, gb - 3u * G, par, cl);            // B = block 2*grp+1 of tile-1
                if constexpr (!ABL::no_stage) {
                    if (!DMA) st.store(hsm, buf ^ 1, tid);
                }
                h_block<R, NQB, true, ABL>(tg + 32 * R::LDS_ROW16, sg + 8, qf, B, A, gb, par, cl);
            }
            tile_barrier<ABL>();
        }
        const unsigned lb = static_cast<unsigned>(tile1 - tile0 - 1) * (4u * R::GPB) +
                            (GR == 1 ? 3u : 2u * static_cast<unsigned>(grp) + 1u) * R::GPB;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int g = 0; g < R::GPB; ++g) R::select(B[qb], par, (lb + static_cast<unsigned>(g)) << 1, cl[qb], g);
    }
    if constexpr (GR == 2) {
        // the two row groups' lists of a query meet in LDS (the tile buffers are idle: every wave is past the last
        // tile's barrier, whose fence also drained the LDS-DMA)
        list* xs = reinterpret_cast<list*>(hsm);
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            R::merge(cl[qb], h);
            if (grp == 1) xs[(gw * NQB + qb) * 64 + lane] = cl[qb];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const list o = xs[(gw * NQB + qb) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) R::put(cl[qb], o[i]);
            const int q = qbase + 32 * qb + r;
            if (q < nq && h == 0) cand_val[(static_cast<size_t>(q) * slots + wg.split * KNN_C) / KNN_C] = cl[qb];
        }
        return;
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int q = qbase + 32 * qb + r;
        R::merge(cl[qb], h);
        if (R::MERGE) {
            if (q < nq && h == 0) cand_val[(static_cast<size_t>(q) * slots + wg.split * KNN_C) / KNN_C] = cl[qb];
        } else {
            if (q < nq) cand_val[(static_cast<size_t>(q) * slots + (wg.split * 2 + h) * KNN_C) / KNN_C] = cl[qb];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Ring form of the row-streaming kernel (round 3).  With 4 k-chunks per block (u8 route) a 128-row tile is only
// ~1000 matrix-pipe cycles per SIMD — less than one LDS-DMA round trip — so the double-buffered form above, which
// requests tile t+1 when tile t starts and waits vmcnt(0) at tile t's barrier, exposes the memory latency once per
// tile.  Here the tiles live in a ring of NBUF LDS buffers (u8: 8 x 19 KiB = the whole 1024-row split of config C3):
//   prologue     query fragments, then tiles 0 .. NBUF-2 are requested back to back;
//   tile t       s_waitcnt vmcnt(own pieces of the tiles after t still allowed in flight) ; s_barrier ;
//                request tile t-1+NBUF into the buffer tile t-1 just left ; 4 blocks of MFMAs + selection.
// One barrier per tile serves both hazards (every wave's pieces of tile t have landed; every wave is done with tile
// t-1), the waits are COUNTED (vmcnt retires in order; a wave issues the same number of pieces for every tile), and no
// fence is involved (a __syncthreads() would drain vmcnt to 0).  GR = 1 only.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `ahead` tiles' worth of this wave's requests are in flight; a wave issues BASE or BASE + 1
// vector-memory operations per tile (`plus`), ahead <= 6
template <int BASE>
__device__ __forceinline__ void wait_tiles(int ahead, bool plus)
{
    static_assert(7 * (BASE + 1) <= 63, "vmcnt is a 6-bit counter");
#define PM_W(A_) case A_: if (plus) wait_vmcnt_imm<A_ * (BASE + 1)>(); else wait_vmcnt_imm<A_ * BASE>(); break;
    switch (ahead) {
        PM_W(1) PM_W(2) PM_W(3) PM_W(4) PM_W(5) PM_W(6) PM_W(7)
        default: wait_vmcnt_imm<0>(); break;
    }
#undef PM_W
}

// 16-byte global load the compiler does not track: no s_waitcnt is generated for it, so it can stay in flight behind
// later LDS-DMA requests (a tracked load issued before a loop of requests makes hipcc wait vmcnt(0) at its first use,
// i.e. for every request of the loop).  The caller waits (wait_tiles) and then passes the registers through
// pin_after_wait() before their first use.
template <typename FRAG>
__device__ __forceinline__ FRAG untracked_load16(const void* p)
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return __builtin_bit_cast(FRAG, v);
}
template <typename FRAG>
__device__ __forceinline__ void pin_after_wait(FRAG& f)
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t = __builtin_bit_cast(u32x4, f);
    asm volatile("" : "+v"(t)::"memory");
    f = __builtin_bit_cast(FRAG, t);
}

// WAVES waves of 32 * NQB queries each: 8 x 32 or 4 x 64 (256 queries per workgroup, as knn_mfma_rows288), or 16 x 32
// (512 queries: four waves per SIMD cover each other's LDS and issue stalls; twice the splits, half the tiles each).
// SPLIT: no workgroup barrier inside the sweep.  Stamps of the barrier form: 2100 cycles per tile with the barrier,
// 1520 without it (8 waves; 3280 against 1710 with 16) — waves that meet every ~1000 matrix-pipe cycles spend a third of
// their time waiting for the slowest of them.  The split-phase form keeps two monotonic counters per ring buffer in LDS:
//   arrive[b]  += 1 by every wave once ITS pieces of the tile in buffer b have landed (counted vmcnt wait), done one
//              tile AHEAD: a wave signals tile t + 1 before it starts computing tile t;
//   done[b]    += 1 by every wave when it has finished the tile in buffer b (only read when a buffer is re-used).
// A wave entering tile t polls arrive[t % NBUF] >= WAVES * (t / NBUF + 1) — normally true for a whole tile time
// already — and a wave requesting tile j >= NBUF polls done[j % NBUF] >= WAVES * (j / NBUF).  Waves drift apart by up
// to a tile (NBUF - 1 for the buffers) instead of meeting at every tile; LDS operations of a wave execute in order,
// so the counter read orders the tile reads behind it.
template <typename R, int NQB, int WAVES, int NBUF, bool SPLIT, typename ABL>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void knn_mfma_ring(
    const uint4* __restrict__ Qh, const uint4* __restrict__ Th, const uint4* __restrict__ seeds_g, int nq, int nt,
    int tiles_per_split, unsigned par, typename R::list* __restrict__ cand_val, int slots, int mode, int pro)
{
    typedef typename R::frag frag;
    typedef typename R::acc acc;
    typedef typename R::list list;
    constexpr int THREADS = WAVES * 64;
    constexpr int QB_WG = WAVES * 32 * NQB;                  // queries per workgroup
    constexpr int TILE_SLOTS = H_TT * R::LDS_ROW16;
    typedef HTileDma<R, THREADS> Dma;
    static_assert(NBUF >= 3 && NBUF <= 8, "ring of three to eight buffers (wait_tiles covers up to 7 tiles in flight)");
    extern __shared__ __attribute__((aligned(16))) uint4 hsm[];                   // [NBUF][H_TT][R::LDS_ROW16] + [NBUF][64] seeds
    uint4* const ssm = hsm + NBUF * TILE_SLOTS;
    int* const arrive = reinterpret_cast<int*>(ssm + NBUF * 64);                  // [NBUF] arrive + [NBUF] done (SPLIT)
    int* const done = arrive + NBUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const WgTile wg = wg_tile((mode & 2) != 0);
    const int qbase = wg.qb * QB_WG + wave * 32 * NQB;
    ABL::stamp(0);
    if constexpr (SPLIT) {                                    // counters start at zero; the only workgroup barrier of the kernel
        if (tid < 2 * NBUF) arrive[tid] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    auto poll = [&](int* ctr, int target) {                  // wave-uniform spin on an LDS counter
        for (;;) {
            const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (v >= target) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    auto bump = [&](int* ctr) {
        if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u32, no return
    };

    const int ntiles = (nt + H_TT - 1) / H_TT;
    const int tile0 = wg.split * tiles_per_split;
    int tile1 = tile0 + tiles_per_split;
    if (tile1 > ntiles) tile1 = ntiles;
    const int ntl = tile1 - tile0;

    // query fragments: register loads, while every later operation is an LDS-DMA piece.  The two kinds do NOT retire in one
    // order (knn_u8_rega, form 5: a counted wait over a mix of them returned early), so the FIRST wait of the sweep is a
    // full drain, vmcnt(0) — it covers the fragments and the prologue's tiles; from then on only pieces are in flight and the
    // waits are counted.
    frag qf[NQB][R::NCH];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int c = 0; c < R::NCH; ++c)
            qf[qb][c] = untracked_load16<frag>(Qh + static_cast<size_t>(qbase + 32 * qb + r) * R::ROW16 + 2 * c + h);

    const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(Th), 0, ntiles * (H_TT * R::ROW16 * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(R::SEEDED ? seeds_g : Th), 0, (ntiles + 1) * SEED_TILE_BYTES, 0x00020000);
    Dma dma;
    dma.init(lane, wave);
    // vector-memory operations this wave issues per tile: its share of the NPIECES pieces (+ the seed piece on the last
    // wave) = BASE or BASE + 1
    constexpr int BASE = Dma::NPIECES / Dma::NW;
    const int own = (Dma::NPIECES - wave + Dma::NW - 1) / Dma::NW + ((R::SEEDED && wave == Dma::NW - 1) ? 1 : 0);
    const bool plus = __builtin_amdgcn_readfirstlane(own) != BASE;
    static_assert(Dma::NPIECES >= Dma::NW, "every wave issues at least one piece");
    auto request = [&](int t) {                              // tile t of the split -> ring buffer t % NBUF
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int buf = t % NBUF;
        dma.issue(t_rsrc, tile0 + t, hsm, buf, wave);        // (buffer stride = TILE_SLOTS: HTileDma::SLOTS)
        if (R::SEEDED && wave == Dma::NW - 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lptr_t)(ssm + 64 * buf), 16, lane * 16, (tile0 + t) * SEED_TILE_BYTES, 0, 0);
    };
    // tile j may be requested once tile j - NBUF is done.  `next` = first tile not yet requested.  The prologue asks
    // for `pro` tiles only (a request costs the issuing wave ~100-200 cycles while the memory pipe is busy: asking for the
    // whole ring up front would keep every wave from its first MFMA for ~4000 cycles); the sweep then asks for up to
    // two more per tile until the ring is full.
    int next = 0;
    auto pump = [&](int done_below, int most) {             // tiles < done_below are finished (by THIS wave when SPLIT)
        for (int n = 0; n < most && next < ntl && next < done_below + NBUF; ++n) {
            if constexpr (SPLIT) {
                if (next >= NBUF) poll(&done[next % NBUF], WAVES * (next / NBUF));    // every wave has left the buffer's last tile
            }
            request(next);
            ++next;
        }
    };
    pump(0, pro);
    if constexpr (SPLIT) {
        if (ntl > 0) {                                       // arrive for tile 0 (later tiles: one tile ahead, inside the sweep)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the full drain: see the query fragments above)
            bump(&arrive[0]);
        }
    }
    ABL::stamp(1);

    list cl[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) cl[qb] = R::empty();

    if (ntl > 0) {
        acc A[NQB], B[NQB];
        constexpr unsigned G = R::GPB;
#pragma unroll 1
        for (int tix = 0; tix < ntl; ++tix) {
            if constexpr (SPLIT) {
                // every wave's pieces of tile tix have landed?  (this wave arrived for it one tile ago)
                if constexpr (!ABL::no_barrier) poll(&arrive[tix % NBUF], WAVES * (tix / NBUF + 1));
                if constexpr (!ABL::no_stage) pump(tix, 2);
                if (tix + 1 < ntl) {                         // arrive for tile tix + 1 (requested: the pump keeps two tiles ahead)
                    wait_tiles<BASE>(next - 2 - tix, plus);
                    bump(&arrive[(tix + 1) % NBUF]);
                }
            } else {
                // tiles after tix that may stay in flight: everything requested so far beyond tix
                if (tix == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the full drain (query fragments: see above)
                else wait_tiles<BASE>(next - 1 - tix, plus);
                if constexpr (!ABL::no_barrier) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int c = 0; c < R::NCH; ++c) pin_after_wait(qf[qb][c]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!SPLIT && !ABL::no_stage) pump(tix, 2);
            if (tix < 14) ABL::stamp(2 + tix);
            const int buf = tix % NBUF;
            const uint4* tb = hsm + (buf * H_TT + r) * R::LDS_ROW16 + h;
            const uint4* sb = ssm + 64 * buf + 4 * h;
            const unsigned lb = static_cast<unsigned>(tix) * (4u * G);
            if (tix == 0) h_tile<R, NQB, true, ABL>(tb, sb, qf, A, B, lb, par, cl);
            else h_tile<R, NQB, false, ABL>(tb, sb, qf, A, B, lb, par, cl);
            if constexpr (SPLIT) {
                if (tix + NBUF < ntl) bump(&done[tix % NBUF]);                   // (only a re-used buffer is ever polled)
            }
        }
        const unsigned lb = static_cast<unsigned>(ntl - 1) * (4u * G) + 3u * G;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int g = 0; g < R::GPB; ++g) R::select(B[qb], par, (lb + static_cast<unsigned>(g)) << 1, cl[qb], g);
    }
    ABL::stamp(16);
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int q = qbase + 32 * qb + r;
        R::merge(cl[qb], h);
        if (R::MERGE) {
            if (q < nq && h == 0) cand_val[(static_cast<size_t>(q) * slots + wg.split * KNN_C) / KNN_C] = cl[qb];
        } else {
            if (q < nq) cand_val[(static_cast<size_t>(q) * slots + (wg.split * 2 + h) * KNN_C) / KNN_C] = cl[qb];
        }
    }
    ABL::stamp(17);
}

// the DMA form addresses a train copy through a buffer descriptor with 32-bit byte offsets
template <typename R>
inline bool rows288_dma_ok(int nt) { return (static_cast<long long>(nt) + H_TT) * (R::ROW16 * 16) < 0x7FFFFFFFLL; }

// PM_OPT_KNN_XCD_TILE: 1 = launch order, 2 = tiled (the default; wg_tile falls back when the grid shape does not divide)
inline bool xcd_tiled(const pm_ctx* ctx) { return ctx->opts[PM_OPT_KNN_XCD_TILE] != 1; }

template <int NCH, bool FULL, int TT, typename ABL>
int launch_mfma(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim, const float* tnorm, int splits,
                int tiles_per_split, unsigned keep_mask, float* cval, int slots, const unsigned long long* stats,
                unsigned epoch, int only_if_ineligible)
{
    constexpr int LDT = NCH * 8 + 4;
    const size_t lds = 2 * TT * LDT * sizeof(float);
    static bool attr_done_dev[PM_MAX_DEVICES] = {};          // the attribute is per device (and per template instance)
    bool& attr_done = attr_done_dev[ctx->device];
    if (!attr_done) {
        PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_l2_mfma<NCH, FULL, TT, ABL>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        attr_done = true;
    }
    dim3 grid((nq + QB - 1) / QB, splits);
    // a grid that is resident at once takes the XCD-tiled workgroup order (see wg_tile)
    const int mode = (only_if_ineligible ? 1 : 0) | (xcd_tiled(ctx) ? 2 : 0);
    pm::ScopedKernelTime t(ctx, "knn_l2_mfma");
    hipLaunchKernelGGL((knn_l2_mfma<NCH, FULL, TT, ABL>), grid, dim3(256), lds, ctx->stream, dq, dt, tnorm, nq, nt, dim,
                       tiles_per_split, keep_mask, cval, slots, stats, epoch, mode);
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

template <typename ABL>
int coarse_f32_dispatch(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim, const float* tnorm,
                      int splits, int tiles_per_split, unsigned keep_mask, float* cval, int slots,
                      const unsigned long long* stats, unsigned epoch, int only_if_ineligible)
{
#define PM_LAUNCH_MFMA(NCH_, FULL_)                                                                              \
    launch_mfma<NCH_, FULL_, TT32, ABL>(ctx, dq, nq, dt, nt, dim, tnorm, splits, tiles_per_split, keep_mask, cval, slots, \
                                   stats, epoch, only_if_ineligible)
    if (dim == 128) return PM_LAUNCH_MFMA(16, true);
    if (dim == 64) return PM_LAUNCH_MFMA(8, true);
    if (dim == 32) return PM_LAUNCH_MFMA(4, true);
    if (dim < 32) return PM_LAUNCH_MFMA(4, false);
    if (dim < 64) return PM_LAUNCH_MFMA(8, false);
    return PM_LAUNCH_MFMA(16, false);
#undef PM_LAUNCH_MFMA
}

template <typename R, typename ABL>
int launch_rows288(pm_ctx* ctx, const char* name, const void* Qh, const void* Th, const void* seeds, int nq, int nq_pad, int nt,
                   int splits, int tiles_per_split, unsigned par, void* cval, int slots, const unsigned long long* stats,
                   unsigned epoch, int mode)
{
    const size_t lds = sizeof(uint4) * (2 * H_TT * R::LDS_ROW16 + (R::SEEDED ? 2 * 64 : 0));
    // few tiles per workgroup: the 8-wave form covers latency better; long sweeps: the 4-wave form halves LDS reads
    const int nqb_opt = ctx->opts[PM_OPT_KNN_F16_WAVES];
    const int nqb = nqb_opt ? nqb_opt : (tiles_per_split <= 8 ? 1 : 2);          // 3: two row groups of 4 waves x 64 queries
    static bool attr_done_dev[PM_MAX_DEVICES] = {};          // the attribute is per device (and per template instance)
    bool& attr_done = attr_done_dev[ctx->device];
    if (!attr_done) {
#define PM_ATTR(NQB_, DMA_, GR_)                                                                                  \
    PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_mfma_rows288<R, NQB_, DMA_, GR_, ABL>),        \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)))
        PM_ATTR(1, true, 1); PM_ATTR(2, true, 1);
        if constexpr (!R::SEEDED) { PM_ATTR(1, false, 1); PM_ATTR(2, false, 1); }
        if constexpr (R::MERGE) {
            PM_ATTR(2, true, 2);
            if constexpr (!R::SEEDED) PM_ATTR(2, false, 2);
        }
#undef PM_ATTR
        attr_done = true;
    }
    pm::ScopedKernelTime t(ctx, name);
    const uint4* q4 = static_cast<const uint4*>(Qh);
    const uint4* t4 = static_cast<const uint4*>(Th);
    const uint4* s4 = static_cast<const uint4*>(seeds);
    typename R::list* out = static_cast<typename R::list*>(cval);
    // train tiles by LDS-DMA unless pinned to register staging (measured: C3 f16 21.3 -> 19.3 us, 32k x 32k f16 233 -> 210 us,
    // C4 i8 211 -> 198 us: the ds_write_b128 pass and 16 staging VGPRs disappear)
    // (the DMA form addresses the train copy through a buffer descriptor with 32-bit byte offsets: below 2 GiB only;
    // the seeded routes exist in the DMA form only: their callers check rows288_dma_ok first)
    const bool dma = R::SEEDED || (ctx->opts[PM_OPT_KNN_STAGING] != 1 && rows288_dma_ok<R>(nt));
    PM_REQUIRE(!R::SEEDED || rows288_dma_ok<R>(nt), PM_E_UNSUPPORTED, "train set too large for the seeded coarse routes");
#define PM_GO(NQB_, DMA_, GR_, THREADS_)                                                                           \
    hipLaunchKernelGGL((knn_mfma_rows288<R, NQB_, DMA_, GR_, ABL>), dim3(nq_pad / H_QB, splits), dim3(THREADS_), lds,  \
                       ctx->stream, q4, t4, s4, nq, nt, tiles_per_split, par, out, slots, stats, epoch, mode)
#define PM_GO2(NQB_, GR_, THREADS_)                                                                                \
    do {                                                                                                           \
        if constexpr (R::SEEDED) { PM_GO(NQB_, true, GR_, THREADS_); }                                             \
        else { if (dma) PM_GO(NQB_, true, GR_, THREADS_); else PM_GO(NQB_, false, GR_, THREADS_); }                \
    } while (0)
    mode = (mode ? 1 : 0) | (xcd_tiled(ctx) ? 2 : 0);     // see wg_tile
    bool grouped = false;
    if constexpr (R::MERGE) {
        if (nqb == 3) {
            grouped = true;
            PM_GO2(2, 2, 512);
        }
    }
    if (grouped) { }
    else if (nqb == 2) PM_GO2(2, 1, 256);
    else PM_GO2(1, 1, 512);
#undef PM_GO2
#undef PM_GO
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

// ring form (knn_mfma_ring): NBUF buffers of one tile (+ 1 KiB of seeds) each.  qb_wg: queries per workgroup the caller
// sized nq_pad and the splits for (ring_qb_wg below).
inline int ring_qb_wg(const pm_ctx* ctx) { return ctx->opts[PM_OPT_KNN_F16_WAVES] == 3 ? 512 : 256; }
template <typename R, int NBUF, typename ABL>
int launch_ring(pm_ctx* ctx, const char* name, const void* Qh, const void* Th, const void* seeds, int nq, int nq_pad, int nt,
                int splits, int tiles_per_split, unsigned par, void* cval, int slots)
{
    const size_t lds = static_cast<size_t>(NBUF) * (sizeof(uint4) * H_TT * R::LDS_ROW16 + 1024) + 64;      // + the counters
    static_assert(static_cast<size_t>(NBUF) * (sizeof(uint4) * H_TT * R::LDS_ROW16 + 1024) + 64 <= 160 * 1024, "ring exceeds the CU's LDS");
    const bool split = ctx->opts[PM_OPT_KNN_RING] == 3;      // 2: workgroup barrier per tile; 3: split-phase LDS counters
    PM_REQUIRE(rows288_dma_ok<R>(nt), PM_E_UNSUPPORTED, "train set too large for the LDS-DMA coarse routes");
    const int form = ctx->opts[PM_OPT_KNN_F16_WAVES];       // 0 / 1: 8 waves x 32 queries, 2: 4 x 64, 3: 16 x 32
    const int qb_wg = ring_qb_wg(ctx);
    PM_REQUIRE(nq_pad % qb_wg == 0, PM_E_INVALID, "query padding does not match the workgroup size");
    static bool attr_done_dev[PM_MAX_DEVICES] = {};
    bool& attr_done = attr_done_dev[ctx->device];
    if (!attr_done) {
#define PM_ATTR(NQB_, WAVES_)                                                                                        \
    PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_mfma_ring<R, NQB_, WAVES_, NBUF, false, ABL>),    \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                \
    PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_mfma_ring<R, NQB_, WAVES_, NBUF, true, ABL>),     \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)))
        PM_ATTR(1, 8); PM_ATTR(2, 4); PM_ATTR(1, 16);
#undef PM_ATTR
        attr_done = true;
    }
    pm::ScopedKernelTime t(ctx, name);
    const uint4* q4 = static_cast<const uint4*>(Qh);
    const uint4* t4 = static_cast<const uint4*>(Th);
    const uint4* s4 = static_cast<const uint4*>(seeds);
    typename R::list* out = static_cast<typename R::list*>(cval);
    const int mode = xcd_tiled(ctx) ? 2 : 0;
    int pro = ctx->opts[PM_OPT_KNN_RING_PROLOGUE] ? ctx->opts[PM_OPT_KNN_RING_PROLOGUE] : 2;   // tiles requested up front
    if (pro < 2) pro = 2;                                    // (the split-phase form signals one tile ahead)
#define PM_GO1(NQB_, WAVES_, SPLIT_)                                                                                \
    hipLaunchKernelGGL((knn_mfma_ring<R, NQB_, WAVES_, NBUF, SPLIT_, ABL>), dim3(nq_pad / qb_wg, splits), dim3(WAVES_ * 64), lds, \
                       ctx->stream, q4, t4, s4, nq, nt, tiles_per_split, par, out, slots, mode, pro)
#define PM_GO(NQB_, WAVES_) do { if (split) PM_GO1(NQB_, WAVES_, true); else PM_GO1(NQB_, WAVES_, false); } while (0)
    if (form == 3) PM_GO(1, 16);
    else if (form == 2) PM_GO(2, 4);
    else PM_GO(1, 8);
#undef PM_GO
#undef PM_GO1
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

// ---------------------------------------------------------------------------------------------
// Register-operand form of the u8 coarse pass (round 3, PM_OPT_KNN_RING = 4).  The stamps and ablations of the two
// LDS forms above say the same thing three times: at the i8 rate the hand-off of a tile between the waves that fill it
// and the waves that read it costs as much as the tile's MFMAs, whatever the primitive.  With 128-byte rows the
// hand-off is not needed at all: a wave can hold 128 queries as B operands for the whole kernel (4 blocks x 4 k-chunks
// x 4 VGPRs = 64 registers) and read its A operands — 32 train rows = 4 KiB per block — STRAIGHT FROM GLOBAL MEMORY into
// registers (4 x global_load_dwordx4 per lane, one block ahead), plus the block's 16 C-in seeds (4 more loads, two
// distinct addresses per instruction).  The eight waves of a workgroup own the same 128 queries and take the 32-row
// blocks of the split in turn (wave w: blocks w, w + 8, ...), so every train row still enters the CU once per workgroup,
// nothing is staged in LDS, and there is no barrier before the final merge of the eight waves' candidate lists.
// Same candidate format and numbering as knn_mfma_rows288<RouteU8T<GPB>> (group id = block-in-split * GPB + g).
// ---------------------------------------------------------------------------------------------
constexpr int RA_NQB = 4;                // query blocks per wave: 128 queries per workgroup
constexpr int RA_WAVES = 8;
constexpr int RA_QB = 32 * RA_NQB;

// PRIV: the A operands go through a PRIVATE LDS buffer of the wave instead (LDS-DMA, five coalesced 1-KiB pieces per block:
// a dwordx4 load whose lanes sit 128 bytes apart touches 32 cache lines for 1 KiB and the texture path takes a line per
// clock — measured: the straight-to-register form is 2x SLOWER than the tile kernels at 32k x 32k).  Still no hand-off:
// a wave waits for its own pieces only (counted vmcnt), reads them back with ds_read_b128 and never meets another wave
// before the final merge.
constexpr int RA_PIECES = 5;                                   // 32 rows x 144 B = 4.5 KiB
constexpr int RA_PRIV_SLOTS = RA_PIECES * 64;                  // 16-byte slots per private buffer

// WSPLIT: every WAVE owns a train split of its own (split = 8 * blockIdx.y + wave: up to 2048 consecutive rows) instead of
// every eighth block of the workgroup's split: the waves of a workgroup then share nothing but their 128 queries, there
// is no merge and no barrier at all, and a wave's entry cost (16 query-fragment loads) is spread over up to 64 blocks
// instead of 8.  Needs long sweeps: the host takes it when a split is >= 1024 rows with every CU busy.
template <int GPB_, bool PRIV, bool WSPLIT, typename ABL>
__global__ __launch_bounds__(RA_WAVES * 64, 1) void knn_u8_rega(const uint4* __restrict__ Q8, const uint4* __restrict__ T8,
                                                               const uint4* __restrict__ seeds_g, int nq, int nt, int tiles_per_split,
                                                               i32x4* __restrict__ cand_val, int slots, int mode)
{
    typedef RouteU8T<GPB_> R;
    typedef typename R::frag frag;
    typedef typename R::acc acc;
    typedef typename R::list list;
    __shared__ list xs[RA_WAVES][RA_NQB][64];
    extern __shared__ __attribute__((aligned(16))) uint4 psm[];       // PRIV: [RA_WAVES][3][RA_PRIV_SLOTS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const WgTile wg = wg_tile((mode & 2) != 0);
    // WSPLIT: waves w and w + 4 sweep the SAME split for two different sets of 128 queries (a workgroup = 256 queries x 4
    // splits): they run in step, so the second request for a piece is served by the CU's L1 and the L2 -> CU traffic per
    // descriptor pair is that of the 256-query tile kernels
    const int qbase = WSPLIT ? wg.qb * (2 * RA_QB) + (wave >> 2) * RA_QB : wg.qb * RA_QB;

    const int ntiles = (nt + H_TT - 1) / H_TT;
    const int my_split = WSPLIT ? wg.split * (RA_WAVES / 2) + (wave & 3) : wg.split;
    const int tile0 = my_split * tiles_per_split;
    int tile1 = tile0 + tiles_per_split;
    if (tile1 > ntiles) tile1 = ntiles;
    const int nblocks = tile1 > tile0 ? (tile1 - tile0) * 4 : 0;     // 32-row blocks of this split
    const int block0 = tile0 * 4;
    constexpr int STEP = WSPLIT ? 1 : RA_WAVES;                      // block stride of a wave
    if (WSPLIT && nblocks == 0) return;                              // (no barrier in this form)

    struct Ops { frag a[U8_NCH]; Seed64 s; };
    // PRIV: source offset (bytes inside the block) of this lane's slot of each DMA piece; a lane that lands in a row's pad
    // slot, or behind the 32nd row, re-reads a valid unit (never used)
    // PRIV: the train copy has 144-byte rows (knn_shared.hpp, "wide" rows): the LDS image of a block IS its memory image,
    // pieces are lane-linear and the block's seeds arrive in the pad slots of its first eight rows — so that EVERY
    // operation a counted wait covers is an LDS-DMA piece.  (The first version of form 5 loaded the seeds to registers
    // with global_load_dwordx4 between the pieces and waited vmcnt(18) for "everything but the two youngest blocks": it
    // returned wrong neighbours for a few queries per launch, differently on every run, and was correct with vmcnt(0) —
    // LDS-DMA pieces and register loads do not retire in one order, whatever the counter's description says.  Found by
    // tools/fuzz_campaign.py, seed 41.)
    constexpr int T_ROW_BYTES = (PRIV ? U8_WIDE_ROW16 : U8_ROW16) * 16;
    int src[RA_PIECES];
#pragma unroll
    for (int p = 0; p < RA_PIECES; ++p) {
        const int sl = 64 * p + lane, row = sl / U8_LDS_ROW16, u = sl % U8_LDS_ROW16;
        (void)row; (void)u;
        src[p] = (sl < 32 * U8_LDS_ROW16 ? sl : 32 * U8_LDS_ROW16 - 1) * 16;
    }
    const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(T8), 0, ntiles * (H_TT * T_ROW_BYTES), 0x00020000);
    uint4* const mine = psm + wave * (3 * RA_PRIV_SLOTS);
    auto load_ops = [&](int blk, Ops& o, int buf) {         // blk: block inside the split (clamped by the caller)
        if constexpr (PRIV) {
            typedef __attribute__((address_space(3))) void* lptr_t;
            const int soff = (block0 + blk) * (32 * T_ROW_BYTES);
#pragma unroll
            for (int p = 0; p < RA_PIECES; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(t_rsrc, (lptr_t)(mine + buf * RA_PRIV_SLOTS + 64 * p), 16, src[p], soff, 0, 0);
        } else {
            const uint4* tp = T8 + (static_cast<size_t>(block0 + blk) * 32 + r) * U8_ROW16 + h;
#pragma unroll
            for (int c = 0; c < U8_NCH; ++c) o.a[c] = *reinterpret_cast<const frag*>(tp + 2 * c);
        }
        const uint4* sp = seeds_g + static_cast<size_t>(block0 + blk) * 8 + 4 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (PRIV) { }                                              // (seeds: with the pieces, read from LDS)
            else o.s.v[i] = sp[i];
        }
    };
    constexpr int OPS_PER_BLOCK = RA_PIECES;                                     // (PRIV: LDS-DMA pieces only)
    ABL::stamp(0);
    Ops o0, o1;
    int b = WSPLIT ? 0 : wave;
    if (!PRIV && b < nblocks) load_ops(b, o0, 0);           // first block's operands, then the query fragments
    frag qf[RA_NQB][U8_NCH];
#pragma unroll
    for (int qb = 0; qb < RA_NQB; ++qb)
#pragma unroll
        for (int c = 0; c < U8_NCH; ++c)
            qf[qb][c] = *reinterpret_cast<const frag*>(Q8 + static_cast<size_t>(qbase + 32 * qb + r) * U8_ROW16 + 2 * c + h);
    if constexpr (PRIV) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the counted waits below start from a clean slate)
#pragma unroll
    for (int qb = 0; qb < RA_NQB; ++qb)
#pragma unroll
        for (int c = 0; c < U8_NCH; ++c) {                  // opaque: never re-read from memory inside the loop
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 t = __builtin_bit_cast(u32x4, qf[qb][c]);
            asm volatile("" : "+v"(t));
            qf[qb][c] = __builtin_bit_cast(frag, t);
        }
    list cl[RA_NQB];
#pragma unroll
    for (int qb = 0; qb < RA_NQB; ++qb) cl[qb] = R::empty();
    ABL::stamp(1);
    if (PRIV && !WSPLIT && b < nblocks) load_ops(b, o0, 0);

    auto block = [&](Ops& o, int blk, int buf) {
        const int sk = blk / STEP - 4;                      // diagnostic builds: the wave's 5th .. 7th block, four stamps each
        if (sk >= 0 && sk < 3) ABL::stamp(2 + 4 * sk);
        if constexpr (PRIV) {
            // this block's pieces and seeds have landed once at most the next TWO blocks' requests are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS_PER_BLOCK) : "memory");
            {
                const uint4* sq = mine + buf * RA_PRIV_SLOTS + 4 * h * U8_LDS_ROW16 + U8_ROW16;     // pad slots of rows 4h .. 4h+3
#pragma unroll
                for (int i = 0; i < 4; ++i) o.s.v[i] = sq[i * U8_LDS_ROW16];
            }
            const uint4* tb = mine + buf * RA_PRIV_SLOTS + r * U8_LDS_ROW16 + h;
#pragma unroll
            for (int c = 0; c < U8_NCH; ++c) o.a[c] = *reinterpret_cast<const frag*>(tb + 2 * c);
        }
        if (sk >= 0 && sk < 3) ABL::stamp(3 + 4 * sk);
        const acc seed = __builtin_bit_cast(acc, o.s);
        acc a[RA_NQB];
        // The two waves of a SIMD start in step and the issue arbiter keeps them there: both in their 16-MFMA chain (sharing
        // the matrix pipe), then both in their selection (sharing the VALU) — nothing overlaps.  Raised priority over the
        // chain lets ONE wave run its chain at full rate while the other selects; they then alternate.
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int c = 0; c < U8_NCH; ++c)
#pragma unroll
            for (int qb = 0; qb < RA_NQB; ++qb) a[qb] = R::mfma(o.a[c], qf[qb][c], c == 0 ? seed : a[qb]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        if (sk >= 0 && sk < 3) ABL::stamp(4 + 4 * sk);
        if constexpr (ABL::no_epi) {
#pragma unroll
            for (int qb = 0; qb < RA_NQB; ++qb)
#pragma unroll
                for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(a[qb][e]));
        } else {
            const unsigned gb = static_cast<unsigned>(blk) * R::GPB;
#pragma unroll
            for (int qb = 0; qb < RA_NQB; ++qb)
#pragma unroll
                for (int g = 0; g < R::GPB; ++g) R::select(a[qb], 0u, (gb + static_cast<unsigned>(g)) << 1, cl[qb], g);
        }
        if (sk >= 0 && sk < 3) {
            asm volatile("" : "+v"(cl[0][0]), "+v"(cl[RA_NQB - 1][3]));
            ABL::stamp(5 + 4 * sk);
        }
    };
    if constexpr (WSPLIT) {
        // Split per wave: pieces two blocks ahead in three private buffers (runtime buffer index), and TWO accumulator sets:
        // the selection of block b - 1 is issued between the MFMAs of block b (chunks 1 .. 3), as in h_block — stamps of the
        // version that selected a block right after its own MFMAs showed the two waves of a SIMD in step, both in their
        // MFMA chain and then both in their selection, the matrix pipe idle for ~40 % of a block.
        acc accA[RA_NQB], accB[RA_NQB];
        auto issue = [&](int blk, int buf) {
            typedef __attribute__((address_space(3))) void* lptr_t;
            const int soff = (block0 + blk) * (32 * T_ROW_BYTES);
#pragma unroll
            for (int p = 0; p < RA_PIECES; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(t_rsrc, (lptr_t)(mine + buf * RA_PRIV_SLOTS + 64 * p), 16, src[p], soff, 0, 0);
        };
        auto select_all = [&](const acc (&p)[RA_NQB], int blk) {
            const unsigned gb = static_cast<unsigned>(blk) * R::GPB;
#pragma unroll
            for (int qb = 0; qb < RA_NQB; ++qb)
#pragma unroll
                for (int g = 0; g < R::GPB; ++g) R::select(p[qb], 0u, (gb + static_cast<unsigned>(g)) << 1, cl[qb], g);
        };
        auto wblock = [&](int buf, int blk, acc (&cur)[RA_NQB], const acc (&prev)[RA_NQB], auto has_prev) {
            const int sk = blk - 4;
            if (sk >= 0 && sk < 3) ABL::stamp(2 + 4 * sk);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RA_PIECES) : "memory");      // this block's pieces have landed
            const uint4* base = mine + buf * RA_PRIV_SLOTS;
            Seed64 sd;
#pragma unroll
            for (int i = 0; i < 4; ++i) sd.v[i] = base[(4 * h + i) * U8_LDS_ROW16 + U8_ROW16];   // pad slots of rows 4h .. 4h+3
            frag a[U8_NCH];
#pragma unroll
            for (int c = 0; c < U8_NCH; ++c) a[c] = *reinterpret_cast<const frag*>(base + r * U8_LDS_ROW16 + h + 2 * c);
            if (sk >= 0 && sk < 3) ABL::stamp(3 + 4 * sk);
            const acc seed = __builtin_bit_cast(acc, sd);
            const unsigned gb = static_cast<unsigned>(blk - 1) * R::GPB;
            constexpr int NGB = R::GPB * RA_NQB;
#pragma unroll
            for (int c = 0; c < U8_NCH; ++c) {
#pragma unroll
                for (int qb = 0; qb < RA_NQB; ++qb) cur[qb] = R::mfma(a[c], qf[qb][c], c == 0 ? seed : cur[qb]);
                if constexpr (decltype(has_prev)::value) {
                    if (c >= 1) {
                        if constexpr (ABL::no_epi) {
                            if (c == 1) {
#pragma unroll
                                for (int qb = 0; qb < RA_NQB; ++qb)
#pragma unroll
                                    for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(prev[qb][e]));
                            }
                        } else {
#pragma unroll
                            for (int e = (c - 1) * NGB / (U8_NCH - 1); e < c * NGB / (U8_NCH - 1); ++e)
                                R::select(prev[e % RA_NQB], 0u, (gb + static_cast<unsigned>(e / RA_NQB)) << 1, cl[e % RA_NQB], e / RA_NQB);
                            // pin the selection to this chunk (hipcc otherwise sinks all of it below the last MFMA)
                            asm volatile("" : "+v"(cl[0][0]), "+v"(cl[1][0]), "+v"(cl[2][0]), "+v"(cl[3][0]));
                        }
                    }
                }
            }
            if (sk >= 0 && sk < 3) { ABL::stamp(4 + 4 * sk); ABL::stamp(5 + 4 * sk); }
        };
        const int last = nblocks - 1;
        int bc = 0, bn = 2;                                  // buffer of the current block, buffer two blocks ahead
        issue(0, 0);
        issue(1 < last ? 1 : last, 1);
        b = 0;
        issue(2 < last ? 2 : last, 2);
        wblock(0, 0, accA, accB, std::false_type{});          // block b accumulates in accA (b even) / accB (b odd)
        b = 1; bc = 1; bn = 0;
        while (true) {
            if (b >= nblocks) break;
            issue(b + 2 < last ? b + 2 : last, bn);
            wblock(bc, b, accB, accA, std::true_type{});
            ++b; bc = bc == 2 ? 0 : bc + 1; bn = bn == 2 ? 0 : bn + 1;
            if (b >= nblocks) break;
            issue(b + 2 < last ? b + 2 : last, bn);
            wblock(bc, b, accA, accB, std::true_type{});
            ++b; bc = bc == 2 ? 0 : bc + 1; bn = bn == 2 ? 0 : bn + 1;
        }
        // the last block's accumulators: in accA if an odd number of blocks ran the loop... (b - 1 is even <=> accA)
        if constexpr (!ABL::no_epi) {
            if (((nblocks - 1) & 1) == 0) select_all(accA, nblocks - 1);
            else select_all(accB, nblocks - 1);
        }
    } else if constexpr (PRIV) {
        // THREE operand sets in turn, two blocks ahead: stamps of the first version (one block ahead) showed a wave waiting
        // ~1200-1500 cycles per block for pieces it had requested one block (~1100 cycles) earlier — under load an
        // L2-served LDS-DMA piece takes ~2000+ cycles.  Requests are clamped to the wave's last block, so every block costs
        // exactly RA_PIECES + 4 vector-memory operations and the waits can be counted.
        const int last = nblocks > 0 ? b + (nblocks - 1 - b) / STEP * STEP : 0;
        Ops o2;
        if (b < nblocks) {
            load_ops(b + STEP < last ? b + STEP : last, o1, 1);
            while (true) {
                load_ops(b + 2 * STEP < last ? b + 2 * STEP : last, o2, 2);
                block(o0, b, 0);
                b += STEP;
                if (b >= nblocks) break;
                load_ops(b + 2 * STEP < last ? b + 2 * STEP : last, o0, 0);
                block(o1, b, 1);
                b += STEP;
                if (b >= nblocks) break;
                load_ops(b + 2 * STEP < last ? b + 2 * STEP : last, o1, 1);
                block(o2, b, 2);
                b += STEP;
                if (b >= nblocks) break;
            }
        }
    } else {
    // two operand sets in turn: block b + 8 is in flight while block b is multiplied
    while (b < nblocks) {
        const int b1 = b + STEP;
        load_ops(b1 < nblocks ? b1 : b, o1, 1);             // (clamped: a valid address either way)
        block(o0, b, 0);
        if (b1 >= nblocks) break;
        const int b2 = b1 + STEP;
        load_ops(b2 < nblocks ? b2 : b1, o0, 0);
        block(o1, b1, 1);
        b = b2;
    }
    }
    if constexpr (PRIV) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA in flight when the wave ends
    ABL::stamp(16);

    if constexpr (WSPLIT) {
#pragma unroll
        for (int qb = 0; qb < RA_NQB; ++qb) {
            R::merge(cl[qb], h);
            const int q = qbase + 32 * qb + r;
            if (q < nq && h == 0) cand_val[(static_cast<size_t>(q) * slots + my_split * KNN_C) / KNN_C] = cl[qb];
        }
        ABL::stamp(17);
        return;
    }
    // the eight waves' lists of a query meet in LDS; wave w < 4 finishes query block w
#pragma unroll
    for (int qb = 0; qb < RA_NQB; ++qb) {
        R::merge(cl[qb], h);
        xs[wave][qb][lane] = cl[qb];
    }
    __syncthreads();
    if (wave >= RA_NQB) return;
    list out = xs[0][wave][lane];
#pragma unroll
    for (int w = 1; w < RA_WAVES; ++w) {
        const list o = xs[w][wave][lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) R::put(out, o[i]);
    }
    const int q = qbase + 32 * wave + r;
    if (q < nq && h == 0) cand_val[(static_cast<size_t>(q) * slots + wg.split * KNN_C) / KNN_C] = out;
}

template <int GPB_, typename ABL>
int launch_rega(pm_ctx* ctx, const char* name, const void* Q8, const void* T8, const void* seeds, int nq, int nq_pad, int nt,
                int splits, int tiles_per_split, void* cval, int slots, int form)
{
    PM_REQUIRE(nq_pad % RA_QB == 0, PM_E_INVALID, "query padding does not match the workgroup size");
    pm::ScopedKernelTime t(ctx, name);
    const int mode = xcd_tiled(ctx) ? 2 : 0;
    const uint4* q4 = static_cast<const uint4*>(Q8);
    const uint4* t4 = static_cast<const uint4*>(T8);
    const uint4* s4 = static_cast<const uint4*>(seeds);
    i32x4* out = static_cast<i32x4*>(cval);
    if (form >= 5) {       // 5: private LDS buffers, blocks dealt round-robin; 6: private LDS buffers, a split per wave
        PM_REQUIRE(rows288_dma_ok<RouteU8T<GPB_>>(nt) && (static_cast<long long>(nt) + 2 * H_TT) * (U8_WIDE_ROW16 * 16) < 0x7FFFFFFFLL,
                   PM_E_UNSUPPORTED, "train set too large for the LDS-DMA coarse routes");
        const size_t lds = sizeof(uint4) * RA_WAVES * 3 * RA_PRIV_SLOTS;
        static bool attr_done_dev[PM_MAX_DEVICES] = {};
        if (!attr_done_dev[ctx->device]) {
            PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_u8_rega<GPB_, true, false, ABL>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_u8_rega<GPB_, true, true, ABL>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            attr_done_dev[ctx->device] = true;
        }
        if (form == 6)
            hipLaunchKernelGGL((knn_u8_rega<GPB_, true, true, ABL>), dim3(nq_pad / (2 * RA_QB), (splits + RA_WAVES / 2 - 1) / (RA_WAVES / 2)),
                               dim3(RA_WAVES * 64), lds, ctx->stream, q4, t4, s4, nq, nt, tiles_per_split, out, slots, mode);
        else
            hipLaunchKernelGGL((knn_u8_rega<GPB_, true, false, ABL>), dim3(nq_pad / RA_QB, splits), dim3(RA_WAVES * 64), lds,
                               ctx->stream, q4, t4, s4, nq, nt, tiles_per_split, out, slots, mode);
    } else {
        hipLaunchKernelGGL((knn_u8_rega<GPB_, false, false, ABL>), dim3(nq_pad / RA_QB, splits), dim3(RA_WAVES * 64), 0, ctx->stream,
                           q4, t4, s4, nq, nt, tiles_per_split, out, slots, mode);
    }
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

// u8 route: group size (rows per candidate group) x staging form
template <typename ABL>
int coarse_u8_dispatch(pm_ctx* ctx, const void* Q8, const void* T8, const int* seeds, int nq, int nq_pad, int nt, int splits,
                       int tiles_per_split, int* cval, int slots, int group_rows, int form)
{
    // form: 0 / 1 two LDS tile buffers, 2 / 3 ring of eight, 4 / 5 / 6 register-operand forms (the caller sized the grid
    // and the splits for 128 queries per workgroup); 2 .. 6 exist for the default group size only
    if (form >= 4 && group_rows == 8)
        return launch_rega<2, ABL>(ctx, "knn_l2_mfma_u8", Q8, T8, seeds, nq, nq_pad, nt, splits, tiles_per_split, cval, slots, form);
    if (form >= 2 && group_rows == 8)
        return launch_ring<RouteU8T<2>, 8, ABL>(ctx, "knn_l2_mfma_u8", Q8, T8, seeds, nq, nq_pad, nt, splits, tiles_per_split, 0u, cval, slots);
#define PM_U8(GPB_)                                                                                                        \
    launch_rows288<RouteU8T<GPB_>, ABL>(ctx, "knn_l2_mfma_u8", Q8, T8, seeds, nq, nq_pad, nt, splits, tiles_per_split, 0u, cval,   \
                                        slots, nullptr, 0u, 0)
    if (group_rows == 16) return PM_U8(1);
    if (group_rows == 8) return PM_U8(2);
    return PM_U8(4);
#undef PM_U8
}

}  // namespace
}  // namespace pm_knn
