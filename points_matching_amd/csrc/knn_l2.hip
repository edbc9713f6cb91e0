// knn_l2.hip — brute-force L2 k-NN over float descriptors on gfx950 (MI355X).
//
// Replaces `matcher.match(imageDesc1, imageDesc2, matchePoints, Mat())` (main.cpp:46) with the
// BruteForceMatcher<L2<float>> of main.cpp:43, generalised to k-NN.  The RESULT is defined by
// docs/SPEC.md S1/S3: distance = sqrtf(canonical sum of squared differences), order =
// (distance bits, trainIdx).  How it is computed here:
//
//   knn_l2_prep    row norms ||q||^2, ||t||^2 (+ max train norm, non-finite flag)
//   knn_l2_mfma    COARSE pass on the matrix cores: v_mfma_f32_32x32x2_f32 accumulates q.t over
//                  D; s = ||t||^2 - 2 q.t ranks the train rows of a query up to a proven error
//                  eps.  Train rows ride the MFMA M dimension and queries the N dimension, so a
//                  lane owns ONE query column and keeps its 4 smallest s (value, index) in
//                  registers for the whole sweep: no cross-lane traffic in the loop.
//   knn_l2_refine  per query: tau = k-th smallest coarse value; every slot within
//                  (tau+eps)(1+2^-20)+eps is re-evaluated in the canonical op order on the VALU
//                  and the k smallest canonical keys are emitted.  A sub-list whose 4th entry
//                  is still inside the window may have dropped a candidate: that query is
//                  re-scanned exactly (rare; always correct).
//   knn_l2_exact   general kernel (any dim, k <= 16): canonical distances for every pair on the
//                  VALU out of LDS tiles.  Also the `PM_KNN_FORCE_EXACT` path.
//
// Both routes are bit-identical by construction (tests/test_knn_l2_gpu.py asserts it).
#include "pm_common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KNN_C = 4;      // coarse candidates kept per (query, split, lane-half)
constexpr int TILE_T = 64;    // train rows per LDS tile
constexpr int QB = 128;       // queries per workgroup in the coarse kernel (4 waves x 32)
constexpr float KNN_INF = __builtin_inff();

__device__ __forceinline__ uint32_t f32_bits(float f) { return __float_as_uint(f); }

// SPEC S3 ordering key; NaN distances are canonicalised so they sort after +inf.
__device__ __forceinline__ uint64_t knn_key(float dist, int idx)
{
    uint32_t b = (dist != dist) ? 0x7FC00000u : f32_bits(dist);
    return (static_cast<uint64_t>(b) << 32) | static_cast<uint32_t>(idx);
}

// SPEC S1 — canonical squared distance: eight lane accumulators over 8-wide strides, unfused
// multiply and add, (acc[l]+acc[l+4]) lane-wise, ((s0+s1)+s2)+s3, scalar tail.  The TU is built
// with -ffp-contract=off, so none of these contract into v_fma/v_mac.
template <bool VEC4>
__device__ __forceinline__ float l2sqr_canonical(const float* __restrict__ a,
                                                 const float* __restrict__ b, int dim)
{
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int j = 0;
    if (VEC4) {
        for (; j + 8 <= dim; j += 8) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(a + j);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(a + j + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + j);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(b + j + 4);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                float t = a0[l] - b0[l];
                float p = t * t;
                acc[l] = acc[l] + p;
                float u = a1[l] - b1[l];
                float w = u * u;
                acc[l + 4] = acc[l + 4] + w;
            }
        }
    } else {
        for (; j + 8 <= dim; j += 8) {
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                float t = a[j + l] - b[j + l];
                float p = t * t;
                acc[l] = acc[l] + p;
            }
        }
    }
    const float s0 = acc[0] + acc[4], s1 = acc[1] + acc[5], s2 = acc[2] + acc[6],
                s3 = acc[3] + acc[7];
    float d = ((s0 + s1) + s2) + s3;
    for (; j < dim; ++j) {
        float t = a[j] - b[j];
        float p = t * t;
        d = d + p;
    }
    return d;
}

__device__ __forceinline__ float wave_min_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// prep: squared norms of the query rows and the train rows in ONE launch (64 rows per block, 16
// lanes per row).  stats words are epoch-tagged 64-bit maxima, (epoch << 32) | payload, so a
// call never has to clear them: values left by earlier calls carry a smaller epoch and lose.
//   stats[0]: payload = float bits of the largest finite train norm
//   stats[1]: payload = 1 when any norm is not finite
// Approximate values: they only feed the coarse pass and its error bound.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_l2_prep(const float* __restrict__ Q, int nq,
                                                   const float* __restrict__ T, int nt, int dim,
                                                   float* __restrict__ qnorm, float* __restrict__ tnorm,
                                                   unsigned long long* __restrict__ stats, unsigned epoch)
{
    __shared__ unsigned wmax[4];
    __shared__ unsigned wbad[4];
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int qblocks = (nq + 63) / 64;
    const bool is_t = static_cast<int>(blockIdx.x) >= qblocks;
    const float* x = is_t ? T : Q;
    const int n = is_t ? nt : nq;
    float* norm = is_t ? tnorm : qnorm;
    const int row0 = (is_t ? blockIdx.x - qblocks : blockIdx.x) * 64;
    unsigned mx = 0u, bad = 0u;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 16 + grp;
        const int rl = row < n ? row : n - 1;
        const float* p = x + static_cast<size_t>(rl) * dim;
        float s = 0.f;
        for (int c = sub; c < dim; c += 16) s = fmaf(p[c], p[c], s);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
        if (sub == 0 && row < n) {
            norm[row] = s;
            if (!(s < KNN_INF)) bad = 1u;
            else if (is_t) mx = max(mx, f32_bits(s));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, static_cast<unsigned>(__shfl_xor(static_cast<int>(mx), o, 64)));
        bad |= static_cast<unsigned>(__shfl_xor(static_cast<int>(bad), o, 64));
    }
    if ((threadIdx.x & 63) == 0) { wmax[threadIdx.x >> 6] = mx; wbad[threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        bad = wbad[0] | wbad[1] | wbad[2] | wbad[3];
        const unsigned long long tag = static_cast<unsigned long long>(epoch) << 32;
        if (is_t) atomicMax(&stats[0], tag | mx);
        if (bad) atomicMax(&stats[1], tag | 1ull);
    }
}

// ---------------------------------------------------------------------------------------------
// coarse pass on the matrix cores
//
// Per (query, train row) the accumulator is seeded with -||t||^2/2 and the MFMA chain adds q.t, so
// it ends as w = q.t - ||t||^2/2 = -(d2a - ||q||^2)/2: the LARGEST w are the nearest rows, and no
// VALU work is needed to form the ranking value.  The row's position in this lane's stream
// (lid = tile_in_split*32 + block*16 + reg) is written into the low `bits` mantissa bits of w, so
// a candidate is ONE float and keeping the 4 largest is branch-free:
//     n0 = max(x,w0); n1 = med3(x,w0,w1); n2 = med3(x,w1,w2); n3 = med3(x,w2,w3)
// (5 VALU per pair incl. the bit insert).  The truncation error 2^(bits-23)*|w| is part of the
// refinement's window (SPEC S1b).  Two accumulator sets: the epilogue of tile t-1 is issued
// between the MFMAs of tile t, so the matrix pipe does not wait for the selection.
// ---------------------------------------------------------------------------------------------
constexpr float KNN_BIG = 3.0e38f;       // finite sentinel: stays finite under the bit insert

__device__ __forceinline__ float embed_lid(float w, unsigned keep_mask, unsigned lid)
{
    return __uint_as_float((__float_as_uint(w) & keep_mask) | lid);
}

__device__ __forceinline__ void top4_insert(f32x4& c, float x)
{
    const float n0 = __builtin_amdgcn_fmed3f(x, c[0], KNN_INF);      // = max(x, c0), no canonicalising v_max
    const float n1 = __builtin_amdgcn_fmed3f(x, c[0], c[1]);
    const float n2 = __builtin_amdgcn_fmed3f(x, c[1], c[2]);
    const float n3 = __builtin_amdgcn_fmed3f(x, c[2], c[3]);
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// NCH = padded dim / 8; FULL = (dim == 8*NCH), which drops the column guards.
// grid = (ceil(nq/QB), splits).  Dynamic LDS: 2 tiles of TILE_T x (8*NCH + 4) floats (row stride
// padded by one 16-B slot: conflict-free ds_read_b128 for the 16 rows of a lane group) + 2 x TILE_T
// seeds (-||t||^2/2, or -KNN_BIG/2 past the last row).
template <int NCH, bool FULL>
struct KnnTile {
    static constexpr int DP = NCH * 8;
    static constexpr int LDT = DP + 4;
    static constexpr int F4_PER_ROW = DP / 4;
    static constexpr int NSTG = TILE_T * F4_PER_ROW / 256;
    static_assert(TILE_T * F4_PER_ROW % 256 == 0, "tile must split evenly over the workgroup");

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
            int g = tile * TILE_T + row;
            g = g < nt ? g : nt - 1;
            int col = 4 * c4;
            if (!FULL) col = col < dim ? col : dim - 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(T + static_cast<size_t>(g) * dim + col);
            if (!FULL && 4 * c4 >= dim) v = f32x4{0.f, 0.f, 0.f, 0.f};
            stg[i] = v;
        }
        const int gn = tile * TILE_T + (tid & (TILE_T - 1));
        const float nrm = tnorm[gn < nt ? gn : nt - 1];
        stg_n = gn < nt ? -0.5f * nrm : -0.5f * KNN_BIG;
    }
    // registers -> LDS buffer
    __device__ __forceinline__ void store(float* __restrict__ Ts, float* __restrict__ Tn, int buf, int tid) const
    {
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            const int f = tid + 256 * i;
            const int row = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
            *reinterpret_cast<f32x4*>(Ts + (buf * TILE_T + row) * LDT + 4 * c4) = stg[i];
        }
        if (tid < TILE_T) Tn[buf * TILE_T + tid] = stg_n;
    }
};

// One tile: seed the accumulators, run the MFMA chain, and (EPI) select the previous tile's
// accumulators p0/p1 in between.  C[i][j] of lane (j = lane&31), register reg is train row
// i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) of the 32-row block.
template <int NCH, bool EPI>
__device__ __forceinline__ void knn_tile_compute(const float* __restrict__ Ts, const float* __restrict__ Tn,
                                                 int buf, int r, int h, const f32x4 (&qf)[NCH], f32x16& a0,
                                                 f32x16& a1, const f32x16& p0, const f32x16& p1, unsigned pbase,
                                                 unsigned keep_mask, f32x4& cl)
{
    constexpr int LDT = NCH * 8 + 4;
    const float* tn = Tn + buf * TILE_T + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 n0 = *reinterpret_cast<const f32x4*>(tn + 8 * g);
        const f32x4 n1 = *reinterpret_cast<const f32x4*>(tn + 32 + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0[4 * g + e] = n0[e]; a1[4 * g + e] = n1[e]; }
    }
    const float* tb = Ts + buf * TILE_T * LDT + r * LDT + 4 * h;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(tb + 8 * c);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(tb + 32 * LDT + 8 * c);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[t], qf[c][t], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[t], qf[c][t], a1, 0, 0, 0);
        }
        if (EPI) {
            constexpr int PER = 32 / NCH;          // selection steps per chunk
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int v = c * PER + u;         // 0..31: block v>>4, register v&15
                const float w = v < 16 ? p0[v & 15] : p1[v & 15];
                top4_insert(cl, embed_lid(w, keep_mask, pbase | static_cast<unsigned>(v)));
            }
            // pin the selection to this chunk: without a use here hipcc sinks all of it below the
            // MFMA chain (in front of the barrier), where nothing hides it.  Placed after the
            // chunk's MFMAs, the ~10 VALU ops issue in the shadow of the last 64-cycle MFMA.
            asm volatile("" : "+v"(cl[0]), "+v"(cl[1]), "+v"(cl[2]), "+v"(cl[3]));
        }
    }
}

__device__ __forceinline__ void knn_select_all(const f32x16& p0, const f32x16& p1, unsigned pbase,
                                               unsigned keep_mask, f32x4& cl)
{
#pragma unroll
    for (int v = 0; v < 32; ++v) {
        const float w = v < 16 ? p0[v & 15] : p1[v & 15];
        top4_insert(cl, embed_lid(w, keep_mask, pbase | static_cast<unsigned>(v)));
    }
}

template <int NCH, bool FULL>
__global__ __launch_bounds__(256, 2) void knn_l2_mfma(
    const float* __restrict__ Q, const float* __restrict__ T, const float* __restrict__ tnorm, int nq,
    int nt, int dim, int tiles_per_split, unsigned keep_mask, float* __restrict__ cand_val, int slots)
{
    using Tile = KnnTile<NCH, FULL>;
    constexpr int LDT = Tile::LDT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ts = smem;                          // [2][TILE_T][LDT]
    float* Tn = smem + 2 * TILE_T * LDT;       // [2][TILE_T]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int qrow = blockIdx.x * QB + wave * 32 + r;
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

    const int ntiles = (nt + TILE_T - 1) / TILE_T;
    const int tile0 = blockIdx.y * tiles_per_split;
    int tile1 = tile0 + tiles_per_split;
    if (tile1 > ntiles) tile1 = ntiles;

    f32x4 cl = {-KNN_BIG, -KNN_BIG, -KNN_BIG, -KNN_BIG};
    if (tile0 < tile1) {                       // block-uniform
        Tile st;
        st.load(T, tnorm, tile0, nt, dim, tid);
        st.store(Ts, Tn, 0, tid);
        __syncthreads();

        // every accumulator has a compile-time name: tiles alternate A, B, A, ...
        f32x16 accA0, accA1, accB0, accB1;
        int tile = tile0;
        {   // first tile -> A, nothing pending
            const bool more = tile + 1 < tile1;
            if (more) st.load(T, tnorm, tile + 1, nt, dim, tid);
            knn_tile_compute<NCH, false>(Ts, Tn, 0, r, h, qf, accA0, accA1, accA0, accA1, 0u, keep_mask, cl);
            if (more) st.store(Ts, Tn, 1, tid);
            __syncthreads();
            ++tile;
        }
        for (;;) {
            if (tile >= tile1) { knn_select_all(accA0, accA1, static_cast<unsigned>(tile - 1 - tile0) << 5, keep_mask, cl); break; }
            {   // tile -> B while selecting A (tile-1)
                const int buf = (tile - tile0) & 1;
                const bool more = tile + 1 < tile1;
                if (more) st.load(T, tnorm, tile + 1, nt, dim, tid);
                knn_tile_compute<NCH, true>(Ts, Tn, buf, r, h, qf, accB0, accB1, accA0, accA1,
                                            static_cast<unsigned>(tile - 1 - tile0) << 5, keep_mask, cl);
                if (more) st.store(Ts, Tn, buf ^ 1, tid);
                __syncthreads();
                ++tile;
            }
            if (tile >= tile1) { knn_select_all(accB0, accB1, static_cast<unsigned>(tile - 1 - tile0) << 5, keep_mask, cl); break; }
            {   // tile -> A while selecting B (tile-1)
                const int buf = (tile - tile0) & 1;
                const bool more = tile + 1 < tile1;
                if (more) st.load(T, tnorm, tile + 1, nt, dim, tid);
                knn_tile_compute<NCH, true>(Ts, Tn, buf, r, h, qf, accA0, accA1, accB0, accB1,
                                            static_cast<unsigned>(tile - 1 - tile0) << 5, keep_mask, cl);
                if (more) st.store(Ts, Tn, buf ^ 1, tid);
                __syncthreads();
                ++tile;
            }
        }
    }

    if (qrow < nq) {
        const size_t o = static_cast<size_t>(qrow) * slots + (blockIdx.y * 2 + h) * KNN_C;
        *reinterpret_cast<f32x4*>(cand_val + o) = cl;
    }
}

// ---------------------------------------------------------------------------------------------
// refinement: one wave per query
// ---------------------------------------------------------------------------------------------
struct Best2 {
    uint64_t k0, k1;
    float d0, d1;
};

__device__ __forceinline__ void best2_insert(Best2& b, uint64_t key, float d)
{
    if (key < b.k1) {
        if (key < b.k0) { b.k1 = b.k0; b.d1 = b.d0; b.k0 = key; b.d0 = d; }
        else { b.k1 = key; b.d1 = d; }
    }
}

template <bool VEC4>
__global__ __launch_bounds__(256) void knn_l2_refine(
    const float* __restrict__ Q, const float* __restrict__ T, const float* __restrict__ qnorm,
    const unsigned long long* __restrict__ stats, unsigned epoch, unsigned* __restrict__ diag, int nq, int nt,
    int dim, int k, int slots, const float* __restrict__ cand_val, int tiles_per_split, unsigned lid_mask,
    float eps_coef, float embed_coef, pm_match* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (q >= nq) return;
    const float* qp = Q + static_cast<size_t>(q) * dim;
    const float na = qnorm[q];
    const unsigned long long s0 = stats[0], s1 = stats[1];
    const float tmax = static_cast<unsigned>(s0 >> 32) == epoch ? __uint_as_float(static_cast<unsigned>(s0)) : 0.f;
    const bool nonfinite = static_cast<unsigned>(s1 >> 32) == epoch && (s1 & 1ull);
    // window half-width: fp error of the coarse value + truncation by the embedded row id
    const float eps = eps_coef * (na + tmax) + embed_coef * (na + 2.f * tmax);
    const float* cv = cand_val + static_cast<size_t>(q) * slots;
    // slot value w = q.t - ||t||^2/2 (+id bits)  ->  coarse squared distance d2a = ||q||^2 - 2w
    auto coarse = [&](int s) -> float {
        const float w = cv[s];
        return w > -1.0e38f ? fmaf(-2.f, w, na) : KNN_INF;
    };
    auto row_of = [&](int s) -> int {
        const unsigned lid = __float_as_uint(cv[s]) & lid_mask;
        const int split = s / (2 * KNN_C), hh = (s / KNN_C) & 1;
        const int reg = lid & 15, blk = (lid >> 4) & 1;
        return (split * tiles_per_split + static_cast<int>(lid >> 5)) * TILE_T + 32 * blk + (reg & 3) +
               8 * (reg >> 2) + 4 * hh;
    };

    // k-th smallest coarse value over all slots (k <= 2)
    float m0 = KNN_INF, m1 = KNN_INF;
    for (int s = lane; s < slots; s += 64) {
        const float v = coarse(s);
        if (v < m1) { if (v < m0) { m1 = m0; m0 = v; } else { m1 = v; } }
    }
    float tau = KNN_INF;
    for (int round = 0; round < k; ++round) {
        tau = wave_min_f32(m0);
        const unsigned long long owners = __ballot(m0 == tau);
        if (owners == 0ull) break;               // NaN guard
        const int first = __ffsll(static_cast<long long>(owners)) - 1;
        if (lane == first) { m0 = m1; m1 = KNN_INF; }
    }
    const float thr = (tau + eps) * 1.00000095367431640625f + eps;

    // a sub-list whose largest kept entry is inside the window may have dropped candidates
    bool spill = false;
    for (int s = lane; s < slots; s += 64)
        if ((s & (KNN_C - 1)) == KNN_C - 1) spill |= coarse(s) <= thr;
    const bool rescan = nonfinite || !(thr < KNN_INF) || __any(spill);
    if (diag && lane == 0) { if (rescan) atomicAdd(&diag[0], 1u); if (nonfinite) diag[1] = 1u; }

    Best2 b{~0ull, ~0ull, KNN_INF, KNN_INF};
    if (!rescan) {
        for (int s = lane; s < slots; s += 64) {
            const float v = coarse(s);
            const int j = row_of(s);
            if (v <= thr && j < nt) {
                const float d = __builtin_sqrtf(
                    l2sqr_canonical<VEC4>(qp, T + static_cast<size_t>(j) * dim, dim));
                best2_insert(b, knn_key(d, j), d);
            }
        }
    } else {
        for (int j = lane; j < nt; j += 64) {
            const float d =
                __builtin_sqrtf(l2sqr_canonical<VEC4>(qp, T + static_cast<size_t>(j) * dim, dim));
            best2_insert(b, knn_key(d, j), d);
        }
    }
    for (int c = 0; c < k; ++c) {
        const uint64_t best = wave_min_u64(b.k0);
        const unsigned long long owners = __ballot(b.k0 == best);
        const int first = __ffsll(static_cast<long long>(owners)) - 1;
        const float dist = __shfl(b.d0, first, 64);
        if (lane == first) { b.k0 = b.k1; b.d0 = b.d1; b.k1 = ~0ull; b.d1 = KNN_INF; }
        if (lane == 0) {
            pm_match m;
            m.queryIdx = q;
            m.imgIdx = 0;
            if (best == ~0ull) { m.trainIdx = -1; m.distance = KNN_INF; }
            else { m.trainIdx = static_cast<int>(static_cast<uint32_t>(best)); m.distance = dist; }
            out[static_cast<size_t>(q) * k + c] = m;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// exact general kernel.  Workgroup = 4 waves = EXQ_WG queries (4 per wave); lanes ride the train
// rows of a 64-row LDS tile; K is walked in chunks of EX_KC columns so any dim fits.  Each lane
// keeps the KL smallest keys per query; k > KL takes ceil(k/KL) passes, pass p only admitting
// keys above the last one emitted by pass p-1.
// ---------------------------------------------------------------------------------------------
constexpr int EX_KC = 128;
constexpr int EX_LD = EX_KC + 4;
constexpr int EX_QPW = 4;
constexpr int EXQ_WG = 4 * EX_QPW;

template <int KL>
struct KeyList {
    uint64_t k[KL];
    float d[KL];
    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int i = 0; i < KL; ++i) { k[i] = ~0ull; d[i] = KNN_INF; }
    }
    __device__ __forceinline__ void insert(uint64_t key, float dist)
    {
        if (key < k[KL - 1]) {
            k[KL - 1] = key; d[KL - 1] = dist;
#pragma unroll
            for (int i = KL - 1; i > 0; --i)
                if (k[i] < k[i - 1]) {
                    uint64_t t = k[i]; k[i] = k[i - 1]; k[i - 1] = t;
                    float u = d[i]; d[i] = d[i - 1]; d[i - 1] = u;
                }
        }
    }
    __device__ __forceinline__ void pop()
    {
#pragma unroll
        for (int i = 0; i + 1 < KL; ++i) { k[i] = k[i + 1]; d[i] = d[i + 1]; }
        k[KL - 1] = ~0ull; d[KL - 1] = KNN_INF;
    }
};

template <int KL>
__global__ __launch_bounds__(256) void knn_l2_exact(const float* __restrict__ Q,
                                                    const float* __restrict__ T, int nq, int nt,
                                                    int dim, int k, pm_match* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float Ts[TILE_T * EX_LD];
    __shared__ __attribute__((aligned(16))) float Qs[EXQ_WG * EX_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * EXQ_WG;
    const int full8 = dim / 8 * 8;              // columns covered by the 8-lane accumulators
    const int ntiles = (nt + TILE_T - 1) / TILE_T;
    const int nchunks = (dim + EX_KC - 1) / EX_KC;
    const bool vec4 = (dim & 3) == 0;

    uint64_t floor_key[EX_QPW];                 // keys <= floor were emitted by earlier passes
    bool have_floor = false;
#pragma unroll
    for (int i = 0; i < EX_QPW; ++i) floor_key[i] = 0ull;

    for (int emitted = 0; emitted < k; emitted += KL) {
        KeyList<KL> best[EX_QPW];
#pragma unroll
        for (int i = 0; i < EX_QPW; ++i) best[i].reset();

        for (int tile = 0; tile < ntiles; ++tile) {
            float acc[EX_QPW][8];
            float tail[EX_QPW];
#pragma unroll
            for (int i = 0; i < EX_QPW; ++i) {
                tail[i] = 0.f;
#pragma unroll
                for (int l = 0; l < 8; ++l) acc[i][l] = 0.f;
            }
            for (int ch = 0; ch < nchunks; ++ch) {
                const int kc0 = ch * EX_KC;
                const int kw = dim - kc0 < EX_KC ? dim - kc0 : EX_KC;   // valid columns in chunk
                __syncthreads();
                if (vec4) {
                    for (int f = tid; f < TILE_T * (EX_KC / 4); f += 256) {
                        const int row = f / (EX_KC / 4), c4 = f % (EX_KC / 4);
                        int g = tile * TILE_T + row;
                        g = g < nt ? g : nt - 1;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (4 * c4 < kw)
                            v = *reinterpret_cast<const f32x4*>(T + static_cast<size_t>(g) * dim + kc0 + 4 * c4);
                        *reinterpret_cast<f32x4*>(Ts + row * EX_LD + 4 * c4) = v;
                    }
                    for (int f = tid; f < EXQ_WG * (EX_KC / 4); f += 256) {
                        const int row = f / (EX_KC / 4), c4 = f % (EX_KC / 4);
                        int g = q0 + row;
                        g = g < nq ? g : nq - 1;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (4 * c4 < kw)
                            v = *reinterpret_cast<const f32x4*>(Q + static_cast<size_t>(g) * dim + kc0 + 4 * c4);
                        *reinterpret_cast<f32x4*>(Qs + row * EX_LD + 4 * c4) = v;
                    }
                } else {
                    for (int f = tid; f < TILE_T * EX_KC; f += 256) {
                        const int row = f / EX_KC, c = f % EX_KC;
                        int g = tile * TILE_T + row;
                        g = g < nt ? g : nt - 1;
                        Ts[row * EX_LD + c] = c < kw ? T[static_cast<size_t>(g) * dim + kc0 + c] : 0.f;
                    }
                    for (int f = tid; f < EXQ_WG * EX_KC; f += 256) {
                        const int row = f / EX_KC, c = f % EX_KC;
                        int g = q0 + row;
                        g = g < nq ? g : nq - 1;
                        Qs[row * EX_LD + c] = c < kw ? Q[static_cast<size_t>(g) * dim + kc0 + c] : 0.f;
                    }
                }
                __syncthreads();
                // full 8-column groups of this chunk
                const int g8 = (full8 - kc0 < kw ? (full8 - kc0 > 0 ? full8 - kc0 : 0) : kw) / 8;
                const float* tr = Ts + lane * EX_LD;
                for (int gi = 0; gi < g8; ++gi) {
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(tr + 8 * gi);
                    const f32x4 t1 = *reinterpret_cast<const f32x4*>(tr + 8 * gi + 4);
#pragma unroll
                    for (int i = 0; i < EX_QPW; ++i) {
                        const float* qr = Qs + (wave * EX_QPW + i) * EX_LD + 8 * gi;
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(qr);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(qr + 4);
#pragma unroll
                        for (int l = 0; l < 4; ++l) {
                            float x = a0[l] - t0[l];
                            float p = x * x;
                            acc[i][l] = acc[i][l] + p;
                            float y = a1[l] - t1[l];
                            float w = y * y;
                            acc[i][l + 4] = acc[i][l + 4] + w;
                        }
                    }
                }
                // scalar tail columns (dim % 8) live in the last chunk; stash them for after the
                // lane combine.  At most 7 values per query: fold them in order later.
                if (kc0 + kw == dim && full8 < dim) {
#pragma unroll
                    for (int i = 0; i < EX_QPW; ++i) {
                        // combine first (all full groups are done once the last chunk is in)
                        const float s0 = acc[i][0] + acc[i][4], s1 = acc[i][1] + acc[i][5],
                                    s2 = acc[i][2] + acc[i][6], s3 = acc[i][3] + acc[i][7];
                        float d = ((s0 + s1) + s2) + s3;
                        for (int c = full8 - kc0; c < kw; ++c) {
                            float x = Qs[(wave * EX_QPW + i) * EX_LD + c] - tr[c];
                            float p = x * x;
                            d = d + p;
                        }
                        tail[i] = d;
                    }
                }
            }
            const int j = tile * TILE_T + lane;
#pragma unroll
            for (int i = 0; i < EX_QPW; ++i) {
                float d2;
                if (full8 < dim) d2 = tail[i];
                else {
                    const float s0 = acc[i][0] + acc[i][4], s1 = acc[i][1] + acc[i][5],
                                s2 = acc[i][2] + acc[i][6], s3 = acc[i][3] + acc[i][7];
                    d2 = ((s0 + s1) + s2) + s3;
                }
                const float d = __builtin_sqrtf(d2);
                const uint64_t key = knn_key(d, j);
                if (j < nt && (!have_floor || key > floor_key[i])) best[i].insert(key, d);
            }
        }
        // merge the lanes' lists: KL rounds of wave-min
#pragma unroll
        for (int i = 0; i < EX_QPW; ++i) {
            const int q = q0 + wave * EX_QPW + i;
            for (int c = 0; c < KL; ++c) {
                const uint64_t bestk = wave_min_u64(best[i].k[0]);
                const unsigned long long owners = __ballot(best[i].k[0] == bestk);
                const int first = __ffsll(static_cast<long long>(owners)) - 1;
                const float dist = __shfl(best[i].d[0], first, 64);
                if (lane == first) best[i].pop();
                if (bestk != ~0ull) floor_key[i] = bestk;
                if (lane == 0 && q < nq && emitted + c < k) {
                    pm_match m;
                    m.queryIdx = q;
                    m.imgIdx = 0;
                    if (bestk == ~0ull) { m.trainIdx = -1; m.distance = KNN_INF; }
                    else { m.trainIdx = static_cast<int>(static_cast<uint32_t>(bestk)); m.distance = dist; }
                    out[static_cast<size_t>(q) * k + emitted + c] = m;
                }
            }
        }
        have_floor = true;
    }
}

template <int NCH, bool FULL>
int launch_mfma(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim,
                const float* tnorm, int splits, int tiles_per_split, unsigned keep_mask, float* cval, int slots)
{
    constexpr int LDT = NCH * 8 + 4;
    const size_t lds = (2 * TILE_T * LDT + 2 * TILE_T) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_l2_mfma<NCH, FULL>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        attr_done = true;
    }
    dim3 grid((nq + QB - 1) / QB, splits);
    pm::ScopedKernelTime t(ctx, "knn_l2_mfma");
    hipLaunchKernelGGL((knn_l2_mfma<NCH, FULL>), grid, dim3(256), lds, ctx->stream, dq, dt, tnorm, nq, nt, dim,
                       tiles_per_split, keep_mask, cval, slots);
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

int run_exact(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim, int k, pm_match* dout)
{
    if (nq == 0) return PM_OK;
    dim3 grid((nq + EXQ_WG - 1) / EXQ_WG);
    pm::ScopedKernelTime t(ctx, "knn_l2_exact");
    if (k == 1)
        hipLaunchKernelGGL(knn_l2_exact<1>, grid, dim3(256), 0, ctx->stream, dq, dt, nq, nt, dim, k, dout);
    else if (k == 2)
        hipLaunchKernelGGL(knn_l2_exact<2>, grid, dim3(256), 0, ctx->stream, dq, dt, nq, nt, dim, k, dout);
    else
        hipLaunchKernelGGL(knn_l2_exact<4>, grid, dim3(256), 0, ctx->stream, dq, dt, nq, nt, dim, k, dout);
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

}  // namespace

extern "C" int pm_bf_knn_l2_f32_dev(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim,
                                    int k, int flags, pm_match* dout)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_REQUIRE(nq >= 0 && nt >= 0 && dim >= 1 && k >= 1 && k <= PM_MAX_K, PM_E_INVALID,
               "need nq,nt >= 0, dim >= 1, 1 <= k <= PM_MAX_K");
    PM_REQUIRE(nq == 0 || (dq && dout), PM_E_INVALID, "null query/output pointer");
    PM_REQUIRE(nt == 0 || dt, PM_E_INVALID, "null train pointer");
    if (nq == 0) return PM_OK;
    PM_HIP_CHECK(hipSetDevice(ctx->device));

    const bool fast = !(flags & PM_KNN_FORCE_EXACT) && k <= 2 && (dim % 4) == 0 && dim <= 128 && nt >= 1;
    if (!fast) return run_exact(ctx, dq, nq, dt, nt, dim, k, dout);

    // split the train rows so that the grid fills the chip (~2 workgroups per CU)
    const int ntiles = (nt + TILE_T - 1) / TILE_T;
    const int nqb = (nq + QB - 1) / QB;
    int splits = (2 * ctx->n_cu + nqb - 1) / nqb;
    if (splits > ntiles) splits = ntiles;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    const int tiles_per_split = (ntiles + splits - 1) / splits;
    splits = (ntiles + tiles_per_split - 1) / tiles_per_split;
    const int slots = splits * 2 * KNN_C;
    // row id inside a lane's stream: tile_in_split*32 + block*16 + reg, in the low mantissa bits
    int lid_bits = 5;
    while ((1 << lid_bits) < tiles_per_split * 32) ++lid_bits;
    if (lid_bits > 16) return run_exact(ctx, dq, nq, dt, nt, dim, k, dout);     // > 2048 tiles per split
    const unsigned lid_mask = (1u << lid_bits) - 1u;

    // scratch: norms, stats, candidate lists.  The arena is carved per call; callers that
    // interleave calls on one context are serialised by the stream.
    const size_t need = pm::align_up(sizeof(float) * nq, 256) + pm::align_up(sizeof(float) * nt, 256) +
                        pm::align_up(sizeof(float) * static_cast<size_t>(nq) * slots, 256) + 1024;
    int rc = pm::arena_reserve(ctx, need);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    float* qnorm = static_cast<float*>(pm::arena_take(ctx, sizeof(float) * nq));
    float* tnorm = static_cast<float*>(pm::arena_take(ctx, sizeof(float) * nt));
    float* cval = static_cast<float*>(pm::arena_take(ctx, sizeof(float) * static_cast<size_t>(nq) * slots));
    PM_REQUIRE(qnorm && tnorm && cval, PM_E_NOMEM, "scratch arena too small");

    unsigned long long* stats = ctx->knn_stats;          // persistent, epoch-tagged: never cleared
    if (++ctx->knn_epoch == 0u) {              // 2^32 calls: restart the epoch tags
        PM_HIP_CHECK(hipMemsetAsync(stats, 0, 16, ctx->stream));
        ctx->knn_epoch = 1u;
    }
    const unsigned epoch = ctx->knn_epoch;
    unsigned* diag = nullptr;
    if (ctx->knn_diag) {
        diag = ctx->knn_diag_words;
        PM_HIP_CHECK(hipMemsetAsync(diag, 0, 8, ctx->stream));
    }
    {
        pm::ScopedKernelTime t(ctx, "knn_l2_prep");
        hipLaunchKernelGGL(knn_l2_prep, dim3((nq + 63) / 64 + (nt + 63) / 64), dim3(256), 0, ctx->stream, dq, nq, dt,
                           nt, dim, qnorm, tnorm, stats, epoch);
        PM_HIP_CHECK(hipGetLastError());
    }
#define PM_LAUNCH_MFMA(NCH_, FULL_) \
    launch_mfma<NCH_, FULL_>(ctx, dq, nq, dt, nt, dim, tnorm, splits, tiles_per_split, ~lid_mask, cval, slots)
    if (dim == 128) rc = PM_LAUNCH_MFMA(16, true);
    else if (dim == 64) rc = PM_LAUNCH_MFMA(8, true);
    else if (dim == 32) rc = PM_LAUNCH_MFMA(4, true);
    else if (dim < 32) rc = PM_LAUNCH_MFMA(4, false);
    else if (dim < 64) rc = PM_LAUNCH_MFMA(8, false);
    else rc = PM_LAUNCH_MFMA(16, false);
#undef PM_LAUNCH_MFMA
    if (rc != PM_OK) return rc;

    // |coarse - canonical| <= (6*dim + 32) * 2^-24 * (||q||^2 + ||t||^2) + 2^(bits-23) * (||q||^2 + 2||t||^2);
    // see docs/SPEC.md S1b
    const float eps_coef = static_cast<float>((6.0 * dim + 32.0) * 5.9604644775390625e-8 * 1.001);
    const float embed_coef = static_cast<float>(static_cast<double>(1u << lid_bits) * 1.1920928955078125e-7 * 1.01);
    {
        pm::ScopedKernelTime t(ctx, "knn_l2_refine");
        hipLaunchKernelGGL(knn_l2_refine<true>, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, dq, dt, qnorm, stats,
                           epoch, diag, nq, nt, dim, k, slots, cval, tiles_per_split, lid_mask, eps_coef, embed_coef,
                           dout);
        PM_HIP_CHECK(hipGetLastError());
    }
    return PM_OK;
}

extern "C" int pm_bf_knn_l2_f32(pm_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim, int k,
                                int flags, pm_match* out)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_REQUIRE(nq >= 0 && nt >= 0 && dim >= 1 && k >= 1 && k <= PM_MAX_K, PM_E_INVALID,
               "need nq,nt >= 0, dim >= 1, 1 <= k <= PM_MAX_K");
    PM_REQUIRE(nq == 0 || (q && out), PM_E_INVALID, "null query/output pointer");
    PM_REQUIRE(nt == 0 || t, PM_E_INVALID, "null train pointer");
    if (nq == 0) return PM_OK;
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t qb = sizeof(float) * static_cast<size_t>(nq) * dim;
    const size_t tb = sizeof(float) * static_cast<size_t>(nt) * dim;
    const size_t ob = sizeof(pm_match) * static_cast<size_t>(nq) * k;
    float *dq = nullptr, *dt = nullptr;
    pm_match* dout = nullptr;
    PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dq), qb));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dt), tb ? tb : 16);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dout), ob);
    int rc = PM_OK;
    if (e != hipSuccess) {
        pm::set_error("hipMalloc failed: %s", hipGetErrorString(e));
        rc = PM_E_NOMEM;
    }
    if (rc == PM_OK) {
        e = hipMemcpyAsync(dq, q, qb, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && tb) e = hipMemcpyAsync(dt, t, tb, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { pm::set_error("H2D copy failed: %s", hipGetErrorString(e)); rc = PM_E_HIP; }
    }
    if (rc == PM_OK) rc = pm_bf_knn_l2_f32_dev(ctx, dq, nq, dt, nt, dim, k, flags, dout);
    if (rc == PM_OK) {
        e = hipMemcpyAsync(out, dout, ob, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { pm::set_error("D2H copy failed: %s", hipGetErrorString(e)); rc = PM_E_HIP; }
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(dq);
    (void)hipFree(dt);
    (void)hipFree(dout);
    return rc;
}
