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
// prep: squared norms, 16 lanes per row.  stats[0] = max norm (float bits), stats[1] |= 1 when a
// norm is not finite.  Approximate values only feed the coarse pass and its error bound.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_l2_prep(const float* __restrict__ x, int n, int dim,
                                                   float* __restrict__ norm,
                                                   unsigned* __restrict__ stats, int track_max)
{
    const int sub = threadIdx.x & 15;
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int rl = row < n ? row : n - 1;
    const float* p = x + static_cast<size_t>(rl) * dim;
    float s = 0.f;
    for (int c = sub; c < dim; c += 16) s = fmaf(p[c], p[c], s);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
    if (sub == 0 && row < n) {
        norm[row] = s;
        if (!(s < KNN_INF)) atomicOr(&stats[1], 1u);
        else if (track_max) atomicMax(&stats[0], f32_bits(s));
    }
}

// ---------------------------------------------------------------------------------------------
// coarse pass on the matrix cores
// ---------------------------------------------------------------------------------------------
struct Cand4 {
    float v0, v1, v2, v3;
    int i0, i1, i2, i3;
};

__device__ __forceinline__ void cand_insert(Cand4& c, float s, int j)
{
    if (s < c.v3) {
        c.v3 = s; c.i3 = j;
        if (c.v3 < c.v2) { float t = c.v2; c.v2 = c.v3; c.v3 = t; int u = c.i2; c.i2 = c.i3; c.i3 = u; }
        if (c.v2 < c.v1) { float t = c.v1; c.v1 = c.v2; c.v2 = t; int u = c.i1; c.i1 = c.i2; c.i2 = u; }
        if (c.v1 < c.v0) { float t = c.v0; c.v0 = c.v1; c.v1 = t; int u = c.i0; c.i0 = c.i1; c.i1 = u; }
    }
}

// NCH = padded dim / 8.  grid = (ceil(nq/QB), splits).  Dynamic LDS: 2 tiles of
// TILE_T x (8*NCH + 4) floats (row stride padded by one 16-B slot: conflict-free ds_read_b128 for
// the 16 rows of a lane group) + 2 x TILE_T train norms.
template <int NCH>
__global__ __launch_bounds__(256, 2) void knn_l2_mfma(
    const float* __restrict__ Q, const float* __restrict__ T, const float* __restrict__ tnorm, int nq,
    int nt, int dim, int tiles_per_split, float* __restrict__ cand_val, int* __restrict__ cand_idx,
    int slots)
{
    constexpr int DP = NCH * 8;
    constexpr int LDT = DP + 4;
    constexpr int F4_PER_ROW = DP / 4;
    constexpr int NSTG = TILE_T * F4_PER_ROW / 256;
    static_assert(TILE_T * F4_PER_ROW % 256 == 0, "tile must split evenly over the workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ts = smem;                          // [2][TILE_T][LDT]
    float* Tn = smem + 2 * TILE_T * LDT;       // [2][TILE_T]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int qrow = blockIdx.x * QB + wave * 32 + r;
    const int qld = qrow < nq ? qrow : nq - 1;

    // B operand: this lane's query row, k = 8c + 4h + {0..3} for chunk c (the k permutation is
    // shared with the A operand below, and a dot product does not care about k order).
    f32x4 qf[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k0 = 8 * c + 4 * h;
        qf[c] = k0 < dim ? *reinterpret_cast<const f32x4*>(Q + static_cast<size_t>(qld) * dim + k0)
                         : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (nt + TILE_T - 1) / TILE_T;
    const int tile0 = blockIdx.y * tiles_per_split;
    int tile1 = tile0 + tiles_per_split;
    if (tile1 > ntiles) tile1 = ntiles;

    Cand4 cl{KNN_INF, KNN_INF, KNN_INF, KNN_INF, -1, -1, -1, -1};

    f32x4 stg[NSTG];
    float stg_n = 0.f;
    auto stage_load = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            const int f = tid + 256 * i;
            const int row = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
            int g = tile * TILE_T + row;
            g = g < nt ? g : nt - 1;
            stg[i] = 4 * c4 < dim
                         ? *reinterpret_cast<const f32x4*>(T + static_cast<size_t>(g) * dim + 4 * c4)
                         : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (tid < TILE_T) {
            int g = tile * TILE_T + tid;
            stg_n = tnorm[g < nt ? g : nt - 1];
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            const int f = tid + 256 * i;
            const int row = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
            *reinterpret_cast<f32x4*>(Ts + (buf * TILE_T + row) * LDT + 4 * c4) = stg[i];
        }
        if (tid < TILE_T) Tn[buf * TILE_T + tid] = stg_n;
    };

    if (tile0 < tile1) {
        stage_load(tile0);
        stage_store(0);
    }
    __syncthreads();

    for (int tile = tile0; tile < tile1; ++tile) {
        const int buf = (tile - tile0) & 1;
        const bool more = tile + 1 < tile1;
        if (more) stage_load(tile + 1);          // global loads fly under the MFMA chain

        f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 acc1 = acc0;
        const float* tb = Ts + buf * TILE_T * LDT + r * LDT + 4 * h;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(tb + 8 * c);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(tb + 32 * LDT + 8 * c);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], qf[c][t], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], qf[c][t], acc1, 0, 0, 0);
            }
        }

        // epilogue: C[i][j] sits in lane (j = lane&31), register reg with
        // i = (reg&3) + 8*(reg>>2) + 4*(lane>>5): 16 train rows per block for this lane's query.
        const float* tn = Tn + buf * TILE_T + 4 * h;
        const int jbase = tile * TILE_T + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 n0 = *reinterpret_cast<const f32x4*>(tn + 8 * g);
            const f32x4 n1 = *reinterpret_cast<const f32x4*>(tn + 32 + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j0 = jbase + 8 * g + e;
                float s0 = fmaf(-2.f, acc0[4 * g + e], n0[e]);
                s0 = j0 < nt ? s0 : KNN_INF;
                cand_insert(cl, s0, j0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j1 = jbase + 32 + 8 * g + e;
                float s1 = fmaf(-2.f, acc1[4 * g + e], n1[e]);
                s1 = j1 < nt ? s1 : KNN_INF;
                cand_insert(cl, s1, j1);
            }
        }

        if (more) stage_store(buf ^ 1);
        __syncthreads();
    }

    if (qrow < nq) {
        const size_t o = static_cast<size_t>(qrow) * slots + (blockIdx.y * 2 + h) * KNN_C;
        *reinterpret_cast<f32x4*>(cand_val + o) = f32x4{cl.v0, cl.v1, cl.v2, cl.v3};
        *reinterpret_cast<int4*>(cand_idx + o) = int4{cl.i0, cl.i1, cl.i2, cl.i3};
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
    unsigned* __restrict__ stats, int nq, int nt, int dim, int k, int slots,
    const float* __restrict__ cand_val, const int* __restrict__ cand_idx, float eps_coef,
    pm_match* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (q >= nq) return;
    const float* qp = Q + static_cast<size_t>(q) * dim;
    const float na = qnorm[q];
    const float tmax = __uint_as_float(stats[0]);
    const bool nonfinite = stats[1] != 0u;
    const float eps = eps_coef * (na + tmax);

    // k-th smallest coarse value over all slots (k <= 2)
    float m0 = KNN_INF, m1 = KNN_INF;
    for (int s = lane; s < slots; s += 64) {
        const float v = cand_val[static_cast<size_t>(q) * slots + s];
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
    const float thr = ((tau + na) + eps) * 1.00000095367431640625f + eps;

    // a sub-list whose largest kept entry is inside the window may have dropped candidates
    bool spill = false;
    for (int s = lane; s < slots; s += 64)
        if ((s & (KNN_C - 1)) == KNN_C - 1) {
            const float v = cand_val[static_cast<size_t>(q) * slots + s];
            spill |= (v + na) <= thr;
        }
    const bool rescan = nonfinite || !(thr < KNN_INF) || __any(spill);
    if (rescan && lane == 0) atomicAdd(&stats[2], 1u);

    Best2 b{~0ull, ~0ull, KNN_INF, KNN_INF};
    if (!rescan) {
        for (int s = lane; s < slots; s += 64) {
            const float v = cand_val[static_cast<size_t>(q) * slots + s];
            const int j = cand_idx[static_cast<size_t>(q) * slots + s];
            if (j >= 0 && (v + na) <= thr) {
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

template <int NCH>
int launch_mfma(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim,
                const float* tnorm, int splits, int tiles_per_split, float* cval, int* cidx, int slots)
{
    constexpr int LDT = NCH * 8 + 4;
    const size_t lds = (2 * TILE_T * LDT + 2 * TILE_T) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_l2_mfma<NCH>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        attr_done = true;
    }
    dim3 grid((nq + QB - 1) / QB, splits);
    pm::ScopedKernelTime t(ctx, "knn_l2_mfma");
    hipLaunchKernelGGL(knn_l2_mfma<NCH>, grid, dim3(256), lds, ctx->stream, dq, dt, tnorm, nq, nt, dim,
                       tiles_per_split, cval, cidx, slots);
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
    if (!fast) { ctx->last_knn_stats = nullptr; return run_exact(ctx, dq, nq, dt, nt, dim, k, dout); }

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

    // scratch: norms, stats, candidate lists.  The arena is carved per call; callers that
    // interleave calls on one context are serialised by the stream.
    const size_t need = pm::align_up(sizeof(float) * nq, 256) + pm::align_up(sizeof(float) * nt, 256) + 256 +
                        2 * pm::align_up(sizeof(float) * static_cast<size_t>(nq) * slots, 256) + 1024;
    int rc = pm::arena_reserve(ctx, need);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    float* qnorm = static_cast<float*>(pm::arena_take(ctx, sizeof(float) * nq));
    float* tnorm = static_cast<float*>(pm::arena_take(ctx, sizeof(float) * nt));
    unsigned* stats = static_cast<unsigned*>(pm::arena_take(ctx, 16));
    float* cval = static_cast<float*>(pm::arena_take(ctx, sizeof(float) * static_cast<size_t>(nq) * slots));
    int* cidx = static_cast<int*>(pm::arena_take(ctx, sizeof(int) * static_cast<size_t>(nq) * slots));
    PM_REQUIRE(qnorm && tnorm && stats && cval && cidx, PM_E_NOMEM, "scratch arena too small");

    PM_HIP_CHECK(hipMemsetAsync(stats, 0, 16, ctx->stream));
    ctx->last_knn_stats = stats;
    {
        pm::ScopedKernelTime t(ctx, "knn_l2_prep");
        hipLaunchKernelGGL(knn_l2_prep, dim3((nq + 15) / 16), dim3(256), 0, ctx->stream, dq, nq, dim, qnorm, stats, 0);
        hipLaunchKernelGGL(knn_l2_prep, dim3((nt + 15) / 16), dim3(256), 0, ctx->stream, dt, nt, dim, tnorm, stats, 1);
        PM_HIP_CHECK(hipGetLastError());
    }
    if (dim <= 32) rc = launch_mfma<4>(ctx, dq, nq, dt, nt, dim, tnorm, splits, tiles_per_split, cval, cidx, slots);
    else if (dim <= 64) rc = launch_mfma<8>(ctx, dq, nq, dt, nt, dim, tnorm, splits, tiles_per_split, cval, cidx, slots);
    else rc = launch_mfma<16>(ctx, dq, nq, dt, nt, dim, tnorm, splits, tiles_per_split, cval, cidx, slots);
    if (rc != PM_OK) return rc;

    // |coarse - canonical| <= (4*dim + 16) * 2^-24 * (||q||^2 + ||t||^2); see docs/SPEC.md S1b
    const float eps_coef = static_cast<float>((4.0 * dim + 16.0) * 5.9604644775390625e-8 * 1.001);
    {
        pm::ScopedKernelTime t(ctx, "knn_l2_refine");
        hipLaunchKernelGGL(knn_l2_refine<true>, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, dq, dt, qnorm, stats,
                           nq, nt, dim, k, slots, cval, cidx, eps_coef, dout);
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
