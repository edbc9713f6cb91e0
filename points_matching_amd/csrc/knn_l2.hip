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
#include <cstdlib>

#include "knn_shared.hpp"

namespace {

using namespace pm_knn;

constexpr int TILE_T = 64;    // train rows per LDS tile of the exact kernel
// u8 route: ratio test + compaction + gather inside the refinement launch?  A kernel boundary costs about what the in-launch
// look-back behind the slowest workgroup costs, so from 1024 queries up the two forms measure the same (call, us, two launches
// / fused: 1024^2 14.95 / 14.83, 2048^2 16.3 / 16.2, 8192^2 26.5 / 26.1) and the separate launch stays the default; up to 512
// queries (<= 32 workgroups, one look-back hop) the fused form is 1.6-1.8 us shorter (512^2 15.1 / 13.3, 128^2 14.1 / 12.4;
// tools/small_fusion.py) and is the default.  PM_OPT_FILTER_FUSION pins either form.
constexpr int PM_U8_FUSED_MAX_NQ = 768;
__host__ inline bool u8_fused_by_default(int nq) { return nq <= PM_U8_FUSED_MAX_NQ; }
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

using pm::wave_min_f32;
using pm::wave_min_u64;

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
        for (int c = 4 * sub; c < dim; c += 64) {            // dim % 4 == 0 on this route
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
            s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
        }
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
// f16 route: integer-valued descriptors (what OpenCV's SIFT emits: 0..255 stored as float)
//
// For integer data with dim * max|x|^2 < 2^24 every product and every partial sum of q.t is an
// integer below 2^24, so v_mfma_f32_32x32x16_f16 (f16 operands, f32 accumulate) computes the dot
// product EXACTLY at 16x the f32 matrix rate.  knn_l2_prep16 writes f16 copies of both matrices in
// rows of H_ROW = 128 data halfs (zero padded) + one 16-half chunk that carries the row seed
// through the matrix pipe:   train row : (A2, A1, A0, 0, ...),  ||t||^2 = 8192*A2 + 64*A1 + A0
//                            query row : (-4096, -32, -0.5, 0, ...)
// so the ninth k-chunk contributes exactly -||t||^2/2 and the accumulator ends as
// w = q.t - ||t||^2/2 with no VALU work.  The copies are padded to whole tiles (pad train rows
// carry A2 = 60000, i.e. w << any real row), which removes every bounds check from the kernel.
// prep16 also VERIFIES the premise on the device (integers, |x| <= 361, finite): if it fails, bit 1
// of stats[1] is raised and this route's result is not used (auto mode: the f32 kernel runs
// instead; hint mode: the refinement re-scans exactly).  Wrong data costs time, never correctness.
//
// (The coarse kernel itself lives in knn_coarse.hip.  One row group per workgroup instead of the
// 4-iteration loop below was measured slower: 11.7 vs 8.3 us at C3.)
// ---------------------------------------------------------------------------------------------
// SEEDED (round 3, hint route): rows of 128 halfs with no seed chunk; -||t||^2/2 goes to `seeds` (seed order,
// knn_shared.hpp) and starts the accumulators of the coarse kernel instead (RouteF16S).
// DP (round 3): padded data columns per row, 128 or 256 — descriptors of up to 256 dimensions (and any dim: the copies
// are zero padded, so dim % 4 != 0 only changes how the f32 rows are READ).  ALIGNED: rows read as 16-byte vectors
// (dim % 4 == 0 and 16-byte aligned base pointers), else element by element.
// Integer premise for DP columns: |x| <= floor(sqrt(2^24 / DP)) keeps every partial sum below 2^24 (361 / 255).
template <int DP>
struct F16Rows {
    static constexpr int ROWH = DP + 16;                     // halfs per row: data + the seed chunk
    static constexpr int NCB = DP / 128;                     // column blocks of 128 a lane group walks
    static constexpr float MAXABS = DP == 128 ? 361.f : 255.f;
};

// the 8 columns cb + 8*sub .. +7 of one row as floats, zero beyond dim; unconditional (clamped) loads
template <bool ALIGNED>
struct Row8 {
    f32x4 a, b;
    float e[8];
    __device__ __forceinline__ void load(const float* __restrict__ p, int c, int dim)
    {
        if constexpr (ALIGNED) {
            const int c1 = c < dim ? c : dim - 4, c2 = c + 4 < dim ? c + 4 : dim - 4;      // dim % 4 == 0, dim >= 4
            a = *reinterpret_cast<const f32x4*>(p + c1);
            b = *reinterpret_cast<const f32x4*>(p + c2);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = p[c + k < dim ? c + k : dim - 1];
        }
    }
    __device__ __forceinline__ void get(int c, int dim, bool live, float (&v)[8]) const
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float x = ALIGNED ? (k < 4 ? a[k & 3] : b[k & 3]) : e[k];
            const bool in = ALIGNED ? (c + (k & ~3) < dim) : (c + k < dim);
            v[k] = (live && in) ? x : 0.f;
        }
    }
};

template <bool SEEDED, int DP, bool ALIGNED>
__global__ __launch_bounds__(256) void knn_l2_prep16(const float* __restrict__ Q, int nq, int nq_pad,
                                                     const float* __restrict__ T, int nt, int nt_pad, int dim,
                                                     float* __restrict__ qnorm, float* __restrict__ tnorm,
                                                     _Float16* __restrict__ Qh, _Float16* __restrict__ Th,
                                                     float* __restrict__ seeds,
                                                     unsigned long long* __restrict__ stats, unsigned epoch)
{
    typedef F16Rows<DP> G;
    constexpr int ROWH = SEEDED ? DP : G::ROWH;
    constexpr int NCB = G::NCB;
    __shared__ unsigned wmax[4];
    __shared__ unsigned wbad[4];
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int qblocks = nq_pad / 64;
    const bool is_t = static_cast<int>(blockIdx.x) >= qblocks;
    const float* x = is_t ? T : Q;
    const int n = is_t ? nt : nq;
    float* norm = is_t ? tnorm : qnorm;
    _Float16* xh = is_t ? Th : Qh;
    const int row0 = (is_t ? blockIdx.x - qblocks : blockIdx.x) * 64;
    unsigned mx = 0u, bad = 0u;
    // all the loads of a thread's four rows are requested before anything is computed (unconditional, clamped
    // addresses: a guarded load makes hipcc wait for each in turn — four dependent memory round trips instead of one)
    const int c0 = 8 * sub;
    Row8<ALIGNED> ld[4][NCB];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 16 + grp;
        const float* p = x + static_cast<size_t>(row < n ? row : n - 1) * dim;       // n >= 1 on this route
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) ld[it][cb].load(p, 128 * cb + c0, dim);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 16 + grp;                 // < n_pad by construction
        const bool live = row < n;
        float s = 0.f;
        bool okrow = true;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float v[8];
            ld[it][cb].get(128 * cb + c0, dim, live, v);
            f16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s = fmaf(v[e], v[e], s);
                okrow &= (v[e] == __builtin_rintf(v[e])) && (__builtin_fabsf(v[e]) <= G::MAXABS);
                hv[e] = static_cast<_Float16>(v[e]);
            }
            *reinterpret_cast<f16x8*>(xh + static_cast<size_t>(row) * ROWH + 128 * cb + c0) = hv;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
        if (!okrow) bad |= 2u;
        if (sub == 0) {
            if (live) {
                norm[row] = s;
                if (!(s < KNN_INF)) bad |= 3u;
                else if (is_t) mx = max(mx, f32_bits(s));
            }
            if constexpr (SEEDED) {
                // exact when eligible (||t||^2 an integer below 2^24); rows padding the last tile sit below every real row
                if (is_t) seeds[seed_pos(row)] = live ? -0.5f * s : -1.0e30f;
                continue;
            }
            f16x8 e0 = {0, 0, 0, 0, 0, 0, 0, 0};
            const f16x8 e1 = {0, 0, 0, 0, 0, 0, 0, 0};
            if (is_t) {
                if (live) {
                    const unsigned tn = s < 16777216.f ? static_cast<unsigned>(s) : 0u;   // exact when eligible
                    e0[0] = static_cast<_Float16>(static_cast<float>(tn >> 13));
                    e0[1] = static_cast<_Float16>(static_cast<float>((tn >> 6) & 127u));
                    e0[2] = static_cast<_Float16>(static_cast<float>(tn & 63u));
                } else {
                    e0[0] = static_cast<_Float16>(60000.f);                               // pad row: w ~ -2.4e8
                }
            } else {
                e0[0] = static_cast<_Float16>(-4096.f);
                e0[1] = static_cast<_Float16>(-32.f);
                e0[2] = static_cast<_Float16>(-0.5f);
            }
            *reinterpret_cast<f16x8*>(xh + static_cast<size_t>(row) * G::ROWH + DP) = e0;
            *reinterpret_cast<f16x8*>(xh + static_cast<size_t>(row) * G::ROWH + DP + 8) = e1;
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
        // stats[1] is a max, so the flags are published as the values 1 (non-finite), 2 (not
        // f16-eligible) or 3 (both): 3 >= 2 >= 1 keeps "not eligible" visible once any block saw it,
        // and a non-finite input is never eligible.
        if (bad) atomicMax(&stats[1], tag | static_cast<unsigned long long>(bad == 1u ? 3u : bad));
    }
}

// ---------------------------------------------------------------------------------------------
// u8 route (round 3): u8-valued descriptors — what OpenCV's SIFT emits, 0..255 stored as float (BASELINE configs 2, 3,
// 5) — are centred to x - 128 and ranked on v_mfma_i32_32x32x32_i8: 128-byte rows (4 k-chunks instead of the f16
// route's 9), exact integer dot products, the per-row term -(||t - 128||^2 >> 1) starting the accumulators from the
// per-tile seed array (knn_shared.hpp).  Coarse squared distance of (q, t): ||q'||^2 - 2w = d2 or d2 - 1.
// knn_l2_prep8 writes the centred byte copies (padded to whole tiles, pad rows zero with the pad seed), ||q'||^2 and
// the seeds, and VERIFIES the premise (integers in [0, 255]); a wrong hint raises bit 1 of stats[1] and the refinement
// re-scans exactly, as on the f16 hint route.
// ---------------------------------------------------------------------------------------------
template <int ITERS>
__global__ __launch_bounds__(256) void knn_l2_prep8(const float* __restrict__ Q, int nq, int nq_pad,
                                                    const float* __restrict__ T, int nt, int nt_pad, int dim,
                                                    float* __restrict__ qnorm, float* __restrict__ tnorm,
                                                    uint2* __restrict__ Q8, uint2* __restrict__ T8,
                                                    int* __restrict__ seeds, unsigned long long* __restrict__ stats,
                                                    unsigned epoch, int t_wide)
{
    __shared__ unsigned wbad[4];
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int qblocks = nq_pad / (16 * ITERS);
    const bool is_t = static_cast<int>(blockIdx.x) >= qblocks;
    const float* x = is_t ? T : Q;
    const int n = is_t ? nt : nq;
    uint2* x8 = is_t ? T8 : Q8;
    const int row0 = (is_t ? blockIdx.x - qblocks : blockIdx.x) * (16 * ITERS);
    unsigned bad = 0u;
    const int c0 = 8 * sub;
    f32x4 ld[ITERS][2];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {                      // all the loads of a thread in flight together (see prep16)
        const int row = row0 + it * 16 + grp;
        const float* p = x + static_cast<size_t>(row < n ? row : n - 1) * dim;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = c0 + 4 * e < dim ? c0 + 4 * e : dim - 4;
            ld[it][e] = *reinterpret_cast<const f32x4*>(p + c);
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int row = row0 + it * 16 + grp;                 // < n_pad by construction
        const bool live = row < n;
        float s = 0.f;
        bool okrow = true;
        unsigned w[2] = {0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool col = live && c0 + (e & ~3) < dim;
            const float v = ld[it][e >> 2][e & 3];
            okrow &= !col || ((v == __builtin_rintf(v)) && v >= 0.f && v <= 255.f);
            const float vc = __builtin_fminf(__builtin_fmaxf(v, 0.f), 255.f);      // (NaN -> 0; the row is flagged anyway)
            const float c = col ? vc - 128.f : 0.f;
            s = fmaf(c, c, s);                                // exact: 128 * 128^2 < 2^24
            const unsigned b = col ? (static_cast<unsigned>(static_cast<int>(vc)) ^ 0x80u) : 0u;   // x - 128 as a signed byte
            w[e >> 2] |= b << (8 * (e & 3));
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
        if (!okrow) bad |= 2u;
        x8[static_cast<size_t>(row) * ((is_t && t_wide) ? U8_WIDE_ROW16 * 2 : U8_DP / 8) + sub] = uint2{w[0], w[1]};
        if (sub == 0) {
            const int si = static_cast<int>(s);
            if (is_t) {
                seeds[seed_pos(row)] = live ? -(si >> 1) : U8_PAD_SEED;
                if (t_wide) reinterpret_cast<int*>(T8)[u8_wide_seed_index(row)] = live ? -(si >> 1) : U8_PAD_SEED;
                if (live) tnorm[row] = s;                     // ||t - 128||^2 (an exact integer: the integer refinement's row term)
            } else if (live) {
                qnorm[row] = s;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad |= static_cast<unsigned>(__shfl_xor(static_cast<int>(bad), o, 64));
    if ((threadIdx.x & 63) == 0) wbad[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        bad = wbad[0] | wbad[1] | wbad[2] | wbad[3];
        if (bad) atomicMax(&stats[1], (static_cast<unsigned long long>(epoch) << 32) | 3ull);   // not u8-valued: scan exactly
    }
}

// The same for u8 INPUT rows (pm_bf_knn_l2_u8: descriptors that never were floats — BASELINE config 5 streams a quarter
// of the bytes): centring is one XOR per dword, the norm one v_dot4_i32_i8 of the centred bytes with themselves, and there is
// no premise to verify.  Rows of `dim` bytes, dim % 4 == 0, dim <= 128, 4-byte aligned.
__global__ __launch_bounds__(256) void knn_l2_prep8_u8(const uint8_t* __restrict__ Q, int nq, int nq_pad,
                                                       const uint8_t* __restrict__ T, int nt, int nt_pad, int dim,
                                                       float* __restrict__ qnorm, float* __restrict__ tnorm,
                                                       uint2* __restrict__ Q8, uint2* __restrict__ T8, int* __restrict__ seeds,
                                                       int t_wide)
{
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int qblocks = nq_pad / 64;
    const bool is_t = static_cast<int>(blockIdx.x) >= qblocks;
    const uint8_t* x = is_t ? T : Q;
    const int n = is_t ? nt : nq;
    uint2* x8 = is_t ? T8 : Q8;
    const int row0 = (is_t ? blockIdx.x - qblocks : blockIdx.x) * 64;
    const int c0 = 8 * sub;
    unsigned ld[4][2];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 16 + grp;
        const uint8_t* p = x + static_cast<size_t>(row < n ? row : n - 1) * dim;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = c0 + 4 * e < dim ? c0 + 4 * e : dim - 4;
            ld[it][e] = *reinterpret_cast<const unsigned*>(p + c);
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 16 + grp;
        const bool live = row < n;
        unsigned w[2];
        int si = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            w[e] = (live && c0 + 4 * e < dim) ? (ld[it][e] ^ 0x80808080u) : 0u;
            si = __builtin_amdgcn_sdot4(static_cast<int>(w[e]), static_cast<int>(w[e]), si, false);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) si += __shfl_xor(si, o, 16);
        x8[static_cast<size_t>(row) * ((is_t && t_wide) ? U8_WIDE_ROW16 * 2 : U8_DP / 8) + sub] = uint2{w[0], w[1]};
        if (sub == 0) {
            if (is_t) {
                seeds[seed_pos(row)] = live ? -(si >> 1) : U8_PAD_SEED;
                if (t_wide) reinterpret_cast<int*>(T8)[u8_wide_seed_index(row)] = live ? -(si >> 1) : U8_PAD_SEED;
                if (live) tnorm[row] = static_cast<float>(si);
            } else if (live) {
                qnorm[row] = static_cast<float>(si);
            }
        }
    }
}

// u8 rows -> f32 rows (shapes the u8 route does not take: the f32 matcher then runs on the widened copies, same bits)
__global__ __launch_bounds__(256) void knn_u8_widen(const uint8_t* __restrict__ x, size_t n, float* __restrict__ y)
{
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256)
        y[i] = static_cast<float>(x[i]);
}

// ---------------------------------------------------------------------------------------------
// f16 route for GENERAL floats (automatic mode, data that failed the integer premise — e.g. SURF's unit-norm
// descriptors, main.cpp:37-40): the same f16 coarse kernel on f16-ROUNDED copies, 16x the f32 matrix rate, with the
// rounding paid for by a wider refinement window (docs/SPEC.md S1c):
//     |fl16(q).fl16(t) - q.t| <= (2^-10 + 2^-22) ||q|| ||t||  <=  2^-11 (1 + 2^-12) (||q||^2 + ||t||^2)
// (every product of two f16 values is exact in the f32 accumulator).  The copies are scaled by powers of two:
//   train rows by ONE factor S_t = 2^(10 - h_t), h_t = ceil(log2 max ||t||) from the norm maximum prep16 published
//     (a query ranks train rows against each other: they need a common scale), so ||S_t t|| < 2^10;
//   query row i by its OWN factor S_i = S_t 2^j, j = clamp(h_t - h_i, -3, 7) (a query's ranking does not depend on
//     its scale): ||S_i q_i|| < 2^10 whenever its norm is within [2^-7, 2^3] of the train maximum, smaller rows lose
//     bits (their window is wide in relation to their distances: candidates, at worst re-scans), rows more than
//     8x larger than every train row are not ranked at all (NaN seed: the refinement scans them exactly).
// The seed chunk carries ||S_t t||^2 rounded to 1/16 (an integer below 2^24 in three exact f16 digits, as on the
// integer route) against query constants (-256, -2, -1/32) 2^j, so the accumulator of (i, t) ends as
// S_i S_t (q_i.t - ||t||^2 / 2) + O(2^j / 64); the refinement multiplies by 1 / (S_i S_t).  Within j in [-3, 7] every
// query constant is an exact f16 and the rows padding the last tile (A2 = A1 = 60000) stay below every real row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int half_exp_ceil(float norm2)          // h with sqrt(norm2) < 2^h
{
    int e;
    (void)__builtin_frexpf(norm2, &e);          // norm2 = m * 2^e, 0.5 <= m < 1
    return (e + 1) >> 1;
}

__device__ __forceinline__ int train_half_exp(unsigned long long s0, unsigned epoch)
{
    const float tmax = static_cast<unsigned>(s0 >> 32) == epoch ? __uint_as_float(static_cast<unsigned>(s0)) : 0.f;
    return tmax > 0.f ? half_exp_ceil(tmax) : 10;
}

struct QueryScale {
    float sq;            // S_i
    float unscale;       // 1 / (S_i S_t)
    float qc[3];         // seed constants of the query row
    bool unranked;       // ||q_i|| > 8 max ||t||: no f16 image
};

__device__ __forceinline__ QueryScale query_scale(int ht, float qnorm2)     // the SAME function in prep16g and the refinement
{
    QueryScale g;
    const int hq = qnorm2 > 0.f ? half_exp_ceil(qnorm2) : ht;
    const int j = ht - hq;
    g.unranked = j < -3;
    const int jc = j < -3 ? -3 : (j > 7 ? 7 : j);
    g.sq = __builtin_ldexpf(1.f, 10 - ht + jc);
    g.unscale = __builtin_ldexpf(1.f, 2 * ht - 20 - jc);
    g.qc[0] = -__builtin_ldexpf(1.f, 8 + jc);
    g.qc[1] = -__builtin_ldexpf(1.f, 1 + jc);
    g.qc[2] = -__builtin_ldexpf(1.f, jc - 5);
    return g;
}

__global__ void knn_gen_off(unsigned long long* __restrict__ stats, unsigned epoch)
{
    if (threadIdx.x == 0) atomicMax(&stats[3], (static_cast<unsigned long long>(epoch) << 32) | 1ull);
}

template <int DP, bool ALIGNED>
__global__ __launch_bounds__(256) void knn_l2_prep16g(const float* __restrict__ Q, int nq, int nq_pad,
                                                      const float* __restrict__ T, int nt, int nt_pad, int dim,
                                                      const float* __restrict__ qnorm, const float* __restrict__ tnorm,
                                                      _Float16* __restrict__ Qh, _Float16* __restrict__ Th,
                                                      const unsigned long long* __restrict__ stats, unsigned epoch)
{
    typedef F16Rows<DP> G;
    constexpr int NCB = G::NCB;
    const unsigned long long s1 = stats[1];
    const bool flagged = static_cast<unsigned>(s1 >> 32) == epoch;
    if (!(flagged && (s1 & 2ull))) return;                  // integer-valued data: prep16's copies stand
    if (s1 & 1ull) return;                                  // a non-finite input: the refinement scans everything anyway
    const int ht = train_half_exp(stats[0], epoch);
    const float st = __builtin_ldexpf(1.f, 10 - ht);
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int qblocks = nq_pad / 64;
    const bool is_t = static_cast<int>(blockIdx.x) >= qblocks;
    const float* x = is_t ? T : Q;
    const int n = is_t ? nt : nq;
    _Float16* xh = is_t ? Th : Qh;
    const int row0 = (is_t ? blockIdx.x - qblocks : blockIdx.x) * 64;
    const int c0 = 8 * sub;
    Row8<ALIGNED> ld[4][NCB];
    float nrm[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 16 + grp;
        const int rr = row < n ? row : n - 1;
        const float* p = x + static_cast<size_t>(rr) * dim;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) ld[it][cb].load(p, 128 * cb + c0, dim);
        nrm[it] = (is_t ? tnorm : qnorm)[rr];
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 16 + grp;
        const bool live = row < n;
        QueryScale qs = query_scale(ht, nrm[it]);
        const float sc = is_t ? st : qs.sq;
        const bool zero = !live || (!is_t && qs.unranked);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float v[8];
            ld[it][cb].get(128 * cb + c0, dim, !zero, v);
            f16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = static_cast<_Float16>(v[e] * sc);     // |v * sc| < 2^10: in range; round to nearest even
            *reinterpret_cast<f16x8*>(xh + static_cast<size_t>(row) * G::ROWH + 128 * cb + c0) = hv;
        }
        if (sub == 0) {
            f16x8 e0 = {0, 0, 0, 0, 0, 0, 0, 0};
            const f16x8 e1 = {0, 0, 0, 0, 0, 0, 0, 0};
            if (is_t) {
                if (live) {
                    // ||S_t t||^2 < 2^20, kept to 1/16: an integer below 2^24
                    const float z = __builtin_rintf(nrm[it] * st * st * 16.f);
                    const unsigned tn = z < 16777216.f ? static_cast<unsigned>(z) : 16777215u;
                    e0[0] = static_cast<_Float16>(static_cast<float>(tn >> 13));
                    e0[1] = static_cast<_Float16>(static_cast<float>((tn >> 6) & 127u));
                    e0[2] = static_cast<_Float16>(static_cast<float>(tn & 63u));
                } else {
                    e0[0] = static_cast<_Float16>(60000.f);  // pad row: w = -1.55e7 2^j, below every real row (>= -2^20 - 2^19 2^j)
                    e0[1] = static_cast<_Float16>(60000.f);
                }
            } else if (live && qs.unranked) {
                e0[0] = static_cast<_Float16>(__builtin_nanf(""));    // every value of this query is NaN: nothing is ranked
            } else {
                e0[0] = static_cast<_Float16>(qs.qc[0]);
                e0[1] = static_cast<_Float16>(qs.qc[1]);
                e0[2] = static_cast<_Float16>(qs.qc[2]);
            }
            *reinterpret_cast<f16x8*>(xh + static_cast<size_t>(row) * G::ROWH + DP) = e0;
            *reinterpret_cast<f16x8*>(xh + static_cast<size_t>(row) * G::ROWH + DP + 8) = e1;
        }
    }
}

// The two passes above in ONE launch for data whose scale the caller states (PM_KNN_HINT_UNIT_NORM, round 3): every train
// row has ||t|| <= 1 (SURF, L2-normalised descriptors: what the reference itself matches, main.cpp:37-40).  The train
// scale then needs no norm maximum: the bound 1 IS the published maximum (stats[0]), S_t = 2^(10 - h(1)), every query row
// still gets its own power of two from its own norm, and the premise is verified here — a train row beyond the bound raises
// the non-finite flag, i.e. every query is scanned exactly (a wrong hint costs time, never a result bit).  Everything
// downstream (coarse pass, refinement, window) reads the same words as on the automatic route with the bound in place of
// the measured maximum: the window is the automatic route's whenever the largest train row really has norm 1.
template <int DP, bool ALIGNED>
__global__ __launch_bounds__(256) void knn_l2_prep16u(const float* __restrict__ Q, int nq, int nq_pad,
                                                      const float* __restrict__ T, int nt, int nt_pad, int dim,
                                                      float* __restrict__ qnorm, float* __restrict__ tnorm,
                                                      _Float16* __restrict__ Qh, _Float16* __restrict__ Th,
                                                      unsigned long long* __restrict__ stats, unsigned epoch)
{
    typedef F16Rows<DP> G;
    constexpr int NCB = G::NCB;
    constexpr float BOUND2 = 1.0009765625f;                 // the stated maximum of ||t||^2: 1, with 2^-10 of slack for rows that
                                                            // were normalised in f32 (same half-exponent as 1.0)
    __shared__ unsigned wbad[4];
    const int ht = half_exp_ceil(BOUND2);
    const float st = __builtin_ldexpf(1.f, 10 - ht);
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int qblocks = nq_pad / 64;
    const bool is_t = static_cast<int>(blockIdx.x) >= qblocks;
    const float* x = is_t ? T : Q;
    const int n = is_t ? nt : nq;
    float* norm = is_t ? tnorm : qnorm;
    _Float16* xh = is_t ? Th : Qh;
    const int row0 = (is_t ? blockIdx.x - qblocks : blockIdx.x) * 64;
    const int c0 = 8 * sub;
    unsigned bad = 0u;
    Row8<ALIGNED> ld[4][NCB];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 16 + grp;
        const float* p = x + static_cast<size_t>(row < n ? row : n - 1) * dim;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) ld[it][cb].load(p, 128 * cb + c0, dim);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 16 + grp;
        const bool live = row < n;
        float s = 0.f;                                      // the row's squared norm, as knn_l2_prep16 computes it
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float v[8];
            ld[it][cb].get(128 * cb + c0, dim, live, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(v[e], v[e], s);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
        if (live && (!(s < KNN_INF) || (is_t && !(s <= BOUND2)))) bad = 3u;      // not finite, or beyond the stated bound
        const QueryScale qs = query_scale(ht, s);
        const float sc = is_t ? st : qs.sq;
        const bool zero = !live || (!is_t && qs.unranked);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float v[8];
            ld[it][cb].get(128 * cb + c0, dim, !zero, v);
            f16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = static_cast<_Float16>(v[e] * sc);
            *reinterpret_cast<f16x8*>(xh + static_cast<size_t>(row) * G::ROWH + 128 * cb + c0) = hv;
        }
        if (sub == 0) {
            if (live) norm[row] = s;
            f16x8 e0 = {0, 0, 0, 0, 0, 0, 0, 0};
            const f16x8 e1 = {0, 0, 0, 0, 0, 0, 0, 0};
            if (is_t) {
                if (live) {
                    const float z = __builtin_rintf(s * st * st * 16.f);
                    const unsigned tn = z < 16777216.f ? static_cast<unsigned>(z) : 16777215u;
                    e0[0] = static_cast<_Float16>(static_cast<float>(tn >> 13));
                    e0[1] = static_cast<_Float16>(static_cast<float>((tn >> 6) & 127u));
                    e0[2] = static_cast<_Float16>(static_cast<float>(tn & 63u));
                } else {
                    e0[0] = static_cast<_Float16>(60000.f);
                    e0[1] = static_cast<_Float16>(60000.f);
                }
            } else if (live && qs.unranked) {
                e0[0] = static_cast<_Float16>(__builtin_nanf(""));
            } else {
                e0[0] = static_cast<_Float16>(qs.qc[0]);
                e0[1] = static_cast<_Float16>(qs.qc[1]);
                e0[2] = static_cast<_Float16>(qs.qc[2]);
            }
            *reinterpret_cast<f16x8*>(xh + static_cast<size_t>(row) * G::ROWH + DP) = e0;
            *reinterpret_cast<f16x8*>(xh + static_cast<size_t>(row) * G::ROWH + DP + 8) = e1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad |= static_cast<unsigned>(__shfl_xor(static_cast<int>(bad), o, 64));
    if ((threadIdx.x & 63) == 0) wbad[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long tag = static_cast<unsigned long long>(epoch) << 32;
        bad = wbad[0] | wbad[1] | wbad[2] | wbad[3];
        if (blockIdx.x == 0) {
            atomicMax(&stats[0], tag | f32_bits(BOUND2));   // the "maximum" every later kernel scales and bounds with
            atomicMax(&stats[1], tag | 2ull);               // general floats (not the integer route)
        }
        if (bad) atomicMax(&stats[1], tag | 3ull);
    }
}

// ---------------------------------------------------------------------------------------------
// refinement: one wave per query
// ---------------------------------------------------------------------------------------------
// the KM smallest (key, distance) pairs a lane has seen, ascending (KM = 2 for k <= 2, 4 for k <= 4)
template <int KM>
struct BestN {
    uint64_t k[KM];
    float d[KM];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int i = 0; i < KM; ++i) { k[i] = ~0ull; d[i] = KNN_INF; }
    }
    __device__ __forceinline__ void insert(uint64_t key, float dist)
    {
        if (key < k[KM - 1]) {
            k[KM - 1] = key; d[KM - 1] = dist;
#pragma unroll
            for (int i = KM - 1; i > 0; --i)
                if (k[i] < k[i - 1]) {
                    const uint64_t t = k[i]; k[i] = k[i - 1]; k[i - 1] = t;
                    const float u = d[i]; d[i] = d[i - 1]; d[i - 1] = u;
                }
        }
    }
    __device__ __forceinline__ void pop()
    {
#pragma unroll
        for (int i = 0; i + 1 < KM; ++i) { k[i] = k[i + 1]; d[i] = d[i + 1]; }
        k[KM - 1] = ~0ull; d[KM - 1] = KNN_INF;
    }
};
// ... and the KM smallest coarse values of a lane's slots (for the k-th smallest over the wave / row)
template <typename V, int KM>
struct MinN {
    V m[KM];
    __device__ __forceinline__ void init(V big)
    {
#pragma unroll
        for (int i = 0; i < KM; ++i) m[i] = big;
    }
    __device__ __forceinline__ void insert(V v)
    {
        if (v < m[KM - 1]) {
            m[KM - 1] = v;
#pragma unroll
            for (int i = KM - 1; i > 0; --i)
                if (m[i] < m[i - 1]) { const V t = m[i]; m[i] = m[i - 1]; m[i - 1] = t; }
        }
    }
    __device__ __forceinline__ void pop(V big)
    {
#pragma unroll
        for (int i = 0; i + 1 < KM; ++i) m[i] = m[i + 1];
        m[KM - 1] = big;
    }
};

// Candidate-list geometry of one coarse route (how a slot index and the embedded row id map back
// to a train row, and how wide the refinement window must be).
struct KnnGeom {
    const float* cand;        // [nq][slots]
    int slots;                // splits * KNN_C
    int tiles_per_split;
    int rows_per_tile;        // 64 (f32 route) or 128 (f16 route); ids per tile = rows/2
    unsigned lid_mask;
    float eps_coef;           // fp error of the coarse value, times (||q||^2 + max||t||^2)
    float embed_coef;         // truncation by the embedded id, times (||q||^2 + 2 max||t||^2)
    float eps_coef_gen;       // f16 route on general floats: rounding of the copies, same factor (SPEC S1c)
    float abs_gen;            // ... plus this many units of the SCALED accumulator (flushed subnormals, seed rounding)
    int int_shift;            // u8 route: candidates are ints (w << int_shift) | id and qnorm holds ||q - 128||^2; else 0
};
enum { ROUTE_F32 = 0, ROUTE_F16_HINT = 1, ROUTE_AUTO = 2, ROUTE_U8_HINT = 3 };

// position of the r-th (0-based) set bit of m; r < popcount(m)
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int r)
{
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const int c = __popcll((m >> pos) & ((1ull << w) - 1ull));
        if (r >= c) { r -= c; pos += w; }
    }
    return pos;
}

// SPEC S1 evaluated by 8 consecutive lanes: lane l (0..7) of the group IS accumulator l of the
// canonical form (same products, same order), the combine uses the canonical association.  The
// 16 loads per operand of a lane are independent, so one candidate costs about one memory round
// trip instead of dim/8 dependent ones.  Result valid in the group's lane 0.
__device__ __forceinline__ float l2sqr_canonical_coop8(const float* __restrict__ a, const float* __restrict__ b,
                                                       int dim, int l)
{
    float acc = 0.f;
    const int full8 = dim & ~7;
    int j0 = 0;
    // Blocks of 16 (then 8) column groups with NO per-element condition: all loads of a block are
    // issued before the first subtraction.  (A guarded or clamped element makes hipcc wait for
    // every pair of loads in turn: 16 dependent memory round trips per candidate.)
    for (; j0 + 128 <= full8; j0 += 128) {
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { av[u] = a[j0 + 8 * u + l]; bv[u] = b[j0 + 8 * u + l]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const float t = av[u] - bv[u]; const float p = t * t; acc = acc + p; }
    }
    for (; j0 + 64 <= full8; j0 += 64) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = a[j0 + 8 * u + l]; bv[u] = b[j0 + 8 * u + l]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const float t = av[u] - bv[u]; const float p = t * t; acc = acc + p; }
    }
    for (; j0 < full8; j0 += 8) {
        const float t = a[j0 + l] - b[j0 + l];
        const float p = t * t;
        acc = acc + p;
    }
    const float s = acc + __shfl_down(acc, 4, 8);            // lanes 0..3: acc[l] + acc[l+4]
    const float s1 = __shfl_down(s, 1, 8), s2 = __shfl_down(s, 2, 8), s3 = __shfl_down(s, 3, 8);
    float d = ((s + s1) + s2) + s3;                          // meaningful in lane 0
    for (int j = full8; j < dim; ++j) {
        const float t = a[j] - b[j];
        const float p = t * t;
        d = d + p;
    }
    return d;
}

// Ratio test + stable compaction + keypoint gather (main.cpp:49-69 in its ratio form, :77-78, :89-91) fused into the
// refinement launch.  Tiles of 32 queries = 8 workgroups.  A workgroup adds (1 << 32 | its 4 keep bits) to the tile's
// arrival word with ONE agent-scope atomic; the workgroup whose add completes the tile owns the tile's compaction: it
// publishes the tile's survivor count (epoch-tagged, so nothing is ever cleared), sums the counts of the tiles before it
// (decoupled look-back: one polling lane per earlier tile; earlier tiles belong to earlier workgroups, which the
// dispatcher started first) and writes the tile's survivors at prefix + rank.  What it needs from the other seven
// workgroups — nearest neighbour and distance of each query — reaches it as one 8-byte write-through store per query
// (agent scope, drained before the arrival add), read back with agent-scope loads.
struct KnnFuse {
    float ratio;
    const float* kp1;             // may be null together with kp2 / xy1 / xy2: match list only
    const float* kp2;
    pm_match* good;
    float* xy1;
    float* xy2;
    int* n_out;
    unsigned long long* pk;       // [nq] (distance bits << 32) | trainIdx of the nearest neighbour
    unsigned long long* tile;     // [tiles] arrival word: arrivals << 32 | 32 keep bits; back to 0 when the tile is taken
    unsigned* tilecnt;            // [tiles] epoch << 8 | survivors (epoch-tagged, never cleared)
    unsigned* err;                // set to the epoch if a look-back gave up (never observed; read by pm_ctx_filter_fusion_status)
    unsigned epoch;
};
constexpr int KF_TILE_BLOCKS = 8;                 // workgroups (of 4 queries) per tile
constexpr unsigned KF_SPIN_LIMIT = 1u << 22;

// NS = candidate-list entries a lane holds (ceil(slots / 64) rounded up to 1, 2, 4 or 8): the slot
// values are read ONCE into registers, the candidates of all lists are compacted into one per-wave
// LDS list and evaluated together (8 rows per round), so neither the number of lists nor the way the
// candidates spread over them adds rounds.  (This kernel is VALU-issue bound: one wave per query.)
// (amdgpu_num_sgpr: the fused tail's pointers pushed the kernel to 95 SGPRs, and above 80 a CU admits 7 instead of 8
// of these workgroups — 1792 of the 2048 at C3, i.e. a second round: 10 -> 20 us.  Capped, the kernel stays at 8.)
// GEN: the automatic route may have ranked general floats on rounded copies (SPEC S1c); the other routes are
// instantiated without that code (its scalar loads and the rescale cost the headline path 0.6 us at C3).
// KM: neighbours a lane keeps (2: k <= 2; 4: k <= 4 — the coarse lists hold 4 groups per (query, split), so the k-th
// smallest coarse value still bounds the k-th neighbour; a full list inside the window re-scans its split, as for k <= 2)
template <bool VEC4, int NS, bool FUSE, bool GEN, int KM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void knn_l2_refine(
    const float* __restrict__ Q, const float* __restrict__ T, const float* __restrict__ qnorm,
    const unsigned long long* __restrict__ stats, unsigned epoch, unsigned* __restrict__ diag, int nq, int nt,
    int dim, int k, KnnGeom g16, KnnGeom g32, int route, pm_match* __restrict__ out, KnnFuse fz)
{
    __shared__ int clist[4][64 * NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    bool ghost = false;                                      // FUSE: waves past the last query stay for the barriers
    if (q >= nq) {                                           // wave-uniform
        if (!FUSE) return;                                   // (no block barriers below without FUSE)
        ghost = true;
        q = nq - 1;
    }
    const float* qp = Q + static_cast<size_t>(q) * dim;
    const float na = qnorm[q];
    const unsigned long long s0 = stats[0], s1 = stats[1];
    const float tmax = static_cast<unsigned>(s0 >> 32) == epoch ? __uint_as_float(static_cast<unsigned>(s0)) : 0.f;
    const bool flagged = static_cast<unsigned>(s1 >> 32) == epoch;
    const bool ineligible = flagged && (s1 & 2ull);          // data not integer-valued / too large for f16
    // which coarse route produced the lists; a wrong integer hint voids them (exact re-scan)
    bool general = false;                                                        // f16-rounded scaled copies
    if constexpr (GEN) {
        const unsigned long long s3 = stats[3];
        const bool gen_off = static_cast<unsigned>(s3 >> 32) == epoch;          // the general-float f16 route withdrew
        general = route == ROUTE_AUTO && ineligible && !gen_off;
    }
    const bool hinted = route == ROUTE_F16_HINT || route == ROUTE_U8_HINT;
    const bool use16 = hinted || (route == ROUTE_AUTO && (!ineligible || general));
    const bool nonfinite = (flagged && (s1 & 1ull)) || (hinted && ineligible);
    const KnnGeom g = use16 ? g16 : g32;
    float unscale = 1.f, eps_c = g.eps_coef, eps_abs = 0.f;
    bool unranked = false;
    if constexpr (GEN) {
        if (general) {
            const QueryScale qs = query_scale(train_half_exp(s0, epoch), na);
            unscale = qs.unscale;
            eps_c = g.eps_coef_gen;
            eps_abs = g.abs_gen * qs.unscale;
            unranked = qs.unranked;
        }
    }
    if (diag && lane == 0 && !ghost && q == 0) diag[2] = general ? 1u : (use16 ? (g.int_shift ? 3u : 0u) : 2u);
    const int slots = g.slots, tiles_per_split = g.tiles_per_split, rows_per_tile = g.rows_per_tile;
    const unsigned lid_mask = g.lid_mask;
    // window half-width: fp error of the coarse value + truncation by the embedded row id
    // (u8 route: exact integers, the coarse value is d2 or d2 - 1)
    const float eps = g.int_shift ? 1.f : eps_c * (na + tmax) + g.embed_coef * (na + 2.f * tmax) + eps_abs;
    const float* cv = g.cand + static_cast<size_t>(q) * slots;
    const int gshift = 31 - __clz(rows_per_tile >> 3);       // groups per tile = rows/8 = 2^gshift

    // slot s = lane + 64*i: value w = q.t - ||t||^2/2 (+id bits) -> coarse squared distance
    // d2a = ||q||^2 - 2w; first of the 4 consecutive rows of the group the slot names
    float val[NS];
    int row0[NS];
    MinN<float, KM> mn;
    mn.init(KNN_INF);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = lane + 64 * i;
        const float w = s < slots ? cv[s] : -KNN_INF;
        const float ws = GEN ? w * unscale : w;                                  // (a power of two: the id bits survive)
        float v = w > -1.0e38f ? fmaf(-2.f, ws, na) : KNN_INF;
        if (g.int_shift) {                                                       // wave-uniform: integer candidates
            const int wi = static_cast<int>(__float_as_uint(w)) >> g.int_shift;
            v = (s < slots && wi > U8_PAD_SEED) ? fmaf(-2.f, static_cast<float>(wi), na) : KNN_INF;
        }
        const unsigned gid2 = __float_as_uint(w) & lid_mask;   // (group id << 1) | lane half
        const unsigned gid = gid2 >> 1;
        const int split = s >> 2, hh = static_cast<int>(gid2 & 1u);      // s / KNN_C
        const int tile = static_cast<int>(gid >> gshift), rem = static_cast<int>(gid & ((1u << gshift) - 1u));
        val[i] = v;
        row0[i] = (split * tiles_per_split + tile) * rows_per_tile + 32 * (rem >> 2) + 8 * (rem & 3) + 4 * hh;
        mn.insert(v);
    }
    static_assert(KNN_C == 4, "slot decoding assumes 4 entries per list");
    // k-th smallest coarse value over all slots (k <= KM)
    float tau = KNN_INF;
    for (int round = 0; round < k; ++round) {
        tau = wave_min_f32(mn.m[0]);
        const unsigned long long owners = __ballot(mn.m[0] == tau);
        if (owners == 0ull) break;               // NaN guard
        const int first = __ffsll(static_cast<long long>(owners)) - 1;
        if (lane == first) mn.pop(KNN_INF);
    }
    const float thr = (tau + eps) * 1.00000095367431640625f + eps;

    // Non-finite inputs (or a window that is not finite) void the coarse ranking: scan everything.
    const bool rescan = nonfinite || !(thr < KNN_INF) || unranked;
    if (diag && lane == 0 && !ghost) { if (rescan) atomicAdd(&diag[0], 1u); if (nonfinite) diag[1] = 1u; }

    BestN<KM> b;
    b.init();
    if (!rescan) {
        const int grp = lane >> 3, l = lane & 7;
        int total = 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (64 * i >= slots) break;                     // wave-uniform
            const int s = lane + 64 * i;
            const bool in = s < slots;
            const float v = val[i];
            const int j = row0[i];
            // a sub-list (4 consecutive slots) whose 4th entry is inside the window may have
            // dropped candidates: its rows are scanned below instead of trusting its slots
            unsigned long long spilled = __ballot(in && (s & (KNN_C - 1)) == KNN_C - 1 && v <= thr);
            const bool mine_spilled = (spilled >> (lane | (KNN_C - 1))) & 1ull;
            const bool is_cand = in && v <= thr && j < nt && !mine_spilled;
            const unsigned long long cand = __ballot(is_cand);
            if (is_cand) clist[wave][total + __popcll(cand & ((1ull << lane) - 1ull))] = j;
            total += __popcll(cand);
            if (diag && lane == 0 && spilled && !ghost) atomicAdd(&diag[0], static_cast<unsigned>(__popcll(spilled)));
            while (spilled) {                               // wave-uniform, rare: every row of that split
                const int split = (64 * i + __ffsll(static_cast<long long>(spilled)) - 1) / KNN_C;
                spilled &= spilled - 1ull;
                const int row_begin = split * tiles_per_split * rows_per_tile;
                for (int lid = lane; lid < tiles_per_split * rows_per_tile; lid += 64) {
                    const int row = row_begin + lid;
                    if (row < nt) {
                        const float d = __builtin_sqrtf(
                            l2sqr_canonical<VEC4>(qp, T + static_cast<size_t>(row) * dim, dim));
                        b.insert(knn_key(d, row), d);
                    }
                }
            }
        }
        // the candidate list was written by some lanes and is read by others of the SAME wave: order the LDS
        // stores before the loads explicitly (no workgroup barrier: waves run on independent queries)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nc = 4 * total;                           // every candidate group is 4 consecutive rows
        for (int r0 = 0; r0 < nc; r0 += 8) {                // 8 rows per round, 8 lanes each
            const int rank = r0 + grp;
            const bool live0 = rank < nc;
            int jj = clist[wave][live0 ? rank >> 2 : 0] + (rank & 3);
            const bool live = live0 && jj < nt;
            jj = jj < nt ? jj : nt - 1;
            const float d2 = l2sqr_canonical_coop8(qp, T + static_cast<size_t>(jj) * dim, dim, l);
            if (live && l == 0) {
                const float d = __builtin_sqrtf(d2);
                b.insert(knn_key(d, jj), d);
            }
        }
    } else {
        for (int j = lane; j < nt; j += 64) {
            const float d =
                __builtin_sqrtf(l2sqr_canonical<VEC4>(qp, T + static_cast<size_t>(j) * dim, dim));
            b.insert(knn_key(d, j), d);
        }
    }
    int nn_idx[2] = {-1, -1};                                // wave-uniform copies of the first two neighbours (FUSE)
    float nn_d[2] = {KNN_INF, KNN_INF};
    for (int c = 0; c < k; ++c) {
        const uint64_t best = wave_min_u64(b.k[0]);
        const unsigned long long owners = __ballot(b.k[0] == best);
        const int first = __ffsll(static_cast<long long>(owners)) - 1;
        const float dist = __shfl(b.d[0], first, 64);
        if (lane == first) b.pop();
        pm_match m;
        m.queryIdx = q;
        m.imgIdx = 0;
        if (best == ~0ull) { m.trainIdx = -1; m.distance = KNN_INF; }
        else { m.trainIdx = static_cast<int>(static_cast<uint32_t>(best)); m.distance = dist; }
        if (c < 2) { nn_idx[c] = m.trainIdx; nn_d[c] = m.distance; }
        if (lane == 0 && !ghost && out) out[static_cast<size_t>(q) * k + c] = m;
    }
    if (!FUSE) return;

    // ---- ratio test + stable compaction + gather, in this launch (see KnnFuse)
    __shared__ unsigned s_keep[4];
    __shared__ unsigned s_bits;
    __shared__ int s_last;
    __shared__ int s_pre[4];
    const int tid = threadIdx.x;
    {
        const float rhs = fz.ratio * nn_d[1];                // pm_filter_ratio: float multiply, strict <
        const bool keep = !ghost && nn_idx[0] >= 0 && nn_idx[1] >= 0 && nn_d[0] < rhs;
        if (lane == 0) {
            if (!ghost)
                __hip_atomic_store(&fz.pk[q], (static_cast<unsigned long long>(f32_bits(nn_d[0])) << 32) |
                                                  static_cast<unsigned long long>(static_cast<uint32_t>(nn_idx[0])),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_keep[wave] = keep ? 1u : 0u;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's 8-byte record is written through
    __syncthreads();
    const int blk = static_cast<int>(blockIdx.x), tile = blk / KF_TILE_BLOCKS;
    const int nblk = static_cast<int>(gridDim.x), ntiles = (nblk + KF_TILE_BLOCKS - 1) / KF_TILE_BLOCKS;
    if (tid == 0) {
        const unsigned bits4 = s_keep[0] | (s_keep[1] << 1) | (s_keep[2] << 2) | (s_keep[3] << 3);
        const unsigned long long add = (1ull << 32) | (static_cast<unsigned long long>(bits4) << (4 * (blk % KF_TILE_BLOCKS)));
        const unsigned long long old = __hip_atomic_fetch_add(&fz.tile[tile], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int in_tile = nblk - tile * KF_TILE_BLOCKS < KF_TILE_BLOCKS ? nblk - tile * KF_TILE_BLOCKS : KF_TILE_BLOCKS;
        const bool last = static_cast<int>(old >> 32) == in_tile - 1;
        s_last = last ? 1 : 0;
        if (last) {
            s_bits = static_cast<unsigned>(old + add);
            __hip_atomic_store(&fz.tile[tile], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
        }
    }
    __syncthreads();
    if (!s_last) return;
    const unsigned bits = s_bits;
    const int cnt = __popc(bits);
    if (tid == 0)
        __hip_atomic_store(&fz.tilecnt[tile], (fz.epoch << 8) | static_cast<unsigned>(cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int before = 0;
    for (int j = tid; j < tile; j += 256) {
        unsigned v = 0u, spins = 0u;
        for (;;) {
            v = __hip_atomic_load(&fz.tilecnt[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 8) == fz.epoch || ++spins > KF_SPIN_LIMIT) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if ((v >> 8) == fz.epoch) before += static_cast<int>(v & 255u);
        else __hip_atomic_store(fz.err, fz.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if (lane == 0) s_pre[wave] = before;
    __syncthreads();
    const int prefix = s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3];
    if (tid < 32 && ((bits >> tid) & 1u)) {
        const int qi = tile * (4 * KF_TILE_BLOCKS) + tid;
        const int off = prefix + __popc(bits & ((1u << tid) - 1u));
        const unsigned long long rec = __hip_atomic_load(&fz.pk[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pm_match m;
        m.queryIdx = qi;
        m.trainIdx = static_cast<int>(static_cast<uint32_t>(rec));
        m.imgIdx = 0;
        m.distance = __uint_as_float(static_cast<uint32_t>(rec >> 32));
        fz.good[off] = m;
        if (fz.kp1) {
            *reinterpret_cast<float2*>(fz.xy1 + 2 * static_cast<size_t>(off)) =
                *reinterpret_cast<const float2*>(fz.kp1 + 2 * static_cast<size_t>(qi));
            *reinterpret_cast<float2*>(fz.xy2 + 2 * static_cast<size_t>(off)) =
                *reinterpret_cast<const float2*>(fz.kp2 + 2 * static_cast<size_t>(m.trainIdx));
        }
    }
    if (tile == ntiles - 1 && tid == 0) *fz.n_out = prefix + cnt;
}

// ---------------------------------------------------------------------------------------------
// u8 route refinement (round 3): one wave per query, ONE LANE PER CANDIDATE ROW.  For u8-valued data every partial sum
// of the canonical f32 distance (SPEC S1) is an integer below 2^24, i.e. the canonical value IS the integer squared
// distance, whatever the summation order — so the rows are re-evaluated on the centred byte copies the coarse pass
// already made: d2 = ||q'||^2 + ||t'||^2 - 2 q'.t' with 32 v_dot4_i32_i8 per row (128 B per row instead of 512 B and
// ~400 f32 operations), distance = sqrtf(float(d2)) (correctly rounded), key = (distance bits, index) as everywhere.
// Candidate groups hold GROUP = 4, 8 or 16 rows (the coarse kernel's selection cost falls with the group size; here
// 64 rows cost what one costs).  A wrong hint (bit 1 of stats[1]) sends every query to the canonical f32 scan.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int u8_row_d2(const uint4 (&qv)[U8_ROW16], const uint4* __restrict__ T8, int row, int qn,
                                         const float* __restrict__ tnorm, int t_row16)
{
    const uint4* tp = T8 + static_cast<size_t>(row) * t_row16;
    uint4 tv[U8_ROW16];
#pragma unroll
    for (int i = 0; i < U8_ROW16; ++i) tv[i] = tp[i];
    const int tn = static_cast<int>(tnorm[row]);
    int d0 = 0, d1 = 0;
#pragma unroll
    for (int i = 0; i < U8_ROW16; ++i) {
        d0 = __builtin_amdgcn_sdot4(static_cast<int>(qv[i].x), static_cast<int>(tv[i].x), d0, false);
        d1 = __builtin_amdgcn_sdot4(static_cast<int>(qv[i].y), static_cast<int>(tv[i].y), d1, false);
        d0 = __builtin_amdgcn_sdot4(static_cast<int>(qv[i].z), static_cast<int>(tv[i].z), d0, false);
        d1 = __builtin_amdgcn_sdot4(static_cast<int>(qv[i].w), static_cast<int>(tv[i].w), d1, false);
    }
    return qn + tn - 2 * (d0 + d1);
}

// FOUR queries per wave, one 16-lane row each: the candidate slots of a query (32 at config C3) never filled a wave, and
// the first form of this kernel (one wave per query, ~300 instructions, like knn_l2_refine) was issue-bound at ~5 us for
// 8192 queries.  Here the k-th-smallest search and the final top-k are 4-step DPP reductions inside a row (row_ror: no
// cross-row traffic, no v_readlane) shared by four queries, and the usual two candidate groups of 8 rows are exactly
// one row of lanes.  A query with an overflowing list, fewer than k ranked groups or a wrong hint scans all train rows
// (rare; 16 lanes wide).
__device__ __forceinline__ unsigned row_min_u32(unsigned v)          // minimum over the lane's 16-lane row, in every lane of it
{
    v = min(v, pm::dpp_u32<0x121>(v));
    v = min(v, pm::dpp_u32<0x122>(v));
    v = min(v, pm::dpp_u32<0x124>(v));
    v = min(v, pm::dpp_u32<0x128>(v));
    return v;
}
__device__ __forceinline__ unsigned long long row_min_u64(unsigned long long v)
{
    const unsigned hi = static_cast<unsigned>(v >> 32), lo = static_cast<unsigned>(v);
    const unsigned mh = row_min_u32(hi);
    const unsigned ml = row_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return (static_cast<unsigned long long>(mh) << 32) | ml;
}

// Exact re-scans of knn_l2_refine8 (round 3).  A list whose 4th entry is inside the window may have dropped candidates
// (SPEC S1b), so the rows of that split are evaluated exactly; runs of identical train rows — repeated texture — make that
// a common case (12 identical rows fill a list with four groups at the same distance).  The first form of this kernel let
// the query's 16 lanes walk ALL train rows: 512 dependent passes at 8192 rows, and because a launch lasts as long as its
// slowest wave the matcher call went from 23 to 225 us with 52 such queries among 8192 (1 ms at 32k x 32k).  Now only the
// spilled SPLITS are evaluated, by all 64 lanes of the wave for one of its four queries at a time (a variant that handed
// them to the whole workgroup cut the tail further but cost every launch a barrier: +1 us at config C3).
template <int KM, int U, int LANES>
__device__ __forceinline__ void r8_scan_rows(BestN<KM>& tb, const uint4 (&qv)[U8_ROW16], int qn, const uint4* __restrict__ T8,
                                             const float* __restrict__ tnorm, int t_row16, int row_begin, int row_end, int me)
{
    for (int base = row_begin; base < row_end; base += LANES * U) {
        int d2[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = base + u * LANES + me;
            ok[u] = row < row_end;
            // the row in two halves of four 16-byte loads: the register peak of this rare path must not cost the common
            // path a wave per SIMD (whole rows in flight: 106 VGPRs, 4 waves; measured +0.25 us per launch at config C3)
            const uint4* tp = T8 + static_cast<size_t>(ok[u] ? row : row_begin) * t_row16;
            int acc = 0;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                uint4 tv[U8_ROW16 / 2];
#pragma unroll
                for (int i = 0; i < U8_ROW16 / 2; ++i) tv[i] = tp[hf * (U8_ROW16 / 2) + i];
#pragma unroll
                for (int i = 0; i < U8_ROW16 / 2; ++i) {
                    const uint4 a = qv[hf * (U8_ROW16 / 2) + i];
                    acc = __builtin_amdgcn_sdot4(static_cast<int>(a.x), static_cast<int>(tv[i].x), acc, false);
                    acc = __builtin_amdgcn_sdot4(static_cast<int>(a.y), static_cast<int>(tv[i].y), acc, false);
                    acc = __builtin_amdgcn_sdot4(static_cast<int>(a.z), static_cast<int>(tv[i].z), acc, false);
                    acc = __builtin_amdgcn_sdot4(static_cast<int>(a.w), static_cast<int>(tv[i].w), acc, false);
                }
                asm volatile("" : "+v"(acc));                 // (keeps the second half's loads behind the first half's use)
            }
            d2[u] = qn + static_cast<int>(tnorm[ok[u] ? row : row_begin]) - 2 * acc;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = base + u * LANES + me;
            const float d = __builtin_sqrtf(static_cast<float>(d2[u]));
            if (ok[u]) tb.insert(knn_key(d, row), d);
        }
    }
}

// FUSE (k == 2): the ratio test, the stable compaction and the keypoint gather (main.cpp:49-69 in its ratio form, :77-78,
// :89-91) ride this launch.  A workgroup holds 16 whole queries, so nothing is handed between workgroups but ONE
// epoch-tagged survivor count each (decoupled look-back over the earlier workgroups, as in filter_ratio_gather); `out`
// may then be null.
template <int NS, int GROUP, int KM, bool FUSE>
__global__ __launch_bounds__(256) void knn_l2_refine8(
    const float* __restrict__ Q, const float* __restrict__ T, const uint4* __restrict__ Q8, const uint4* __restrict__ T8,
    const float* __restrict__ qnorm, const float* __restrict__ tnorm, const int* __restrict__ cand,
    const unsigned long long* __restrict__ stats, unsigned epoch, unsigned* __restrict__ diag, int nq, int nt, int dim, int k,
    int slots, int tiles_per_split, pm_match* __restrict__ out, KnnFuse fz, int t_row16)
{
    constexpr int GPB = 16 / GROUP;                          // groups per 32-row block and lane half
    constexpr int IMAX = 0x7FFFFFFF;
    __shared__ int clist[4][4][16 * NS];                     // [wave][query of the wave][candidate]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = lane >> 4, l = lane & 15;
    const int q = blockIdx.x * 16 + wave * 4 + qi;
    const bool live = q < nq;
    const int qc = live ? q : nq - 1;                        // rows past the last query shadow it (no early exit: DPP rows stay whole)
    const unsigned long long s1 = stats[1];
    const bool wrong_hint = static_cast<unsigned>(s1 >> 32) == epoch;       // any flag: not u8-valued (or not finite)
    if (diag && threadIdx.x == 0 && blockIdx.x == 0) diag[2] = 3u;
    const int qn = static_cast<int>(qnorm[qc]);
    const int* cv = cand + static_cast<size_t>(qc) * slots;
    uint4 qv[U8_ROW16];                                      // the query's centred bytes (one address per row of lanes)
#pragma unroll
    for (int i = 0; i < U8_ROW16; ++i) qv[i] = Q8[static_cast<size_t>(qc) * U8_ROW16 + i];

    // slot s = l + 16*i: candidate (w << U8_SHIFT) | (group id << 1 | lane half); coarse squared distance
    // d2a = ||q'||^2 - 2w = d2 or d2 - 1 of the group's best row
    int val[NS], code[NS];
    MinN<int, KM> mn;
    mn.init(IMAX);
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int s = l + 16 * i;
        const int c = s < slots ? cv[s] : I8_EMPTY;
        const int wi = c >> U8_SHIFT;
        const int v = wi > U8_PAD_SEED ? qn - 2 * wi : IMAX;
        const int gid2 = c & ((1 << U8_SHIFT) - 1);
        const int gid = gid2 >> 1, hh = gid2 & 1;
        const int split = s >> 2;
        const int tile = gid / (4 * GPB), rem = gid % (4 * GPB);
        val[i] = v;
        // first row of the group's 32-row block | (group inside the block << 1) | lane half   (block rows are multiples of 32)
        code[i] = ((split * tiles_per_split + tile) * H_TT + 32 * (rem / GPB)) | ((rem % GPB) << 1) | hh;
        mn.insert(v);
    }
    static_assert(KNN_C == 4, "slot decoding assumes 4 entries per list");
    int tau = IMAX;
    for (int round = 0; round < k; ++round) {                // k <= KM
        tau = static_cast<int>(row_min_u32(static_cast<unsigned>(mn.m[0]) ^ 0x80000000u) ^ 0x80000000u);
        const unsigned owners = static_cast<unsigned>(__ballot(mn.m[0] == tau) >> (16 * qi)) & 0xFFFFu;
        if (l == __ffs(static_cast<int>(owners)) - 1) mn.pop(IMAX);
    }
    const bool full = wrong_hint || tau == IMAX;             // fewer than k ranked groups (nt < k, ...): scan everything
    const int thr = tau == IMAX ? 0 : tau + 1;

    // a list whose 4th entry is inside the window may have dropped candidates: the rows of that SPLIT are evaluated exactly
    // (by the workgroup, below); its ranked groups are then not expanded a second time
    unsigned long long spill = 0ull;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (16 * i >= slots) break;                          // wave-uniform
        const int s = l + 16 * i;
        unsigned sp = static_cast<unsigned>(__ballot(s < slots && !full && (s & (KNN_C - 1)) == KNN_C - 1 && val[i] <= thr) >> (16 * qi)) & 0xFFFFu;
        while (sp) {                                         // (at most four lists per 16 slots)
            const int bit = __ffs(static_cast<int>(sp)) - 1;
            sp &= sp - 1u;
            spill |= 1ull << ((bit + 16 * i) >> 2);
        }
    }
    int total = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (16 * i >= slots) break;                          // wave-uniform
        const int s = l + 16 * i;
        const bool is_cand = s < slots && !full && val[i] <= thr && !((spill >> (s >> 2)) & 1ull);
        const unsigned rm = static_cast<unsigned>(__ballot(is_cand) >> (16 * qi)) & 0xFFFFu;
        if (is_cand) clist[wave][qi][total + __popc(rm & ((1u << l) - 1u))] = code[i];
        total += __popc(rm);
    }
    if (diag && l == 0 && live && (full || spill)) { atomicAdd(&diag[0], 1u); if (wrong_hint) diag[1] = 1u; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    BestN<KM> b;
    b.init();
    auto take = [&](int row) {
        const int d2 = u8_row_d2(qv, T8, row, qn, tnorm, t_row16);
        const float d = __builtin_sqrtf(static_cast<float>(d2));
        b.insert(knn_key(d, row), d);
    };
    const int nrows = full ? 0 : total * GROUP;
    for (int r0 = 0; __any(r0 < nrows); r0 += 16) {          // one row per lane, 16 per query and pass
        const int idx = r0 + l;
        if (idx < nrows) {
            const int cd = clist[wave][qi][idx / GROUP];
            const int reg = ((cd >> 1) & 15) * GROUP + idx % GROUP;
            const int row = (cd & ~31) + (reg & 3) + 8 * (reg >> 2) + 4 * (cd & 1);
            if (row < nt) take(row);
        }
    }
    if (wrong_hint) {                                        // canonical scan of the f32 rows (a wrong hint only costs time)
        const float* qp = Q + static_cast<size_t>(qc) * dim;
        for (int j = l; j < nt; j += 16) {
            const float d = __builtin_sqrtf(l2sqr_canonical<true>(qp, T + static_cast<size_t>(j) * dim, dim));
            b.insert(knn_key(d, j), d);
        }
    }
    // ---- exact re-scans, by the whole WAVE for one of its four queries at a time (no barrier, no LDS: the common case pays
    // one ballot).  The owner row's masks reach the other 48 lanes by shuffles, its bytes are read again; every lane keeps the KM best keys
    // of the rows it evaluated, and KM wave-wide minimum reductions hand the best of them to the owner.
    {
        const bool need = live && !wrong_hint && (full || spill != 0ull);
        unsigned long long needy = __ballot(need && l == 0);             // bit 16*qi: row qi needs a scan
        while (needy) {
            const int src = __ffsll(static_cast<long long>(needy)) - 1;      // the owner row's first lane
            needy &= needy - 1ull;
            const int oqi = __shfl(qc, src, 64);                            // the owner's query: its bytes again (one address per wave)
            uint4 oq[U8_ROW16];
#pragma unroll
            for (int i = 0; i < U8_ROW16; ++i) oq[i] = Q8[static_cast<size_t>(oqi) * U8_ROW16 + i];
            const int oqn = __shfl(qn, src, 64);
            const int oall = __shfl(full ? 1 : 0, src, 64);
            const unsigned mlo = __shfl(static_cast<unsigned>(spill), src, 64), mhi = __shfl(static_cast<unsigned>(spill >> 32), src, 64);
            unsigned long long mm = oall ? 0ull : ((static_cast<unsigned long long>(mhi) << 32) | mlo);
            BestN<KM> tb;
            tb.init();
            const int rows_per_split = tiles_per_split * H_TT;
            if (oall) {
                r8_scan_rows<KM, 1, 64>(tb, oq, oqn, T8, tnorm, t_row16, 0, nt, lane);
            } else {
                while (mm) {
                    const int sp = __ffsll(static_cast<long long>(mm)) - 1;
                    mm &= mm - 1ull;
                    const int r0 = sp * rows_per_split;
                    r8_scan_rows<KM, 1, 64>(tb, oq, oqn, T8, tnorm, t_row16, r0, r0 + rows_per_split < nt ? r0 + rows_per_split : nt, lane);
                }
            }
            for (int j = 0; j < KM; ++j) {                               // the KM best of the wave's keys -> the owner's first lane
                const uint64_t m = pm::wave_min_u64(tb.k[0]);
                if (tb.k[0] == m && m != ~0ull) tb.pop();                // (keys are distinct rows)
                if (lane == src && m != ~0ull) b.insert(m, __uint_as_float(static_cast<unsigned>(m >> 32)));
            }
        }
    }
    int nn_idx[2] = {-1, -1};                                // the first two neighbours of the row's query (FUSE)
    float nn_d[2] = {KNN_INF, KNN_INF};
    for (int c = 0; c < k; ++c) {
        const uint64_t best = row_min_u64(b.k[0]);
        const unsigned owners = static_cast<unsigned>(__ballot(b.k[0] == best) >> (16 * qi)) & 0xFFFFu;
        const int first = __ffs(static_cast<int>(owners)) - 1;
        const float dist = __shfl(b.d[0], 16 * qi + first, 64);
        if (l == first) b.pop();
        pm_match m;
        m.queryIdx = q;
        m.imgIdx = 0;
        if (best == ~0ull) { m.trainIdx = -1; m.distance = KNN_INF; }
        else { m.trainIdx = static_cast<int>(static_cast<uint32_t>(best)); m.distance = dist; }
        if (c < 2) { nn_idx[c] = m.trainIdx; nn_d[c] = m.distance; }
        if (l == 0 && live && out) out[static_cast<size_t>(q) * k + c] = m;
    }
    if constexpr (FUSE) {
        __shared__ unsigned s_bits[4];
        __shared__ unsigned long long s_rec[16];
        __shared__ int s_pre[4];
        const int tid = threadIdx.x;
        const float rhs = fz.ratio * nn_d[1];                // pm_filter_ratio: float multiply, strict <
        const bool keep = live && nn_idx[0] >= 0 && nn_idx[1] >= 0 && nn_d[0] < rhs;
        if (l == 0)
            s_rec[wave * 4 + qi] = (static_cast<unsigned long long>(f32_bits(nn_d[0])) << 32) |
                                   static_cast<unsigned long long>(static_cast<uint32_t>(nn_idx[0]));
        const unsigned long long bal = __ballot(l == 0 && keep);
        if (lane == 0)
            s_bits[wave] = static_cast<unsigned>(bal & 1ull) | (static_cast<unsigned>((bal >> 16) & 1ull) << 1) |
                           (static_cast<unsigned>((bal >> 32) & 1ull) << 2) | (static_cast<unsigned>((bal >> 48) & 1ull) << 3);
        __syncthreads();
        const unsigned bits = s_bits[0] | (s_bits[1] << 4) | (s_bits[2] << 8) | (s_bits[3] << 12);
        const int cnt = __popc(bits);
        const int blk = static_cast<int>(blockIdx.x);
        if (tid == 0)
            __hip_atomic_store(&fz.tilecnt[blk], (fz.epoch << 8) | static_cast<unsigned>(cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int before = 0;
        for (int j = tid; j < blk; j += 256) {               // earlier workgroups were dispatched earlier: bounded spin anyway
            unsigned v = 0u, spins = 0u;
            for (;;) {
                v = __hip_atomic_load(&fz.tilecnt[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 8) == fz.epoch || ++spins > KF_SPIN_LIMIT) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if ((v >> 8) == fz.epoch) before += static_cast<int>(v & 255u);
            else __hip_atomic_store(fz.err, fz.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
        if (lane == 0) s_pre[wave] = before;
        __syncthreads();
        const int prefix = s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3];
        if (tid < 16 && ((bits >> tid) & 1u)) {
            const int qg = blk * 16 + tid;
            const int off = prefix + __popc(bits & ((1u << tid) - 1u));
            const unsigned long long rec = s_rec[tid];
            pm_match m;
            m.queryIdx = qg;
            m.trainIdx = static_cast<int>(static_cast<uint32_t>(rec));
            m.imgIdx = 0;
            m.distance = __uint_as_float(static_cast<uint32_t>(rec >> 32));
            fz.good[off] = m;
            if (fz.kp1) {
                *reinterpret_cast<float2*>(fz.xy1 + 2 * static_cast<size_t>(off)) =
                    *reinterpret_cast<const float2*>(fz.kp1 + 2 * static_cast<size_t>(qg));
                *reinterpret_cast<float2*>(fz.xy2 + 2 * static_cast<size_t>(off)) =
                    *reinterpret_cast<const float2*>(fz.kp2 + 2 * static_cast<size_t>(m.trainIdx));
            }
        }
        if (blk == static_cast<int>(gridDim.x) - 1 && tid == 0) *fz.n_out = prefix + cnt;
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
    const bool vec4 = (dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(T)) & 15) == 0;

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

int run_exact(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim, int k, pm_match* dout)
{
    if (nq == 0) return PM_OK;
    PM_REQUIRE(dout != nullptr, PM_E_INVALID, "this shape takes the exact kernel, which needs the k-NN record buffer (d_knn)");
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


// per-context words of the fused compaction: arrival word + epoch-tagged survivor count per 32-query tile
int kf_prepare(pm_ctx* ctx, int nq, KnnFuse& fz, int queries_per_tile = 32)
{
    const int tiles = (nq + queries_per_tile - 1) / queries_per_tile;
    if (tiles > ctx->kf_cap) {
        PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->kf_tile) PM_HIP_CHECK(hipFree(ctx->kf_tile));
        ctx->kf_tile = nullptr;
        ctx->kf_cap = 0;
        const int cap = tiles < 4096 ? 4096 : tiles + tiles / 2;
        PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->kf_tile), static_cast<size_t>(cap) * 12));
        PM_HIP_CHECK(hipMemsetAsync(ctx->kf_tile, 0, static_cast<size_t>(cap) * 12, ctx->stream));
        ctx->kf_cap = cap;
        ctx->kf_epoch = 0;
    }
    if (++ctx->kf_epoch >= (1u << 24)) {          // 24-bit tag: restart
        PM_HIP_CHECK(hipMemsetAsync(ctx->kf_tile, 0, static_cast<size_t>(ctx->kf_cap) * 12, ctx->stream));
        ctx->kf_epoch = 1;
    }
    fz.tile = ctx->kf_tile;
    fz.tilecnt = reinterpret_cast<unsigned*>(ctx->kf_tile + ctx->kf_cap);
    fz.err = reinterpret_cast<unsigned*>(ctx->knn_stats) + 12;      // byte 48 of the side-band block
    fz.epoch = ctx->kf_epoch;
    return PM_OK;
}

// the matcher; `fuse` (k == 2): ratio test + compaction + gather ride the refinement launch (MFMA routes) or follow as
// pm_filter_ratio_gather_dev (exact kernel)
// uq / ut != null: the rows are u8 (pm_bf_knn_l2_u8); only the u8 route takes them — the return value 2 tells the caller
// that this shape needs the f32 path on widened copies (dq / dt are not read in that mode).
int knn_l2_enqueue(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim, int k, int flags, pm_match* dout,
                   KnnFuse* fuse, const uint8_t* uq = nullptr, const uint8_t* ut = nullptr)
{
    const bool u8in = uq != nullptr;
    if (u8in) {
        const bool al = ((reinterpret_cast<uintptr_t>(uq) | reinterpret_cast<uintptr_t>(ut)) & 3) == 0;
        if (!(k <= 4 && (dim % 4) == 0 && dim <= 128 && nt >= 1 && al)) return 2;
        flags = PM_KNN_HINT_U8;
    }

    // the MFMA routes read rows as 16-byte vectors: dim % 4 == 0 AND 16-byte aligned base pointers (anything else
    // takes the exact kernel, whose loads are scalar unless both hold)
    // Rows are read as 16-byte vectors when dim % 4 == 0 and the base pointers are 16-byte aligned; the f16 passes (round 3)
    // also take any other layout (element loads in the prep and refinement kernels: the padded copies do not care) and up
    // to 256 dimensions (17 k-chunks).  The f32-input pass and the u8 route stay at dim % 4 == 0, dim <= 128.
    const bool aligned16 = u8in || ((reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dt)) & 15) == 0;
    const bool vec = (dim % 4) == 0 && aligned16;
    const bool narrow = vec && dim <= 128;
    const bool wide16 = !narrow && !(flags & PM_KNN_FORCE_F32) && ctx->opts[PM_OPT_KNN_WIDE] != 1;
    const bool fast = !(flags & PM_KNN_FORCE_EXACT) && (k <= 2 || (k <= 4 && ctx->opts[PM_OPT_KNN_WIDE] != 1)) && dim <= 256 &&
                      nt >= 1 && (narrow || wide16) && (dim >= 4 || !vec);
    if (!fast) {
        if (u8in) return 2;                                  // (u8 rows have no f32 image here: the caller widens first)
        const int rx = run_exact(ctx, dq, nq, dt, nt, dim, k, dout);
        return rx == PM_OK && fuse ? 1 : rx;                 // 1: done, but the caller still has to filter
    }
    int route = (flags & PM_KNN_FORCE_F32) ? ROUTE_F32 : ((flags & PM_KNN_HINT_U8) && narrow) ? ROUTE_U8_HINT :
                (flags & (PM_KNN_HINT_INTEGER | PM_KNN_HINT_U8)) ? ROUTE_F16_HINT : ROUTE_AUTO;
    const int dp16 = dim <= 128 ? 128 : 256;                 // padded columns of the f16 copies
    // automatic route: general floats rank on rounded f16 copies too (SPEC S1c); the f32-input pass is enqueued only when
    // forced (PM_KNN_FORCE_F32) or when PM_OPT_KNN_GENERAL_F16 = 1 keeps it as the automatic route's pass for such data
    // PM_KNN_HINT_UNIT_NORM: the automatic route's general-float form with its two prep launches in one (knn_l2_prep16u)
    const bool unit_hint = (flags & PM_KNN_HINT_UNIT_NORM) && route == ROUTE_AUTO;
    const bool gen32 = route == ROUTE_AUTO && !unit_hint && ctx->opts[PM_OPT_KNN_GENERAL_F16] == 1 && narrow;
    const bool want32 = route == ROUTE_F32 || gen32, want16 = route != ROUTE_F32;

    // ---- f32 route geometry: 64-row tiles, 128 queries per workgroup, two workgroups per CU.
    // (A 128-row tile with one workgroup per CU measured 172 us against 153 us at C3.)
    constexpr int TT = TT32;
    KnnGeom g32{}, g16{};
    int splits32 = 1, splits16 = 1, lid_bits32 = 3, lid_bits16 = 4;
    {
        const int nqb = (nq + QB - 1) / QB;
        const int ntiles = (nt + TT - 1) / TT;
        int splits = (2 * ctx->n_cu + nqb - 1) / nqb;
        // at most 2048 train rows per split: the id embedded in a candidate costs mantissa bits, and with them
        // the window widens (more candidates, overflowing lists -> split re-scans): 9-10 id bits at most
        if (splits < (ntiles + 31) / 32) splits = (ntiles + 31) / 32;
        if (splits > ntiles) splits = ntiles;
        if (splits > 64) splits = 64;
        if (splits < 1) splits = 1;
        g32.tiles_per_split = (ntiles + splits - 1) / splits;
        splits32 = (ntiles + g32.tiles_per_split - 1) / g32.tiles_per_split;
        g32.slots = splits32 * KNN_C;
        g32.rows_per_tile = TT;
        // candidate id = (row-group id inside a lane's stream: tile_in_split*(TT/8) + block*4 + group) * 2 + lane half,
        // in the low mantissa bits
        while ((1 << lid_bits32) < g32.tiles_per_split * (TT / 8) * 2) ++lid_bits32;
        g32.lid_mask = (1u << lid_bits32) - 1u;
        // |coarse - canonical| <= (6*dim + 32) * 2^-24 * (||q||^2 + ||t||^2), plus the id truncation
        // 2^(bits-23) * (||q||^2 + 2||t||^2); see docs/SPEC.md S1b
        g32.eps_coef = static_cast<float>((6.0 * dim + 32.0) * 5.9604644775390625e-8 * 1.001);
        g32.embed_coef = static_cast<float>(static_cast<double>(1u << lid_bits32) * 1.1920928955078125e-7 * 1.01);
    }
    // ---- f16 route geometry: 128-row tiles, 256 queries per workgroup (4 waves x 64)
    // (u8 ring kernel in its 16-wave form, PM_OPT_KNN_F16_WAVES = 3: 512 queries per workgroup)
    // u8 coarse kernel form (PM_OPT_KNN_RING): 1 two LDS tile buffers, 2 / 3 ring, 4 / 5 register-operand forms (128 queries
    // per workgroup), 6 register-operand form with a split per WAVE — long sweeps only: taken when every CU stays busy with
    // splits of at least 8 tiles (1024 rows), else the two-buffer tile kernel runs
    const bool u8_default_group = ctx->opts[PM_OPT_KNN_U8_GROUP] != 1 && ctx->opts[PM_OPT_KNN_U8_GROUP] != 3;
    const bool u8_asked = ((flags & PM_KNN_HINT_U8) || u8in) && !(flags & PM_KNN_FORCE_F32);
    int u8_form = (u8_asked && u8_default_group) ? ctx->opts[PM_OPT_KNN_RING] : 1;
    int ws_splits = 0;
    if (u8_form == 6) {
        const int ntl = (nt + H_TT - 1) / H_TT, nqb128 = (nq + H_QB - 1) / H_QB * 2;
        int sp = (8 * ctx->n_cu + nqb128 - 1) / nqb128;                // 8 waves per workgroup, one split each
        if (sp < (ntl + 15) / 16) sp = (ntl + 15) / 16;
        if (sp > 64) sp = 64;
        if (sp > ntl) sp = ntl;
        if (sp < 1) sp = 1;
        const int tps = (ntl + sp - 1) / sp;
        const bool fits = (static_cast<long long>(nt) + 3 * H_TT) * (U8_WIDE_ROW16 * 16) < 0x7FFFFFFFLL;   // 32-bit DMA offsets
        if (tps >= 8 && tps <= 16 && fits) ws_splits = sp; else u8_form = 1;
    }
    if (u8_form == 5 && (static_cast<long long>(nt) + 3 * H_TT) * (U8_WIDE_ROW16 * 16) >= 0x7FFFFFFFLL) u8_form = 1;
    const int qb_wg = u8_form >= 4 ? 128 : (u8_form >= 2 && ctx->opts[PM_OPT_KNN_F16_WAVES] == 3) ? 512 : H_QB;
    const int q_unit = qb_wg > H_QB ? qb_wg : H_QB;           // (a multiple of 256 also when workgroups take 128 queries)
    const int nq_pad = (nq + q_unit - 1) / q_unit * q_unit, nt_pad = (nt + H_TT - 1) / H_TT * H_TT;
    {
        const int nqb = nq_pad / qb_wg;
        const int ntiles = nt_pad / H_TT;
        // train splits sized for ONE workgroup per CU: with LDS-DMA staging a lone workgroup keeps the matrix pipe as busy as
        // two co-resident ones did with register staging (C3: 18.9 vs 19.0-21.7 us, 4096 x 4096: 10.0 vs 11.6 us), and half
        // the splits are half the candidate lists the refinement has to read.  PM_OPT_KNN_WG_PER_CU = 2: two per CU.
        const int wg_per_cu = ctx->opts[PM_OPT_KNN_WG_PER_CU] == 2 ? 2 : 1;
        int splits = ws_splits ? ws_splits : (wg_per_cu * ctx->n_cu + nqb - 1) / nqb;
        if (splits < (ntiles + 15) / 16) splits = (ntiles + 15) / 16;          // <= 2048 rows per split (see above)
        if (splits > ntiles) splits = ntiles;
        if (splits > 64) splits = 64;
        if (splits < 1) splits = 1;
        g16.tiles_per_split = (ntiles + splits - 1) / splits;
        splits16 = (ntiles + g16.tiles_per_split - 1) / g16.tiles_per_split;
        g16.slots = splits16 * KNN_C;
        g16.rows_per_tile = H_TT;
        while ((1 << lid_bits16) < g16.tiles_per_split * (H_TT / 8) * 2) ++lid_bits16;
        g16.lid_mask = (1u << lid_bits16) - 1u;
        g16.eps_coef = 0.f;           // integer data: the f16 products and f32 sums are exact
        // general floats through the same kernel (SPEC S1c).  d2a = ||q||^2 - 2w, so the window pays TWICE the error of
        // w: 2 (2^-10 + 2^-22) ||q|| ||t|| <= 2^-10 (1 + 2^-12) (||q||^2 + ||t||^2) for the two roundings, an eighth on top
        // for the matrix core's internal summation order, plus the f32 route's term for the accumulation and the norms
        g16.eps_coef_gen = static_cast<float>(9.765625e-4 * 1.125 + (6.0 * dim + 32.0) * 5.9604644775390625e-8 * 1.001);
        // ... and, in units of the SCALED accumulator: f16 subnormals flushed on either operand (2 * 2^-14 * 2^10 per
        // element) and the seed's 1/16 rounding times r / 2 <= 64, both doubled
        g16.abs_gen = static_cast<float>(dim) / 4.f + 4.f;
        g16.embed_coef = static_cast<float>(static_cast<double>(1u << lid_bits16) * 1.1920928955078125e-7 * 1.01);
    }
    // the seeded forms (round 3) of the two hint routes: LDS-DMA staging only, and the u8 route's integer candidates
    // leave 9 bits for the id (<= 2048 train rows per split, which the split rule above keeps below 64 splits)
    const int seeded_opt = ctx->opts[PM_OPT_KNN_SEEDED];
    if (route == ROUTE_U8_HINT && (lid_bits16 > U8_SHIFT || (static_cast<long long>(nt_pad) + H_TT) * U8_DP >= 0x7FFFFFFFLL ||
                                   seeded_opt == 1))
    {
        if (u8in) return 2;                                     // (u8 rows: the caller widens and takes the f32 entry point)
        route = ROUTE_F16_HINT;                                 // u8-valued data satisfy the integer premise too
    }
    // rows per candidate group of the u8 route: PM_OPT_KNN_U8_GROUP 1 / 2 / 3 = 4 / 8 / 16 (0: 8)
    const int u8_group = ctx->opts[PM_OPT_KNN_U8_GROUP] == 1 ? 4 : (ctx->opts[PM_OPT_KNN_U8_GROUP] == 3 ? 16 : 8);
    // u8 refinement: integer re-evaluation on the byte copies (default) or the canonical f32 kernel (4-row groups only)
    const bool u8_int_refine = !(ctx->opts[PM_OPT_KNN_U8_REFINE] == 1 && u8_group == 4);
    // (f16 pass: the seeded form measured SLOWER than the seed chunk — C3 21.1 vs 18.7 us, 32k x 32k 199 vs 203 us: the four
    // C-in reads per block cost what the ninth MFMA cost — so it runs only when PM_OPT_KNN_SEEDED = 2 asks for it)
    const bool f16s = route == ROUTE_F16_HINT && seeded_opt == 2 && narrow &&
                      (static_cast<long long>(nt_pad) + H_TT) * (F16S_ROW16 * 16) < 0x7FFFFFFFLL;
    const bool u8r = route == ROUTE_U8_HINT;
    if (u8r) {
        g16.lid_mask = (1u << U8_SHIFT) - 1u;
        g16.int_shift = U8_SHIFT;
        g16.embed_coef = 0.f;
    }
    if (u8in && !(route == ROUTE_U8_HINT && u8_int_refine)) return 2;
    if ((want32 && lid_bits32 > 16) || (want16 && lid_bits16 > 16)) {    // > 64k rows per lane stream
        const int rx = run_exact(ctx, dq, nq, dt, nt, dim, k, dout);
        return rx == PM_OK && fuse ? 1 : rx;
    }

    // scratch: norms, f16 copies, candidate lists.  The arena is carved per call; callers that
    // interleave calls on one context are serialised by the stream.
    const size_t c32 = want32 ? sizeof(float) * static_cast<size_t>(nq) * g32.slots : 0;
    const size_t c16 = want16 ? sizeof(float) * static_cast<size_t>(nq) * g16.slots : 0;
    const size_t rowb = u8r ? U8_DP : sizeof(_Float16) * (f16s ? H_DP : dp16 + 16);   // bytes per row of the coarse copies
    const size_t qh = want16 ? rowb * static_cast<size_t>(nq_pad) : 0;
    const int t_wide = (u8r && u8_form >= 5) ? 1 : 0;            // 144-byte train rows with the seeds in the pad slots (knn_u8_rega)
    const size_t th = want16 ? (t_wide ? static_cast<size_t>(U8_WIDE_ROW16) * 16 : rowb) * static_cast<size_t>(nt_pad) : 0;
    const size_t sdb = (u8r || f16s) ? 4 * static_cast<size_t>(nt_pad + H_TT) : 0;       // seeds (+ one tile of slack)
    const size_t pkb = fuse ? sizeof(unsigned long long) * static_cast<size_t>(nq) : 0;
    const size_t need = pm::align_up(sizeof(float) * nq, 256) + pm::align_up(sizeof(float) * nt, 256) +
                        pm::align_up(c32, 256) + pm::align_up(c16, 256) + pm::align_up(qh, 256) + pm::align_up(th, 256) +
                        pm::align_up(sdb, 256) + pm::align_up(pkb, 256) + 2048;
    int rc = pm::arena_reserve(ctx, need);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    float* qnorm = static_cast<float*>(pm::arena_take(ctx, sizeof(float) * nq));
    float* tnorm = static_cast<float*>(pm::arena_take(ctx, sizeof(float) * nt));
    float* cval32 = want32 ? static_cast<float*>(pm::arena_take(ctx, c32)) : nullptr;
    float* cval16 = want16 ? static_cast<float*>(pm::arena_take(ctx, c16)) : nullptr;
    _Float16* Qh = want16 ? static_cast<_Float16*>(pm::arena_take(ctx, qh)) : nullptr;
    _Float16* Th = want16 ? static_cast<_Float16*>(pm::arena_take(ctx, th)) : nullptr;
    void* seeds = sdb ? pm::arena_take(ctx, sdb) : nullptr;
    PM_REQUIRE(qnorm && tnorm && (!want32 || cval32) && (!want16 || (cval16 && Qh && Th)) && (!sdb || seeds), PM_E_NOMEM,
               "scratch arena too small");
    KnnFuse fz{};
    if (fuse) {
        fz = *fuse;
        fz.pk = static_cast<unsigned long long*>(pm::arena_take(ctx, pkb));
        PM_REQUIRE(fz.pk != nullptr, PM_E_NOMEM, "scratch arena too small");
        rc = kf_prepare(ctx, nq, fz, (u8r && u8_int_refine) ? 16 : 32);      // (refine8: one count per 16-query workgroup)
        if (rc != PM_OK) return rc;
    }
    g32.cand = cval32;
    g16.cand = cval16;

    unsigned long long* stats = ctx->knn_stats;          // persistent, epoch-tagged: never cleared
    PM_REFUSE_CAPTURE(ctx);
    if (++ctx->knn_epoch == 0u) {              // 2^32 calls: restart the epoch tags
        PM_HIP_CHECK(hipMemsetAsync(stats, 0, 32, ctx->stream));
        ctx->knn_epoch = 1u;
    }
    const unsigned epoch = ctx->knn_epoch;
    unsigned* diag = nullptr;
    if (ctx->knn_diag) {
        diag = ctx->knn_diag_words;
        PM_HIP_CHECK(hipMemsetAsync(diag, 0, 12, ctx->stream));
    }
    {
        pm::ScopedKernelTime t(ctx, "knn_l2_prep");
        if (u8in)
            hipLaunchKernelGGL(knn_l2_prep8_u8, dim3(nq_pad / 64 + nt_pad / 64), dim3(256), 0, ctx->stream, uq, nq, nq_pad, ut, nt,
                               nt_pad, dim, qnorm, tnorm, reinterpret_cast<uint2*>(Qh), reinterpret_cast<uint2*>(Th),
                               static_cast<int*>(seeds), t_wide);
        else if (u8r && ctx->opts[PM_OPT_KNN_PREP_ROWS] != 1)      // 16 rows per workgroup: matcher call 23.5 -> 22.2 us at C3, 15.1 -> 14.2 at C2
            hipLaunchKernelGGL(knn_l2_prep8<1>, dim3(nq_pad / 16 + nt_pad / 16), dim3(256), 0, ctx->stream, dq, nq, nq_pad, dt,
                               nt, nt_pad, dim, qnorm, tnorm, reinterpret_cast<uint2*>(Qh), reinterpret_cast<uint2*>(Th),
                               static_cast<int*>(seeds), stats, epoch, t_wide);
        else if (u8r)
            hipLaunchKernelGGL(knn_l2_prep8<4>, dim3(nq_pad / 64 + nt_pad / 64), dim3(256), 0, ctx->stream, dq, nq, nq_pad, dt,
                               nt, nt_pad, dim, qnorm, tnorm, reinterpret_cast<uint2*>(Qh), reinterpret_cast<uint2*>(Th),
                               static_cast<int*>(seeds), stats, epoch, t_wide);
        else if (f16s)
            hipLaunchKernelGGL((knn_l2_prep16<true, 128, true>), dim3(nq_pad / 64 + nt_pad / 64), dim3(256), 0, ctx->stream, dq, nq,
                               nq_pad, dt, nt, nt_pad, dim, qnorm, tnorm, Qh, Th, static_cast<float*>(seeds), stats, epoch);
        else if (unit_hint) {
#define PM_PREP16U(DP_, AL_)                                                                                              \
    hipLaunchKernelGGL((knn_l2_prep16u<DP_, AL_>), dim3(nq_pad / 64 + nt_pad / 64), dim3(256), 0, ctx->stream, dq, nq, nq_pad, dt, \
                       nt, nt_pad, dim, qnorm, tnorm, Qh, Th, stats, epoch)
            if (dp16 == 128) { if (vec) PM_PREP16U(128, true); else PM_PREP16U(128, false); }
            else { if (vec) PM_PREP16U(256, true); else PM_PREP16U(256, false); }
#undef PM_PREP16U
        }
        else if (want16) {
#define PM_PREP16(DP_, AL_)                                                                                               \
    hipLaunchKernelGGL((knn_l2_prep16<false, DP_, AL_>), dim3(nq_pad / 64 + nt_pad / 64), dim3(256), 0, ctx->stream, dq, nq,    \
                       nq_pad, dt, nt, nt_pad, dim, qnorm, tnorm, Qh, Th, nullptr, stats, epoch)
            if (dp16 == 128) { if (vec) PM_PREP16(128, true); else PM_PREP16(128, false); }
            else { if (vec) PM_PREP16(256, true); else PM_PREP16(256, false); }
#undef PM_PREP16
        }
        else
            hipLaunchKernelGGL(knn_l2_prep, dim3((nq + 63) / 64 + (nt + 63) / 64), dim3(256), 0, ctx->stream, dq, nq, dt,
                               nt, dim, qnorm, tnorm, stats, epoch);
        // automatic route: data that failed the integer premise get f16-ROUNDED scaled copies instead (the train scale
        // needs the norm maximum of the pass above, hence a launch of its own; it returns at once for integer data)
        if (route == ROUTE_AUTO && !gen32 && !unit_hint) {
#define PM_PREP16G(DP_, AL_)                                                                                              \
    hipLaunchKernelGGL((knn_l2_prep16g<DP_, AL_>), dim3(nq_pad / 64 + nt_pad / 64), dim3(256), 0, ctx->stream, dq, nq, nq_pad, dt, \
                       nt, nt_pad, dim, qnorm, tnorm, Qh, Th, stats, epoch)
            if (dp16 == 128) { if (vec) PM_PREP16G(128, true); else PM_PREP16G(128, false); }
            else { if (vec) PM_PREP16G(256, true); else PM_PREP16G(256, false); }
#undef PM_PREP16G
        }
        else if (gen32)
            hipLaunchKernelGGL(knn_gen_off, dim3(1), dim3(64), 0, ctx->stream, stats, epoch);
        PM_HIP_CHECK(hipGetLastError());
    }
    if (u8r) {
        rc = launch_coarse_u8(ctx, Qh, Th, static_cast<const int*>(seeds), nq, nq_pad, nt, splits16, g16.tiles_per_split,
                              reinterpret_cast<int*>(cval16), g16.slots, u8_group, u8_form);
        if (rc != PM_OK) return rc;
    } else if (f16s) {
        rc = launch_coarse_f16s(ctx, Qh, Th, static_cast<const float*>(seeds), nq, nq_pad, nt, splits16, g16.tiles_per_split,
                                ~g16.lid_mask, cval16, g16.slots);
        if (rc != PM_OK) return rc;
    } else if (want16) {
        rc = launch_coarse_f16(ctx, Qh, Th, nq, nq_pad, nt, splits16, g16.tiles_per_split, ~g16.lid_mask, cval16,
                               g16.slots, stats, epoch, route == ROUTE_AUTO ? 1 : 0, dp16);
        if (rc != PM_OK) return rc;
    }
    if (want32) {
        rc = launch_coarse_f32(ctx, dq, nq, dt, nt, dim, tnorm, splits32, g32.tiles_per_split, ~g32.lid_mask, cval32,
                               g32.slots, stats, epoch, route == ROUTE_AUTO ? 1 : 0);
        if (rc != PM_OK) return rc;
    }
    if (u8r && u8_int_refine) {
        pm::ScopedKernelTime t(ctx, "knn_l2_refine");
#define PM_R8K(NS_, GROUP_, KM_, FUSE_)                                                                                    \
    hipLaunchKernelGGL((knn_l2_refine8<NS_, GROUP_, KM_, FUSE_>), dim3((nq + 15) / 16), dim3(256), 0, ctx->stream, dq, dt,  \
                       reinterpret_cast<const uint4*>(Qh), reinterpret_cast<const uint4*>(Th), qnorm, tnorm,               \
                       reinterpret_cast<const int*>(cval16), stats, epoch, diag, nq, nt, dim, k, g16.slots, g16.tiles_per_split, dout, fz, \
                       t_wide ? U8_WIDE_ROW16 : U8_ROW16)
#define PM_R8(NS_, GROUP_) do { if (fuse) PM_R8K(NS_, GROUP_, 2, true); else if (k <= 2) PM_R8K(NS_, GROUP_, 2, false); else PM_R8K(NS_, GROUP_, 4, false); } while (0)
#define PM_R8G(NS_) do { if (u8_group == 4) PM_R8(NS_, 4); else if (u8_group == 8) PM_R8(NS_, 8); else PM_R8(NS_, 16); } while (0)
        if (g16.slots <= 16) PM_R8G(1);                      // slots of a query per lane of its 16-lane row
        else if (g16.slots <= 32) PM_R8G(2);
        else if (g16.slots <= 64) PM_R8G(4);
        else if (g16.slots <= 128) PM_R8G(8);
        else PM_R8G(16);
#undef PM_R8G
#undef PM_R8
#undef PM_R8K
        PM_HIP_CHECK(hipGetLastError());
        return PM_OK;
    }
    {
        pm::ScopedKernelTime t(ctx, "knn_l2_refine");
        const int max_slots = (want16 ? g16.slots : 0) > (want32 ? g32.slots : 0) ? g16.slots : g32.slots;
#define PM_REFINE5(VEC_, NS_, FUSE_, GEN_, KM_)                                                                      \
    hipLaunchKernelGGL((knn_l2_refine<VEC_, NS_, FUSE_, GEN_, KM_>), dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, dq, dt,  \
                       qnorm, stats, epoch, diag, nq, nt, dim, k, g16, g32, route, dout, fz)
#define PM_REFINE4(VEC_, NS_, FUSE_, GEN_) do { if (k <= 2) PM_REFINE5(VEC_, NS_, FUSE_, GEN_, 2); else PM_REFINE5(VEC_, NS_, false, GEN_, 4); } while (0)
#define PM_REFINE3(NS_, FUSE_, GEN_) do { if (vec) PM_REFINE4(true, NS_, FUSE_, GEN_); else PM_REFINE4(false, NS_, FUSE_, GEN_); } while (0)
#define PM_REFINE2(NS_, FUSE_) do { if (route == ROUTE_AUTO) PM_REFINE3(NS_, FUSE_, true); else PM_REFINE3(NS_, FUSE_, false); } while (0)
#define PM_REFINE(NS_) do { if (fuse) PM_REFINE2(NS_, true); else PM_REFINE2(NS_, false); } while (0)
        if (max_slots <= 64) PM_REFINE(1);
        else if (max_slots <= 128) PM_REFINE(2);
        else if (max_slots <= 256) PM_REFINE(4);
        else PM_REFINE(8);
#undef PM_REFINE5
#undef PM_REFINE4
#undef PM_REFINE3
#undef PM_REFINE2
#undef PM_REFINE
        PM_HIP_CHECK(hipGetLastError());
    }
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
    return knn_l2_enqueue(ctx, dq, nq, dt, nt, dim, k, flags, dout, nullptr);
}

// main.cpp:46 + :49-69 (ratio form) + :77-78 + :89-91 in one call: 2-NN, ratio test, stable compaction and keypoint
// gather, as the matcher launches followed by the filter launch, or (no record buffer given, or PM_OPT_FILTER_FUSION
// = 2) with the last three riding the refinement launch.  Same outputs either way.
extern "C" int pm_bf_knn_l2_ratio_dev(pm_ctx* ctx, const float* d_q, int nq, const float* d_t, int nt, int dim, int flags,
                                      float ratio, const float* d_kp1_xy, const float* d_kp2_xy, pm_match* d_knn,
                                      pm_match* d_good, float* d_xy1, float* d_xy2, int32_t* d_n_good)
{
    PM_REQUIRE(ctx != nullptr && d_n_good != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(nq >= 0 && nt >= 0 && dim >= 1, PM_E_INVALID, "need nq,nt >= 0, dim >= 1");
    PM_REQUIRE(nq == 0 || (d_q && d_good), PM_E_INVALID, "null query/output pointer");
    PM_REQUIRE(nt == 0 || d_t, PM_E_INVALID, "null train pointer");
    PM_REQUIRE((d_kp1_xy == nullptr) == (d_kp2_xy == nullptr), PM_E_INVALID, "give both keypoint arrays or none");
    PM_REQUIRE(d_kp1_xy == nullptr || (d_xy1 && d_xy2), PM_E_INVALID, "null point outputs");
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    if (nq == 0) {
        PM_HIP_CHECK(hipMemsetAsync(d_n_good, 0, sizeof(int32_t), ctx->stream));
        return PM_OK;
    }
    KnnFuse fz{};
    fz.ratio = ratio; fz.kp1 = d_kp1_xy; fz.kp2 = d_kp2_xy; fz.good = d_good; fz.xy1 = d_xy1; fz.xy2 = d_xy2; fz.n_out = d_n_good;
    // Measured at C3 (8192 queries): refinement 10.0 us + filter launch 6.4 us against 19.8 us for the fused launch — the
    // in-launch hand-off (write-through store, drain, arrival atomic, look-back, dependent loads) costs about what the
    // launch boundary it replaces costs, and it lengthens every workgroup.  So the two-launch form is the default
    // whenever the caller provides the record buffer; PM_OPT_FILTER_FUSION = 2 selects the fused launch, 1 the two launches.
    // u8 route (PM_KNN_HINT_U8): its refinement holds 16 whole queries per workgroup, so the fused tail is one count per
    // workgroup and a look-back — PM_OPT_FILTER_FUSION = 0 takes the form that measured faster there.
    const int fusion = ctx->opts[PM_OPT_FILTER_FUSION];
    const bool u8_hint = (flags & PM_KNN_HINT_U8) && !(flags & (PM_KNN_FORCE_F32 | PM_KNN_FORCE_EXACT));
    const bool separate = fusion == 1 || (fusion == 0 && d_knn != nullptr && !(u8_hint && u8_fused_by_default(nq)));
    int rc;
    if (!separate) {
        rc = knn_l2_enqueue(ctx, d_q, nq, d_t, nt, dim, 2, flags, d_knn, &fz);
        if (rc <= 0) return rc;                                      // done (fused) or failed
        // rc == 1: the exact kernel ran (shape outside the MFMA routes) into d_knn; the filter follows as its own launch
        return pm_filter_ratio_gather_dev(ctx, d_knn, nq, 2, ratio, d_kp1_xy, d_kp2_xy, d_good, d_xy1, d_xy2, d_n_good);
    }
    PM_REQUIRE(d_knn != nullptr, PM_E_INVALID, "the two-launch form needs d_knn");
    rc = knn_l2_enqueue(ctx, d_q, nq, d_t, nt, dim, 2, flags, d_knn, nullptr);
    if (rc != PM_OK) return rc;
    return pm_filter_ratio_gather_dev(ctx, d_knn, nq, 2, ratio, d_kp1_xy, d_kp2_xy, d_good, d_xy1, d_xy2, d_n_good);
}

// ---- u8 descriptor rows (pm_bf_knn_l2_u8*): the u8 route with nothing to convert but one XOR; other shapes go through
// the f32 matcher on widened copies.  Result = pm_bf_knn_l2_f32 on the same values, bit for bit.
namespace {
int u8_widened(pm_ctx* ctx, const uint8_t* dq, int nq, const uint8_t* dt, int nt, int dim, const float** fq, const float** ft)
{
    const size_t a = pm::align_up(static_cast<size_t>(nq) * dim, 64), b = static_cast<size_t>(nt) * dim;
    if ((a + b) * sizeof(float) > ctx->widen_cap) {
        PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->widen) PM_HIP_CHECK(hipFree(ctx->widen));
        ctx->widen = nullptr;
        ctx->widen_cap = 0;
        const size_t cap = pm::align_up((a + b) * sizeof(float) * 5 / 4, size_t(1) << 20);
        PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->widen), cap));
        ctx->widen_cap = cap;
    }
    if (static_cast<size_t>(nq) * dim)
        hipLaunchKernelGGL(knn_u8_widen, dim3(256), dim3(256), 0, ctx->stream, dq, static_cast<size_t>(nq) * dim, ctx->widen);
    if (b) hipLaunchKernelGGL(knn_u8_widen, dim3(256), dim3(256), 0, ctx->stream, dt, b, ctx->widen + a);
    PM_HIP_CHECK(hipGetLastError());
    *fq = ctx->widen;
    *ft = ctx->widen + a;
    return PM_OK;
}
}  // namespace

extern "C" int pm_bf_knn_l2_u8_dev(pm_ctx* ctx, const uint8_t* dq, int nq, const uint8_t* dt, int nt, int dim, int k,
                                   pm_match* dout)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_REQUIRE(nq >= 0 && nt >= 0 && dim >= 1 && k >= 1 && k <= PM_MAX_K, PM_E_INVALID,
               "need nq,nt >= 0, dim >= 1, 1 <= k <= PM_MAX_K");
    PM_REQUIRE(nq == 0 || (dq && dout), PM_E_INVALID, "null query/output pointer");
    PM_REQUIRE(nt == 0 || dt, PM_E_INVALID, "null train pointer");
    if (nq == 0) return PM_OK;
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = nt >= 1 ? knn_l2_enqueue(ctx, nullptr, nq, nullptr, nt, dim, k, 0, dout, nullptr, dq, dt) : 2;
    if (rc != 2) return rc;
    const float *fq = nullptr, *ft = nullptr;
    rc = u8_widened(ctx, dq, nq, dt, nt, dim, &fq, &ft);
    if (rc != PM_OK) return rc;
    return knn_l2_enqueue(ctx, fq, nq, ft, nt, dim, k, PM_KNN_HINT_INTEGER, dout, nullptr);
}

extern "C" int pm_bf_knn_l2_u8_ratio_dev(pm_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int dim, float ratio,
                                         const float* d_kp1_xy, const float* d_kp2_xy, pm_match* d_knn, pm_match* d_good,
                                         float* d_xy1, float* d_xy2, int32_t* d_n_good)
{
    PM_REQUIRE(ctx != nullptr && d_n_good != nullptr && d_knn != nullptr, PM_E_INVALID, "null argument (the u8 form needs the record buffer)");
    PM_REQUIRE(nq >= 0 && nt >= 0 && dim >= 1, PM_E_INVALID, "need nq,nt >= 0, dim >= 1");
    PM_REQUIRE(nq == 0 || (d_q && d_good), PM_E_INVALID, "null query/output pointer");
    PM_REQUIRE((d_kp1_xy == nullptr) == (d_kp2_xy == nullptr), PM_E_INVALID, "give both keypoint arrays or none");
    PM_REQUIRE(d_kp1_xy == nullptr || (d_xy1 && d_xy2), PM_E_INVALID, "null point outputs");
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    if (nq == 0) {
        PM_HIP_CHECK(hipMemsetAsync(d_n_good, 0, sizeof(int32_t), ctx->stream));
        return PM_OK;
    }
    const int fusion = ctx->opts[PM_OPT_FILTER_FUSION];
    if (fusion == 2 || (fusion == 0 && u8_fused_by_default(nq))) {
        KnnFuse fz{};
        fz.ratio = ratio; fz.kp1 = d_kp1_xy; fz.kp2 = d_kp2_xy; fz.good = d_good; fz.xy1 = d_xy1; fz.xy2 = d_xy2; fz.n_out = d_n_good;
        const int rf = nt >= 1 ? knn_l2_enqueue(ctx, nullptr, nq, nullptr, nt, dim, 2, 0, d_knn, &fz, d_q, d_t) : 2;
        if (rf != 2) return rf;                              // done (fused) or failed; 2: a shape for the widened f32 path
    }
    const int rc = pm_bf_knn_l2_u8_dev(ctx, d_q, nq, d_t, nt, dim, 2, d_knn);
    if (rc != PM_OK) return rc;
    return pm_filter_ratio_gather_dev(ctx, d_knn, nq, 2, ratio, d_kp1_xy, d_kp2_xy, d_good, d_xy1, d_xy2, d_n_good);
}

extern "C" int pm_bf_knn_l2_u8(pm_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int dim, int k, pm_match* out)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_REQUIRE(nq >= 0 && nt >= 0 && dim >= 1 && k >= 1 && k <= PM_MAX_K, PM_E_INVALID,
               "need nq,nt >= 0, dim >= 1, 1 <= k <= PM_MAX_K");
    PM_REQUIRE(nq == 0 || (q && out), PM_E_INVALID, "null query/output pointer");
    PM_REQUIRE(nt == 0 || t, PM_E_INVALID, "null train pointer");
    if (nq == 0) return PM_OK;
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t qb = static_cast<size_t>(nq) * dim, tb = static_cast<size_t>(nt) * dim;
    const size_t ob = sizeof(pm_match) * static_cast<size_t>(nq) * k;
    uint8_t *dq = nullptr, *dt = nullptr;
    pm_match* dout = nullptr;
    int rc = PM_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dq), qb);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dt), tb ? tb : 16);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dout), ob);
    if (e != hipSuccess) { pm::set_error("hipMalloc failed: %s", hipGetErrorString(e)); rc = PM_E_NOMEM; }
    if (rc == PM_OK) {
        e = hipMemcpyAsync(dq, q, qb, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && tb) e = hipMemcpyAsync(dt, t, tb, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { pm::set_error("H2D copy failed: %s", hipGetErrorString(e)); rc = PM_E_HIP; }
    }
    if (rc == PM_OK) rc = pm_bf_knn_l2_u8_dev(ctx, dq, nq, dt, nt, dim, k, dout);
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
