// ransac.hip — RANSAC fundamental-matrix estimation on gfx950 (MI355X).
//
// Replaces `cv::findFundamentalMat(Mat(selPoints1), Mat(selPoints2), CV_FM_7POINT)`
// (main.cpp:95-98) by the estimator BASELINE.json names: H fixed hypotheses, each = sample 8
// correspondences (counter-based sampler keyed by (seed, h): docs/SPEC.md S6), Hartley-normalised
// 8-point solve with rank-2 enforcement in fp64 (S7), Sampson scoring of ALL n correspondences in
// fp32 (S8), keep the hypothesis with most inliers, ties -> lowest h (S9).
//
//   ransac_solve   one lane per hypothesis: sampler + 8-point solve, F rounded to fp32
//   ransac_score   lane = hypothesis (F in 9 VGPRs), correspondences are wave-uniform and arrive
//                  as scalar operands; grid = hypothesis blocks x correspondence chunks, partial
//                  inlier counts meet in one integer atomicAdd per (lane, chunk)
//                  Single-shard runs finish inside this launch: the last workgroup of a hypothesis
//                  block reduces the block's best key = (inliers << 32) | (0xFFFFFFFF - h), the last
//                  block overall picks the winner, publishes its stored fp64 model and the mask.
//   ransac_select  (sharded runs) key only: block max, one 64-bit atomicMax; the caller all-reduces it
//   ransac_final_* re-derive F (fp64) and the inlier mask of ONE hypothesis id — every rank of a
//                  multi-GPU run does this for the all-reduced winner, so no model is broadcast.
//
// Every operation below is written in the order docs/SPEC.md fixes (explicit fma, unfused
// elsewhere; the TU is built with -ffp-contract=off) so that the CPU restatement reproduces the
// same bits.
#include <cstdlib>

#include "pm_common.hpp"

namespace {

constexpr int MODEL_STRIDE = 12;   // 9 x fp32 F, valid flag, 2 pad

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// SPEC S6: 8 distinct indices in [0, n) as a pure function of (seed, h, n).
__device__ __forceinline__ void sample8(uint64_t seed, uint64_t h, int n, int (&idx)[8])
{
    const uint64_t stream = mix64(seed ^ 0x9E3779B97F4A7C15ULL) ^ mix64(h + 0xD1B54A32D192ED03ULL);
#pragma unroll
    for (int s = 0; s < 8; ++s) idx[s] = -1;
    int cnt = 0;
    for (uint64_t d = 0; d < 64 && cnt < 8; ++d) {
        const uint64_t r = mix64(stream + (d + 1) * 0x9E3779B97F4A7C15ULL);
        const int c = static_cast<int>(((r >> 32) * static_cast<uint64_t>(static_cast<uint32_t>(n))) >> 32);
        bool dup = false;
#pragma unroll
        for (int s = 0; s < 8; ++s) dup |= (s < cnt) && (idx[s] == c);
        if (!dup) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s == cnt) idx[s] = c;
            ++cnt;
        }
    }
    for (int c = 0; cnt < 8; ++c) {
        bool dup = false;
#pragma unroll
        for (int s = 0; s < 8; ++s) dup |= (s < cnt) && (idx[s] == c);
        if (!dup) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s == cnt) idx[s] = c;
            ++cnt;
        }
    }
}

// SPEC S7, Hartley normalisation of 8 points: centroid to the origin, mean distance sqrt(2).
__device__ __forceinline__ bool hartley8(const double (&px)[8], const double (&py)[8], double (&nx)[8],
                                         double (&ny)[8], double& s, double& tx, double& ty)
{
    double cx = px[0], cy = py[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) { cx = cx + px[i]; cy = cy + py[i]; }
    cx = cx * 0.125; cy = cy * 0.125;
    double md = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double dx = px[i] - cx, dy = py[i] - cy;
        md = md + sqrt(fma(dx, dx, dy * dy));
    }
    md = md * 0.125;
    if (!(md > 0.0) || !(md < __builtin_inf())) return false;
    s = 1.4142135623730951 / md;
#pragma unroll
    for (int i = 0; i < 8; ++i) { nx[i] = (px[i] - cx) * s; ny[i] = (py[i] - cy) * s; }
    tx = -(s * cx); ty = -(s * cy);
    return true;
}

// One Jacobi (Hestenes) rotation of columns P,Q of G (and V).  Compile-time column indices keep
// everything in registers.
template <int P, int Qc>
__device__ __forceinline__ void jacobi_pair(double (&G)[3][3], double (&V)[3][3])
{
    double al = G[0][P] * G[0][P]; al = fma(G[1][P], G[1][P], al); al = fma(G[2][P], G[2][P], al);
    double be = G[0][Qc] * G[0][Qc]; be = fma(G[1][Qc], G[1][Qc], be); be = fma(G[2][Qc], G[2][Qc], be);
    double ga = G[0][P] * G[0][Qc]; ga = fma(G[1][P], G[1][Qc], ga); ga = fma(G[2][P], G[2][Qc], ga);
    // converged pair: |ga| <= 2^-52 * sqrt(al*be), tested without the root.  Same bits on both
    // sides, so the skip is deterministic; late sweeps then cost three dot products per pair.
    if (!(ga * ga > 4.930380657631324e-32 * (al * be))) return;
    const double zeta = (be - al) / (2.0 * ga);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
    const double c = 1.0 / sqrt(fma(t, t, 1.0));
    const double s = c * t;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double gp = G[i][P], gq = G[i][Qc];
        G[i][P] = fma(c, gp, -(s * gq));
        G[i][Qc] = fma(s, gp, c * gq);
        const double vp = V[i][P], vq = V[i][Qc];
        V[i][P] = fma(c, vp, -(s * vq));
        V[i][Qc] = fma(s, vp, c * vq);
    }
}

// SPEC S7: normalised 8-point solve.  Returns false for a degenerate sample (F is then 0).
__device__ __forceinline__ bool solve8(const double (&x1)[8], const double (&y1)[8], const double (&x2)[8],
                                    const double (&y2)[8], double (&F)[9])
{
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = 0.0;
    double ax[8], ay[8], bx[8], by[8], s1, t1x, t1y, s2, t2x, t2y;
    if (!hartley8(x1, y1, ax, ay, s1, t1x, t1y)) return false;
    if (!hartley8(x2, y2, bx, by, s2, t2x, t2y)) return false;

    // B = A^T (9 x 8): column c is the epipolar constraint row of correspondence c
    double B[9][8], beta[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        B[0][c] = bx[c] * ax[c]; B[1][c] = bx[c] * ay[c]; B[2][c] = bx[c];
        B[3][c] = by[c] * ax[c]; B[4][c] = by[c] * ay[c]; B[5][c] = by[c];
        B[6][c] = ax[c];         B[7][c] = ay[c];         B[8][c] = 1.0;
    }
    // Householder QR of B; reflector j stays in column j (v0 on the diagonal)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double sigma = 0.0;
#pragma unroll
        for (int i = j + 1; i < 9; ++i) sigma = fma(B[i][j], B[i][j], sigma);
        const double alpha = B[j][j];
        const double nrm = sqrt(fma(alpha, alpha, sigma));
        if (!(nrm > 0.0)) { beta[j] = 0.0; continue; }
        const double v0 = alpha + (alpha >= 0.0 ? nrm : -nrm);
        const double vtv = fma(v0, v0, sigma);
        beta[j] = 2.0 / vtv;
        B[j][j] = v0;
#pragma unroll
        for (int c = j + 1; c < 8; ++c) {
            double dot = v0 * B[j][c];
#pragma unroll
            for (int i = j + 1; i < 9; ++i) dot = fma(B[i][j], B[i][c], dot);
            const double w = beta[j] * dot;
            B[j][c] = fma(-w, v0, B[j][c]);
#pragma unroll
            for (int i = j + 1; i < 9; ++i) B[i][c] = fma(-w, B[i][j], B[i][c]);
        }
    }
    // null vector f = H0 H1 ... H7 e8
    double f[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0};
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        if (beta[j] == 0.0) continue;
        double dot = B[j][j] * f[j];
#pragma unroll
        for (int i = j + 1; i < 9; ++i) dot = fma(B[i][j], f[i], dot);
        const double w = beta[j] * dot;
        f[j] = fma(-w, B[j][j], f[j]);
#pragma unroll
        for (int i = j + 1; i < 9; ++i) f[i] = fma(-w, B[i][j], f[i]);
    }
    // rank 2: one-sided Jacobi on the columns of G, six fixed sweeps, then drop the smallest column
    double G[3][3], V[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) G[i][j] = f[3 * i + j];
    for (int sweep = 0; sweep < 6; ++sweep) {
        jacobi_pair<0, 1>(G, V);
        jacobi_pair<0, 2>(G, V);
        jacobi_pair<1, 2>(G, V);
    }
    double cn[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        double a = G[0][p] * G[0][p]; a = fma(G[1][p], G[1][p], a); a = fma(G[2][p], G[2][p], a);
        cn[p] = a;
    }
    int m = 0;
    double cm = cn[0];
    if (cn[1] < cm) { m = 1; cm = cn[1]; }
    if (cn[2] < cm) { m = 2; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
            if (p == m) G[i][p] = 0.0;
    double Fn[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double a = G[i][0] * V[j][0]; a = fma(G[i][1], V[j][1], a); a = fma(G[i][2], V[j][2], a);
            Fn[i][j] = a;
        }
    // denormalise: F = T2^T Fn T1
    double M[3][3], Fo[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M[i][0] = Fn[i][0] * s1;
        M[i][1] = Fn[i][1] * s1;
        M[i][2] = fma(Fn[i][0], t1x, fma(Fn[i][1], t1y, Fn[i][2]));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        Fo[j] = s2 * M[0][j];
        Fo[3 + j] = s2 * M[1][j];
        Fo[6 + j] = fma(t2x, M[0][j], fma(t2y, M[1][j], M[2][j]));
    }
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) ss = fma(Fo[i], Fo[i], ss);
    const double nrm = sqrt(ss);
    if (!(nrm > 0.0) || !(nrm < __builtin_inf())) return false;
    double inv = 1.0 / nrm;
    if (Fo[8] < 0.0) inv = -inv;
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = Fo[i] * inv;
    return true;
}

__device__ __forceinline__ bool hyp_model(const float* __restrict__ xy1, const float* __restrict__ xy2, int n,
                                          uint64_t seed, uint64_t h, double (&F)[9])
{
    int idx[8];
    sample8(seed, h, n, idx);
    double x1[8], y1[8], x2[8], y2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float2 a = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(idx[i]));
        const float2 b = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(idx[i]));
        x1[i] = static_cast<double>(a.x); y1[i] = static_cast<double>(a.y);
        x2[i] = static_cast<double>(b.x); y2[i] = static_cast<double>(b.y);
    }
    return solve8(x1, y1, x2, y2, F);
}

// SPEC S8: fp32 inlier test of one correspondence against one model.
template <int KIND>
__device__ __forceinline__ bool inlier32(const float (&f)[9], float x, float y, float xp, float yp, float thr2)
{
    const float a = fmaf(f[0], x, fmaf(f[1], y, f[2]));
    const float b = fmaf(f[3], x, fmaf(f[4], y, f[5]));
    const float c = fmaf(f[6], x, fmaf(f[7], y, f[8]));
    const float num = fmaf(xp, a, fmaf(yp, b, c));
    const float at = fmaf(f[0], xp, fmaf(f[3], yp, f[6]));
    const float bt = fmaf(f[1], xp, fmaf(f[4], yp, f[7]));
    const float n2 = num * num;
    if (KIND == PM_ERR_SAMPSON) {
        const float den = fmaf(a, a, fmaf(b, b, fmaf(at, at, bt * bt)));
        return n2 <= thr2 * den;
    } else {
        const float d2 = fmaf(a, a, b * b);
        const float d1 = fmaf(at, at, bt * bt);
        return (n2 <= thr2 * d2) && (n2 <= thr2 * d1);
    }
}

// n = min(*d_n, n_max) when the count lives on the device (matcher -> filter -> RANSAC batches).
__device__ __forceinline__ int resolve_n(int n_max, const int* __restrict__ d_n)
{
    if (!d_n) return n_max;
    const int v = *d_n;
    return v < n_max ? (v < 0 ? 0 : v) : n_max;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x8v __attribute__((ext_vector_type(8)));
constexpr int PAIR_PAD = 8;          // NaN records behind the last pair: the scalar-operand scorer prefetches past its range

// Scratch of one shard: fp32 models for the scorer, fp64 models for a local finalisation,
// inlier counters.  The solve kernel also clears the counters and the shard key, so the run
// needs no memset nodes.
__global__ __launch_bounds__(64) void ransac_solve(const float* __restrict__ xy1, const float* __restrict__ xy2,
                                                   int n_max, const int* __restrict__ d_n, uint64_t seed,
                                                   int64_t hyp_begin, int nh, float* __restrict__ models,
                                                   double* __restrict__ models64, int* __restrict__ counts,
                                                   unsigned long long* __restrict__ key, int* __restrict__ tickets,
                                                   int n_tickets, float* __restrict__ pairs)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t == 0) *key = 0ull;
    if (t < n_tickets) tickets[t] = 0;          // arrival counters of the fused scorer (grid >= n_tickets threads)
    const int n = resolve_n(n_max, d_n);
    // pair records for the scorer: record r = (x_a, x_b, y_a, y_b, x'_a, x'_b, y'_a, y'_b) of points
    // a = 2r, b = 2r+1; an odd tail and PAIR_PAD records behind the end are NaN (never inliers)
    if (pairs) {
        const int nrec = (n + 1) / 2 + PAIR_PAD;
        for (int r = t; r < nrec; r += static_cast<int>(gridDim.x) * 64) {
            const float nanv = __builtin_nanf("");
            const int a = 2 * r, b = 2 * r + 1;
            const int ac = a < n ? a : 0, bc = b < n ? b : 0;                 // clamped: unconditional loads (n >= 1 below)
            float2 pa = {nanv, nanv}, qa = pa, pb = pa, qb = pa;
            if (n > 0) {
                pa = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(ac));
                qa = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(ac));
                pb = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(bc));
                qb = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(bc));
            }
            const bool va = a < n, vb = b < n;
            f32x4v lo = {va ? pa.x : nanv, vb ? pb.x : nanv, va ? pa.y : nanv, vb ? pb.y : nanv};
            f32x4v hi = {va ? qa.x : nanv, vb ? qb.x : nanv, va ? qa.y : nanv, vb ? qb.y : nanv};
            *reinterpret_cast<f32x4v*>(pairs + 8 * static_cast<size_t>(r)) = lo;
            *reinterpret_cast<f32x4v*>(pairs + 8 * static_cast<size_t>(r) + 4) = hi;
        }
    }
    if (t >= nh) return;
    counts[t] = 0;
    float* m = models + static_cast<size_t>(t) * MODEL_STRIDE;
    if (n < 8) { m[9] = 0.f; return; }
    double F[9];
    const bool ok = hyp_model(xy1, xy2, n, seed, static_cast<uint64_t>(hyp_begin + t), F);
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = static_cast<float>(F[i]);
    m[9] = ok ? 1.f : 0.f;
    double* m64 = models64 + static_cast<size_t>(t) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) m64[i] = F[i];
}

__device__ __forceinline__ unsigned long long hyp_key(const float* __restrict__ models, const int* __restrict__ counts,
                                                      int t, int64_t hyp_begin)
{
    const float valid = models[static_cast<size_t>(t) * MODEL_STRIDE + 9];   // both loads unconditional
    const uint32_t cnt = static_cast<uint32_t>(counts[t]);
    const uint32_t h = static_cast<uint32_t>(hyp_begin + t);
    const unsigned long long key = (static_cast<unsigned long long>(cnt) << 32) |
                                   static_cast<unsigned long long>(0xFFFFFFFFu - h);
    return valid != 0.f ? key : 0ull;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long key)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(key, o, 64);
        key = w > key ? w : key;
    }
    return key;
}

// result block in device memory
struct FinalOut {
    double F[9];
    int valid;
    int n_inliers;
    float F32[9];
    int pad;
};

__device__ __forceinline__ void publish_model(FinalOut* __restrict__ fo, double* __restrict__ F_out,
                                              int* __restrict__ n_out, const double (&F)[9], bool ok)
{
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const double v = ok ? F[i] : 0.0;
        fo->F[i] = v;
        fo->F32[i] = static_cast<float>(v);
        if (F_out) F_out[i] = v;
    }
    fo->valid = ok ? 1 : 0;
    fo->n_inliers = 0;
    if (n_out) *n_out = 0;
}

constexpr int SCORE_CHUNK = 256;      // correspondences staged per pass (4 KB of LDS)
constexpr int SCORE_MIN_CHUNK = 128;  // smaller chunks = more counter atomics and tickets: slower even though the chip is
                                      // underfilled (C3: 96 -> 29.8 us, 128 -> 24.6, 160 -> 24.3, 256 -> 25.9)


// SPEC S8 on TWO correspondences at once: every operation is the packed-f32 form of the scalar
// one (v_pk_fma_f32 / v_pk_mul_f32 are IEEE per component, so the bits equal inlier32), which
// halves the VALU issue slots of the scorer.  Returns how many of the two are inliers.
template <int KIND>
__device__ __forceinline__ int inlier32_x2(const float (&f)[9], f32x2 x, f32x2 y, f32x2 xp, f32x2 yp, float thr2)
{
#define PM_SPLAT(v) f32x2{(v), (v)}
    const f32x2 a = __builtin_elementwise_fma(PM_SPLAT(f[0]), x, __builtin_elementwise_fma(PM_SPLAT(f[1]), y, PM_SPLAT(f[2])));
    const f32x2 b = __builtin_elementwise_fma(PM_SPLAT(f[3]), x, __builtin_elementwise_fma(PM_SPLAT(f[4]), y, PM_SPLAT(f[5])));
    const f32x2 c = __builtin_elementwise_fma(PM_SPLAT(f[6]), x, __builtin_elementwise_fma(PM_SPLAT(f[7]), y, PM_SPLAT(f[8])));
    const f32x2 num = __builtin_elementwise_fma(xp, a, __builtin_elementwise_fma(yp, b, c));
    const f32x2 at = __builtin_elementwise_fma(PM_SPLAT(f[0]), xp, __builtin_elementwise_fma(PM_SPLAT(f[3]), yp, PM_SPLAT(f[6])));
    const f32x2 bt = __builtin_elementwise_fma(PM_SPLAT(f[1]), xp, __builtin_elementwise_fma(PM_SPLAT(f[4]), yp, PM_SPLAT(f[7])));
    const f32x2 n2 = num * num;
    const f32x2 t2 = PM_SPLAT(thr2);
#undef PM_SPLAT
    if (KIND == PM_ERR_SAMPSON) {
        const f32x2 den = __builtin_elementwise_fma(a, a, __builtin_elementwise_fma(b, b, __builtin_elementwise_fma(at, at, bt * bt)));
        const f32x2 rhs = t2 * den;
        return (n2[0] <= rhs[0] ? 1 : 0) + (n2[1] <= rhs[1] ? 1 : 0);
    } else {
        const f32x2 r2 = t2 * __builtin_elementwise_fma(a, a, b * b);
        const f32x2 r1 = t2 * __builtin_elementwise_fma(at, at, bt * bt);
        return ((n2[0] <= r2[0]) && (n2[0] <= r1[0]) ? 1 : 0) + ((n2[1] <= r2[1]) && (n2[1] <= r1[1]) ? 1 : 0);
    }
}

// What the fused (single-shard) scorer needs to finish the run inside the same launch.
struct FusedTail {
    const double* models64;
    int* tickets;                 // [0] = hypothesis blocks done, [1 + b] = chunks of block b done
    unsigned long long* best;
    FinalOut* fo;
    double* F_out;                // may be null
    uint8_t* mask;
    int* n_out;                   // may be null
    int64_t hyp_begin;
};

// FUSED: the workgroup that completes a hypothesis block (last of its chunks to arrive) reduces the
// block's best key; the workgroup that completes the last block picks the winner, publishes its
// stored fp64 model and writes the inlier mask — score, pick and mask in ONE launch.  Arrival
// order is whatever it is: every counter is an agent-scope atomic, each wave drains its own
// atomics (s_waitcnt vmcnt(0)) before the workgroup's ticket is drawn, and the finishing
// workgroups read counters/keys with agent-scope loads (CDNA guide, Guideline 16).
template <int KIND, bool FUSED, bool SMEM>
__global__ __launch_bounds__(256) void ransac_score(const float* __restrict__ xy1, const float* __restrict__ xy2,
                                                    const float* __restrict__ pairs, int n_max, const int* __restrict__ d_n, int min_chunk,
                                                    const float* __restrict__ models, int nh, float thr2,
                                                    int* __restrict__ counts, FusedTail ft)
{
    // pair p of the chunk: pts[2p] = (x_a, x_b, y_a, y_b), pts[2p+1] = (x'_a, x'_b, y'_a, y'_b)
    __shared__ __attribute__((aligned(16))) float pts[SCORE_CHUNK * 4];
    const int n = resolve_n(n_max, d_n);
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int tl = t < nh ? t : nh - 1;
    const int tt = threadIdx.x;
    // The correspondences are cut into chunks of chunk_len (<= SCORE_CHUNK, even, chosen here from the
    // DEVICE-side count so that the gridDim.y workgroup columns all have work), column y taking chunks
    // y, y + gridDim.y, ...: few big chunks when there are many hypothesis blocks (prologue, counter
    // update and ticket amortised), more, smaller ones when there are few.
    const int ycols = static_cast<int>(gridDim.y);
    int chunk_len = ((n + ycols - 1) / ycols + 1) & ~1;
    if (SMEM) chunk_len = chunk_len < 64 ? 64 : chunk_len;      // no LDS stage: one contiguous range per column
    else chunk_len = chunk_len < min_chunk ? min_chunk : (chunk_len > SCORE_CHUNK ? SCORE_CHUNK : chunk_len);
    // columns past the last chunk have nothing to add: they leave at once and are not counted by the
    // tickets (column 0 always stays, so that the tail runs even for an empty input)
    int active = (n + chunk_len - 1) / chunk_len;
    active = active < 1 ? 1 : (active > ycols ? ycols : active);
    if (static_cast<int>(blockIdx.y) >= active) return;
    const int stride = active * chunk_len;
    int i0 = blockIdx.y * chunk_len;
    float2 sp = {0.f, 0.f}, spp = sp;
    if (!SMEM) {
        // this thread's point of the first chunk is requested BEFORE the model, and the next chunk's
        // point while the current chunk is scored: neither load sits behind the other's wait
        if (i0 < n) {
            const int idx = i0 + tt < n ? i0 + tt : n - 1;           // clamped: the load stays unconditional
            sp = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(idx));
            spp = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(idx));
        }
    }
    float f[9];
    const float* m = models + static_cast<size_t>(tl) * MODEL_STRIDE;
#pragma unroll
    for (int i = 0; i < 9; ++i) f[i] = m[i];
    int cnt = 0;
    if (SMEM) {
        // Correspondences as SCALAR operands: the pair records written by ransac_solve are wave-uniform,
        // so they arrive through the scalar cache (s_load_dwordx8) and feed v_pk_fma_f32 as SGPR pairs —
        // no LDS staging, no broadcast ds_reads, no barriers.  Four records are in flight while four
        // are scored (SMEM returns out of order: a wait covers everything outstanding).
        const f32x8v* __restrict__ P = reinterpret_cast<const f32x8v*>(pairs);
#define PM_X2(c) inlier32_x2<KIND>(f, f32x2{c[0], c[1]}, f32x2{c[2], c[3]}, f32x2{c[4], c[5]}, f32x2{c[6], c[7]}, thr2)
        // PM_AFTER(idx, r0, r1): the loads at record idx are requested only after the set (r0, r1) has
        // ARRIVED and the previous set is fully scored (the empty asm consumes the count, passes the set
        // through and produces idx), so the wait for the set in use never covers the set in flight and
        // the scheduler cannot pull the next set's arithmetic in front of that wait
#define PM_AFTER(idx, r0, r1) asm volatile("" : "+s"(idx), "+s"(r0), "+s"(r1) : "v"(cnt))
        for (; i0 < n; i0 += stride) {
            int i1 = i0 + chunk_len;
            if (i1 > n) i1 = n;
            int p = i0 >> 1;
            const int p1 = (i1 + 1) >> 1;
            f32x8v a0 = P[p], a1 = P[p + 1], b0, b1;
            for (; p + 4 <= p1; p += 4) {
                int ob = p + 2;
                PM_AFTER(ob, a0, a1);
                b0 = P[ob]; b1 = P[ob + 1];
                __builtin_amdgcn_sched_barrier(0);          // the request stays HERE, ahead of the scoring of a
                cnt += PM_X2(a0);
                cnt += PM_X2(a1);
                int oa = p + 4;
                PM_AFTER(oa, b0, b1);
                a0 = P[oa]; a1 = P[oa + 1];
                __builtin_amdgcn_sched_barrier(0);
                cnt += PM_X2(b0);
                cnt += PM_X2(b1);
            }
            if (p < p1) cnt += PM_X2(a0);
            if (p + 1 < p1) cnt += PM_X2(a1);
            if (p + 2 < p1) { b0 = P[p + 2]; cnt += PM_X2(b0); }
        }
#undef PM_AFTER
#undef PM_X2
    } else {
        for (; i0 < n; i0 += stride) {
            int i1 = i0 + chunk_len;
            if (i1 > n) i1 = n;
            const int len = i1 - i0;
            const int len2 = (len + 1) & ~1;
            // one coalesced pass into LDS; every lane then reads the same address (broadcast), so a pair of
            // points costs two ds_read_b128 per wave.  An odd tail is padded with NaN (never an inlier).
            if (tt < len2) {
                const float nanv = __builtin_nanf("");
                const bool real = tt < len;
                float* d = pts + (tt >> 1) * 8 + (tt & 1);
                d[0] = real ? sp.x : nanv; d[2] = real ? sp.y : nanv; d[4] = real ? spp.x : nanv; d[6] = real ? spp.y : nanv;
            }
            if (i0 + stride < n) {
                const int idx = i0 + stride + tt < n ? i0 + stride + tt : n - 1;
                sp = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(idx));
                spp = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(idx));
            }
            __syncthreads();
            const int npair = len2 >> 1;
            int p = 0;
            for (; p + 2 <= npair; p += 2) {
                const f32x4v u0 = *reinterpret_cast<const f32x4v*>(pts + p * 8);
                const f32x4v v0 = *reinterpret_cast<const f32x4v*>(pts + p * 8 + 4);
                const f32x4v u1 = *reinterpret_cast<const f32x4v*>(pts + p * 8 + 8);
                const f32x4v v1 = *reinterpret_cast<const f32x4v*>(pts + p * 8 + 12);
                cnt += inlier32_x2<KIND>(f, f32x2{u0[0], u0[1]}, f32x2{u0[2], u0[3]}, f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}, thr2);
                cnt += inlier32_x2<KIND>(f, f32x2{u1[0], u1[1]}, f32x2{u1[2], u1[3]}, f32x2{v1[0], v1[1]}, f32x2{v1[2], v1[3]}, thr2);
            }
            for (; p < npair; ++p) {
                const f32x4v u0 = *reinterpret_cast<const f32x4v*>(pts + p * 8);
                const f32x4v v0 = *reinterpret_cast<const f32x4v*>(pts + p * 8 + 4);
                cnt += inlier32_x2<KIND>(f, f32x2{u0[0], u0[1]}, f32x2{u0[2], u0[3]}, f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}, thr2);
            }
            __syncthreads();                        // the stage is rewritten by the next chunk
        }
    }
    if (t < nh && cnt) atomicAdd(&counts[t], cnt);
    if (!FUSED) return;

    __shared__ int role;
    __shared__ unsigned long long wkey[4];
    __shared__ float Fsh[9];
    __shared__ int vsh;
    __shared__ int wcnt[4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's counter updates are performed
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tk = __hip_atomic_fetch_add(&ft.tickets[1 + blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        role = tk == active - 1 ? 1 : 0;
    }
    __syncthreads();
    if (role == 0) return;
    // ---- last chunk of this hypothesis block: the block's best key
    unsigned long long key = 0ull;
    if (t < nh && models[static_cast<size_t>(t) * MODEL_STRIDE + 9] != 0.f) {
        const uint32_t c = static_cast<uint32_t>(__hip_atomic_load(&counts[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        key = (static_cast<unsigned long long>(c) << 32) |
              static_cast<unsigned long long>(0xFFFFFFFFu - static_cast<uint32_t>(ft.hyp_begin + t));
    }
    key = wave_max_u64(key);
    if ((threadIdx.x & 63) == 0) wkey[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) key = wkey[w] > key ? wkey[w] : key;
        if (key) atomicMax(ft.best, key);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int g = __hip_atomic_fetch_add(&ft.tickets[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        role = g == static_cast<int>(gridDim.x) - 1 ? 2 : 0;
    }
    __syncthreads();
    if (role != 2) return;
    // ---- every block is done: pick the winner, publish its stored fp64 model, write the mask
    if (threadIdx.x == 0) {
        const unsigned long long k = __hip_atomic_load(ft.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = k != 0ull && n >= 8;
        const int tw = ok ? static_cast<int>(static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(k)) - ft.hyp_begin) : 0;
        double F[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) F[i] = ok ? ft.models64[static_cast<size_t>(tw) * 9 + i] : 0.0;
        publish_model(ft.fo, ft.F_out, nullptr, F, ok);
#pragma unroll
        for (int i = 0; i < 9; ++i) Fsh[i] = ok ? static_cast<float>(F[i]) : 0.f;
        vsh = ok ? 1 : 0;
    }
    __syncthreads();
    float fw[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) fw[i] = Fsh[i];
    int mine = 0;
    for (int i = threadIdx.x; i < n_max; i += 256) {
        bool in = false;
        if (i < n && vsh) {
            const float2 a = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(i));
            const float2 b = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(i));
            in = inlier32<KIND>(fw, a.x, a.y, b.x, b.y, thr2);
        }
        ft.mask[i] = in ? 1 : 0;
        mine += in ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        ft.fo->n_inliers = tot;
        if (ft.n_out) *ft.n_out = tot;
    }
}

// shard key only (multi-GPU: the caller all-reduces it)
__global__ __launch_bounds__(256) void ransac_select(const float* __restrict__ models, const int* __restrict__ counts,
                                                     int nh, int64_t hyp_begin, unsigned long long* __restrict__ best)
{
    __shared__ unsigned long long wbest[4];
    const int t = blockIdx.x * 256 + threadIdx.x;
    unsigned long long key = t < nh ? hyp_key(models, counts, t, hyp_begin) : 0ull;
    key = wave_max_u64(key);
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long k = wbest[0];
        for (int w = 1; w < 4; ++w) k = wbest[w] > k ? wbest[w] : k;
        if (k) atomicMax(best, k);
    }
}

// any-shard finalisation: re-derive the model of the id inside *key from (seed, id)
__global__ __launch_bounds__(64) void ransac_final_solve(const float* __restrict__ xy1, const float* __restrict__ xy2,
                                                         int n_max, const int* __restrict__ d_n, uint64_t seed,
                                                         const unsigned long long* __restrict__ key,
                                                         FinalOut* __restrict__ fo, double* __restrict__ F_out,
                                                         int* __restrict__ n_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = resolve_n(n_max, d_n);
    const unsigned long long k = n >= 8 ? *key : 0ull;
    double F[9];
    bool ok = false;
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = 0.0;
    if (k != 0ull) {
        const uint32_t h = 0xFFFFFFFFu - static_cast<uint32_t>(k);
        ok = hyp_model(xy1, xy2, n, seed, static_cast<uint64_t>(h), F);
    }
    publish_model(fo, F_out, n_out, F, ok);
}

template <int KIND>
__global__ __launch_bounds__(256) void ransac_final_mask(const float* __restrict__ xy1, const float* __restrict__ xy2,
                                                         int n_max, const int* __restrict__ d_n, float thr2,
                                                         FinalOut* __restrict__ fo, uint8_t* __restrict__ mask,
                                                         int* __restrict__ n_out)
{
    const int n = resolve_n(n_max, d_n);
    const int i = blockIdx.x * 256 + threadIdx.x;
    float f[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) f[j] = fo->F32[j];
    bool in = false;
    if (i < n && fo->valid) {
        const float2 p = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(i));
        const float2 pp = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(i));
        in = inlier32<KIND>(f, p.x, p.y, pp.x, pp.y, thr2);
    }
    if (i < n_max) mask[i] = in ? 1 : 0;
    const unsigned long long b = __ballot(in);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_out ? n_out : &fo->n_inliers, __popcll(b));
}

int check_params(const pm_ransac_params* p)
{
    PM_REQUIRE(p != nullptr, PM_E_INVALID, "params is null");
    PM_REQUIRE(p->hyp_begin >= 0 && p->hyp_end >= p->hyp_begin && p->hyp_end <= 0x100000000LL, PM_E_INVALID,
               "hypothesis ids must satisfy 0 <= begin <= end <= 2^32");
    PM_REQUIRE(p->hyp_end - p->hyp_begin <= 0x7FFFFFFFLL / MODEL_STRIDE, PM_E_INVALID, "hypothesis shard too large");
    PM_REQUIRE(p->error_kind == PM_ERR_SAMPSON || p->error_kind == PM_ERR_SYM_EPIPOLAR, PM_E_INVALID,
               "unknown error_kind");
    return PM_OK;
}

struct ShardScratch {
    float* pairs;                 // interleaved pair records for the scalar-operand scorer (may be null)
    float* models;
    double* models64;
    int* counts;
    int* tickets;
    int n_tickets;
    FinalOut* fo;
};

size_t pairs_bytes(int n_max) { return sizeof(float) * 8 * (static_cast<size_t>(n_max + 1) / 2 + PAIR_PAD); }

size_t shard_scratch_bytes(const pm_ransac_params* p, int n_max)
{
    const size_t nh = static_cast<size_t>(p->hyp_end - p->hyp_begin);
    return pm::align_up(pairs_bytes(n_max), 256) + pm::align_up(sizeof(float) * MODEL_STRIDE * nh, 256) + pm::align_up(sizeof(double) * 9 * nh, 256) +
           pm::align_up(sizeof(int) * nh, 256) + pm::align_up(sizeof(int) * (nh / 256 + 2), 256) + 1024;
}

int take_scratch(pm_ctx* ctx, const pm_ransac_params* p, int n_max, ShardScratch& sc)
{
    const size_t nh = static_cast<size_t>(p->hyp_end - p->hyp_begin);
    sc.pairs = static_cast<float*>(pm::arena_take(ctx, pairs_bytes(n_max)));
    sc.models = static_cast<float*>(pm::arena_take(ctx, sizeof(float) * MODEL_STRIDE * nh + 16));
    sc.models64 = static_cast<double*>(pm::arena_take(ctx, sizeof(double) * 9 * nh + 16));
    sc.counts = static_cast<int*>(pm::arena_take(ctx, sizeof(int) * nh + 16));
    sc.n_tickets = static_cast<int>((nh + 255) / 256) + 1;
    sc.tickets = static_cast<int*>(pm::arena_take(ctx, sizeof(int) * (nh / 256 + 2)));
    sc.fo = static_cast<FinalOut*>(pm::arena_take(ctx, sizeof(FinalOut)));
    PM_REQUIRE(sc.pairs && sc.models && sc.models64 && sc.counts && sc.tickets && sc.fo, PM_E_NOMEM, "scratch arena too small");
    return PM_OK;
}

// solve + score the shard.  On return (stream order) counts[] hold the inlier counts.
int score_shard(pm_ctx* ctx, const float* dxy1, const float* dxy2, int n, const int* d_n,
                const pm_ransac_params* p, unsigned long long* d_key, const ShardScratch& sc,
                const FusedTail* fused = nullptr)
{
    const int nh = static_cast<int>(p->hyp_end - p->hyp_begin);
    if (nh == 0) {
        PM_HIP_CHECK(hipMemsetAsync(d_key, 0, sizeof(unsigned long long), ctx->stream));
        return PM_OK;
    }
    // workgroup columns over the correspondences: about 16 workgroups per CU in total, never more
    // columns than minimum-size chunks of the largest possible count
    const int hb = (nh + 255) / 256;
    int chunks = (16 * ctx->n_cu + hb - 1) / hb;
    constexpr int min_chunk = SCORE_MIN_CHUNK;
    const int max_chunks = (n + min_chunk - 1) / min_chunk;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    if (chunks > 65535) chunks = 65535;
    // Large shards (the columns would each loop over several LDS stages) take the scalar-operand
    // scorer: C4 373 -> 320 us.  Small ones keep the LDS stage: the pair records cost a cold scalar-cache
    // round trip that a 20 us kernel does not amortise (C3 22.0 vs 23.5 us, C2 14.0 vs 17.0 us).
    const char* smem_str = getenv("PM_SCORE_SMEM");             // "0"/"1" pins a scorer (tests, A/B timing)
    const int smem_env = smem_str ? atoi(smem_str) : -1;
    const bool use_smem = smem_env >= 0 ? smem_env != 0 : static_cast<long long>(chunks) * SCORE_CHUNK < n;
    {
        pm::ScopedKernelTime t(ctx, "ransac_solve");
        hipLaunchKernelGGL(ransac_solve, dim3((nh + 63) / 64), dim3(64), 0, ctx->stream, dxy1, dxy2, n, d_n, p->seed,
                           p->hyp_begin, nh, sc.models, sc.models64, sc.counts, d_key, sc.tickets, sc.n_tickets,
                           use_smem ? sc.pairs : nullptr);
        PM_HIP_CHECK(hipGetLastError());
    }
    const float thr2 = p->thresh_px * p->thresh_px;
    {
        pm::ScopedKernelTime t(ctx, "ransac_score");
        const FusedTail none{};
        const FusedTail& ft = fused ? *fused : none;
#define PM_SCORE(KIND_, FUSED_, SMEM_)                                                                             \
    hipLaunchKernelGGL((ransac_score<KIND_, FUSED_, SMEM_>), dim3(hb, chunks), dim3(256), 0, ctx->stream, dxy1, dxy2,  \
                       sc.pairs, n, d_n, min_chunk, sc.models, nh, thr2, sc.counts, ft)
#define PM_SCORE2(KIND_, FUSED_) do { if (use_smem) PM_SCORE(KIND_, FUSED_, true); else PM_SCORE(KIND_, FUSED_, false); } while (0)
        if (p->error_kind == PM_ERR_SAMPSON) { if (fused) PM_SCORE2(PM_ERR_SAMPSON, true); else PM_SCORE2(PM_ERR_SAMPSON, false); }
        else { if (fused) PM_SCORE2(PM_ERR_SYM_EPIPOLAR, true); else PM_SCORE2(PM_ERR_SYM_EPIPOLAR, false); }
#undef PM_SCORE2
#undef PM_SCORE
        PM_HIP_CHECK(hipGetLastError());
    }
    return PM_OK;
}

int select_shard(pm_ctx* ctx, const pm_ransac_params* p, unsigned long long* d_key, const ShardScratch& sc)
{
    const int nh = static_cast<int>(p->hyp_end - p->hyp_begin);
    if (nh == 0) return PM_OK;
    pm::ScopedKernelTime t(ctx, "ransac_select");
    hipLaunchKernelGGL(ransac_select, dim3((nh + 255) / 256), dim3(256), 0, ctx->stream, sc.models, sc.counts, nh,
                       p->hyp_begin, d_key);
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

int launch_mask(pm_ctx* ctx, const float* dxy1, const float* dxy2, int n, const int* d_n, const pm_ransac_params* p,
                FinalOut* d_fo, uint8_t* d_mask, int* d_ninl)
{
    const float thr2 = p->thresh_px * p->thresh_px;
    if (p->error_kind == PM_ERR_SAMPSON)
        hipLaunchKernelGGL(ransac_final_mask<PM_ERR_SAMPSON>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dxy1,
                           dxy2, n, d_n, thr2, d_fo, d_mask, d_ninl);
    else
        hipLaunchKernelGGL(ransac_final_mask<PM_ERR_SYM_EPIPOLAR>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                           dxy1, dxy2, n, d_n, thr2, d_fo, d_mask, d_ninl);
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

// finalise from a key that may name ANY id (re-solves it)
int finalize_from_key(pm_ctx* ctx, const float* dxy1, const float* dxy2, int n, const int* d_n,
                      const pm_ransac_params* p, const unsigned long long* d_key, FinalOut* d_fo, double* d_F,
                      uint8_t* d_mask, int* d_ninl)
{
    pm::ScopedKernelTime t(ctx, "ransac_final");
    hipLaunchKernelGGL(ransac_final_solve, dim3(1), dim3(64), 0, ctx->stream, dxy1, dxy2, n, d_n, p->seed, d_key,
                       d_fo, d_F, d_ninl);
    return launch_mask(ctx, dxy1, dxy2, n, d_n, p, d_fo, d_mask, d_ninl);
}

// single-shard run: solve, then score + pick + mask in one launch (no second solve, no extra launches)
int run_local(pm_ctx* ctx, const float* dxy1, const float* dxy2, int n, const int* d_n, const pm_ransac_params* p,
              unsigned long long* d_key, const ShardScratch& sc, double* d_F, uint8_t* d_mask, int* d_ninl)
{
    FusedTail ft{};
    ft.models64 = sc.models64;
    ft.tickets = sc.tickets;
    ft.best = d_key;
    ft.fo = sc.fo;
    ft.F_out = d_F;
    ft.mask = d_mask;
    ft.n_out = d_ninl;
    ft.hyp_begin = p->hyp_begin;
    return score_shard(ctx, dxy1, dxy2, n, d_n, p, d_key, sc, &ft);
}

// Shared host-pointer driver: run the shard (hyp < 0) or take the given hypothesis, then finalise.
int host_run(pm_ctx* ctx, const float* xy1, const float* xy2, int n, const pm_ransac_params* p, int64_t hyp,
             double F[9], uint8_t* mask, int* n_inliers, uint64_t* best_key)
{
    if (F) for (int i = 0; i < 9; ++i) F[i] = 0.0;
    if (mask && n > 0) memset(mask, 0, static_cast<size_t>(n));
    if (n_inliers) *n_inliers = 0;
    if (best_key) *best_key = 0;
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    int rc = check_params(p);
    if (rc != PM_OK) return rc;
    PM_REQUIRE(n >= 0 && (n == 0 || (xy1 && xy2)), PM_E_INVALID, "bad point arrays");
    if (n < 8) { pm::set_error("need at least 8 correspondences, got %d", n); return PM_E_TOO_FEW; }
    PM_REQUIRE(hyp < 0x100000000LL, PM_E_INVALID, "hypothesis id must be < 2^32");
    PM_HIP_CHECK(hipSetDevice(ctx->device));

    const size_t xyb = sizeof(float) * 2 * static_cast<size_t>(n);
    const size_t need = 2 * pm::align_up(xyb, 256) + pm::align_up(static_cast<size_t>(n), 256) + 512 +
                        shard_scratch_bytes(p, n) + 2048;
    rc = pm::arena_reserve(ctx, need);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    float* dxy1 = static_cast<float*>(pm::arena_take(ctx, xyb));
    float* dxy2 = static_cast<float*>(pm::arena_take(ctx, xyb));
    uint8_t* dmask = static_cast<uint8_t*>(pm::arena_take(ctx, static_cast<size_t>(n)));
    unsigned long long* dkey = static_cast<unsigned long long*>(pm::arena_take(ctx, 8));
    PM_REQUIRE(dxy1 && dxy2 && dmask && dkey, PM_E_NOMEM, "scratch arena too small");
    ShardScratch sc;
    rc = take_scratch(ctx, p, n, sc);
    if (rc != PM_OK) return rc;
    rc = pm::pinned_reserve(ctx, sizeof(FinalOut) + 8 + static_cast<size_t>(n));
    if (rc != PM_OK) return rc;

    PM_HIP_CHECK(hipMemcpyAsync(dxy1, xy1, xyb, hipMemcpyHostToDevice, ctx->stream));
    PM_HIP_CHECK(hipMemcpyAsync(dxy2, xy2, xyb, hipMemcpyHostToDevice, ctx->stream));
    if (hyp < 0) {
        if (p->hyp_end > p->hyp_begin) {
            rc = run_local(ctx, dxy1, dxy2, n, nullptr, p, dkey, sc, nullptr, dmask, nullptr);
        } else {
            rc = score_shard(ctx, dxy1, dxy2, n, nullptr, p, dkey, sc);
            if (rc == PM_OK) rc = finalize_from_key(ctx, dxy1, dxy2, n, nullptr, p, dkey, sc.fo, nullptr, dmask, nullptr);
        }
    } else {
        const unsigned long long k = pm_ransac_key(0u, static_cast<uint32_t>(hyp)) | (1ull << 32);  // non-zero
        unsigned long long* hk = static_cast<unsigned long long*>(ctx->pinned);
        *hk = k;
        PM_HIP_CHECK(hipMemcpyAsync(dkey, hk, 8, hipMemcpyHostToDevice, ctx->stream));
        rc = finalize_from_key(ctx, dxy1, dxy2, n, nullptr, p, dkey, sc.fo, nullptr, dmask, nullptr);
    }
    if (rc != PM_OK) return rc;
    char* hp = static_cast<char*>(ctx->pinned);
    FinalOut* hfo = reinterpret_cast<FinalOut*>(hp + 8);
    uint8_t* hmask = reinterpret_cast<uint8_t*>(hp + 8 + sizeof(FinalOut));
    unsigned long long* hkey = reinterpret_cast<unsigned long long*>(hp);
    PM_HIP_CHECK(hipMemcpyAsync(hkey, dkey, 8, hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP_CHECK(hipMemcpyAsync(hfo, sc.fo, sizeof(FinalOut), hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP_CHECK(hipMemcpyAsync(hmask, dmask, static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (best_key && hyp < 0) *best_key = *hkey;
    if (!hfo->valid) {
        pm::set_error("no valid model (all hypotheses degenerate)");
        return PM_E_NO_MODEL;
    }
    if (F) memcpy(F, hfo->F, sizeof(double) * 9);
    if (mask) memcpy(mask, hmask, static_cast<size_t>(n));
    if (n_inliers) *n_inliers = hfo->n_inliers;
    return PM_OK;
}

int dev_prologue(pm_ctx* ctx, const pm_ransac_params* p, int n_max, const float* d_xy1, const float* d_xy2,
                 ShardScratch& sc)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    int rc = check_params(p);
    if (rc != PM_OK) return rc;
    PM_REQUIRE(n_max >= 1 && d_xy1 && d_xy2, PM_E_INVALID, "bad point arrays");
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    rc = pm::arena_reserve(ctx, shard_scratch_bytes(p, n_max) + 2048);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    return take_scratch(ctx, p, n_max, sc);
}

}  // namespace

extern "C" int pm_ransac_fundamental(pm_ctx* ctx, const float* xy1, const float* xy2, int n,
                                     const pm_ransac_params* p, double F[9], uint8_t* mask, int* n_inliers,
                                     uint64_t* best_key)
{
    return host_run(ctx, xy1, xy2, n, p, -1, F, mask, n_inliers, best_key);
}

extern "C" int pm_ransac_model_from_hyp(pm_ctx* ctx, const float* xy1, const float* xy2, int n,
                                        const pm_ransac_params* p, int64_t hyp, double F[9], uint8_t* mask,
                                        int* n_inliers)
{
    if (hyp < 0) { pm::set_error("hyp must be >= 0"); return PM_E_INVALID; }
    pm_ransac_params q;
    if (p) { q = *p; q.hyp_begin = 0; q.hyp_end = 0; }
    return host_run(ctx, xy1, xy2, n, p ? &q : nullptr, hyp, F, mask, n_inliers, nullptr);
}

extern "C" int pm_ransac_score_dev(pm_ctx* ctx, const float* d_xy1, const float* d_xy2, int n,
                                   const pm_ransac_params* p, uint64_t* d_best_key)
{
    PM_REQUIRE(d_best_key != nullptr, PM_E_INVALID, "null argument");
    if (n >= 0 && n < 8) { pm::set_error("need at least 8 correspondences, got %d", n); return PM_E_TOO_FEW; }
    ShardScratch sc;
    int rc = dev_prologue(ctx, p, n, d_xy1, d_xy2, sc);
    if (rc != PM_OK) return rc;
    unsigned long long* key = reinterpret_cast<unsigned long long*>(d_best_key);
    rc = score_shard(ctx, d_xy1, d_xy2, n, nullptr, p, key, sc);
    if (rc != PM_OK) return rc;
    return select_shard(ctx, p, key, sc);
}

extern "C" int pm_ransac_score_devn(pm_ctx* ctx, const float* d_xy1, const float* d_xy2, int n_max,
                                    const int32_t* d_n, const pm_ransac_params* p, uint64_t* d_best_key)
{
    PM_REQUIRE(d_best_key != nullptr && d_n != nullptr, PM_E_INVALID, "null argument");
    ShardScratch sc;
    int rc = dev_prologue(ctx, p, n_max, d_xy1, d_xy2, sc);
    if (rc != PM_OK) return rc;
    unsigned long long* key = reinterpret_cast<unsigned long long*>(d_best_key);
    rc = score_shard(ctx, d_xy1, d_xy2, n_max, d_n, p, key, sc);
    if (rc != PM_OK) return rc;
    return select_shard(ctx, p, key, sc);
}

extern "C" int pm_ransac_run_dev(pm_ctx* ctx, const float* d_xy1, const float* d_xy2, int n_max, const int32_t* d_n,
                                 const pm_ransac_params* p, uint64_t* d_best_key, double* d_F, uint8_t* d_mask,
                                 int32_t* d_n_inliers)
{
    PM_REQUIRE(d_best_key && d_F && d_mask && d_n_inliers, PM_E_INVALID, "null argument");
    ShardScratch sc;
    int rc = dev_prologue(ctx, p, n_max, d_xy1, d_xy2, sc);
    if (rc != PM_OK) return rc;
    PM_REQUIRE(p->hyp_end > p->hyp_begin, PM_E_INVALID, "empty hypothesis range");
    unsigned long long* key = reinterpret_cast<unsigned long long*>(d_best_key);
    return run_local(ctx, d_xy1, d_xy2, n_max, d_n, p, key, sc, d_F, d_mask, d_n_inliers);
}

extern "C" int pm_ransac_model_from_key_dev(pm_ctx* ctx, const float* d_xy1, const float* d_xy2, int n_max,
                                            const int32_t* d_n, const pm_ransac_params* p, const uint64_t* d_key,
                                            double* d_F, uint8_t* d_mask, int32_t* d_n_inliers)
{
    PM_REQUIRE(ctx && d_xy1 && d_xy2 && d_key && d_F && d_mask && d_n_inliers, PM_E_INVALID, "null argument");
    int rc = check_params(p);
    if (rc != PM_OK) return rc;
    PM_REQUIRE(n_max >= 1, PM_E_INVALID, "n_max must be positive");
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    rc = pm::arena_reserve(ctx, sizeof(FinalOut) + 1024);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    FinalOut* dfo = static_cast<FinalOut*>(pm::arena_take(ctx, sizeof(FinalOut)));
    PM_REQUIRE(dfo != nullptr, PM_E_NOMEM, "scratch arena too small");
    return finalize_from_key(ctx, d_xy1, d_xy2, n_max, d_n, p, reinterpret_cast<const unsigned long long*>(d_key), dfo,
                             d_F, d_mask, d_n_inliers);
}
