// ransac_core.hpp — device functions shared by the RANSAC-F translation units (ransac.hip: hypothesis-per-lane
// kernels for large correspondence sets and the sharded legacy entry points; ransac_fused.hip: the one-launch
// solve + score + pick + mask kernel).  Every operation is written in the order docs/SPEC.md fixes (S6 sampler,
// S7 normalised 8-point solve, S8 inlier test); both TUs are built with -ffp-contract=off, so the only fused
// multiply-adds are the explicit fma()/fmaf() calls and the CPU restatement reproduces the same bits.
// Replaces the arithmetic behind cv::findFundamentalMat (main.cpp:95-98) with the estimator BASELINE.json names.
#pragma once
#include "pm_common.hpp"

namespace pm_ransac {

// Diagnostics policy of the RANSAC kernels.  The product instantiates everything with NoDiag (empty inline hooks,
// no code); tools/ablation/ransac_fused_stamps.hip supplies a policy that writes shader-clock stamps.
struct NoDiag {
    static constexpr bool scalar_fma = false;                       // score with unpacked v_fma_f32 (experiment)
    static __device__ __forceinline__ void phase(int) {}            // kernel phase boundary i
    static __device__ __forceinline__ void solve(int) {}            // solver phase boundary i
};

constexpr int MODEL_STRIDE = 12;   // 9 x fp32 F, valid flag, 2 pad

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// SPEC S6: 8 distinct indices in [0, n) as a pure function of (seed, h, n): draws d = 0, 1, ... give candidates
// c_d = floor(n * hi32(mix64(stream + (d+1)*phi)) / 2^32); a candidate equal to an earlier one is skipped; after 64
// draws the smallest unused integers complete the sample.  Same results as the sequential loop of the oracle, computed
// so that the 64-bit multiplies of the first eight draws are independent (they dominated: 5.3k of the solver's 24k
// cycles with the draw-by-draw loop) and the common case — eight distinct candidates — touches no select chains.
__device__ __forceinline__ void sample8(uint64_t seed, uint64_t h, int n, int (&idx)[8])
{
    const uint64_t stream = mix64(seed ^ 0x9E3779B97F4A7C15ULL) ^ mix64(h + 0xD1B54A32D192ED03ULL);
    int cand[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const uint64_t r = mix64(stream + static_cast<uint64_t>(d + 1) * 0x9E3779B97F4A7C15ULL);
        cand[d] = static_cast<int>(((r >> 32) * static_cast<uint64_t>(static_cast<uint32_t>(n))) >> 32);
    }
    unsigned dup = 0u;                         // bit d: candidate d repeats an earlier one (so it is skipped)
#pragma unroll
    for (int d = 1; d < 8; ++d)
#pragma unroll
        for (int e = 0; e < d; ++e) dup |= (cand[e] == cand[d] ? 1u : 0u) << d;
#pragma unroll
    for (int s = 0; s < 8; ++s) idx[s] = cand[s];
    if (dup == 0u) return;
    // rare: keep the first occurrences in draw order, then go on drawing
#pragma unroll
    for (int s = 0; s < 8; ++s) idx[s] = -1;
    int cnt = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        if (!((dup >> d) & 1u)) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s == cnt) idx[s] = cand[d];
            ++cnt;
        }
    }
    for (uint64_t d = 8; d < 64 && cnt < 8; ++d) {
        const uint64_t r = mix64(stream + (d + 1) * 0x9E3779B97F4A7C15ULL);
        const int c = static_cast<int>(((r >> 32) * static_cast<uint64_t>(static_cast<uint32_t>(n))) >> 32);
        bool rep = false;
#pragma unroll
        for (int s = 0; s < 8; ++s) rep |= (s < cnt) && (idx[s] == c);
        if (!rep) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s == cnt) idx[s] = c;
            ++cnt;
        }
    }
    for (int c = 0; cnt < 8; ++c) {
        bool rep = false;
#pragma unroll
        for (int s = 0; s < 8; ++s) rep |= (s < cnt) && (idx[s] == c);
        if (!rep) {
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
template <typename DIAG = NoDiag>
__device__ __forceinline__ bool solve8(const double (&x1)[8], const double (&y1)[8], const double (&x2)[8],
                                    const double (&y2)[8], double (&F)[9])
{
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = 0.0;
    double ax[8], ay[8], bx[8], by[8], s1, t1x, t1y, s2, t2x, t2y;
    if (!hartley8(x1, y1, ax, ay, s1, t1x, t1y)) return false;
    if (!hartley8(x2, y2, bx, by, s2, t2x, t2y)) return false;

    DIAG::solve(1);
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
    DIAG::solve(2);
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
    DIAG::solve(3);
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
    DIAG::solve(4);
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

// SPEC S8 on TWO correspondences at once: every operation is the packed-f32 form of the scalar
// one (v_pk_fma_f32 / v_pk_mul_f32 are IEEE per component, so the bits equal inlier32), which
// halves the VALU issue slots of the scorer.  ia / ib: the two verdicts.
template <int KIND>
__device__ __forceinline__ void inlier32_x2_flags(const float (&f)[9], f32x2 x, f32x2 y, f32x2 xp, f32x2 yp, float thr2,
                                                  bool& ia, bool& ib)
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
        ia = n2[0] <= rhs[0];
        ib = n2[1] <= rhs[1];
    } else {
        const f32x2 r2 = t2 * __builtin_elementwise_fma(a, a, b * b);
        const f32x2 r1 = t2 * __builtin_elementwise_fma(at, at, bt * bt);
        ia = (n2[0] <= r2[0]) && (n2[0] <= r1[0]);
        ib = (n2[1] <= r2[1]) && (n2[1] <= r1[1]);
    }
}
// Same, as the number of inliers among the two (the hypothesis-per-lane scorers add it to a lane counter).
template <int KIND>
__device__ __forceinline__ int inlier32_x2(const float (&f)[9], f32x2 x, f32x2 y, f32x2 xp, f32x2 yp, float thr2)
{
    bool ia, ib;
    inlier32_x2_flags<KIND>(f, x, y, xp, yp, thr2, ia, ib);
    return (ia ? 1 : 0) + (ib ? 1 : 0);
}

}  // namespace pm_ransac
