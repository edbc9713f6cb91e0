// lmeds.hip — 7-point minimal solver + LMedS on gfx950.
//
// This is the estimator the reference's call LITERALLY selects:
// `cv::findFundamentalMat(Mat(selPoints1), Mat(selPoints2), CV_FM_7POINT)` (main.cpp:95-98) with
// more than 7 correspondences runs OpenCV 2.4's least-median-of-squares loop over 7-point minimal
// solves [recalled; OpenCV is not in the image].  The arithmetic is frozen by docs/SPEC.md S13-S15
// and restated by oracle/pm_oracle.c; the kernels below reproduce it bit for bit (explicit fma,
// unfused elsewhere, -ffp-contract=off, no transcendental functions: the cubic is solved by
// bisection + deflation so that only IEEE-exact operations decide the result).
//
//   lmeds_solve    one lane per hypothesis: sampler (S13) + normalised 7-point solve (S14):
//                  null space by Householder QR, cubic det(x*G1 + y*G2) = 0, up to three models
//   lmeds_median   one workgroup per model: fp64 symmetric-epipolar residuals of all n
//                  correspondences (as float keys in LDS), exact median by 4-pass radix select
//   lmeds_final    one workgroup: smallest (median, model id), robust sigma, inlier mask (S15)
#include <cfloat>
#include <cmath>

#include "pm_common.hpp"

namespace {

constexpr int LM_MAX_N = 32768;          // residual keys of one model live in LDS (128 KB)
constexpr double LM_INF = __builtin_inf();

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// SPEC S13: 7 distinct indices in [0, n) as a pure function of (seed, h, n).
__device__ __forceinline__ void sample7(uint64_t seed, uint64_t h, int n, int (&idx)[7])
{
    const uint64_t stream = mix64(seed ^ 0x7F4A7C159E3779B9ULL) ^ mix64(h + 0xD1B54A32D192ED03ULL);
#pragma unroll
    for (int s = 0; s < 7; ++s) idx[s] = -1;
    int cnt = 0;
    for (uint64_t d = 0; d < 64 && cnt < 7; ++d) {
        const uint64_t r = mix64(stream + (d + 1) * 0x9E3779B97F4A7C15ULL);
        const int c = static_cast<int>(((r >> 32) * static_cast<uint64_t>(static_cast<uint32_t>(n))) >> 32);
        bool dup = false;
#pragma unroll
        for (int s = 0; s < 7; ++s) dup |= (s < cnt) && (idx[s] == c);
        if (!dup) {
#pragma unroll
            for (int s = 0; s < 7; ++s)
                if (s == cnt) idx[s] = c;
            ++cnt;
        }
    }
    for (int c = 0; cnt < 7; ++c) {
        bool dup = false;
#pragma unroll
        for (int s = 0; s < 7; ++s) dup |= (s < cnt) && (idx[s] == c);
        if (!dup) {
#pragma unroll
            for (int s = 0; s < 7; ++s)
                if (s == cnt) idx[s] = c;
            ++cnt;
        }
    }
}

__device__ __forceinline__ bool hartley7(const double (&px)[7], const double (&py)[7], double (&nx)[7],
                                         double (&ny)[7], double& s, double& tx, double& ty)
{
    double cx = px[0], cy = py[0];
#pragma unroll
    for (int i = 1; i < 7; ++i) { cx = cx + px[i]; cy = cy + py[i]; }
    cx = cx / 7.0; cy = cy / 7.0;
    double md = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const double dx = px[i] - cx, dy = py[i] - cy;
        md = md + sqrt(fma(dx, dx, dy * dy));
    }
    md = md / 7.0;
    if (!(md > 0.0) || !(md < LM_INF)) return false;
    s = 1.4142135623730951 / md;
#pragma unroll
    for (int i = 0; i < 7; ++i) { nx[i] = (px[i] - cx) * s; ny[i] = (py[i] - cy) * s; }
    tx = -(s * cx); ty = -(s * cy);
    return true;
}

__device__ __forceinline__ double det3(double r00, double r01, double r02, double r10, double r11, double r12,
                                       double r20, double r21, double r22)
{
    const double m0 = r11 * r22 - r12 * r21;
    const double m1 = r10 * r22 - r12 * r20;
    const double m2 = r10 * r21 - r11 * r20;
    double d = r00 * m0;
    d = d - r01 * m1;
    d = d + r02 * m2;
    return d;
}

// SPEC S14.  F: up to three models (row-major, unit Frobenius norm, F[8] >= 0); valid[r].
__device__ __forceinline__ void solve7(const double (&x1)[7], const double (&y1)[7], const double (&x2)[7],
                                       const double (&y2)[7], double (&F)[3][9], bool (&valid)[3])
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        valid[r] = false;
#pragma unroll
        for (int i = 0; i < 9; ++i) F[r][i] = 0.0;
    }
    double ax[7], ay[7], bx[7], by[7], s1, t1x, t1y, s2, t2x, t2y;
    if (!hartley7(x1, y1, ax, ay, s1, t1x, t1y)) return;
    if (!hartley7(x2, y2, bx, by, s2, t2x, t2y)) return;
    double B[9][7], beta[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        B[0][c] = bx[c] * ax[c]; B[1][c] = bx[c] * ay[c]; B[2][c] = bx[c];
        B[3][c] = by[c] * ax[c]; B[4][c] = by[c] * ay[c]; B[5][c] = by[c];
        B[6][c] = ax[c];         B[7][c] = ay[c];         B[8][c] = 1.0;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
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
        for (int c = j + 1; c < 7; ++c) {
            double dot = v0 * B[j][c];
#pragma unroll
            for (int i = j + 1; i < 9; ++i) dot = fma(B[i][j], B[i][c], dot);
            const double w = beta[j] * dot;
            B[j][c] = fma(-w, v0, B[j][c]);
#pragma unroll
            for (int i = j + 1; i < 9; ++i) B[i][c] = fma(-w, B[i][j], B[i][c]);
        }
    }
    // null space: g[0] = H0..H6 e7, g[1] = H0..H6 e8
    double g[2][9];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
        for (int i = 0; i < 9; ++i) g[v][i] = (i == 7 + v) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 6; j >= 0; --j) {
            if (beta[j] == 0.0) continue;
            double dot = B[j][j] * g[v][j];
#pragma unroll
            for (int i = j + 1; i < 9; ++i) dot = fma(B[i][j], g[v][i], dot);
            const double w = beta[j] * dot;
            g[v][j] = fma(-w, B[j][j], g[v][j]);
#pragma unroll
            for (int i = j + 1; i < 9; ++i) g[v][i] = fma(-w, B[i][j], g[v][i]);
        }
    }
#define PM_A(i) g[0][i]
#define PM_B(i) g[1][i]
#define PM_DET(R0, R1, R2) det3(R0(0), R0(1), R0(2), R1(3), R1(4), R1(5), R2(6), R2(7), R2(8))
    const double c3 = PM_DET(PM_A, PM_A, PM_A), c0 = PM_DET(PM_B, PM_B, PM_B);
    double c2 = PM_DET(PM_B, PM_A, PM_A); c2 = c2 + PM_DET(PM_A, PM_B, PM_A); c2 = c2 + PM_DET(PM_A, PM_A, PM_B);
    double c1 = PM_DET(PM_A, PM_B, PM_B); c1 = c1 + PM_DET(PM_B, PM_A, PM_B); c1 = c1 + PM_DET(PM_B, PM_B, PM_A);
#undef PM_DET
#undef PM_B
#undef PM_A
    const bool swap = !(fabs(c3) >= fabs(c0));
    const double k3 = swap ? c0 : c3, k2 = swap ? c1 : c2, k1 = swap ? c2 : c1, k0 = swap ? c3 : c0;
    if (!(fabs(k3) > 0.0)) return;
    const double a = k2 / k3, b = k1 / k3, c = k0 / k3;
    double R = fabs(a);
    if (fabs(b) > R) R = fabs(b);
    if (fabs(c) > R) R = fabs(c);
    R = 1.0 + R;
    if (!(R < LM_INF)) return;
    double lo = -R, hi = R;
    for (int it = 0; it < 100; ++it) {
        const double mid = 0.5 * (lo + hi);
        double v = mid + a; v = fma(v, mid, b); v = fma(v, mid, c);
        if (v < 0.0) lo = mid; else hi = mid;
    }
    double roots[3];
    int nr = 1;
    roots[0] = 0.5 * (lo + hi);
    roots[1] = 0.0; roots[2] = 0.0;
    const double q1 = a + roots[0], q0 = fma(q1, roots[0], b);
    const double disc = fma(q1, q1, -(4.0 * q0));
    if (disc >= 0.0) {
        const double sq = sqrt(disc);
        const double t = -0.5 * (q1 + (q1 >= 0.0 ? sq : -sq));
        roots[1] = t;
        roots[2] = (t != 0.0) ? q0 / t : 0.0;
        nr = 3;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (r >= nr) continue;
        const double z = roots[r];
        double Fn[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double P = swap ? g[1][3 * i + j] : g[0][3 * i + j];
                const double Q = swap ? g[0][3 * i + j] : g[1][3 * i + j];
                Fn[i][j] = fma(z, P, Q);
            }
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
        if (!(nrm > 0.0) || !(nrm < LM_INF)) continue;
        double inv = 1.0 / nrm;
        if (Fo[8] < 0.0) inv = -inv;
#pragma unroll
        for (int i = 0; i < 9; ++i) F[r][i] = Fo[i] * inv;
        valid[r] = true;
    }
}

__global__ __launch_bounds__(64) void lmeds_solve(const float* __restrict__ xy1, const float* __restrict__ xy2, int n,
                                                  uint64_t seed, int64_t hyp_begin, int nh,
                                                  double* __restrict__ models, int* __restrict__ mvalid)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= nh) return;
    int idx[7];
    sample7(seed, static_cast<uint64_t>(hyp_begin + t), n, idx);
    double x1[7], y1[7], x2[7], y2[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float2 a = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(idx[i]));
        const float2 b = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(idx[i]));
        x1[i] = static_cast<double>(a.x); y1[i] = static_cast<double>(a.y);
        x2[i] = static_cast<double>(b.x); y2[i] = static_cast<double>(b.y);
    }
    double F[3][9];
    bool valid[3];
    solve7(x1, y1, x2, y2, F, valid);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double* m = models + (static_cast<size_t>(t) * 3 + r) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) m[i] = F[r][i];
        mvalid[t * 3 + r] = valid[r] ? 1 : 0;
    }
}

// SPEC S15 residual: max(d1^2/|l1|^2, d2^2/|l2|^2) in fp64, rounded to float; NaN -> +inf.
__device__ __forceinline__ float lmeds_err(const double (&F)[9], float2 p, float2 pp)
{
    const double x = static_cast<double>(p.x), y = static_cast<double>(p.y);
    const double xp = static_cast<double>(pp.x), yp = static_cast<double>(pp.y);
    const double a = fma(F[0], x, fma(F[1], y, F[2]));
    const double b = fma(F[3], x, fma(F[4], y, F[5]));
    const double c = fma(F[6], x, fma(F[7], y, F[8]));
    const double d2 = fma(xp, a, fma(yp, b, c));
    const double s2 = 1.0 / fma(a, a, b * b);
    const double at = fma(F[0], xp, fma(F[3], yp, F[6]));
    const double bt = fma(F[1], xp, fma(F[4], yp, F[7]));
    const double ct = fma(F[2], xp, fma(F[5], yp, F[8]));
    const double d1 = fma(x, at, fma(y, bt, ct));
    const double s1 = 1.0 / fma(at, at, bt * bt);
    const double e1 = (d1 * d1) * s1, e2 = (d2 * d2) * s2;
    const float e = static_cast<float>(e1 >= e2 ? e1 : e2);
    return (e != e) ? __builtin_inff() : e;
}

// rank-th smallest (0-based) of the n non-negative float keys in LDS: MSB-first radix select, 8 bits
// per pass.  hist[256], wsum[4] and ctl[2] are LDS scratch.  All 256 threads of the workgroup call
// it; thread b owns histogram bin b (block-wide exclusive scan: wave shuffles + 4 wave totals).
__device__ __forceinline__ uint32_t radix_select(const uint32_t* __restrict__ keys, int n, int rank, uint32_t* hist,
                                                 uint32_t* wsum, uint32_t* ctl)
{
    uint32_t prefix = 0u, mask = 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int pass = 3; pass >= 0; --pass) {
        const int sh = 8 * pass;
        hist[threadIdx.x] = 0u;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            const uint32_t k = keys[i];
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> sh) & 255u], 1u);
        }
        __syncthreads();
        const uint32_t h = hist[threadIdx.x];
        uint32_t inc = h;                              // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t base = 0u;
#pragma unroll
        for (int w = 0; w < 4; ++w) base += (w < wave) ? wsum[w] : 0u;
        const uint32_t before = base + inc - h;        // keys in lower bins
        if (static_cast<uint32_t>(rank) >= before && static_cast<uint32_t>(rank) < before + h) {   // exactly one bin
            ctl[0] = threadIdx.x;
            ctl[1] = before;
        }
        __syncthreads();
        prefix |= ctl[0] << sh;
        mask |= 255u << sh;
        rank -= static_cast<int>(ctl[1]);
        __syncthreads();
    }
    return prefix;
}

__global__ __launch_bounds__(256) void lmeds_median(const float* __restrict__ xy1, const float* __restrict__ xy2, int n,
                                                    const double* __restrict__ models, const int* __restrict__ mvalid,
                                                    double* __restrict__ meds)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lm_lds[];     // [n] keys, [256] hist, [4] wsum, [2] ctl
    uint32_t* keys = lm_lds;
    uint32_t* hist = lm_lds + n;
    uint32_t* wsum = hist + 256;
    uint32_t* ctl = wsum + 4;
    const int model = blockIdx.x;
    if (!mvalid[model]) {                              // uniform
        if (threadIdx.x == 0) meds[model] = LM_INF;
        return;
    }
    double F[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = models[static_cast<size_t>(model) * 9 + i];
    for (int i = threadIdx.x; i < n; i += 256) {
        const float2 p = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(i));
        const float2 pp = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(i));
        keys[i] = __float_as_uint(lmeds_err(F, p, pp));        // >= +0: the bit pattern orders like the value
    }
    __syncthreads();
    const uint32_t hi = radix_select(keys, n, n / 2, hist, wsum, ctl);
    double med = static_cast<double>(__uint_as_float(hi));
    if ((n & 1) == 0) {
        // v[n/2 - 1] in one more pass: with L = #keys strictly below v[n/2] (L <= n/2), it is the largest
        // of those keys when L == n/2, and v[n/2] itself (a duplicate) when L < n/2
        if (threadIdx.x < 2) ctl[threadIdx.x] = 0u;
        __syncthreads();
        uint32_t less = 0u, mx = 0u;
        for (int i = threadIdx.x; i < n; i += 256) {
            const uint32_t k = keys[i];
            if (k < hi) { ++less; mx = k > mx ? k : mx; }
        }
        if (less) { atomicAdd(&ctl[0], less); atomicMax(&ctl[1], mx); }
        __syncthreads();
        const uint32_t lo = ctl[0] >= static_cast<uint32_t>(n / 2) ? ctl[1] : hi;
        med = (static_cast<double>(__uint_as_float(lo)) + med) * 0.5;
    }
    if (threadIdx.x == 0) meds[model] = med;
}

struct LmedsOut {
    double F[9];
    double median;
    long long best_model;
    int n_inliers;
    int found;
};

__global__ __launch_bounds__(256) void lmeds_final(const float* __restrict__ xy1, const float* __restrict__ xy2, int n,
                                                   const double* __restrict__ models, const double* __restrict__ meds,
                                                   int n_models, int64_t hyp_begin, LmedsOut* __restrict__ out,
                                                   double* __restrict__ F_out, uint8_t* __restrict__ mask,
                                                   int* __restrict__ n_out, long long* __restrict__ best_out,
                                                   double* __restrict__ med_out)
{
    __shared__ double s_med[256];
    __shared__ int s_id[256];
    __shared__ int wcnt[4];
    // smallest (median, model id); a median of +inf (invalid model / all-NaN residuals) never wins
    double bm = LM_INF;
    int bi = -1;
    for (int m = threadIdx.x; m < n_models; m += 256) {
        const double v = meds[m];
        if (v < bm) { bm = v; bi = m; }               // ascending m: the first minimum is the lowest id
    }
    s_med[threadIdx.x] = bm;
    s_id[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (static_cast<int>(threadIdx.x) < o) {
            const double v = s_med[threadIdx.x + o];
            const int id = s_id[threadIdx.x + o];
            const double u = s_med[threadIdx.x];
            const int ui = s_id[threadIdx.x];
            if (id >= 0 && (ui < 0 || v < u || (v == u && id < ui))) { s_med[threadIdx.x] = v; s_id[threadIdx.x] = id; }
        }
        __syncthreads();
    }
    const int win = s_id[0];
    const double med = s_med[0];
    const bool found = win >= 0 && med < LM_INF;
    double F[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = found ? models[static_cast<size_t>(win) * 9 + i] : 0.0;
    double sigma = ((2.5 * 1.4826) * (1.0 + 5.0 / static_cast<double>(n - 7))) * sqrt(found ? med : 0.0);
    sigma = sigma > 0.001 ? sigma : 0.001;
    const double thr = sigma * sigma;
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        bool in = false;
        if (found) {
            const float2 p = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(i));
            const float2 pp = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(i));
            in = static_cast<double>(lmeds_err(F, p, pp)) <= thr;
        }
        if (mask) mask[i] = in ? 1 : 0;
        mine += in ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        const long long id = found ? 3ll * hyp_begin + win : -1ll;
#pragma unroll
        for (int i = 0; i < 9; ++i) { out->F[i] = F[i]; if (F_out) F_out[i] = F[i]; }
        out->median = found ? med : LM_INF;
        out->best_model = id;
        out->n_inliers = tot;
        out->found = found ? 1 : 0;
        if (n_out) *n_out = tot;
        if (best_out) *best_out = id;
        if (med_out) *med_out = found ? med : LM_INF;
    }
}

// ---- adaptive-iteration RANSAC over the same 7-point models (SPEC S16) --------------------------
// one workgroup per model: number of correspondences with residual <= thr (fp64 compare of the
// float residual, as OpenCV's findInliers)
__global__ __launch_bounds__(256) void fm_count(const float* __restrict__ xy1, const float* __restrict__ xy2, int n,
                                                const double* __restrict__ models, const int* __restrict__ mvalid,
                                                double thr, int* __restrict__ counts)
{
    __shared__ int wcnt[4];
    const int model = blockIdx.x;
    if (!mvalid[model]) {                              // uniform
        if (threadIdx.x == 0) counts[model] = -1;
        return;
    }
    double F[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = models[static_cast<size_t>(model) * 9 + i];
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float2 p = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(i));
        const float2 pp = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(i));
        mine += static_cast<double>(lmeds_err(F, p, pp)) <= thr ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) counts[model] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// F and mask of one given model (slot `win` of the current batch), one workgroup
__global__ __launch_bounds__(256) void fm_mask(const float* __restrict__ xy1, const float* __restrict__ xy2, int n,
                                               const double* __restrict__ models, int win, double thr,
                                               LmedsOut* __restrict__ out, uint8_t* __restrict__ mask)
{
    __shared__ int wcnt[4];
    double F[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i] = models[static_cast<size_t>(win) * 9 + i];
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float2 p = *reinterpret_cast<const float2*>(xy1 + 2 * static_cast<size_t>(i));
        const float2 pp = *reinterpret_cast<const float2*>(xy2 + 2 * static_cast<size_t>(i));
        const bool in = static_cast<double>(lmeds_err(F, p, pp)) <= thr;
        mask[i] = in ? 1 : 0;
        mine += in ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) out->F[i] = F[i];
        out->n_inliers = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        out->found = 1;
    }
}

// OpenCV cvRANSACUpdateNumIters [recalled]; host libm, as in the oracle
int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0.0 ? p : 0.0; p = p < 1.0 ? p : 1.0;
    ep = ep > 0.0 ? ep : 0.0; ep = ep < 1.0 ? ep : 1.0;
    double num = 1.0 - p > DBL_MIN ? 1.0 - p : DBL_MIN;
    double denom = 1.0 - pow(1.0 - ep, static_cast<double>(model_points));
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return (denom >= 0.0 || -num >= static_cast<double>(max_iters) * (-denom)) ? max_iters
                                                                               : static_cast<int>(lround(num / denom));
}

int check_lmeds(const pm_lmeds_params* p, int n)
{
    PM_REQUIRE(p != nullptr, PM_E_INVALID, "params is null");
    PM_REQUIRE(p->hyp_begin >= 0 && p->hyp_end >= p->hyp_begin && p->hyp_end <= 0x40000000LL &&
                   p->hyp_end - p->hyp_begin <= 0x2000000LL,
               PM_E_INVALID, "need 0 <= hyp_begin <= hyp_end <= 2^30 and at most 2^25 hypotheses per call");
    PM_REQUIRE(n <= LM_MAX_N, PM_E_UNSUPPORTED, "LMedS keeps one model's residuals in LDS: at most 32768 correspondences");
    return PM_OK;
}

// device-resident core; d_out is a LmedsOut in the arena
int run_lmeds(pm_ctx* ctx, const float* dxy1, const float* dxy2, int n, const pm_lmeds_params* p, double* d_F,
              uint8_t* d_mask, int* d_ninl, long long* d_best, double* d_med, LmedsOut** d_out_ret)
{
    const int nh = static_cast<int>(p->hyp_end - p->hyp_begin);
    const int nm = 3 * nh;
    const size_t need = pm::align_up(sizeof(double) * 9 * static_cast<size_t>(nm > 0 ? nm : 1), 256) +
                        pm::align_up(sizeof(int) * static_cast<size_t>(nm > 0 ? nm : 1), 256) +
                        pm::align_up(sizeof(double) * static_cast<size_t>(nm > 0 ? nm : 1), 256) + 1024;
    int rc = pm::arena_reserve(ctx, need);
    if (rc != PM_OK) return rc;
    double* models = static_cast<double*>(pm::arena_take(ctx, sizeof(double) * 9 * static_cast<size_t>(nm > 0 ? nm : 1)));
    int* mvalid = static_cast<int*>(pm::arena_take(ctx, sizeof(int) * static_cast<size_t>(nm > 0 ? nm : 1)));
    double* meds = static_cast<double*>(pm::arena_take(ctx, sizeof(double) * static_cast<size_t>(nm > 0 ? nm : 1)));
    LmedsOut* d_out = static_cast<LmedsOut*>(pm::arena_take(ctx, sizeof(LmedsOut)));
    PM_REQUIRE(models && mvalid && meds && d_out, PM_E_NOMEM, "scratch arena too small");
    if (d_out_ret) *d_out_ret = d_out;
    if (nh > 0) {
        {
            pm::ScopedKernelTime t(ctx, "lmeds_solve");
            hipLaunchKernelGGL(lmeds_solve, dim3((nh + 63) / 64), dim3(64), 0, ctx->stream, dxy1, dxy2, n, p->seed,
                               p->hyp_begin, nh, models, mvalid);
            PM_HIP_CHECK(hipGetLastError());
        }
        const size_t lds = sizeof(uint32_t) * (static_cast<size_t>(n) + 256 + 4 + 2);
        static size_t lds_set_dev[PM_MAX_DEVICES] = {};        // per device: largest size granted so far
        size_t& lds_set = lds_set_dev[ctx->device];
        if (lds > lds_set) {
            PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lmeds_median),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            lds_set = lds;
        }
        pm::ScopedKernelTime t(ctx, "lmeds_median");
        hipLaunchKernelGGL(lmeds_median, dim3(nm), dim3(256), lds, ctx->stream, dxy1, dxy2, n, models, mvalid, meds);
        PM_HIP_CHECK(hipGetLastError());
    }
    {
        pm::ScopedKernelTime t(ctx, "lmeds_final");
        hipLaunchKernelGGL(lmeds_final, dim3(1), dim3(256), 0, ctx->stream, dxy1, dxy2, n, models, meds, nm, p->hyp_begin,
                           d_out, d_F, d_mask, d_ninl, d_best, d_med);
        PM_HIP_CHECK(hipGetLastError());
    }
    return PM_OK;
}

}  // namespace

extern "C" int pm_lmeds_fundamental_dev(pm_ctx* ctx, const float* d_xy1, const float* d_xy2, int n,
                                        const pm_lmeds_params* p, double* d_F, uint8_t* d_mask, int32_t* d_n_inliers,
                                        int64_t* d_best_model, double* d_median)
{
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_REQUIRE(n >= 8 && d_xy1 && d_xy2, PM_E_INVALID, "need n >= 8 device correspondences");
    int rc = check_lmeds(p, n);
    if (rc != PM_OK) return rc;
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    pm::arena_reset(ctx);
    return run_lmeds(ctx, d_xy1, d_xy2, n, p, d_F, d_mask, d_n_inliers, reinterpret_cast<long long*>(d_best_model), d_median,
                     nullptr);
}

extern "C" int pm_lmeds_fundamental(pm_ctx* ctx, const float* xy1, const float* xy2, int n, const pm_lmeds_params* p,
                                    double F[9], uint8_t* mask, int* n_inliers, int64_t* best_model, double* median)
{
    if (F) for (int i = 0; i < 9; ++i) F[i] = 0.0;
    if (mask && n > 0) memset(mask, 0, static_cast<size_t>(n));
    if (n_inliers) *n_inliers = 0;
    if (best_model) *best_model = -1;
    if (median) *median = LM_INF;
    PM_REQUIRE(ctx != nullptr, PM_E_INVALID, "ctx is null");
    PM_REQUIRE(n >= 0 && (n == 0 || (xy1 && xy2)), PM_E_INVALID, "bad point arrays");
    if (n < 8) { pm::set_error("need at least 8 correspondences, got %d", n); return PM_E_TOO_FEW; }
    int rc = check_lmeds(p, n);
    if (rc != PM_OK) return rc;
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t xyb = sizeof(float) * 2 * static_cast<size_t>(n);
    const int nm = 3 * static_cast<int>(p->hyp_end - p->hyp_begin);
    const size_t need = 2 * pm::align_up(xyb, 256) + pm::align_up(static_cast<size_t>(n), 256) +
                        pm::align_up((sizeof(double) * 10 + sizeof(int)) * static_cast<size_t>(nm > 0 ? nm : 1), 256) + 4096;
    rc = pm::arena_reserve(ctx, need);
    if (rc != PM_OK) return rc;
    rc = pm::pinned_reserve(ctx, sizeof(LmedsOut));
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    float* dxy1 = static_cast<float*>(pm::arena_take(ctx, xyb));
    float* dxy2 = static_cast<float*>(pm::arena_take(ctx, xyb));
    uint8_t* dmask = static_cast<uint8_t*>(pm::arena_take(ctx, static_cast<size_t>(n)));
    PM_REQUIRE(dxy1 && dxy2 && dmask, PM_E_NOMEM, "scratch arena too small");
    PM_HIP_CHECK(hipMemcpyAsync(dxy1, xy1, xyb, hipMemcpyHostToDevice, ctx->stream));
    PM_HIP_CHECK(hipMemcpyAsync(dxy2, xy2, xyb, hipMemcpyHostToDevice, ctx->stream));
    LmedsOut* d_out = nullptr;
    rc = run_lmeds(ctx, dxy1, dxy2, n, p, nullptr, dmask, nullptr, nullptr, nullptr, &d_out);
    if (rc != PM_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    LmedsOut* h = static_cast<LmedsOut*>(ctx->pinned);
    PM_HIP_CHECK(hipMemcpyAsync(h, d_out, sizeof(LmedsOut), hipMemcpyDeviceToHost, ctx->stream));
    if (mask) PM_HIP_CHECK(hipMemcpyAsync(mask, dmask, static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (!h->found) {
        if (mask) memset(mask, 0, static_cast<size_t>(n));
        pm::set_error("no valid 7-point model in the hypothesis range");
        return PM_E_NO_MODEL;
    }
    if (F) memcpy(F, h->F, sizeof(h->F));
    if (n_inliers) *n_inliers = h->n_inliers;
    if (best_model) *best_model = h->best_model;
    if (median) *median = h->median;
    return PM_OK;
}

// SPEC S16.  The hypotheses are solved and counted on the device in batches of ADAPT_BATCH ids; the
// host replays OpenCV's sequential loop over the counts (best model so far, shrinking iteration
// budget) and stops requesting batches once the budget is reached.
extern "C" int pm_ransac7_adaptive(pm_ctx* ctx, const float* xy1, const float* xy2, int n, const pm_adaptive_params* p,
                                   double F[9], uint8_t* mask, int* n_inliers, int64_t* best_model, int* iters_run)
{
    constexpr int ADAPT_BATCH = 512;
    if (F) for (int i = 0; i < 9; ++i) F[i] = 0.0;
    if (mask && n > 0) memset(mask, 0, static_cast<size_t>(n));
    if (n_inliers) *n_inliers = 0;
    if (best_model) *best_model = -1;
    if (iters_run) *iters_run = 0;
    PM_REQUIRE(ctx != nullptr && p != nullptr, PM_E_INVALID, "null ctx / params");
    PM_REQUIRE(n >= 0 && (n == 0 || (xy1 && xy2)), PM_E_INVALID, "bad point arrays");
    PM_REQUIRE(p->max_iters >= 1 && p->max_iters <= (1 << 20) && p->thresh_px > 0.f, PM_E_INVALID,
               "need 1 <= max_iters <= 2^20 and thresh_px > 0");
    if (n < 8) { pm::set_error("need at least 8 correspondences, got %d", n); return PM_E_TOO_FEW; }
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t xyb = sizeof(float) * 2 * static_cast<size_t>(n);
    const int nm = 3 * ADAPT_BATCH;
    const size_t need = 2 * pm::align_up(xyb, 256) + pm::align_up(static_cast<size_t>(n), 256) +
                        3 * pm::align_up(sizeof(double) * 9 * nm, 256) + 2 * pm::align_up(sizeof(int) * nm, 256) + 4096;
    int rc = pm::arena_reserve(ctx, need);
    if (rc != PM_OK) return rc;
    rc = pm::pinned_reserve(ctx, sizeof(LmedsOut) + sizeof(int) * nm);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    float* dxy1 = static_cast<float*>(pm::arena_take(ctx, xyb));
    float* dxy2 = static_cast<float*>(pm::arena_take(ctx, xyb));
    uint8_t* dmask = static_cast<uint8_t*>(pm::arena_take(ctx, static_cast<size_t>(n)));
    double* models = static_cast<double*>(pm::arena_take(ctx, sizeof(double) * 9 * nm));
    double* best_models = static_cast<double*>(pm::arena_take(ctx, sizeof(double) * 9));
    int* mvalid = static_cast<int*>(pm::arena_take(ctx, sizeof(int) * nm));
    int* counts = static_cast<int*>(pm::arena_take(ctx, sizeof(int) * nm));
    LmedsOut* d_out = static_cast<LmedsOut*>(pm::arena_take(ctx, sizeof(LmedsOut)));
    PM_REQUIRE(dxy1 && dxy2 && dmask && models && best_models && mvalid && counts && d_out, PM_E_NOMEM,
               "scratch arena too small");
    LmedsOut* h_out = static_cast<LmedsOut*>(ctx->pinned);
    int* h_counts = reinterpret_cast<int*>(h_out + 1);
    PM_HIP_CHECK(hipMemcpyAsync(dxy1, xy1, xyb, hipMemcpyHostToDevice, ctx->stream));
    PM_HIP_CHECK(hipMemcpyAsync(dxy2, xy2, xyb, hipMemcpyHostToDevice, ctx->stream));
    const double thr = static_cast<double>(p->thresh_px) * static_cast<double>(p->thresh_px);
    int niters = static_cast<int>(p->max_iters), best = 6, h = 0;
    long long best_id = -1;
    while (h < niters) {
        const int h0 = h;
        int nb = niters - h0;
        if (nb > ADAPT_BATCH) nb = ADAPT_BATCH;
        {
            pm::ScopedKernelTime t(ctx, "lmeds_solve");
            hipLaunchKernelGGL(lmeds_solve, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, dxy1, dxy2, n, p->seed,
                               static_cast<int64_t>(h0), nb, models, mvalid);
            PM_HIP_CHECK(hipGetLastError());
        }
        {
            pm::ScopedKernelTime t(ctx, "fm_count");
            hipLaunchKernelGGL(fm_count, dim3(3 * nb), dim3(256), 0, ctx->stream, dxy1, dxy2, n, models, mvalid, thr, counts);
            PM_HIP_CHECK(hipGetLastError());
        }
        PM_HIP_CHECK(hipMemcpyAsync(h_counts, counts, sizeof(int) * 3 * nb, hipMemcpyDeviceToHost, ctx->stream));
        PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        int batch_best = -1;
        for (; h < niters && h < h0 + nb; ++h)
            for (int r = 0; r < 3; ++r) {
                const int c = h_counts[3 * (h - h0) + r];
                if (c > best) {
                    best = c;
                    best_id = 3ll * h + r;
                    batch_best = 3 * (h - h0) + r;
                    niters = update_num_iters(p->confidence, static_cast<double>(n - c) / static_cast<double>(n), 7, niters);
                }
            }
        if (batch_best >= 0)      // keep the running winner's model: the batch buffer is overwritten by the next batch
            PM_HIP_CHECK(hipMemcpyAsync(best_models, models + static_cast<size_t>(batch_best) * 9, sizeof(double) * 9,
                                        hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (iters_run) *iters_run = h;
    if (best_id < 0) {
        (void)hipStreamSynchronize(ctx->stream);
        pm::set_error("no 7-point model with more than 6 inliers");
        return PM_E_NO_MODEL;
    }
    hipLaunchKernelGGL(fm_mask, dim3(1), dim3(256), 0, ctx->stream, dxy1, dxy2, n, best_models, 0, thr, d_out, dmask);
    PM_HIP_CHECK(hipGetLastError());
    PM_HIP_CHECK(hipMemcpyAsync(h_out, d_out, sizeof(LmedsOut), hipMemcpyDeviceToHost, ctx->stream));
    if (mask) PM_HIP_CHECK(hipMemcpyAsync(mask, dmask, static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
    PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (F) memcpy(F, h_out->F, sizeof(h_out->F));
    if (n_inliers) *n_inliers = h_out->n_inliers;
    if (best_model) *best_model = best_id;
    return PM_OK;
}
