/* pm_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the hot path of wenxiaoshuai/Points-Matching
 * (`Points Matching/main.cpp:42-123`, cited as main.cpp:N) in the arithmetic frozen by
 * docs/SPEC.md.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; nothing under points_matching_amd/ includes, links or calls it.
 *
 * PARITY STATUS: "parity unpinned" for everything that lives inside OpenCV 2.4.13 in the
 * reference (descriptor matcher main.cpp:44-46, findFundamentalMat main.cpp:95-98): OpenCV is
 * an un-vendored prebuilt dependency that is absent from /root/reference and from this image,
 * the reference has no tests / golden vectors / recorded output, and its binary cannot run
 * here.  The parts of the path whose arithmetic IS in the reference tree are
 * pinned exactly and tested as such: the midpoint filter (main.cpp:49-69), the match-list
 * format (main.cpp:73-76), the index/gather order (main.cpp:77-78, :89-91) and the residual
 * report (main.cpp:103-123).
 *
 * Build: see oracle/Makefile  (gcc -O3 -ffp-contract=off -march=x86-64-v3; every fused
 * multiply-add in the spec is an explicit fma()/fmaf() call, everything else is unfused).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct pmo_match {      /* == pm_match == cv::DMatch (main.cpp:45) */
    int32_t queryIdx, trainIdx, imgIdx;
    float distance;
} pmo_match;

typedef struct pmo_ransac_params {  /* == pm_ransac_params */
    int64_t hyp_begin, hyp_end;
    uint64_t seed;
    float thresh_px;
    int32_t error_kind;
} pmo_ransac_params;

static uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ------------------------------------------------------------------------------------------
 * SPEC S1 — canonical squared L2 distance.  Follows the (recalled, unverified) SSE2 path of
 * OpenCV 2.4.13 normL2Sqr_ that BruteForceMatcher<L2<float>> (main.cpp:43) runs: eight lane
 * accumulators over 8-wide strides (d0 = lanes 0..3, d1 = lanes 4..7), separate multiply and
 * add, d0+d1 lane-wise, then ((s0+s1)+s2)+s3, then a scalar tail.
 * ------------------------------------------------------------------------------------------ */
float pmo_l2sqr(const float* a, const float* b, int dim)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int j = 0;
    for (; j + 8 <= dim; j += 8)
        for (int l = 0; l < 8; ++l) {
            float t = a[j + l] - b[j + l];
            float p = t * t;
            acc[l] = acc[l] + p;
        }
    float s0 = acc[0] + acc[4], s1 = acc[1] + acc[5], s2 = acc[2] + acc[6], s3 = acc[3] + acc[7];
    float d = ((s0 + s1) + s2) + s3;
    for (; j < dim; ++j) {
        float t = a[j] - b[j];
        float p = t * t;
        d = d + p;
    }
    return d;
}

/* SPEC S3 — ordering key: (distance bits, trainIdx); NaN distances are canonicalised to the
 * quiet-NaN pattern so they sort after +inf. */
static uint64_t knn_key(float dist, int32_t idx)
{
    uint32_t b = (dist != dist) ? 0x7FC00000u : f32_bits(dist);
    return ((uint64_t)b << 32) | (uint32_t)idx;
}

static void knn_insert(uint64_t* keys, float* dists, int k, uint64_t key, float dist)
{
    if (key >= keys[k - 1]) return;
    int p = k - 1;
    while (p > 0 && keys[p - 1] > key) { keys[p] = keys[p - 1]; dists[p] = dists[p - 1]; --p; }
    keys[p] = key; dists[p] = dist;
}

static void knn_emit(const uint64_t* keys, const float* dists, int k, int qi, pmo_match* out)
{
    for (int c = 0; c < k; ++c) {
        out[c].queryIdx = qi;
        out[c].imgIdx = 0;
        if (keys[c] == UINT64_MAX) { out[c].trainIdx = -1; out[c].distance = INFINITY; }
        else { out[c].trainIdx = (int32_t)(uint32_t)keys[c]; out[c].distance = dists[c]; }
    }
}

/* main.cpp:46 with the BF-L2 matcher of main.cpp:43, generalised to k-NN. */
int pmo_bf_knn_l2_f32(const float* q, int nq, const float* t, int nt, int dim, int k,
                      pmo_match* out, int nthreads)
{
    if (nq < 0 || nt < 0 || dim < 1 || k < 1 || k > 16) return -1;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int i = 0; i < nq; ++i) {
        uint64_t keys[16]; float dists[16];
        for (int c = 0; c < k; ++c) { keys[c] = UINT64_MAX; dists[c] = INFINITY; }
        const float* a = q + (size_t)i * dim;
        for (int j = 0; j < nt; ++j) {
            float d2 = pmo_l2sqr(a, t + (size_t)j * dim, dim);
            float d = sqrtf(d2);
            knn_insert(keys, dists, k, knn_key(d, j), d);
        }
        knn_emit(keys, dists, k, i, out + (size_t)i * k);
    }
    (void)nthreads;
    return 0;
}

/* SPEC S2 — Hamming distance of `bytes`-byte binary descriptors, reported as float. */
int pmo_bf_knn_hamming_u8(const uint8_t* q, int nq, const uint8_t* t, int nt, int bytes, int k,
                          pmo_match* out, int nthreads)
{
    if (nq < 0 || nt < 0 || bytes < 4 || (bytes & 3) || k < 1 || k > 16) return -1;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int i = 0; i < nq; ++i) {
        uint64_t keys[16]; float dists[16];
        for (int c = 0; c < k; ++c) { keys[c] = UINT64_MAX; dists[c] = INFINITY; }
        const uint8_t* a = q + (size_t)i * bytes;
        for (int j = 0; j < nt; ++j) {
            const uint8_t* b = t + (size_t)j * bytes;
            int h = 0;
            for (int w = 0; w < bytes; w += 4) {
                uint32_t x, y; memcpy(&x, a + w, 4); memcpy(&y, b + w, 4);
                h += __builtin_popcount(x ^ y);
            }
            float d = (float)h;
            knn_insert(keys, dists, k, knn_key(d, j), d);
        }
        knn_emit(keys, dists, k, i, out + (size_t)i * k);
    }
    (void)nthreads;
    return 0;
}

/* main.cpp:49-69, literally (incl. the initial values 1 and 0 and the promotion to double). */
int pmo_filter_midpoint(const pmo_match* m, int n, double* min_out, double* max_out,
                        pmo_match* out, int* n_out)
{
    double minMatch = 1, maxMatch = 0;
    for (int i = 0; i < n; ++i) {
        minMatch = minMatch > m[i].distance ? m[i].distance : minMatch;
        maxMatch = maxMatch < m[i].distance ? m[i].distance : maxMatch;
    }
    int c = 0;
    for (int i = 0; i < n; ++i)
        if (m[i].distance < minMatch + (maxMatch - minMatch) / 2) out[c++] = m[i];
    if (min_out) *min_out = minMatch;
    if (max_out) *max_out = maxMatch;
    *n_out = c;
    return 0;
}

/* SPEC S4 — ratio test in the slot of main.cpp:49-69. */
int pmo_filter_ratio(const pmo_match* knn, int nq, int k, float ratio, pmo_match* out, int* n_out)
{
    if (k < 2) return -1;
    int c = 0;
    for (int i = 0; i < nq; ++i) {
        const pmo_match* r = knn + (size_t)i * k;
        if (r[0].trainIdx < 0 || r[1].trainIdx < 0) continue;
        float rhs = ratio * r[1].distance;
        if (r[0].distance < rhs) out[c++] = r[0];
    }
    *n_out = c;
    return 0;
}

/* main.cpp:73-79 / :89-91 */
int pmo_gather_points(const float* kp_xy, int n_kp, const int32_t* idx, int n, float* out_xy)
{
    for (int i = 0; i < n; ++i) {
        if (idx[i] < 0 || idx[i] >= n_kp) return -1;
        out_xy[2 * i] = kp_xy[2 * idx[i]];
        out_xy[2 * i + 1] = kp_xy[2 * idx[i] + 1];
    }
    return 0;
}

long pmo_format_match_list(const pmo_match* m, int n, char* buf, size_t cap)
{
    long need = 0;
    char line[128];
    int w = snprintf(line, sizeof line, "Good Matches are:\n");
    if (buf && (size_t)(need + w) < cap) memcpy(buf + need, line, (size_t)w);
    need += w;
    for (int i = 0; i < n; ++i) {
        w = snprintf(line, sizeof line, "-- Good Match [%d] Keypoint 1: %d  -- Keypoint 2: %d  \n",
                     i, m[i].queryIdx, m[i].trainIdx);   /* main.cpp:76 */
        if (buf && (size_t)(need + w) < cap) memcpy(buf + need, line, (size_t)w);
        need += w;
    }
    if (buf && cap) buf[(size_t)need < cap ? (size_t)need : cap - 1] = 0;
    return need;
}

/* ------------------------------------------------------------------------------------------
 * SPEC S6 — counter-based sampler.
 * ------------------------------------------------------------------------------------------ */
static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void pmo_sample8(uint64_t seed, uint64_t h, int n, int32_t idx[8])
{
    uint64_t stream = mix64(seed ^ 0x9E3779B97F4A7C15ULL) ^ mix64(h + 0xD1B54A32D192ED03ULL);
    int cnt = 0;
    for (uint64_t d = 0; d < 64 && cnt < 8; ++d) {
        uint64_t r = mix64(stream + (d + 1) * 0x9E3779B97F4A7C15ULL);
        int32_t c = (int32_t)(((r >> 32) * (uint64_t)(uint32_t)n) >> 32);
        int dup = 0;
        for (int s = 0; s < cnt; ++s) dup |= (idx[s] == c);
        if (!dup) idx[cnt++] = c;
    }
    for (int32_t c = 0; cnt < 8; ++c) {     /* deterministic completion; n >= 8 guarantees it ends */
        int dup = 0;
        for (int s = 0; s < cnt; ++s) dup |= (idx[s] == c);
        if (!dup) idx[cnt++] = c;
    }
}

/* ------------------------------------------------------------------------------------------
 * SPEC S7 — Hartley-normalised 8-point solve with rank-2 enforcement, all fp64.
 * p1/p2: the 8 sampled points (x,y) of image 1 / image 2.  Returns 1 if valid.
 * ------------------------------------------------------------------------------------------ */
#define PMO_SQRT2 1.4142135623730951
#define PMO_JACOBI_SWEEPS 6

static int hartley(const double* p, double* pn, double* s_out, double* tx_out, double* ty_out)
{
    double cx = p[0], cy = p[1];
    for (int i = 1; i < 8; ++i) { cx = cx + p[2 * i]; cy = cy + p[2 * i + 1]; }
    cx = cx * 0.125; cy = cy * 0.125;
    double md = 0.0;
    for (int i = 0; i < 8; ++i) {
        double dx = p[2 * i] - cx, dy = p[2 * i + 1] - cy;
        md = md + sqrt(fma(dx, dx, dy * dy));
    }
    md = md * 0.125;
    if (!(md > 0.0) || !(md < INFINITY)) return 0;
    double s = PMO_SQRT2 / md;
    for (int i = 0; i < 8; ++i) {
        pn[2 * i] = (p[2 * i] - cx) * s;
        pn[2 * i + 1] = (p[2 * i + 1] - cy) * s;
    }
    *s_out = s; *tx_out = -(s * cx); *ty_out = -(s * cy);
    return 1;
}

int pmo_solve8(const double* p1, const double* p2, double F[9])
{
    double n1[16], n2[16], s1, t1x, t1y, s2, t2x, t2y;
    for (int i = 0; i < 9; ++i) F[i] = 0.0;
    if (!hartley(p1, n1, &s1, &t1x, &t1y)) return 0;
    if (!hartley(p2, n2, &s2, &t2x, &t2y)) return 0;

    /* B = A^T (9 x 8): column c is the constraint row of correspondence c. */
    double B[9][8], beta[8];
    for (int c = 0; c < 8; ++c) {
        double x1 = n1[2 * c], y1 = n1[2 * c + 1], x2 = n2[2 * c], y2 = n2[2 * c + 1];
        B[0][c] = x2 * x1; B[1][c] = x2 * y1; B[2][c] = x2;
        B[3][c] = y2 * x1; B[4][c] = y2 * y1; B[5][c] = y2;
        B[6][c] = x1;      B[7][c] = y1;      B[8][c] = 1.0;
    }
    /* Householder QR of B; reflector j is kept in column j (v0 on the diagonal). */
    for (int j = 0; j < 8; ++j) {
        double sigma = 0.0;
        for (int i = j + 1; i < 9; ++i) sigma = fma(B[i][j], B[i][j], sigma);
        double alpha = B[j][j];
        double nrm = sqrt(fma(alpha, alpha, sigma));
        if (!(nrm > 0.0)) { beta[j] = 0.0; continue; }
        double v0 = alpha + (alpha >= 0.0 ? nrm : -nrm);
        double vtv = fma(v0, v0, sigma);
        beta[j] = 2.0 / vtv;
        B[j][j] = v0;
        for (int c = j + 1; c < 8; ++c) {
            double dot = v0 * B[j][c];
            for (int i = j + 1; i < 9; ++i) dot = fma(B[i][j], B[i][c], dot);
            double w = beta[j] * dot;
            B[j][c] = fma(-w, v0, B[j][c]);
            for (int i = j + 1; i < 9; ++i) B[i][c] = fma(-w, B[i][j], B[i][c]);
        }
    }
    /* null vector f = H0 H1 ... H7 e8 */
    double f[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1.0};
    for (int j = 7; j >= 0; --j) {
        if (beta[j] == 0.0) continue;
        double dot = B[j][j] * f[j];
        for (int i = j + 1; i < 9; ++i) dot = fma(B[i][j], f[i], dot);
        double w = beta[j] * dot;
        f[j] = fma(-w, B[j][j], f[j]);
        for (int i = j + 1; i < 9; ++i) f[i] = fma(-w, B[i][j], f[i]);
    }
    /* rank-2 enforcement: one-sided (Hestenes) Jacobi on the columns of G, fixed sweeps. */
    double G[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) G[i][j] = f[3 * i + j];
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < PMO_JACOBI_SWEEPS; ++sweep)
        for (int r = 0; r < 3; ++r) {
            int p = PQ[r][0], q = PQ[r][1];
            double al = G[0][p] * G[0][p]; al = fma(G[1][p], G[1][p], al); al = fma(G[2][p], G[2][p], al);
            double be = G[0][q] * G[0][q]; be = fma(G[1][q], G[1][q], be); be = fma(G[2][q], G[2][q], be);
            double ga = G[0][p] * G[0][q]; ga = fma(G[1][p], G[1][q], ga); ga = fma(G[2][p], G[2][q], ga);
            if (!(ga * ga > 4.930380657631324e-32 * (al * be))) continue;   /* converged (or NaN) pair */
            double zeta = (be - al) / (2.0 * ga);
            double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
            double c = 1.0 / sqrt(fma(t, t, 1.0));
            double s = c * t;
            for (int i = 0; i < 3; ++i) {
                double gp = G[i][p], gq = G[i][q];
                G[i][p] = fma(c, gp, -(s * gq));
                G[i][q] = fma(s, gp, c * gq);
                double vp = V[i][p], vq = V[i][q];
                V[i][p] = fma(c, vp, -(s * vq));
                V[i][q] = fma(s, vp, c * vq);
            }
        }
    double cn[3];
    for (int p = 0; p < 3; ++p) {
        double a = G[0][p] * G[0][p]; a = fma(G[1][p], G[1][p], a); a = fma(G[2][p], G[2][p], a);
        cn[p] = a;
    }
    int m = 0;
    if (cn[1] < cn[m]) m = 1;
    if (cn[2] < cn[m]) m = 2;
    for (int i = 0; i < 3; ++i) G[i][m] = 0.0;
    double Fn[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = G[i][0] * V[j][0]; a = fma(G[i][1], V[j][1], a); a = fma(G[i][2], V[j][2], a);
            Fn[i][j] = a;
        }
    /* denormalise: F = T2^T * Fn * T1 */
    double M[3][3], Fo[9];
    for (int i = 0; i < 3; ++i) {
        M[i][0] = Fn[i][0] * s1;
        M[i][1] = Fn[i][1] * s1;
        M[i][2] = fma(Fn[i][0], t1x, fma(Fn[i][1], t1y, Fn[i][2]));
    }
    for (int j = 0; j < 3; ++j) {
        Fo[j] = s2 * M[0][j];
        Fo[3 + j] = s2 * M[1][j];
        Fo[6 + j] = fma(t2x, M[0][j], fma(t2y, M[1][j], M[2][j]));
    }
    double ss = 0.0;
    for (int i = 0; i < 9; ++i) ss = fma(Fo[i], Fo[i], ss);
    double nrm = sqrt(ss);
    if (!(nrm > 0.0) || !(nrm < INFINITY)) return 0;
    double inv = 1.0 / nrm;
    if (Fo[8] < 0.0) inv = -inv;
    for (int i = 0; i < 9; ++i) F[i] = Fo[i] * inv;
    return 1;
}

/* hypothesis h -> F (fp64) and its fp32 rounding used for scoring. */
int pmo_hyp_model(const float* xy1, const float* xy2, int n, uint64_t seed, uint64_t h,
                  double F[9], float F32[9])
{
    int32_t idx[8];
    double p1[16], p2[16];
    pmo_sample8(seed, h, n, idx);
    for (int i = 0; i < 8; ++i) {
        p1[2 * i] = (double)xy1[2 * idx[i]]; p1[2 * i + 1] = (double)xy1[2 * idx[i] + 1];
        p2[2 * i] = (double)xy2[2 * idx[i]]; p2[2 * i + 1] = (double)xy2[2 * idx[i] + 1];
    }
    int ok = pmo_solve8(p1, p2, F);
    for (int i = 0; i < 9; ++i) F32[i] = (float)F[i];
    return ok;
}

/* SPEC S8 — fp32 scoring of one correspondence against one model. */
static int inlier32(const float* f, float x, float y, float xp, float yp, float thr2, int kind)
{
    float a = fmaf(f[0], x, fmaf(f[1], y, f[2]));
    float b = fmaf(f[3], x, fmaf(f[4], y, f[5]));
    float c = fmaf(f[6], x, fmaf(f[7], y, f[8]));
    float num = fmaf(xp, a, fmaf(yp, b, c));
    float at = fmaf(f[0], xp, fmaf(f[3], yp, f[6]));
    float bt = fmaf(f[1], xp, fmaf(f[4], yp, f[7]));
    float n2 = num * num;
    if (kind == 0) {
        float den = fmaf(a, a, fmaf(b, b, fmaf(at, at, bt * bt)));
        return n2 <= thr2 * den;
    } else {
        float d2 = fmaf(a, a, b * b);
        float d1 = fmaf(at, at, bt * bt);
        return (n2 <= thr2 * d2) && (n2 <= thr2 * d1);
    }
}

int pmo_score(const float F32[9], const float* xy1, const float* xy2, int n, float thresh_px,
              int kind, uint8_t* mask)
{
    float thr2 = thresh_px * thresh_px;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        int in = inlier32(F32, xy1[2 * i], xy1[2 * i + 1], xy2[2 * i], xy2[2 * i + 1], thr2, kind);
        if (mask) mask[i] = (uint8_t)in;
        cnt += in;
    }
    return cnt;
}

/* SPEC S9/S10 — RANSAC-F over a hypothesis range (replaces main.cpp:95-98). */
int pmo_ransac_fundamental(const float* xy1, const float* xy2, int n, const pmo_ransac_params* p,
                           double F[9], uint8_t* mask, int* n_inliers, uint64_t* best_key,
                           int nthreads)
{
    if (F) for (int i = 0; i < 9; ++i) F[i] = 0.0;
    if (mask) memset(mask, 0, (size_t)(n > 0 ? n : 0));
    if (n_inliers) *n_inliers = 0;
    if (best_key) *best_key = 0;
    if (n < 8) return -2;
    if (p->hyp_begin < 0 || p->hyp_end < p->hyp_begin || p->hyp_end > 0x100000000LL) return -1;
    uint64_t best = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        uint64_t lbest = 0;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t h = p->hyp_begin; h < p->hyp_end; ++h) {
            double Fh[9]; float F32[9];
            if (!pmo_hyp_model(xy1, xy2, n, p->seed, (uint64_t)h, Fh, F32)) continue;
            int cnt = pmo_score(F32, xy1, xy2, n, p->thresh_px, p->error_kind, NULL);
            uint64_t key = ((uint64_t)(uint32_t)cnt << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)h);
            if (key > lbest) lbest = key;
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        { if (lbest > best) best = lbest; }
    }
    (void)nthreads;
    if (best_key) *best_key = best;
    if (best == 0) return -3;
    uint64_t h = 0xFFFFFFFFu - (uint32_t)best;
    double Fh[9]; float F32[9];
    pmo_hyp_model(xy1, xy2, n, p->seed, h, Fh, F32);
    int cnt = pmo_score(F32, xy1, xy2, n, p->thresh_px, p->error_kind, mask);
    if (F) memcpy(F, Fh, sizeof Fh);
    if (n_inliers) *n_inliers = cnt;
    return 0;
}

int pmo_ransac_model_from_hyp(const float* xy1, const float* xy2, int n, const pmo_ransac_params* p,
                              int64_t hyp, double F[9], uint8_t* mask, int* n_inliers)
{
    if (n < 8) return -2;
    double Fh[9]; float F32[9];
    int ok = pmo_hyp_model(xy1, xy2, n, p->seed, (uint64_t)hyp, Fh, F32);
    if (!ok) {
        if (F) for (int i = 0; i < 9; ++i) F[i] = 0.0;
        if (mask) memset(mask, 0, (size_t)n);
        if (n_inliers) *n_inliers = 0;
        return -3;
    }
    int cnt = pmo_score(F32, xy1, xy2, n, p->thresh_px, p->error_kind, mask);
    if (F) memcpy(F, Fh, sizeof Fh);
    if (n_inliers) *n_inliers = cnt;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * SURVEY 8f-3 — the estimator the reference literally calls: findFundamentalMat(..., CV_FM_7POINT)
 * with more than 7 points runs OpenCV 2.4's LMedS loop over 7-point minimal solves
 * (main.cpp:95-98) [recalled; OpenCV is absent: parity unpinned, arithmetic frozen by SPEC
 * S13-S15].
 * ------------------------------------------------------------------------------------------ */
/* SPEC S13 — 7 distinct indices, same generator as S6 with a domain-separated stream. */
void pmo_sample7(uint64_t seed, uint64_t h, int n, int32_t idx[7])
{
    uint64_t stream = mix64(seed ^ 0x7F4A7C159E3779B9ULL) ^ mix64(h + 0xD1B54A32D192ED03ULL);
    int cnt = 0;
    for (uint64_t d = 0; d < 64 && cnt < 7; ++d) {
        uint64_t r = mix64(stream + (d + 1) * 0x9E3779B97F4A7C15ULL);
        int32_t c = (int32_t)(((r >> 32) * (uint64_t)(uint32_t)n) >> 32);
        int dup = 0;
        for (int s = 0; s < cnt; ++s) dup |= (idx[s] == c);
        if (!dup) idx[cnt++] = c;
    }
    for (int32_t c = 0; cnt < 7; ++c) {
        int dup = 0;
        for (int s = 0; s < cnt; ++s) dup |= (idx[s] == c);
        if (!dup) idx[cnt++] = c;
    }
}

static int hartley7(const double* p, double* pn, double* s_out, double* tx_out, double* ty_out)
{
    double cx = p[0], cy = p[1];
    for (int i = 1; i < 7; ++i) { cx = cx + p[2 * i]; cy = cy + p[2 * i + 1]; }
    cx = cx / 7.0; cy = cy / 7.0;
    double md = 0.0;
    for (int i = 0; i < 7; ++i) {
        double dx = p[2 * i] - cx, dy = p[2 * i + 1] - cy;
        md = md + sqrt(fma(dx, dx, dy * dy));
    }
    md = md / 7.0;
    if (!(md > 0.0) || !(md < INFINITY)) return 0;
    double s = PMO_SQRT2 / md;
    for (int i = 0; i < 7; ++i) {
        pn[2 * i] = (p[2 * i] - cx) * s;
        pn[2 * i + 1] = (p[2 * i + 1] - cy) * s;
    }
    *s_out = s; *tx_out = -(s * cx); *ty_out = -(s * cy);
    return 1;
}

static double det3(const double* r0, const double* r1, const double* r2)
{
    double m0 = r1[1] * r2[2] - r1[2] * r2[1];
    double m1 = r1[0] * r2[2] - r1[2] * r2[0];
    double m2 = r1[0] * r2[1] - r1[1] * r2[0];
    double d = r0[0] * m0;
    d = d - r0[1] * m1;
    d = d + r0[2] * m2;
    return d;
}

/* SPEC S14 — normalised 7-point solve: up to three models; F[r] (9 doubles each) and valid[r]. */
int pmo_solve7(const double* p1, const double* p2, double F[27], int valid[3])
{
    double n1[14], n2[14], s1, t1x, t1y, s2, t2x, t2y;
    for (int i = 0; i < 27; ++i) F[i] = 0.0;
    valid[0] = valid[1] = valid[2] = 0;
    if (!hartley7(p1, n1, &s1, &t1x, &t1y)) return 0;
    if (!hartley7(p2, n2, &s2, &t2x, &t2y)) return 0;
    double B[9][7], beta[7];
    for (int c = 0; c < 7; ++c) {
        double x1 = n1[2 * c], y1 = n1[2 * c + 1], x2 = n2[2 * c], y2 = n2[2 * c + 1];
        B[0][c] = x2 * x1; B[1][c] = x2 * y1; B[2][c] = x2;
        B[3][c] = y2 * x1; B[4][c] = y2 * y1; B[5][c] = y2;
        B[6][c] = x1;      B[7][c] = y1;      B[8][c] = 1.0;
    }
    for (int j = 0; j < 7; ++j) {
        double sigma = 0.0;
        for (int i = j + 1; i < 9; ++i) sigma = fma(B[i][j], B[i][j], sigma);
        double alpha = B[j][j];
        double nrm = sqrt(fma(alpha, alpha, sigma));
        if (!(nrm > 0.0)) { beta[j] = 0.0; continue; }
        double v0 = alpha + (alpha >= 0.0 ? nrm : -nrm);
        double vtv = fma(v0, v0, sigma);
        beta[j] = 2.0 / vtv;
        B[j][j] = v0;
        for (int c = j + 1; c < 7; ++c) {
            double dot = v0 * B[j][c];
            for (int i = j + 1; i < 9; ++i) dot = fma(B[i][j], B[i][c], dot);
            double w = beta[j] * dot;
            B[j][c] = fma(-w, v0, B[j][c]);
            for (int i = j + 1; i < 9; ++i) B[i][c] = fma(-w, B[i][j], B[i][c]);
        }
    }
    /* null space: g[0] = H0..H6 e7, g[1] = H0..H6 e8 */
    double g[2][9];
    for (int v = 0; v < 2; ++v) {
        for (int i = 0; i < 9; ++i) g[v][i] = 0.0;
        g[v][7 + v] = 1.0;
        for (int j = 6; j >= 0; --j) {
            if (beta[j] == 0.0) continue;
            double dot = B[j][j] * g[v][j];
            for (int i = j + 1; i < 9; ++i) dot = fma(B[i][j], g[v][i], dot);
            double w = beta[j] * dot;
            g[v][j] = fma(-w, B[j][j], g[v][j]);
            for (int i = j + 1; i < 9; ++i) g[v][i] = fma(-w, B[i][j], g[v][i]);
        }
    }
    const double *A0 = g[0], *A1 = g[0] + 3, *A2 = g[0] + 6, *B0 = g[1], *B1 = g[1] + 3, *B2 = g[1] + 6;
    double c3 = det3(A0, A1, A2), c0 = det3(B0, B1, B2);
    double c2 = det3(B0, A1, A2); c2 = c2 + det3(A0, B1, A2); c2 = c2 + det3(A0, A1, B2);
    double c1 = det3(A0, B1, B2); c1 = c1 + det3(B0, A1, B2); c1 = c1 + det3(B0, B1, A2);
    /* det(x*A + y*B) = c3 x^3 + c2 x^2 y + c1 x y^2 + c0 y^3; solve in the better-scaled variable */
    int swap = !(fabs(c3) >= fabs(c0));
    double k3 = swap ? c0 : c3, k2 = swap ? c1 : c2, k1 = swap ? c2 : c1, k0 = swap ? c3 : c0;
    const double* P = swap ? g[1] : g[0];     /* model = z*P + Q */
    const double* Q = swap ? g[0] : g[1];
    if (!(fabs(k3) > 0.0)) return 0;
    double a = k2 / k3, b = k1 / k3, c = k0 / k3;
    double R = fabs(a); if (fabs(b) > R) R = fabs(b); if (fabs(c) > R) R = fabs(c);
    R = 1.0 + R;
    if (!(R < INFINITY)) return 0;
    double lo = -R, hi = R;
    for (int it = 0; it < 100; ++it) {
        double mid = 0.5 * (lo + hi);
        double v = mid + a; v = fma(v, mid, b); v = fma(v, mid, c);
        if (v < 0.0) lo = mid; else hi = mid;
    }
    double roots[3]; int nr = 1;
    roots[0] = 0.5 * (lo + hi);
    double q1 = a + roots[0], q0 = fma(q1, roots[0], b);
    double disc = fma(q1, q1, -(4.0 * q0));
    if (disc >= 0.0) {
        double sq = sqrt(disc);
        double t = -0.5 * (q1 + (q1 >= 0.0 ? sq : -sq));
        roots[1] = t;
        roots[2] = (t != 0.0) ? q0 / t : 0.0;
        nr = 3;
    }
    int any = 0;
    for (int r = 0; r < nr; ++r) {
        double z = roots[r], Fn[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Fn[i][j] = fma(z, P[3 * i + j], Q[3 * i + j]);
        double M[3][3], Fo[9];
        for (int i = 0; i < 3; ++i) {
            M[i][0] = Fn[i][0] * s1;
            M[i][1] = Fn[i][1] * s1;
            M[i][2] = fma(Fn[i][0], t1x, fma(Fn[i][1], t1y, Fn[i][2]));
        }
        for (int j = 0; j < 3; ++j) {
            Fo[j] = s2 * M[0][j];
            Fo[3 + j] = s2 * M[1][j];
            Fo[6 + j] = fma(t2x, M[0][j], fma(t2y, M[1][j], M[2][j]));
        }
        double ss = 0.0;
        for (int i = 0; i < 9; ++i) ss = fma(Fo[i], Fo[i], ss);
        double nrm = sqrt(ss);
        if (!(nrm > 0.0) || !(nrm < INFINITY)) continue;
        double inv = 1.0 / nrm;
        if (Fo[8] < 0.0) inv = -inv;
        for (int i = 0; i < 9; ++i) F[9 * r + i] = Fo[i] * inv;
        valid[r] = 1; any = 1;
    }
    return any;
}

int pmo_hyp_models7(const float* xy1, const float* xy2, int n, uint64_t seed, uint64_t h, double F[27], int valid[3])
{
    int32_t idx[7];
    double p1[14], p2[14];
    pmo_sample7(seed, h, n, idx);
    for (int i = 0; i < 7; ++i) {
        p1[2 * i] = (double)xy1[2 * idx[i]]; p1[2 * i + 1] = (double)xy1[2 * idx[i] + 1];
        p2[2 * i] = (double)xy2[2 * idx[i]]; p2[2 * i + 1] = (double)xy2[2 * idx[i] + 1];
    }
    return pmo_solve7(p1, p2, F, valid);
}

/* SPEC S15 — symmetric epipolar residual max(d1^2/|l1|^2, d2^2/|l2|^2), fp64 -> float
 * (cv CvFMEstimator::computeReprojError [recalled]); NaN counts as +inf. */
static float lmeds_err(const double* F, float xf, float yf, float xpf, float ypf)
{
    double x = (double)xf, y = (double)yf, xp = (double)xpf, yp = (double)ypf;
    double a = fma(F[0], x, fma(F[1], y, F[2]));
    double b = fma(F[3], x, fma(F[4], y, F[5]));
    double c = fma(F[6], x, fma(F[7], y, F[8]));
    double d2 = fma(xp, a, fma(yp, b, c));
    double s2 = 1.0 / fma(a, a, b * b);
    double at = fma(F[0], xp, fma(F[3], yp, F[6]));
    double bt = fma(F[1], xp, fma(F[4], yp, F[7]));
    double ct = fma(F[2], xp, fma(F[5], yp, F[8]));
    double d1 = fma(x, at, fma(y, bt, ct));
    double s1 = 1.0 / fma(at, at, bt * bt);
    double e1 = (d1 * d1) * s1, e2 = (d2 * d2) * s2;
    float e = (float)(e1 >= e2 ? e1 : e2);
    return (e != e) ? INFINITY : e;
}

static int cmp_float(const void* a, const void* b)
{
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

typedef struct pmo_lmeds_params { int64_t hyp_begin, hyp_end; uint64_t seed; } pmo_lmeds_params;

/* median of the residuals of one model over all n correspondences (as double) */
double pmo_lmeds_median(const double F[9], const float* xy1, const float* xy2, int n, float* scratch)
{
    for (int i = 0; i < n; ++i) scratch[i] = lmeds_err(F, xy1[2 * i], xy1[2 * i + 1], xy2[2 * i], xy2[2 * i + 1]);
    qsort(scratch, (size_t)n, sizeof(float), cmp_float);
    return (n & 1) ? (double)scratch[n / 2] : ((double)scratch[n / 2 - 1] + (double)scratch[n / 2]) * 0.5;
}

int pmo_lmeds_fundamental(const float* xy1, const float* xy2, int n, const pmo_lmeds_params* p, double F[9],
                          uint8_t* mask, int* n_inliers, int64_t* best_model, double* median_out, int nthreads)
{
    if (F) for (int i = 0; i < 9; ++i) F[i] = 0.0;
    if (mask) memset(mask, 0, (size_t)(n > 0 ? n : 0));
    if (n_inliers) *n_inliers = 0;
    if (best_model) *best_model = -1;
    if (median_out) *median_out = INFINITY;
    if (n < 8) return -2;
    if (p->hyp_begin < 0 || p->hyp_end < p->hyp_begin || p->hyp_end > 0x40000000LL) return -1;
    double best = INFINITY; int64_t best_id = -1; double bestF[9] = {0};
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        float* scratch = (float*)malloc(sizeof(float) * (size_t)n);
        double lbest = INFINITY; int64_t lid = -1; double lF[9] = {0};
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t h = p->hyp_begin; h < p->hyp_end; ++h) {
            double Fm[27]; int valid[3];
            if (!pmo_hyp_models7(xy1, xy2, n, p->seed, (uint64_t)h, Fm, valid)) continue;
            for (int r = 0; r < 3; ++r) {
                if (!valid[r]) continue;
                double med = pmo_lmeds_median(Fm + 9 * r, xy1, xy2, n, scratch);
                int64_t id = 3 * h + r;
                if (med < lbest || (med == lbest && lid >= 0 && id < lid)) { lbest = med; lid = id; memcpy(lF, Fm + 9 * r, sizeof lF); }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        { if (lid >= 0 && (lbest < best || (lbest == best && (best_id < 0 || lid < best_id)))) { best = lbest; best_id = lid; memcpy(bestF, lF, sizeof lF); } }
        free(scratch);
    }
    (void)nthreads;
    if (best_id < 0 || !(best < INFINITY)) return -3;
    double sigma = ((2.5 * 1.4826) * (1.0 + 5.0 / (double)(n - 7))) * sqrt(best);
    sigma = sigma > 0.001 ? sigma : 0.001;
    double thr = sigma * sigma;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        float e = lmeds_err(bestF, xy1[2 * i], xy1[2 * i + 1], xy2[2 * i], xy2[2 * i + 1]);
        int in = (double)e <= thr;
        if (mask) mask[i] = (uint8_t)in;
        cnt += in;
    }
    if (F) memcpy(F, bestF, sizeof bestF);
    if (n_inliers) *n_inliers = cnt;
    if (best_model) *best_model = best_id;
    if (median_out) *median_out = best;
    return 0;
}

/* SPEC S16 — OpenCV 2.4's adaptive-iteration RANSAC over 7-point models (CV_FM_RANSAC)
 * [structure recalled: CvModelEstimator2::runRANSAC + cvRANSACUpdateNumIters]. */
typedef struct pmo_adaptive_params { int64_t max_iters; double confidence; float thresh_px; int32_t pad; uint64_t seed; } pmo_adaptive_params;

int pmo_ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0.0 ? p : 0.0; p = p < 1.0 ? p : 1.0;
    ep = ep > 0.0 ? ep : 0.0; ep = ep < 1.0 ? ep : 1.0;
    double num = 1.0 - p > DBL_MIN ? 1.0 - p : DBL_MIN;
    double denom = 1.0 - pow(1.0 - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return (denom >= 0.0 || -num >= (double)max_iters * (-denom)) ? max_iters : (int)lround(num / denom);
}

int pmo_ransac7_adaptive(const float* xy1, const float* xy2, int n, const pmo_adaptive_params* p, double F[9],
                         uint8_t* mask, int* n_inliers, int64_t* best_model, int* iters_run)
{
    if (F) for (int i = 0; i < 9; ++i) F[i] = 0.0;
    if (mask) memset(mask, 0, (size_t)(n > 0 ? n : 0));
    if (n_inliers) *n_inliers = 0;
    if (best_model) *best_model = -1;
    if (iters_run) *iters_run = 0;
    if (n < 8) return -2;
    if (p->max_iters < 1 || p->max_iters > (1 << 20)) return -1;
    const double thr = (double)p->thresh_px * (double)p->thresh_px;
    int niters = (int)p->max_iters, best = 6, h = 0;
    int64_t best_id = -1;
    double bestF[9] = {0};
    for (; h < niters; ++h) {
        double Fm[27]; int valid[3];
        if (!pmo_hyp_models7(xy1, xy2, n, p->seed, (uint64_t)h, Fm, valid)) continue;
        for (int r = 0; r < 3; ++r) {
            if (!valid[r]) continue;
            int c = 0;
            for (int i = 0; i < n; ++i)
                c += (double)lmeds_err(Fm + 9 * r, xy1[2 * i], xy1[2 * i + 1], xy2[2 * i], xy2[2 * i + 1]) <= thr;
            if (c > best) {
                best = c; best_id = 3 * (int64_t)h + r; memcpy(bestF, Fm + 9 * r, sizeof bestF);
                niters = pmo_ransac_update_num_iters(p->confidence, (double)(n - c) / (double)n, 7, niters);
            }
        }
    }
    if (iters_run) *iters_run = h;
    if (best_id < 0) return -3;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        int in = (double)lmeds_err(bestF, xy1[2 * i], xy1[2 * i + 1], xy2[2 * i], xy2[2 * i + 1]) <= thr;
        if (mask) mask[i] = (uint8_t)in;
        cnt += in;
    }
    if (F) memcpy(F, bestF, sizeof bestF);
    if (n_inliers) *n_inliers = cnt;
    if (best_model) *best_model = best_id;
    return 0;
}

/* main.cpp:103-123 — residual report.  transposed != 0 is the reference literally:
 * temp1 = [x1 y1 1] (1x3, main.cpp:110-112), temp2 = [x2 y2 1]^T (main.cpp:113-115),
 * result = temp1 * F * temp2 (main.cpp:117), sum += abs(result) (main.cpp:120). */
int pmo_epipolar_residuals(const float* xy1, const float* xy2, int n, const double F[9],
                           int transposed, double* r, double* mean_abs)
{
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        double xa, ya, xb, yb;
        if (transposed) { xa = xy1[2 * i]; ya = xy1[2 * i + 1]; xb = xy2[2 * i]; yb = xy2[2 * i + 1]; }
        else            { xa = xy2[2 * i]; ya = xy2[2 * i + 1]; xb = xy1[2 * i]; yb = xy1[2 * i + 1]; }
        double v0 = (xa * F[0] + ya * F[3]) + F[6];
        double v1 = (xa * F[1] + ya * F[4]) + F[7];
        double v2 = (xa * F[2] + ya * F[5]) + F[8];
        double res = (v0 * xb + v1 * yb) + v2;
        if (r) r[i] = res;
        sum += fabs(res);
    }
    if (mean_abs) *mean_abs = sum / n;      /* n == 0 -> NaN, as main.cpp:123 would print */
    return 0;
}

int pmo_f_scale_f33(double F[9])
{
    if (fabs(F[8]) > DBL_EPSILON) {
        double inv = 1.0 / F[8];
        for (int i = 0; i < 9; ++i) F[i] = F[i] * inv;
        return 1;
    }
    return 0;
}

/* main.cpp:127-142 — epipolar lines and the two end points handed to cv::line. */
int pmo_epilines(const float* xy, int n, int which_image, const double F[9], float* lines)
{
    for (int i = 0; i < n; ++i) {
        double x = xy[2 * i], y = xy[2 * i + 1], a, b, c;
        if (which_image == 1) {
            a = (F[0] * x + F[1] * y) + F[2]; b = (F[3] * x + F[4] * y) + F[5]; c = (F[6] * x + F[7] * y) + F[8];
        } else {
            a = (F[0] * x + F[3] * y) + F[6]; b = (F[1] * x + F[4] * y) + F[7]; c = (F[2] * x + F[5] * y) + F[8];
        }
        double nu = a * a + b * b;
        nu = nu != 0.0 ? 1.0 / sqrt(nu) : 1.0;
        lines[3 * i] = (float)(a * nu); lines[3 * i + 1] = (float)(b * nu); lines[3 * i + 2] = (float)(c * nu);
    }
    return 0;
}

static int32_t trunc_to_int(float v)
{
    if (!(v == v)) return INT32_MIN;
    if (v >= 2147483648.0f || v <= -2147483904.0f) return INT32_MIN;   /* x86 cvttss2si result */
    return (int32_t)v;
}

int pmo_epiline_endpoints(const float* lines, int n, int cols, int32_t* xyxy)
{
    for (int i = 0; i < n; ++i) {
        float a = lines[3 * i], b = lines[3 * i + 1], c = lines[3 * i + 2];
        xyxy[4 * i] = 0;
        xyxy[4 * i + 1] = trunc_to_int(-c / b);                       /* main.cpp:138 */
        xyxy[4 * i + 2] = cols;
        xyxy[4 * i + 3] = trunc_to_int(-(c + a * (float)cols) / b);   /* main.cpp:139-140 */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * SPEC S17 — search of a FLANN-style kd-forest (what `FlannBasedMatcher matcher;` at main.cpp:44 does behind
 * matcher.match at main.cpp:46 [recalled: FLANN 1.6.11's KDTreeIndex::getNeighbors / searchLevel, checks = 32,
 * eps = 0]).  Written recursively over a node pointer walk with an explicit array heap, i.e. NOT in the shape of the
 * HIP kernel (iterative descent, interleaved heap): an independent restatement of the same algorithm over the SAME
 * forest (pm_flann_export).  The forest's construction is checked by structural properties in the tests.
 * ------------------------------------------------------------------------------------------ */
typedef struct pmo_flann_node { int32_t child1, child2, divfeat; float divval; } pmo_flann_node;

typedef struct fl_state {
    const pmo_flann_node* nodes;
    const float* T; int dim, k, checks;
    const float* q;
    float rd[16]; int ri[16]; int rcount;
    float* hd; int* hn; int hcount, hcap;
    uint8_t* checked; int check_count;
} fl_state;

/* heap order: (bound, node id) — the id breaks ties, so the pop order does not depend on the heap's internals */
static int fl_before(const fl_state* s, int a, int b)
{
    return s->hd[a] < s->hd[b] || (s->hd[a] == s->hd[b] && s->hn[a] < s->hn[b]);
}

static void fl_heap_push(fl_state* s, float d, int node)
{
    if (s->hcount >= s->hcap) return;
    int i = s->hcount++;
    s->hd[i] = d; s->hn[i] = node;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!fl_before(s, i, p)) break;
        float td = s->hd[p]; s->hd[p] = s->hd[i]; s->hd[i] = td;
        int tn = s->hn[p]; s->hn[p] = s->hn[i]; s->hn[i] = tn;
        i = p;
    }
}

static int fl_heap_pop(fl_state* s, float* d, int* node)
{
    if (s->hcount == 0) return 0;
    *d = s->hd[0]; *node = s->hn[0];
    s->hcount--;
    if (s->hcount > 0) {
        s->hd[0] = s->hd[s->hcount]; s->hn[0] = s->hn[s->hcount];
        int i = 0;
        for (;;) {
            int l = 2 * i + 1, r = l + 1, m = i;
            if (l < s->hcount && fl_before(s, l, m)) m = l;
            if (r < s->hcount && fl_before(s, r, m)) m = r;
            if (m == i) break;
            float td = s->hd[m]; s->hd[m] = s->hd[i]; s->hd[i] = td;
            int tn = s->hn[m]; s->hn[m] = s->hn[i]; s->hn[i] = tn;
            i = m;
        }
    }
    return 1;
}

static void fl_search_level(fl_state* s, int node, float mindist)
{
    const float worst = s->rd[s->k - 1];
    if (worst < mindist) return;
    const pmo_flann_node* nd = &s->nodes[node];
    if (nd->child1 < 0) {
        const int idx = nd->divfeat;
        if (s->checked[idx] || (s->check_count >= s->checks && s->rcount >= s->k)) return;
        s->checked[idx] = 1;
        s->check_count++;
        const float d = pmo_l2sqr(s->q, s->T + (size_t)idx * s->dim, s->dim);
        if (d < s->rd[s->k - 1]) {                  /* KNNResultSet::addPoint: behind equal distances */
            int p = s->k - 1;
            while (p > 0 && s->rd[p - 1] > d) { s->rd[p] = s->rd[p - 1]; s->ri[p] = s->ri[p - 1]; --p; }
            s->rd[p] = d; s->ri[p] = idx;
            if (s->rcount < s->k) s->rcount++;
        }
        return;
    }
    const float val = s->q[nd->divfeat];
    const float diff = val - nd->divval;
    const int best = diff < 0.f ? nd->child1 : nd->child2;
    const int other = diff < 0.f ? nd->child2 : nd->child1;
    const float p = diff * diff;
    const float new_d = mindist + p;
    if (new_d < s->rd[s->k - 1] || s->rcount < s->k) fl_heap_push(s, new_d, other);
    fl_search_level(s, best, mindist);
}

int pmo_flann_search(const pmo_flann_node* nodes, const int32_t* roots, int trees, const float* T, int nt, int dim,
                     const float* Q, int nq, int k, int checks, int heap_cap, pmo_match* out)
{
    if (k < 1 || k > 16 || trees < 1 || nt < 1) return -1;
    float* hd = (float*)malloc(sizeof(float) * (size_t)heap_cap);
    int* hn = (int*)malloc(sizeof(int) * (size_t)heap_cap);
    uint8_t* checked = (uint8_t*)malloc((size_t)nt);
    if (!hd || !hn || !checked) { free(hd); free(hn); free(checked); return -2; }
    for (int i = 0; i < nq; ++i) {
        fl_state s;
        memset(&s, 0, sizeof s);
        s.nodes = nodes; s.T = T; s.dim = dim; s.k = k; s.checks = checks; s.q = Q + (size_t)i * dim;
        s.hd = hd; s.hn = hn; s.hcap = heap_cap; s.checked = checked;
        memset(checked, 0, (size_t)nt);
        for (int c = 0; c < 16; ++c) { s.rd[c] = INFINITY; s.ri[c] = -1; }
        for (int t = 0; t < trees; ++t) fl_search_level(&s, roots[t], 0.f);
        float bd; int bn;
        while (fl_heap_pop(&s, &bd, &bn) && (s.check_count < checks || s.rcount < k)) fl_search_level(&s, bn, bd);
        for (int c = 0; c < k; ++c) {
            pmo_match* m = &out[(size_t)i * k + c];
            m->queryIdx = i; m->imgIdx = 0; m->trainIdx = s.ri[c];
            m->distance = s.ri[c] >= 0 ? sqrtf(s.rd[c]) : INFINITY;
        }
    }
    free(hd); free(hn); free(checked);
    return 0;
}
