// ransac_fused_kernels.hpp — RANSAC-F in ONE launch on gfx950 (MI355X): sample + normalised 8-point solve + Sampson scoring
// of every hypothesis against every correspondence + winner + inlier mask.  Replaces
// `cv::findFundamentalMat(Mat(selPoints1), Mat(selPoints2), ...)` (main.cpp:95-98) with the estimator
// BASELINE.json names (docs/SPEC.md S6-S9).
//
// Mapping (MI355X-first: the correspondences are scored out of registers):
//   * a workgroup owns `hb` consecutive hypothesis ids and ALL correspondences.  It is two TEAMS of four waves (one
//     wave per SIMD each); every team holds the whole tile — lane = correspondence pair (two points per packed-f32
//     instruction), 2*PPT2 points per thread = 2560 or 5120 correspondences (config C3: one tile, loaded once; longer
//     sets are walked tile by tile) — and scores every second hypothesis;
//   * solve phase: lane s < hb samples and solves hypothesis s (fp64, SPEC S7) and leaves the f32 model in LDS;
//   * score phase: the model of hypothesis s is a wave-wide LDS broadcast, every lane tests its own points
//     (v_pk_fma_f32), the verdicts leave the VALU as a v_cmp mask and are counted on the SCALAR unit
//     (s_bcnt1_i32_b64 of the wavefront ballot) — no atomics, no LDS traffic per point;
//   * the 4 partial counts per hypothesis meet in LDS, the workgroup's best key = (inliers << 32) | ~id and the
//     fp64 model behind it go to the workgroup's slot (write-through stores), one ticket per workgroup;
//   * the workgroup that draws the last ticket scans the slots, publishes the winner's stored fp64 model and
//     writes the inlier mask straight from its registers (local run), or emits the 80-byte (key, F) record a
//     multi-GPU run exchanges (sharded run: one all-gather, then pm_ransac_finish_parts_dev on every rank).
// The correspondences may be given as `parts` padded blocks with device-side counts (the all-gathered survivors
// of a query-row-sharded matcher): the view is resolved while loading, no concatenation pass exists.
//
// Everything here is templated on a diagnostics policy (ransac_core.hpp: NoDiag).  Three translation units include this
// header: csrc/ransac_fused.hip (the product launch: NoDiag, nothing else instantiated), csrc/ransac_shard.hip (finish
// kernel + the sharded C-ABI entry points; it shares the view helpers) and tools/ablation/ransac_fused_stamps.hip (a
// diagnostic launch with in-kernel phase stamps; replaces ransac_fused.o in a library of its own, never shipped).
#pragma once
#include "ransac_core.hpp"
#include "ransac_internal.hpp"

namespace pm_ransac {
namespace {

constexpr int RF_THREADS = 256;                      // a TEAM: four waves, one per SIMD, holding one tile of correspondences
constexpr int RF_TEAMS = 2;                          // teams per workgroup: team t scores the hypotheses s = t (mod 2)
constexpr int RF_WG = RF_THREADS * RF_TEAMS;         // 512 threads: two waves per SIMD, all in step (one barrier domain)
constexpr int RF_PTS_PER_SLOT = 2 * RF_THREADS;      // points covered by one register slot (pair) of every thread of a team
constexpr int RF_PPT2 = 5;                           // register slots per thread: tiles of 2560 correspondences (config C3)
constexpr int RF_PPT2_BIG = 10;                      // ... of 5120 for capacities beyond one small tile
static_assert(RF_PPT2_BIG <= 10, "score_tile is instantiated for 1..10 slots");

struct RfSlot {                   // one per workgroup, 128 B apart (never shares a line with another writer)
    unsigned long long key;
    double F[9];
    unsigned long long pad[6];
};
static_assert(sizeof(RfSlot) == 128, "slot stride");

struct RfOut {
    int shard;                    // 0: finish here (key, F, mask, count); 1: emit the record only
    unsigned long long* key;      // local
    double* F;                    // local, may be null
    uint8_t* mask;                // local
    int mask_len;                 // local: bytes of `mask` (zero beyond n)
    int* n_inliers;               // local, may be null
    FinalOut* fo;                 // local, may be null (host entry point reads it back)
    pm_ransac_record* rec;        // shard
};

// ---- correspondence view -------------------------------------------------------------------------------------
// offs[p] = first global index of part p, offs[parts] = n.  One wave computes it (lane p <-> part p).
// (parts == 1, a plain array, needs none of this: view_count() reads the one count as a wave-uniform scalar.)
__device__ __forceinline__ int view_count1(const pm_points_view& v)
{
    int c = v.cap;
    if (v.counts) { const int raw = *v.counts; c = raw < 0 ? 0 : (raw > v.cap ? v.cap : raw); }
    return c;
}
__device__ __forceinline__ void view_offsets(const pm_points_view& v, int* __restrict__ offs, int tid)
{
    if (tid < 64) {
        int c = 0;
        if (tid < v.parts) {
            c = v.cap;
            if (v.counts) {
                const int raw = v.counts[static_cast<size_t>(tid) * v.pitch_cnt];
                c = raw < 0 ? 0 : (raw > v.cap ? v.cap : raw);
            }
        }
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (tid >= o) incl += up;
        }
        if (tid < v.parts) offs[tid] = incl - c;
        if (tid == v.parts - 1) offs[v.parts] = incl;
    }
}

// point i of the concatenated order (i < n)
__device__ __forceinline__ void view_point(const pm_points_view& v, const int* __restrict__ offs, int i, float2& a, float2& b)
{
    size_t e = 2 * static_cast<size_t>(i);
    if (v.parts > 1) {
        int p = 0;
        while (p + 1 < v.parts && i >= offs[p + 1]) ++p;      // parts <= 64
        e = static_cast<size_t>(p) * static_cast<size_t>(v.pitch_xy) + 2 * static_cast<size_t>(i - offs[p]);
    }
    a = *reinterpret_cast<const float2*>(v.xy1 + e);
    b = *reinterpret_cast<const float2*>(v.xy2 + e);
}

template <typename DIAG>
__device__ __forceinline__ bool hyp_model_view(const pm_points_view& v, const int* __restrict__ offs, int n, uint64_t seed,
                                               uint64_t h, double (&F)[9])
{
    int idx[8];
    DIAG::solve(5);
    sample8(seed, h, n, idx);
    DIAG::solve(6);
    double x1[8], y1[8], x2[8], y2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float2 a, b;
        view_point(v, offs, idx[i], a, b);
        x1[i] = static_cast<double>(a.x); y1[i] = static_cast<double>(a.y);
        x2[i] = static_cast<double>(b.x); y2[i] = static_cast<double>(b.y);
    }
    DIAG::solve(0);
    const bool ok = solve8<DIAG>(x1, y1, x2, y2, F);
    DIAG::solve(7);
    return ok;
}

// The model of a hypothesis is wave-uniform, so its coefficients ride in SGPR pairs and reach v_pk_fma_f32 through the
// instruction's op_sel bits (bit i of op_sel / op_sel_hi picks the low or high dword of source i for the low / high
// result lane): no register splats (hipcc hoists 21 v_mov per hypothesis otherwise) and 80 fewer VGPRs than holding the
// model in vector registers.  gfx950 allows one scalar register (pair) per VALU instruction, so the pairs are laid out
// such that the two coefficients of every inner operation share a pair.  (Measured: the score phase runs at ~6 cycles
// per VALU instruction per SIMD with two OR four waves per SIMD, with VGPR or SGPR coefficients alike — the packed-f32
// pipe, not occupancy or operand fetch, bounds it; unpacked v_fma_f32 took 1.5x longer.)
//   sfma_in (q, A, y, C)   : splat(q[A]) * y + splat(q[C])
//   sfma_out(q, A, x, acc) : splat(q[A]) * x + acc
// One IEEE fma per component, exactly like __builtin_elementwise_fma on the same values.
template <int A, int C>
__device__ __forceinline__ f32x2 sfma_in(unsigned long long q, f32x2 y)
{
    f32x2 d;
    if (A == 0 && C == 1) asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(d) : "s"(q), "v"(y));
    if (A == 1 && C == 0) asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "s"(q), "v"(y));
    return d;
}
template <int A>
__device__ __forceinline__ f32x2 sfma_out(unsigned long long q, f32x2 x, f32x2 acc)
{
    f32x2 d;
    if (A == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "s"(q), "v"(x), "v"(acc));
    if (A == 1) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "s"(q), "v"(x), "v"(acc));
    return d;
}

// the six scalar pairs of a model: (f1,f2) (f4,f5) (f7,f8) (f3,f6) (f4,f7) (f0,f3)
struct ModelS {
    unsigned long long q12, q45, q78, q36, q47, q03;
};
__device__ __forceinline__ unsigned long long spair(float lo, float hi)
{
    const unsigned l = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(__float_as_uint(lo))));
    const unsigned h = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(__float_as_uint(hi))));
    return (static_cast<unsigned long long>(h) << 32) | l;
}
__device__ __forceinline__ ModelS model_to_sgprs(f32x4v m0, f32x4v m1, f32x2 m2)
{
    ModelS s;
    s.q12 = spair(m0[1], m0[2]); s.q45 = spair(m1[0], m1[1]); s.q78 = spair(m1[3], m2[0]);
    s.q36 = spair(m0[3], m1[2]); s.q47 = spair(m1[0], m1[3]); s.q03 = spair(m0[0], m0[3]);
    return s;
}

// SPEC S8 on two correspondences (the packed form of inlier32, same operations bit for bit).
template <int KIND>
__device__ __forceinline__ void inlier_pk_model(const ModelS& m, f32x2 x, f32x2 y, f32x2 xp, f32x2 yp, float thr2, bool& ia, bool& ib)
{
    const f32x2 a = sfma_out<0>(m.q03, x, sfma_in<0, 1>(m.q12, y));        // f0*x + (f1*y + f2)
    const f32x2 b = sfma_out<1>(m.q03, x, sfma_in<0, 1>(m.q45, y));        // f3*x + (f4*y + f5)
    const f32x2 c = sfma_out<1>(m.q36, x, sfma_in<0, 1>(m.q78, y));        // f6*x + (f7*y + f8)
    const f32x2 num = __builtin_elementwise_fma(xp, a, __builtin_elementwise_fma(yp, b, c));
    const f32x2 at = sfma_out<0>(m.q03, xp, sfma_in<0, 1>(m.q36, yp));     // f0*x' + (f3*y' + f6)
    const f32x2 bt = sfma_out<0>(m.q12, xp, sfma_in<0, 1>(m.q47, yp));     // f1*x' + (f4*y' + f7)
    const f32x2 n2 = num * num;
    const f32x2 t2 = f32x2{thr2, thr2};
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

// One tile against this team's share of the `hcount` models of the workgroup, KM register slots in use (compile-time: no per-slot
// branches).  The model of hypothesis s+1 is requested from LDS before hypothesis s is scored; the ten verdict masks
// of a hypothesis are collected first and counted afterwards, so the scalar unit waits for the vector pipe once per
// hypothesis, not once per v_cmp.
template <int KIND, int PPT2, int KM, typename DIAG>
__device__ __forceinline__ void score_tile(const float (*__restrict__ mdl)[12], int (*__restrict__ cnt)[RF_HB_MAX], int hcount,
                                           const f32x2 (&X)[PPT2], const f32x2 (&Y)[PPT2], const f32x2 (&XP)[PPT2],
                                           const f32x2 (&YP)[PPT2], float thr2, bool first, int lane, int wave, int team)
{
    if (team >= hcount) return;
    f32x4v m0 = *reinterpret_cast<const f32x4v*>(&mdl[team][0]);
    f32x4v m1 = *reinterpret_cast<const f32x4v*>(&mdl[team][4]);
    f32x2 m2 = *reinterpret_cast<const f32x2*>(&mdl[team][8]);
    for (int s = team; s < hcount; s += RF_TEAMS) {
        const f32x4v c0 = m0, c1 = m1;
        const f32x2 c2 = m2;
        const int sn = s + RF_TEAMS < hcount ? s + RF_TEAMS : s;
        m0 = *reinterpret_cast<const f32x4v*>(&mdl[sn][0]);
        m1 = *reinterpret_cast<const f32x4v*>(&mdl[sn][4]);
        m2 = *reinterpret_cast<const f32x2*>(&mdl[sn][8]);
        const ModelS ms = model_to_sgprs(c0, c1, c2);
        unsigned long long va[KM], vb[KM];
        const float fs[9] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3], c2[0]};   // (dead unless DIAG::scalar_fma)
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            bool ia, ib;
            if constexpr (DIAG::scalar_fma) {
                ia = inlier32<KIND>(fs, X[k][0], Y[k][0], XP[k][0], YP[k][0], thr2);
                ib = inlier32<KIND>(fs, X[k][1], Y[k][1], XP[k][1], YP[k][1], thr2);
            } else {
                inlier_pk_model<KIND>(ms, X[k], Y[k], XP[k], YP[k], thr2, ia, ib);
            }
            va[k] = __ballot(ia);
            vb[k] = __ballot(ib);
        }
        int c = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k) c += __popcll(va[k]) + __popcll(vb[k]);
        if (lane == 0) {
            if (first) cnt[wave][s] = c;
            else __hip_atomic_fetch_add(&cnt[wave][s], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add, no return: no wait
        }
    }
}

template <int NWAVES>
__device__ __forceinline__ unsigned long long wg_max_u64(unsigned long long key, unsigned long long* __restrict__ wk, int tid)
{
    key = pm::wave_max_u64(key);
    __syncthreads();                                   // wk may still be read from an earlier use
    if ((tid & 63) == 0) wk[tid >> 6] = key;
    __syncthreads();
    unsigned long long k = wk[0];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) k = wk[w] > k ? wk[w] : k;
    return k;
}

// ---- the kernel ------------------------------------------------------------------------------------------------
template <int KIND, int PPT2, typename DIAG>
__global__ __launch_bounds__(RF_WG) void ransac_fused(pm_points_view v, uint64_t seed, int64_t hyp_begin, int nh, int hb,
                                                           float thr2, RfSlot* __restrict__ slots, int* __restrict__ ticket,
                                                           RfOut out)
{
    __shared__ __attribute__((aligned(16))) float s_mdl[RF_HB_MAX][12];    // f32 model + valid flag of hypothesis s
    __shared__ double s_m64[RF_HB_MAX][9];
    __shared__ int s_cnt[RF_THREADS / 64][RF_HB_MAX];         // [wave of the scoring team][hypothesis]
    __shared__ int s_offs[PM_MAX_PARTS + 1];
    __shared__ unsigned long long s_wk[RF_WG / 64];
    __shared__ double s_F64[9];
    __shared__ int s_role;
    __shared__ int s_wc[RF_THREADS / 64];

    // tid: thread of the workgroup; ttid / twave: thread / wave inside its team (the tile geometry is a team's)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int team = tid / RF_THREADS, ttid = tid % RF_THREADS, twave = wave % (RF_THREADS / 64);
    DIAG::phase(0);
    int n;
    if (v.parts == 1) {
        n = view_count1(v);
    } else {
        view_offsets(v, s_offs, tid);
        __syncthreads();
        n = s_offs[v.parts];
    }
    DIAG::phase(1);

    // ---- correspondences -> registers, one tile of RF_TILE points at a time: slot k of thread t holds points
    // base + 2*(k*256 + t) and +1 as packed pairs (x_a, x_b), (y_a, y_b), (x'_a, x'_b), (y'_a, y'_b); points >= n
    // are NaN (never inliers)
    constexpr int RF_TILE = RF_PTS_PER_SLOT * PPT2;
    const int ntiles = n > RF_TILE ? (n + RF_TILE - 1) / RF_TILE : 1;
    f32x2 X[PPT2], Y[PPT2], XP[PPT2], YP[PPT2];
    const float nanv = __builtin_nanf("");
    auto load_tile = [&](int t) -> int {
        const int base = t * RF_TILE;
        int kmax = (n - base + RF_PTS_PER_SLOT - 1) / RF_PTS_PER_SLOT;
        kmax = kmax < 0 ? 0 : (kmax > PPT2 ? PPT2 : kmax);
#pragma unroll
        for (int k = 0; k < PPT2; ++k) {
            X[k] = f32x2{nanv, nanv}; Y[k] = X[k]; XP[k] = X[k]; YP[k] = X[k];
            if (k < kmax) {                                      // workgroup-uniform
                const int i0 = base + 2 * (k * RF_THREADS + ttid);
                float2 a0 = {nanv, nanv}, b0 = a0, a1 = a0, b1 = a0;
                if (v.parts == 1) {
                    // plain array: unconditional clamped loads (n >= 1 here), all of a tile's requests in flight together
                    const int j0 = i0 < n ? i0 : n - 1, j1 = i0 + 1 < n ? i0 + 1 : n - 1;
                    a0 = *reinterpret_cast<const float2*>(v.xy1 + 2 * static_cast<size_t>(j0));
                    a1 = *reinterpret_cast<const float2*>(v.xy1 + 2 * static_cast<size_t>(j1));
                    b0 = *reinterpret_cast<const float2*>(v.xy2 + 2 * static_cast<size_t>(j0));
                    b1 = *reinterpret_cast<const float2*>(v.xy2 + 2 * static_cast<size_t>(j1));
                    if (i0 >= n) { a0 = float2{nanv, nanv}; b0 = a0; }
                    if (i0 + 1 >= n) { a1 = float2{nanv, nanv}; b1 = a1; }
                } else {
                    if (i0 < n) view_point(v, s_offs, i0, a0, b0);
                    if (i0 + 1 < n) view_point(v, s_offs, i0 + 1, a1, b1);
                }
                X[k] = f32x2{a0.x, a1.x}; Y[k] = f32x2{a0.y, a1.y};
                XP[k] = f32x2{b0.x, b1.x}; YP[k] = f32x2{b0.y, b1.y};
            }
        }
        return kmax;
    };
    DIAG::phase(2);

    // ---- solve: lane s solves hypothesis h0 + s (SPEC S6, S7); models stay in LDS
    const int h0 = static_cast<int>(blockIdx.x) * hb;
    const int hcount = nh - h0 < hb ? nh - h0 : hb;
    // Lane s of the workgroup solves hypothesis s: ONE wave per 64 ids.  The solve is a ~3000-instruction fp64 stream
    // that costs its SIMD ~24k cycles whatever the number of active lanes (measured: spreading the ids over the four
    // waves of both co-resident workgroups doubled the phase), so ids are packed into as few waves as possible.
    if (tid < hcount) {
        double F[9];
        bool ok = false;
#pragma unroll
        for (int i = 0; i < 9; ++i) F[i] = 0.0;
        if (n >= 8) ok = hyp_model_view<DIAG>(v, s_offs, n, seed, static_cast<uint64_t>(hyp_begin + h0 + tid), F);
#pragma unroll
        for (int i = 0; i < 9; ++i) { s_mdl[tid][i] = static_cast<float>(F[i]); s_m64[tid][i] = F[i]; }
        s_mdl[tid][9] = ok ? 1.f : 0.f;
        s_mdl[tid][10] = 0.f; s_mdl[tid][11] = 0.f;
    }
    DIAG::phase(3);
    // the tile is loaded AFTER the solve: its 40 registers would otherwise be live (spilled) across the solver, which
    // alone decides the kernel's register count (128 at four waves per SIMD)
    int kmax = load_tile(0);
    __syncthreads();
    DIAG::phase(4);

    // ---- score: model = LDS broadcast, lane = its own points, count = popcount of the wavefront ballot (SALU)
    for (int t = 0; t < ntiles; ++t) {
        if (t > 0) kmax = load_tile(t);
        switch (kmax) {                                      // workgroup-uniform
#define PM_CASE(KM_) case KM_: score_tile<KIND, PPT2, (KM_ <= PPT2 ? KM_ : PPT2), DIAG>(s_mdl, s_cnt, hcount, X, Y, XP, YP, thr2, t == 0, lane, twave, team); break;
            PM_CASE(1) PM_CASE(2) PM_CASE(3) PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10)
#undef PM_CASE
            default:
                if (team == 0) for (int s = lane; s < hcount; s += 64) s_cnt[twave][s] = t == 0 ? 0 : s_cnt[twave][s];   // empty tile (n == 0)
                break;
        }
    }
    DIAG::phase(5);
    __syncthreads();

    // ---- the workgroup's best key (SPEC S9: most inliers, then lowest id) and its slot
    unsigned long long key = 0ull;
    if (tid < hcount && s_mdl[tid][9] != 0.f) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < RF_THREADS / 64; ++w) c += s_cnt[w][tid];
        key = (static_cast<unsigned long long>(static_cast<uint32_t>(c)) << 32) |
              static_cast<unsigned long long>(0xFFFFFFFFu - static_cast<uint32_t>(hyp_begin + h0 + tid));
    }
    const unsigned long long kbest = wg_max_u64<RF_WG / 64>(key, s_wk, tid);
    if (wave == 0) {
        const int sb = kbest ? static_cast<int>(static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(kbest)) - hyp_begin) - h0 : 0;
        RfSlot* sl = slots + blockIdx.x;
        if (lane < 9) __hip_atomic_store(&sl->F[lane], kbest ? s_m64[sb][lane] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 9) __hip_atomic_store(&sl->key, kbest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the slot is written through before the ticket is drawn
        if (lane == 0) {
            const int tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_role = tk == static_cast<int>(gridDim.x) - 1 ? 1 : 0;
        }
    }
    __syncthreads();
    DIAG::phase(6);
    if (s_role == 0) { DIAG::phase(9); return; }

    // ---- last workgroup: every slot is complete.  Winner = max key over the slots.
    if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    // every thread requests key AND model of its slots at once (one memory round trip), then the owner of the
    // winning key hands its model over through LDS
    unsigned long long kb = 0ull;
    double fb[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) fb[i] = 0.0;
    for (int j = tid; j < static_cast<int>(gridDim.x); j += RF_WG) {
        const unsigned long long kj = __hip_atomic_load(&slots[j].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double fj[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) fj[i] = __hip_atomic_load(&slots[j].F[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (kj > kb) {
            kb = kj;
#pragma unroll
            for (int i = 0; i < 9; ++i) fb[i] = fj[i];
        }
    }
    const unsigned long long kwin = wg_max_u64<RF_WG / 64>(kb, s_wk, tid);
    const bool ok = kwin != 0ull && n >= 8;
    if (tid < 9) s_F64[tid] = 0.0;
    __syncthreads();
    if (ok && kb == kwin) {                                  // exactly one thread: keys of distinct ids differ
#pragma unroll
        for (int i = 0; i < 9; ++i) s_F64[i] = fb[i];
    }
    __syncthreads();
    DIAG::phase(7);
    if (out.shard) {
        if (tid < 9) out.rec->F[tid] = s_F64[tid];
        if (tid == 9) out.rec->key = ok ? kwin : 0ull;
        return;
    }
    if (tid < 9) {
        if (out.F) out.F[tid] = s_F64[tid];
        if (out.fo) { out.fo->F[tid] = s_F64[tid]; out.fo->F32[tid] = static_cast<float>(s_F64[tid]); }
    }
    if (tid == 9) {
        *out.key = ok ? kwin : 0ull;
        if (out.fo) out.fo->valid = ok ? 1 : 0;
    }
    float fw[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) fw[i] = static_cast<float>(s_F64[i]);
    int mine = 0;
    if (team == 0) {                                         // team 0's registers cover every correspondence
        for (int t = 0; t < ntiles; ++t) {
            if (ntiles > 1) kmax = load_tile(t);             // a single tile is still in the registers
#pragma unroll
            for (int k = 0; k < PPT2; ++k) {
                if (k < kmax) {
                    bool ia, ib;
                    inlier32_x2_flags<KIND>(fw, X[k], Y[k], XP[k], YP[k], thr2, ia, ib);
                    ia = ia && ok; ib = ib && ok;
                    const int i0 = t * RF_TILE + 2 * (k * RF_THREADS + ttid);
                    if (i0 < out.mask_len) out.mask[i0] = ia ? 1 : 0;
                    if (i0 + 1 < out.mask_len) out.mask[i0 + 1] = ib ? 1 : 0;
                    mine += __popcll(__ballot(ia)) + __popcll(__ballot(ib));     // wave-uniform
                }
            }
        }
        if (lane == 0) s_wc[twave] = mine;
    }
    const int covered = n > 0 ? (n + RF_PTS_PER_SLOT - 1) / RF_PTS_PER_SLOT * RF_PTS_PER_SLOT : 0;
    for (int i = covered + tid; i < out.mask_len; i += RF_WG) out.mask[i] = 0;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < RF_THREADS / 64; ++w) tot += s_wc[w];
        if (out.n_inliers) *out.n_inliers = tot;
        if (out.fo) out.fo->n_inliers = tot;
    }
    DIAG::phase(8);
    DIAG::phase(9);
}

// ---- round 3: the same one-launch run with the correspondences in LDS and ONE WAVE PER HYPOTHESIS -------------------------
// Stamps of the register-tile kernel above (profiles/r02_ransac_phase_stamps.txt, r03_ransac_stamps_vs_ids_per_workgroup.txt):
// the fp64 solve is a ~21k-cycle stream per wave WHATEVER its lane count (40, 20, 10, 5 or 1 ids per workgroup: 20.9k-22.4k), the
// two teams then need 18.3k + 13.5k cycles for 40 hypotheses at two waves per SIMD (~6 cycles per instruction, four partial
// counts per hypothesis to combine, the second team trailing the first), and the tile load waits for the solve.  Here
//   * the tile lives in LDS as packed pairs (x_a x_b | y_a y_b | x'_a x'_b | y'_a y'_b per lane and 128-point slot: 2 KiB per
//     slot, config C3's 2275 correspondences = 36 KiB), loaded by the waves that do not solve WHILE the solver waves solve;
//   * 12 waves = 3 per SIMD (768 threads: the 156-VGPR solver still fits without spills) each score WHOLE hypotheses
//     (wave w: ids w, w + 12, ...): operands by ds_read_b64, model in SGPR pairs as before, the inlier count of a hypothesis
//     is a scalar sum of s_bcnt1 in ONE wave — no partial counts, no team skew;
//   * key, slot, ticket, winner and mask as above (the last workgroup reads the mask's operands from LDS again).
// Same operations on the same values: keys, masks and models are bit-identical to the register-tile kernel and the oracle.
constexpr int RL_WAVES = 12;
constexpr int RL_THREADS = RL_WAVES * 64;
constexpr int RL_SLOT_PTS = 128;             // points per slot: 64 lanes x 2 (packed)
constexpr int RL_MAX_SLOTS = 64;             // slots per LDS tile: 128 KiB dynamic (+ ~18 KiB static) of the CU's 160 KiB

// NH hypotheses of one wave (ids s0, s0 + RL_WAVES, ...) over the slots of the LDS tile in ONE pass: each operand read serves NH
// models (the LDS port, 4 x 512 B per wave and slot, would otherwise be ~3/4 as busy as the VALU and the two contend).
template <int KIND, int NH>
__device__ __forceinline__ void score_lds(const float (*s_mdl)[12], int* s_cnt, const f32x2* pp, int kslots, int s0, float thr2, int lane)
{
    ModelS ms[NH];
    int c[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const float* m = s_mdl[s0 + h * RL_WAVES];
        ms[h] = model_to_sgprs(*reinterpret_cast<const f32x4v*>(m), *reinterpret_cast<const f32x4v*>(m + 4),
                               *reinterpret_cast<const f32x2*>(m + 8));
        c[h] = 0;
    }
    for (int slot = 0; slot < kslots; ++slot) {
        const f32x2 x = pp[0], y = pp[64], xp = pp[128], yp = pp[192];
        pp += 256;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            bool ia, ib;
            inlier_pk_model<KIND>(ms[h], x, y, xp, yp, thr2, ia, ib);
            c[h] += __popcll(__ballot(ia)) + __popcll(__ballot(ib));
        }
    }
    if (lane == 0) {                                                       // (this wave owns these hypotheses: no other writer)
#pragma unroll
        for (int h = 0; h < NH; ++h) s_cnt[s0 + h * RL_WAVES] += c[h];
    }
}

template <int KIND, typename DIAG>
__global__ __launch_bounds__(RL_THREADS) void ransac_fused_lds(pm_points_view v, uint64_t seed, int64_t hyp_begin, int nh, int hb,
                                                               float thr2, int tile_slots, RfSlot* __restrict__ slots,
                                                               int* __restrict__ ticket, RfOut out)
{
    extern __shared__ __attribute__((aligned(16))) f32x2 s_pts[];           // [tile_slots][4][64]: X, Y, XP, YP pairs
    __shared__ __attribute__((aligned(16))) float s_mdl[RF_HB_MAX][12];    // f32 model + valid flag of hypothesis s
    __shared__ double s_m64[RF_HB_MAX][9];
    __shared__ int s_cnt[RF_HB_MAX];
    __shared__ int s_offs[PM_MAX_PARTS + 1];
    __shared__ unsigned long long s_wk[RL_WAVES];
    __shared__ double s_F64[9];
    __shared__ int s_role;
    __shared__ int s_wc[RL_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    DIAG::phase(0);
    int n;
    if (v.parts == 1) {
        n = view_count1(v);
    } else {
        view_offsets(v, s_offs, tid);
        __syncthreads();
        n = s_offs[v.parts];
    }
    DIAG::phase(1);
    const int tile_pts = tile_slots * RL_SLOT_PTS;
    const int ntiles = n > tile_pts ? (n + tile_pts - 1) / tile_pts : 1;
    const float nanv = __builtin_nanf("");
    // slot `slot`, lane `l` of the tile that starts at point `base`: points base + 2*(64*slot + l) and +1 (NaN past n)
    auto load_pair = [&](int base, int slot, int l) {
        const int i0 = base + 2 * (64 * slot + l);
        float2 a0 = {nanv, nanv}, b0 = a0, a1 = a0, b1 = a0;
        if (v.parts == 1) {
            if (n > 0) {
                const int j0 = i0 < n ? i0 : n - 1, j1 = i0 + 1 < n ? i0 + 1 : n - 1;      // unconditional clamped loads
                a0 = *reinterpret_cast<const float2*>(v.xy1 + 2 * static_cast<size_t>(j0));
                a1 = *reinterpret_cast<const float2*>(v.xy1 + 2 * static_cast<size_t>(j1));
                b0 = *reinterpret_cast<const float2*>(v.xy2 + 2 * static_cast<size_t>(j0));
                b1 = *reinterpret_cast<const float2*>(v.xy2 + 2 * static_cast<size_t>(j1));
            }
            if (i0 >= n) { a0 = float2{nanv, nanv}; b0 = a0; }
            if (i0 + 1 >= n) { a1 = float2{nanv, nanv}; b1 = a1; }
        } else {
            if (i0 < n) view_point(v, s_offs, i0, a0, b0);
            if (i0 + 1 < n) view_point(v, s_offs, i0 + 1, a1, b1);
        }
        f32x2* d = s_pts + (static_cast<size_t>(slot) * 4) * 64 + l;
        d[0] = f32x2{a0.x, a1.x}; d[64] = f32x2{a0.y, a1.y}; d[128] = f32x2{b0.x, b1.x}; d[192] = f32x2{b0.y, b1.y};
    };
    auto tile_kslots = [&](int t) {
        int k = (n - t * tile_pts + RL_SLOT_PTS - 1) / RL_SLOT_PTS;
        return k < 0 ? 0 : (k > tile_slots ? tile_slots : k);
    };

    // ---- solve (threads < hcount: SPEC S6, S7) || tile 0 -> LDS (the other waves)
    const int h0 = static_cast<int>(blockIdx.x) * hb;
    const int hcount = nh - h0 < hb ? nh - h0 : hb;
    const int solver_waves = (hcount + 63) / 64;                            // 1 (or 2 beyond 64 ids per workgroup)
    DIAG::phase(2);
    if (wave < solver_waves) {
        if (tid < hcount) {
            double F[9];
            bool ok = false;
#pragma unroll
            for (int i = 0; i < 9; ++i) F[i] = 0.0;
            if (n >= 8) ok = hyp_model_view<DIAG>(v, s_offs, n, seed, static_cast<uint64_t>(hyp_begin + h0 + tid), F);
#pragma unroll
            for (int i = 0; i < 9; ++i) { s_mdl[tid][i] = static_cast<float>(F[i]); s_m64[tid][i] = F[i]; }
            s_mdl[tid][9] = ok ? 1.f : 0.f;
            s_mdl[tid][10] = 0.f; s_mdl[tid][11] = 0.f;
            s_cnt[tid] = 0;
        }
    } else {
        const int k0 = tile_kslots(0);
        for (int slot = wave - solver_waves; slot < k0; slot += RL_WAVES - solver_waves) load_pair(0, slot, lane);
    }
    DIAG::phase(3);
    __syncthreads();
    DIAG::phase(4);

    // ---- score: wave w takes whole hypotheses w, w + 12, ...; the count of a hypothesis is a scalar sum in one wave
    for (int t = 0; t < ntiles; ++t) {
        const int kslots = tile_kslots(t);
        if (t > 0) {
            __syncthreads();                                               // everyone is done with the previous tile
            for (int slot = wave; slot < kslots; slot += RL_WAVES) load_pair(t * tile_pts, slot, lane);
            __syncthreads();
        }
        for (int s = wave; s < hcount; s += 4 * RL_WAVES) {
            const int left = (hcount - s + RL_WAVES - 1) / RL_WAVES;       // hypotheses of this wave from s on
            if (left >= 4) score_lds<KIND, 4>(s_mdl, s_cnt, s_pts + lane, kslots, s, thr2, lane);
            else if (left == 3) score_lds<KIND, 3>(s_mdl, s_cnt, s_pts + lane, kslots, s, thr2, lane);
            else if (left == 2) score_lds<KIND, 2>(s_mdl, s_cnt, s_pts + lane, kslots, s, thr2, lane);
            else score_lds<KIND, 1>(s_mdl, s_cnt, s_pts + lane, kslots, s, thr2, lane);
        }
    }
    DIAG::phase(5);
    __syncthreads();

    // ---- the workgroup's best key (SPEC S9: most inliers, then lowest id) and its slot
    unsigned long long key = 0ull;
    if (tid < hcount && s_mdl[tid][9] != 0.f)
        key = (static_cast<unsigned long long>(static_cast<uint32_t>(s_cnt[tid])) << 32) |
              static_cast<unsigned long long>(0xFFFFFFFFu - static_cast<uint32_t>(hyp_begin + h0 + tid));
    const unsigned long long kbest = wg_max_u64<RL_WAVES>(key, s_wk, tid);
    if (wave == 0) {
        const int sb = kbest ? static_cast<int>(static_cast<int64_t>(0xFFFFFFFFu - static_cast<uint32_t>(kbest)) - hyp_begin) - h0 : 0;
        // slots as ten arrays of gridDim.x words (F[0] .. F[8], key): the last workgroup's scan reads them coalesced
        double* sf = reinterpret_cast<double*>(slots) + static_cast<size_t>(lane) * gridDim.x + blockIdx.x;
        if (lane < 9) __hip_atomic_store(sf, kbest ? s_m64[sb][lane] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 9) __hip_atomic_store(reinterpret_cast<unsigned long long*>(sf), kbest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the slot is written through before the ticket is drawn
        if (lane == 0) {
            const int tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_role = tk == static_cast<int>(gridDim.x) - 1 ? 1 : 0;
        }
    }
    __syncthreads();
    DIAG::phase(6);
    if (s_role == 0) { DIAG::phase(9); return; }

    // ---- last workgroup: every slot is complete.  Winner = max key over the slots.
    if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    unsigned long long kb = 0ull;
    double fb[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) fb[i] = 0.0;
    for (int j = tid; j < static_cast<int>(gridDim.x); j += RL_THREADS) {
        const double* sf = reinterpret_cast<const double*>(slots) + j;
        const unsigned long long kj = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(sf + 9 * static_cast<size_t>(gridDim.x)),
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double fj[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) fj[i] = __hip_atomic_load(sf + i * static_cast<size_t>(gridDim.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (kj > kb) {
            kb = kj;
#pragma unroll
            for (int i = 0; i < 9; ++i) fb[i] = fj[i];
        }
    }
    const unsigned long long kwin = wg_max_u64<RL_WAVES>(kb, s_wk, tid);
    const bool ok = kwin != 0ull && n >= 8;
    if (tid < 9) s_F64[tid] = 0.0;
    __syncthreads();
    if (ok && kb == kwin) {                                  // exactly one thread: keys of distinct ids differ
#pragma unroll
        for (int i = 0; i < 9; ++i) s_F64[i] = fb[i];
    }
    __syncthreads();
    DIAG::phase(7);
    if (out.shard) {
        if (tid < 9) out.rec->F[tid] = s_F64[tid];
        if (tid == 9) out.rec->key = ok ? kwin : 0ull;
        return;
    }
    if (tid < 9) {
        if (out.F) out.F[tid] = s_F64[tid];
        if (out.fo) { out.fo->F[tid] = s_F64[tid]; out.fo->F32[tid] = static_cast<float>(s_F64[tid]); }
    }
    if (tid == 9) {
        *out.key = ok ? kwin : 0ull;
        if (out.fo) out.fo->valid = ok ? 1 : 0;
    }
    float fw[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) fw[i] = static_cast<float>(s_F64[i]);
    int mine = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int kslots = tile_kslots(t);
        if (ntiles > 1) {                                    // (a single tile is still in LDS)
            __syncthreads();
            for (int slot = wave; slot < kslots; slot += RL_WAVES) load_pair(t * tile_pts, slot, lane);
            __syncthreads();
        }
        for (int slot = wave; slot < kslots; slot += RL_WAVES) {
            const f32x2* pp = s_pts + static_cast<size_t>(slot) * 256 + lane;
            bool ia, ib;
            inlier32_x2_flags<KIND>(fw, pp[0], pp[64], pp[128], pp[192], thr2, ia, ib);
            ia = ia && ok; ib = ib && ok;
            const int i0 = t * tile_pts + 2 * (64 * slot + lane);
            if (i0 < out.mask_len) out.mask[i0] = ia ? 1 : 0;
            if (i0 + 1 < out.mask_len) out.mask[i0 + 1] = ib ? 1 : 0;
            mine += __popcll(__ballot(ia)) + __popcll(__ballot(ib));     // wave-uniform
        }
    }
    if (lane == 0) s_wc[wave] = mine;
    const int covered = n > 0 ? (n + RL_SLOT_PTS - 1) / RL_SLOT_PTS * RL_SLOT_PTS : 0;
    for (int i = covered + tid; i < out.mask_len; i += RL_THREADS) out.mask[i] = 0;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < RL_WAVES; ++w) tot += s_wc[w];
        if (out.n_inliers) *out.n_inliers = tot;
        if (out.fo) out.fo->n_inliers = tot;
    }
    DIAG::phase(8);
    DIAG::phase(9);
}

int check_view(const pm_points_view* v)
{
    PM_REQUIRE(v != nullptr && v->xy1 && v->xy2, PM_E_INVALID, "null correspondence view");
    PM_REQUIRE(v->parts >= 1 && v->parts <= PM_MAX_PARTS && v->cap >= 1, PM_E_INVALID, "need 1 <= parts <= 64, cap >= 1");
    PM_REQUIRE(v->parts == 1 || (v->pitch_xy >= 2LL * v->cap), PM_E_INVALID, "pitch_xy smaller than a part");
    PM_REQUIRE(static_cast<long long>(v->parts) * v->cap <= 0x7FFFFFFFLL, PM_E_INVALID, "view too large");
    return PM_OK;
}

int sync_words(pm_ctx* ctx, int** out)
{
    if (!ctx->sync_words) {
        PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->sync_words), 256));
        PM_HIP_CHECK(hipMemsetAsync(ctx->sync_words, 0, 256, ctx->stream));
    }
    *out = ctx->sync_words;
    return PM_OK;
}

// Enqueue the one-launch run.  The arena must already be reserved for fused_scratch_bytes(); it is carved here.
// shard != 0: only *rec is written.  Otherwise key / F / mask[mask_len] / n_inliers (F, n_inliers, fo_out optional).
template <typename DIAG>
int fused_launch_t(pm_ctx* ctx, const pm_points_view& v, const pm_ransac_params* p, int shard, pm_ransac_record* d_rec,
                 unsigned long long* d_key, double* d_F, uint8_t* d_mask, int mask_len, int* d_ninl, FinalOut** fo_out)
{
    const long long nh = p->hyp_end - p->hyp_begin;
    const long long cap_total = static_cast<long long>(v.parts) * v.cap;
    PM_REQUIRE(nh >= 1, PM_E_INVALID, "fused_launch: empty hypothesis range");
    // the kernel counts the ids of a launch in an int: one shard holds at most 2^31 - 1 of them (ids themselves go to 2^32)
    PM_REQUIRE(nh <= 0x7FFFFFFFLL, PM_E_INVALID, "a single launch takes at most 2^31 - 1 hypothesis ids: split the range");
    const int hb = fused_hb(ctx, nh);
    const int nwg = static_cast<int>((nh + hb - 1) / hb);
    RfSlot* slots = static_cast<RfSlot*>(pm::arena_take(ctx, sizeof(RfSlot) * static_cast<size_t>(nwg)));
    FinalOut* fo = static_cast<FinalOut*>(pm::arena_take(ctx, sizeof(FinalOut)));
    PM_REQUIRE(slots && fo, PM_E_NOMEM, "scratch arena too small");
    int* sync = nullptr;
    int rc = sync_words(ctx, &sync);
    if (rc != PM_OK) return rc;
    RfOut out{};
    out.shard = shard;
    out.key = d_key; out.F = d_F; out.mask = d_mask; out.mask_len = mask_len; out.n_inliers = d_ninl;
    out.fo = fo_out ? fo : nullptr;
    out.rec = d_rec;
    if (fo_out) *fo_out = fo;
    const float thr2 = p->thresh_px * p->thresh_px;
    pm::ScopedKernelTime t(ctx, "ransac_fused");
    // PM_OPT_RANSAC_FORM: 1 = correspondences in registers, two teams (round 2), 2 = correspondences in LDS, one wave per
    // hypothesis (round 3, the default)
    if (ctx->opts[PM_OPT_RANSAC_FORM] != 1) {
        long long need = (cap_total + RL_SLOT_PTS - 1) / RL_SLOT_PTS;
        const int tile_slots = static_cast<int>(need < 1 ? 1 : (need > RL_MAX_SLOTS ? RL_MAX_SLOTS : need));
        const size_t lds = static_cast<size_t>(tile_slots) * 4 * 64 * sizeof(f32x2);
        static bool attr_done_dev[PM_MAX_DEVICES] = {};
        if (!attr_done_dev[ctx->device]) {
            const int lmax = RL_MAX_SLOTS * 4 * 64 * static_cast<int>(sizeof(f32x2));
            PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ransac_fused_lds<PM_ERR_SAMPSON, DIAG>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lmax));
            PM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ransac_fused_lds<PM_ERR_SYM_EPIPOLAR, DIAG>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lmax));
            attr_done_dev[ctx->device] = true;
        }
        if (p->error_kind == PM_ERR_SAMPSON)
            hipLaunchKernelGGL((ransac_fused_lds<PM_ERR_SAMPSON, DIAG>), dim3(nwg), dim3(RL_THREADS), lds, ctx->stream, v, p->seed,
                               p->hyp_begin, static_cast<int>(nh), hb, thr2, tile_slots, slots, sync, out);
        else
            hipLaunchKernelGGL((ransac_fused_lds<PM_ERR_SYM_EPIPOLAR, DIAG>), dim3(nwg), dim3(RL_THREADS), lds, ctx->stream, v, p->seed,
                               p->hyp_begin, static_cast<int>(nh), hb, thr2, tile_slots, slots, sync, out);
        PM_HIP_CHECK(hipGetLastError());
        return PM_OK;
    }
    // 2*RF_PPT2 points per thread: a 2560-point tile (config C3 fits one); small capacities take the 2-slot build
#define PM_RF(KIND_, PPT2_)                                                                                              \
    hipLaunchKernelGGL((ransac_fused<KIND_, PPT2_, DIAG>), dim3(nwg), dim3(RF_WG), 0, ctx->stream, v, p->seed, p->hyp_begin, \
                       static_cast<int>(nh), hb, thr2, slots, sync, out)
    const int depth = cap_total <= 2 * RF_PTS_PER_SLOT ? 2 : (cap_total <= RF_PPT2 * RF_PTS_PER_SLOT ? RF_PPT2 : RF_PPT2_BIG);
#define PM_RF_K(KIND_)                                                                   \
    do {                                                                                 \
        if (depth == 2) PM_RF(KIND_, 2);                                                 \
        else if (depth == RF_PPT2) PM_RF(KIND_, RF_PPT2);                                \
        else PM_RF(KIND_, RF_PPT2_BIG);                                                  \
    } while (0)
    if (p->error_kind == PM_ERR_SAMPSON) PM_RF_K(PM_ERR_SAMPSON);
    else PM_RF_K(PM_ERR_SYM_EPIPOLAR);
#undef PM_RF_K
#undef PM_RF
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

}  // namespace
}  // namespace pm_ransac
