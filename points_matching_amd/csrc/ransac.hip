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
#include "ransac_internal.hpp"

namespace {

using namespace pm_ransac;

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

constexpr int SCORE_CHUNK = 256;      // correspondences staged per pass (4 KB of LDS)
constexpr int SCORE_MIN_CHUNK = 128;  // smaller chunks = more counter atomics and tickets: slower even though the chip is
                                      // underfilled (C3: 96 -> 29.8 us, 128 -> 24.6, 160 -> 24.3, 256 -> 25.9)


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
    const int operands = ctx->opts[PM_OPT_SCORE_OPERANDS];      // 1 / 2 pins a scorer (tests, A/B timing)
    const bool use_smem = operands ? operands == 2 : static_cast<long long>(chunks) * SCORE_CHUNK < n;
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

// One-launch kernel (ransac_fused.hip) or the hypothesis-per-lane solve + score launches below?  The one-launch kernel
// wins wherever launches, atomics and tickets dominate (C3: 10k ids x 2.3k matches, C5: 2k x 1.1k); for very large
// runs the lane-per-hypothesis scorer with scalar-operand points is ahead (C4, 100k ids x 9.2k matches on one GPU:
// 346 us against 370 us), so those keep it.  The rule uses the capacity (the count may live on the device).
// PM_OPT_RANSAC_PATH pins either (tests run both: two independent implementations of the same spec).
bool use_fused(const pm_ctx* ctx, long long n_cap, long long nh)
{
    const int opt = ctx->opts[PM_OPT_RANSAC_PATH];
    if (opt) return opt == 2;
    return n_cap * nh < (1LL << 30);
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
    const bool fused = hyp < 0 && p->hyp_end > p->hyp_begin && use_fused(ctx, n, p->hyp_end - p->hyp_begin);
    const size_t need = 2 * pm::align_up(xyb, 256) + pm::align_up(static_cast<size_t>(n), 256) + 512 +
                        (fused ? fused_scratch_bytes(ctx, p) : shard_scratch_bytes(p, n)) + 2048;
    rc = pm::arena_reserve(ctx, need);
    if (rc != PM_OK) return rc;
    pm::arena_reset(ctx);
    float* dxy1 = static_cast<float*>(pm::arena_take(ctx, xyb));
    float* dxy2 = static_cast<float*>(pm::arena_take(ctx, xyb));
    uint8_t* dmask = static_cast<uint8_t*>(pm::arena_take(ctx, static_cast<size_t>(n)));
    unsigned long long* dkey = static_cast<unsigned long long*>(pm::arena_take(ctx, 8));
    PM_REQUIRE(dxy1 && dxy2 && dmask && dkey, PM_E_NOMEM, "scratch arena too small");
    ShardScratch sc{};
    if (!fused) {
        rc = take_scratch(ctx, p, n, sc);
        if (rc != PM_OK) return rc;
    }
    rc = pm::pinned_reserve(ctx, sizeof(FinalOut) + 8 + static_cast<size_t>(n));
    if (rc != PM_OK) return rc;

    PM_HIP_CHECK(hipMemcpyAsync(dxy1, xy1, xyb, hipMemcpyHostToDevice, ctx->stream));
    PM_HIP_CHECK(hipMemcpyAsync(dxy2, xy2, xyb, hipMemcpyHostToDevice, ctx->stream));
    if (fused) {
        const pm_points_view v{dxy1, dxy2, nullptr, 1, n, 0, 1, 0};
        rc = fused_launch(ctx, v, p, 0, nullptr, dkey, nullptr, dmask, n, nullptr, &sc.fo);
    } else if (hyp < 0) {
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
    unsigned long long* key = reinterpret_cast<unsigned long long*>(d_best_key);
    if (ctx && p && n_max >= 1 && d_xy1 && d_xy2 && use_fused(ctx, n_max, p->hyp_end - p->hyp_begin)) {
        int rc = check_params(p);
        if (rc != PM_OK) return rc;
        PM_REQUIRE(p->hyp_end > p->hyp_begin, PM_E_INVALID, "empty hypothesis range");
        PM_HIP_CHECK(hipSetDevice(ctx->device));
        rc = pm::arena_reserve(ctx, fused_scratch_bytes(ctx, p) + 1024);
        if (rc != PM_OK) return rc;
        pm::arena_reset(ctx);
        const pm_points_view v{d_xy1, d_xy2, d_n, 1, n_max, 0, 1, 0};
        return fused_launch(ctx, v, p, 0, nullptr, key, d_F, d_mask, n_max, d_n_inliers, nullptr);
    }
    ShardScratch sc;
    int rc = dev_prologue(ctx, p, n_max, d_xy1, d_xy2, sc);
    if (rc != PM_OK) return rc;
    PM_REQUIRE(p->hyp_end > p->hyp_begin, PM_E_INVALID, "empty hypothesis range");
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
