// filter_gather.hip — device-resident form of the glue between the two hot kernels:
// ratio test in the slot of main.cpp:49-69, index extraction main.cpp:77-78 and
// KeyPoint::convert main.cpp:89-91, fused into one stable compaction so that a matched batch
// never leaves HBM between the matcher and the RANSAC kernels.  Survivors keep query order
// (the reference's push_back order, main.cpp:63-68).  Same predicate as pm_filter_ratio:
// second neighbour present and d1 < ratio * d2 (float multiply, strict).
#include "pm_common.hpp"

namespace {

// Stable compaction in ONE launch, decoupled look-back style.  Block b owns rows
// [b*FG_ROWS, (b+1)*FG_ROWS): it evaluates its own rows, publishes its survivor count as an
// epoch-tagged word (agent-scope relaxed store), and obtains its output offset from the counts of
// blocks 0..b-1, which it polls with agent-scope relaxed loads (one word per polling lane).
// Blocks only ever wait for LOWER block ids and the grid (nq/256 blocks) is far below what the
// chip keeps resident, so the wait terminates; it is bounded anyway: a poller that gives up
// re-derives the missing count from the k-NN records itself (always correct, just slower).
// The tag (epoch << 10 | count) makes clearing the words between calls unnecessary.
constexpr int FG_ROWS = 256;
constexpr int FG_MAX_BLOCKS = 4096;          // 1M query rows; more -> the caller is told so
constexpr unsigned FG_SPIN_LIMIT = 200000u;

enum { FG_RATIO = 0, FG_MIDPOINT = 1 };
struct FgParam {
    float ratio;       // FG_RATIO: d1 < ratio * d2
    double cut;        // FG_MIDPOINT: (double)d < cut, cut = min + (max - min)/2 (main.cpp:65)
};

template <int MODE>
__device__ __forceinline__ bool fg_keep(const pm_match* __restrict__ knn, int i, int k, const FgParam& prm, pm_match& best)
{
    const uint4 a = *reinterpret_cast<const uint4*>(knn + static_cast<size_t>(i) * k);
    best.queryIdx = static_cast<int>(a.x); best.trainIdx = static_cast<int>(a.y);
    best.imgIdx = static_cast<int>(a.z);   best.distance = __uint_as_float(a.w);
    if (MODE == FG_RATIO) {
        const uint4 b = *reinterpret_cast<const uint4*>(knn + static_cast<size_t>(i) * k + 1);
        const float rhs = prm.ratio * __uint_as_float(b.w);
        return static_cast<int>(a.y) >= 0 && static_cast<int>(b.y) >= 0 && best.distance < rhs;
    }
    return static_cast<double>(best.distance) < prm.cut;
}

// order-preserving map float -> uint (negative values reversed, positives above them)
__device__ __forceinline__ unsigned fg_ord(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fg_unord(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// main.cpp:49-56: minMatch starts at 1, maxMatch at 0, then min/max over the distances (NaN never
// wins a comparison).  mm[0] = epoch-tagged max of ~ord(d) (the minimum), mm[1] = max of ord(d).
__global__ __launch_bounds__(FG_ROWS) void midpoint_minmax(const pm_match* __restrict__ m, int n, int k,
                                                           unsigned long long* __restrict__ mm, unsigned epoch)
{
    const int i = blockIdx.x * FG_ROWS + threadIdx.x;
    float d = __uint_as_float(reinterpret_cast<const unsigned*>(m + static_cast<size_t>(i < n ? i : n - 1) * k)[3]);
    const bool ok = i < n && d == d;
    unsigned lo = fg_ord(1.0f), hi = fg_ord(0.0f);
    if (ok) { const unsigned o = fg_ord(d); lo = o < lo ? o : lo; hi = o > hi ? o : hi; }
    lo = pm::wave_min_u32(lo);
    hi = pm::wave_max_u32(hi);
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long tag = static_cast<unsigned long long>(epoch) << 32;
        atomicMax(&mm[0], tag | static_cast<unsigned>(~lo));
        atomicMax(&mm[1], tag | hi);
    }
}

template <int MODE>
__global__ __launch_bounds__(FG_ROWS) void filter_ratio_gather(const pm_match* __restrict__ knn, int nq, int k,
                                                               float ratio, const float* __restrict__ kp1,
                                                               const float* __restrict__ kp2,
                                                               pm_match* __restrict__ good, float* __restrict__ xy1,
                                                               float* __restrict__ xy2, int* __restrict__ n_out,
                                                               unsigned* __restrict__ blk_counts, unsigned epoch,
                                                               const unsigned long long* __restrict__ mm,
                                                               double* __restrict__ minmax_out)
{
    FgParam prm{ratio, 0.0};
    double lo_d = 1.0, hi_d = 0.0;
    if (MODE == FG_MIDPOINT) {                     // the words were completed by midpoint_minmax (previous launch)
        lo_d = static_cast<double>(fg_unord(~static_cast<unsigned>(mm[0])));
        hi_d = static_cast<double>(fg_unord(static_cast<unsigned>(mm[1])));
        prm.cut = lo_d + (hi_d - lo_d) / 2;
    }
    __shared__ int wave_cnt[FG_ROWS / 64];
    __shared__ int wave_pre[FG_ROWS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int i = b * FG_ROWS + tid;

    // own row (clamped load, guard on the flag) and the keypoints it would carry
    pm_match best;
    const bool keep = fg_keep<MODE>(knn, i < nq ? i : nq - 1, k, prm, best) && i < nq;
    float2 pa = {0.f, 0.f}, pb = {0.f, 0.f};
    if (kp1) {
        pa = *reinterpret_cast<const float2*>(kp1 + 2 * static_cast<size_t>(best.queryIdx));
        pb = *reinterpret_cast<const float2*>(kp2 + 2 * static_cast<size_t>(best.trainIdx >= 0 ? best.trainIdx : 0));
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int mine = 0;
#pragma unroll
    for (int w = 0; w < FG_ROWS / 64; ++w) mine += wave_cnt[w];
    if (tid == 0)
        __hip_atomic_store(&blk_counts[b], (epoch << 10) | static_cast<unsigned>(mine), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);

    // counts of the blocks before this one: lane t polls block t, t+256, ...
    int before = 0;
    for (int pb_ = tid; pb_ < b; pb_ += FG_ROWS) {
        unsigned v = 0u, spins = 0u;
        for (;;) {
            v = __hip_atomic_load(&blk_counts[pb_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 10) == epoch || ++spins > FG_SPIN_LIMIT) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if ((v >> 10) == epoch) {
            before += static_cast<int>(v & 1023u);
        } else {                                   // gave up: count that block's survivors directly
            pm_match tmp;
            for (int r = 0; r < FG_ROWS; ++r) {
                const int row = pb_ * FG_ROWS + r;
                before += fg_keep<MODE>(knn, row, k, prm, tmp) ? 1 : 0;   // row < b*FG_ROWS <= nq
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if (lane == 0) wave_pre[wave] = before;
    __syncthreads();
    int off = __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
    for (int w = 0; w < FG_ROWS / 64; ++w) {
        off += wave_pre[w];
        if (w < wave) off += wave_cnt[w];
    }
    if (keep) {
        good[off] = best;
        if (kp1) {
            *reinterpret_cast<float2*>(xy1 + 2 * static_cast<size_t>(off)) = pa;
            *reinterpret_cast<float2*>(xy2 + 2 * static_cast<size_t>(off)) = pb;
        }
    }
    if (b == static_cast<int>(gridDim.x) - 1 && tid == 0) {
        int tot = mine;
        for (int w = 0; w < FG_ROWS / 64; ++w) tot += wave_pre[w];
        *n_out = tot;
        if (MODE == FG_MIDPOINT && minmax_out) { minmax_out[0] = lo_d; minmax_out[1] = hi_d; }
    }
}

// Concatenates `parts` padded point blocks (each `stride` points, counts[p] valid) into one
// contiguous array in part order; used after the all-gather of the query-row-sharded matcher.
__global__ __launch_bounds__(256) void concat_points(const float* __restrict__ xy1_parts,
                                                     const float* __restrict__ xy2_parts,
                                                     const int* __restrict__ counts, int parts, int stride,
                                                     float* __restrict__ xy1, float* __restrict__ xy2,
                                                     int* __restrict__ n_out)
{
    const int p = blockIdx.y;
    int off = 0, total = 0;
    for (int r = 0; r < parts; ++r) {
        int c = counts[r];
        c = c < 0 ? 0 : (c > stride ? stride : c);
        if (r < p) off += c;
        total += c;
    }
    int cp = counts[p];
    cp = cp < 0 ? 0 : (cp > stride ? stride : cp);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < cp) {
        const size_t s = 2 * (static_cast<size_t>(p) * stride + i), d = 2 * static_cast<size_t>(off + i);
        xy1[d] = xy1_parts[s]; xy1[d + 1] = xy1_parts[s + 1];
        xy2[d] = xy2_parts[s]; xy2[d + 1] = xy2_parts[s + 1];
    }
    if (p == 0 && i == 0) *n_out = total;
}

}  // namespace

namespace {

int fg_prepare(pm_ctx* ctx, int nq, int* nblk)
{
    *nblk = (nq + FG_ROWS - 1) / FG_ROWS;
    PM_REFUSE_CAPTURE(ctx);
    PM_REQUIRE(*nblk <= FG_MAX_BLOCKS, PM_E_UNSUPPORTED, "more than 1M query rows per compaction call");
    if (!ctx->fg_counts) {
        // [FG_MAX_BLOCKS] survivor counts + 2 x 64-bit min/max words (8-byte aligned: FG_MAX_BLOCKS is even)
        PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->fg_counts), sizeof(unsigned) * (FG_MAX_BLOCKS + 4)));
        PM_HIP_CHECK(hipMemsetAsync(ctx->fg_counts, 0, sizeof(unsigned) * (FG_MAX_BLOCKS + 4), ctx->stream));
        ctx->fg_epoch = 0;
    }
    if (++ctx->fg_epoch >= (1u << 22)) {          // 22-bit tag: restart
        PM_HIP_CHECK(hipMemsetAsync(ctx->fg_counts, 0, sizeof(unsigned) * (FG_MAX_BLOCKS + 4), ctx->stream));
        ctx->fg_epoch = 1;
    }
    return PM_OK;
}

}  // namespace

extern "C" int pm_filter_ratio_gather_dev(pm_ctx* ctx, const pm_match* d_knn, int nq, int k, float ratio,
                                          const float* d_kp1_xy, const float* d_kp2_xy, pm_match* d_good,
                                          float* d_xy1, float* d_xy2, int32_t* d_n_good)
{
    PM_REQUIRE(ctx != nullptr && d_n_good != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(nq >= 0 && k >= 2, PM_E_INVALID, "ratio test needs k >= 2");
    PM_REQUIRE(nq == 0 || (d_knn && d_good), PM_E_INVALID, "null match buffers");
    PM_REQUIRE((d_kp1_xy == nullptr) == (d_kp2_xy == nullptr), PM_E_INVALID, "give both keypoint arrays or none");
    PM_REQUIRE(d_kp1_xy == nullptr || (d_xy1 && d_xy2), PM_E_INVALID, "null point outputs");
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    pm::ScopedKernelTime t(ctx, "filter_gather");
    if (nq == 0) {
        PM_HIP_CHECK(hipMemsetAsync(d_n_good, 0, sizeof(int32_t), ctx->stream));
        return PM_OK;
    }
    int nblk = 0;
    const int rc = fg_prepare(ctx, nq, &nblk);
    if (rc != PM_OK) return rc;
    hipLaunchKernelGGL(filter_ratio_gather<FG_RATIO>, dim3(nblk), dim3(FG_ROWS), 0, ctx->stream, d_knn, nq, k, ratio,
                       d_kp1_xy, d_kp2_xy, d_good, d_xy1, d_xy2, d_n_good, ctx->fg_counts, ctx->fg_epoch, nullptr, nullptr);
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

// The reference's own filter (main.cpp:49-69) in device-resident form: min/max scan, then the same
// stable compaction with the predicate (double)d < min + (max - min)/2.
extern "C" int pm_filter_midpoint_gather_dev(pm_ctx* ctx, const pm_match* d_m, int n, int k, const float* d_kp1_xy,
                                             const float* d_kp2_xy, pm_match* d_good, float* d_xy1, float* d_xy2,
                                             int32_t* d_n_good, double* d_minmax)
{
    PM_REQUIRE(ctx != nullptr && d_n_good != nullptr, PM_E_INVALID, "null argument");
    PM_REQUIRE(n >= 0 && k >= 1, PM_E_INVALID, "need n >= 0, k >= 1");
    PM_REQUIRE(n == 0 || (d_m && d_good), PM_E_INVALID, "null match buffers");
    PM_REQUIRE((d_kp1_xy == nullptr) == (d_kp2_xy == nullptr), PM_E_INVALID, "give both keypoint arrays or none");
    PM_REQUIRE(d_kp1_xy == nullptr || (d_xy1 && d_xy2), PM_E_INVALID, "null point outputs");
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    pm::ScopedKernelTime t(ctx, "filter_gather");
    if (n == 0) {
        PM_HIP_CHECK(hipMemsetAsync(d_n_good, 0, sizeof(int32_t), ctx->stream));
        if (d_minmax) {
            const double init[2] = {1.0, 0.0};
            PM_HIP_CHECK(hipMemcpyAsync(d_minmax, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
            PM_HIP_CHECK(hipStreamSynchronize(ctx->stream));       // `init` lives on this stack frame
        }
        return PM_OK;
    }
    int nblk = 0;
    const int rc = fg_prepare(ctx, n, &nblk);
    if (rc != PM_OK) return rc;
    unsigned long long* mm = reinterpret_cast<unsigned long long*>(ctx->fg_counts + FG_MAX_BLOCKS);
    hipLaunchKernelGGL(midpoint_minmax, dim3(nblk), dim3(FG_ROWS), 0, ctx->stream, d_m, n, k, mm, ctx->fg_epoch);
    PM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(filter_ratio_gather<FG_MIDPOINT>, dim3(nblk), dim3(FG_ROWS), 0, ctx->stream, d_m, n, k, 0.f,
                       d_kp1_xy, d_kp2_xy, d_good, d_xy1, d_xy2, d_n_good, ctx->fg_counts, ctx->fg_epoch, mm, d_minmax);
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}

extern "C" int pm_concat_points_dev(pm_ctx* ctx, const float* d_xy1_parts, const float* d_xy2_parts,
                                    const int32_t* d_counts, int parts, int stride, float* d_xy1, float* d_xy2,
                                    int32_t* d_n_total)
{
    PM_REQUIRE(ctx && d_xy1_parts && d_xy2_parts && d_counts && d_xy1 && d_xy2 && d_n_total, PM_E_INVALID,
               "null argument");
    PM_REQUIRE(parts >= 1 && parts <= 65535 && stride >= 1, PM_E_INVALID, "bad parts/stride");
    PM_HIP_CHECK(hipSetDevice(ctx->device));
    pm::ScopedKernelTime t(ctx, "concat_points");
    hipLaunchKernelGGL(concat_points, dim3((stride + 255) / 256, parts), dim3(256), 0, ctx->stream, d_xy1_parts,
                       d_xy2_parts, d_counts, parts, stride, d_xy1, d_xy2, d_n_total);
    PM_HIP_CHECK(hipGetLastError());
    return PM_OK;
}
