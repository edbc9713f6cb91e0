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

__device__ __forceinline__ bool fg_keep(const pm_match* __restrict__ knn, int i, int k, float ratio, pm_match& best)
{
    const uint4 a = *reinterpret_cast<const uint4*>(knn + static_cast<size_t>(i) * k);
    const uint4 b = *reinterpret_cast<const uint4*>(knn + static_cast<size_t>(i) * k + 1);
    best.queryIdx = static_cast<int>(a.x); best.trainIdx = static_cast<int>(a.y);
    best.imgIdx = static_cast<int>(a.z);   best.distance = __uint_as_float(a.w);
    const float rhs = ratio * __uint_as_float(b.w);
    return static_cast<int>(a.y) >= 0 && static_cast<int>(b.y) >= 0 && best.distance < rhs;
}

__global__ __launch_bounds__(FG_ROWS) void filter_ratio_gather(const pm_match* __restrict__ knn, int nq, int k,
                                                               float ratio, const float* __restrict__ kp1,
                                                               const float* __restrict__ kp2,
                                                               pm_match* __restrict__ good, float* __restrict__ xy1,
                                                               float* __restrict__ xy2, int* __restrict__ n_out,
                                                               unsigned* __restrict__ blk_counts, unsigned epoch)
{
    __shared__ int wave_cnt[FG_ROWS / 64];
    __shared__ int wave_pre[FG_ROWS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int i = b * FG_ROWS + tid;

    // own row (clamped load, guard on the flag) and the keypoints it would carry
    pm_match best;
    const bool keep = fg_keep(knn, i < nq ? i : nq - 1, k, ratio, best) && i < nq;
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
                before += fg_keep(knn, row, k, ratio, tmp) ? 1 : 0;       // row < b*FG_ROWS <= nq
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
    const int nblk = (nq + FG_ROWS - 1) / FG_ROWS;
    PM_REQUIRE(nblk <= FG_MAX_BLOCKS, PM_E_UNSUPPORTED, "more than 1M query rows per compaction call");
    if (!ctx->fg_counts) {
        PM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->fg_counts), sizeof(unsigned) * FG_MAX_BLOCKS));
        PM_HIP_CHECK(hipMemsetAsync(ctx->fg_counts, 0, sizeof(unsigned) * FG_MAX_BLOCKS, ctx->stream));
        ctx->fg_epoch = 0;
    }
    if (++ctx->fg_epoch >= (1u << 22)) {          // 22-bit tag: restart
        PM_HIP_CHECK(hipMemsetAsync(ctx->fg_counts, 0, sizeof(unsigned) * FG_MAX_BLOCKS, ctx->stream));
        ctx->fg_epoch = 1;
    }
    hipLaunchKernelGGL(filter_ratio_gather, dim3(nblk), dim3(FG_ROWS), 0, ctx->stream, d_knn, nq, k, ratio, d_kp1_xy,
                       d_kp2_xy, d_good, d_xy1, d_xy2, d_n_good, ctx->fg_counts, ctx->fg_epoch);
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
