// filter_gather.hip — device-resident form of the glue between the two hot kernels:
// ratio test in the slot of main.cpp:49-69, index extraction main.cpp:77-78 and
// KeyPoint::convert main.cpp:89-91, fused into one stable compaction so that a matched batch
// never leaves HBM between the matcher and the RANSAC kernels.  Survivors keep query order
// (the reference's push_back order, main.cpp:63-68).  Same predicate as pm_filter_ratio:
// second neighbour present and d1 < ratio * d2 (float multiply, strict).
#include "pm_common.hpp"

namespace {

// One workgroup of 1024 threads; a thread owns FG_ITEMS consecutive rows (a contiguous 256-byte
// read at k = 2), so 8192 rows need a single scan: ballot-free local prefix + one wave scan + a
// 16-entry scan across waves.  Longer inputs loop with a running base.
constexpr int FG_ITEMS = 8;

__global__ __launch_bounds__(1024) void filter_ratio_gather(const pm_match* __restrict__ knn, int nq, int k,
                                                            float ratio, const float* __restrict__ kp1,
                                                            const float* __restrict__ kp2,
                                                            pm_match* __restrict__ good, float* __restrict__ xy1,
                                                            float* __restrict__ xy2, int* __restrict__ n_out)
{
    __shared__ int wave_tot[16];
    __shared__ int base_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int start = 0; start < nq; start += 1024 * FG_ITEMS) {
        const int i0 = start + tid * FG_ITEMS;
        pm_match best[FG_ITEMS];
        unsigned keep = 0u;
        // unconditional loads from clamped rows (a load under a per-element branch makes hipcc wait
        // for each one in turn); the row guard is applied to the flag instead
        pm_match second[FG_ITEMS];
#pragma unroll
        for (int e = 0; e < FG_ITEMS; ++e) {
            const int i = i0 + e < nq ? i0 + e : nq - 1;
            best[e] = knn[static_cast<size_t>(i) * k];
            second[e] = knn[static_cast<size_t>(i) * k + 1];
        }
        float2 pa[FG_ITEMS], pb[FG_ITEMS];
#pragma unroll
        for (int e = 0; e < FG_ITEMS; ++e) {
            const float rhs = ratio * second[e].distance;
            const bool ok = i0 + e < nq && best[e].trainIdx >= 0 && second[e].trainIdx >= 0 && best[e].distance < rhs;
            if (ok) keep |= 1u << e;
            if (kp1) {
                const int ti = best[e].trainIdx >= 0 ? best[e].trainIdx : 0;
                pa[e] = *reinterpret_cast<const float2*>(kp1 + 2 * static_cast<size_t>(best[e].queryIdx));
                pb[e] = *reinterpret_cast<const float2*>(kp2 + 2 * static_cast<size_t>(ti));
            }
        }
        const int mine = __popc(keep);
        int incl = mine;                                    // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int off = base_sh + incl - mine;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
#pragma unroll
        for (int e = 0; e < FG_ITEMS; ++e)
            if (keep & (1u << e)) {
                good[off] = best[e];
                if (kp1) {
                    *reinterpret_cast<float2*>(xy1 + 2 * static_cast<size_t>(off)) = pa[e];
                    *reinterpret_cast<float2*>(xy2 + 2 * static_cast<size_t>(off)) = pb[e];
                }
                ++off;
            }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wave_tot[w];
            base_sh += tot;
        }
        __syncthreads();
    }
    if (tid == 0) *n_out = base_sh;
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
    hipLaunchKernelGGL(filter_ratio_gather, dim3(1), dim3(1024), 0, ctx->stream, d_knn, nq, k, ratio, d_kp1_xy,
                       d_kp2_xy, d_good, d_xy1, d_xy2, d_n_good);
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
