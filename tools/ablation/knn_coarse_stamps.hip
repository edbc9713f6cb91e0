// knn_coarse_stamps.hip — diagnostic build of the coarse kNN kernels with in-kernel shader-clock stamps (MI355X guide,
// "In-kernel stamps"): where a workgroup of the ring kernel (csrc/knn_coarse_kernels.hpp: knn_mfma_ring) spends its
// cycles.  This translation unit replaces csrc/knn_coarse.hip in a library of its own (tools/build_stamps.sh); it
// instantiates the same kernel bodies with a stamping policy and exports the read-back entry point
// tools/prof_knn_stamps.py uses.  Never part of libpm_hip.so; the stamps go to a buffer nothing else reads.
#include "knn_coarse_kernels.hpp"

// [workgroup][slot]: 0 entry, 1 requests issued, 2 + t tile t's barrier passed (t < 14), 16 sweep done, 17 lists stored,
// 20 / 21 s_memrealtime at entry / exit (100 MHz, chip-wide), 22 XCC id
__device__ unsigned long long g_knn_stamps[4096 * 24];

namespace pm_knn {

#ifndef ABL_NO_EPI
#define ABL_NO_EPI 0
#endif
#ifndef ABL_NO_STAGE
#define ABL_NO_STAGE 0
#endif
#ifndef ABL_NO_BARRIER
#define ABL_NO_BARRIER 0
#endif
#ifndef ABL_NO_LDSREAD
#define ABL_NO_LDSREAD 0
#endif
// (with -DABL_*: timing-only variants, outputs wrong — what a tile costs without its selection / LDS operand reads / ...)
struct AblStamp : Abl<ABL_NO_EPI != 0, ABL_NO_STAGE != 0, ABL_NO_BARRIER != 0, ABL_NO_LDSREAD != 0> {
    static __device__ __forceinline__ void stamp(int i)
    {
        const unsigned wgid = blockIdx.y * gridDim.x + blockIdx.x;
        if (threadIdx.x == 0 && wgid < 4096) {
            g_knn_stamps[wgid * 24 + i] = __builtin_amdgcn_s_memtime();
            if (i == 0) {
                g_knn_stamps[wgid * 24 + 20] = __builtin_amdgcn_s_memrealtime();
                g_knn_stamps[wgid * 24 + 22] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) ;   // HW_REG_XCC_ID[3:0]
            }
            if (i == 17) g_knn_stamps[wgid * 24 + 21] = __builtin_amdgcn_s_memrealtime();
        }
    }
};

int launch_coarse_f32(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim, const float* tnorm,
                      int splits, int tiles_per_split, unsigned keep_mask, float* cval, int slots,
                      const unsigned long long* stats, unsigned epoch, int only_if_ineligible)
{
    return coarse_f32_dispatch<AblNone>(ctx, dq, nq, dt, nt, dim, tnorm, splits, tiles_per_split, keep_mask, cval, slots,
                                        stats, epoch, only_if_ineligible);
}

int launch_coarse_f16(pm_ctx* ctx, const _Float16* Qh, const _Float16* Th, int nq, int nq_pad, int nt, int splits,
                      int tiles_per_split, unsigned keep_mask, float* cval, int slots,
                      const unsigned long long* stats, unsigned epoch, int mode, int dp)
{
    if (dp == 256)
        return launch_rows288<RouteF16T<256>, AblNone>(ctx, "knn_l2_mfma_f16", Qh, Th, nullptr, nq, nq_pad, nt, splits, tiles_per_split,
                                                  keep_mask, cval, slots, stats, epoch, mode);
    return launch_rows288<RouteF16, AblNone>(ctx, "knn_l2_mfma_f16", Qh, Th, nullptr, nq, nq_pad, nt, splits, tiles_per_split, keep_mask,
                                        cval, slots, stats, epoch, mode);
}

int launch_coarse_i8(pm_ctx* ctx, const void* Qe, const void* Te, int nq, int nq_pad, int nt, int splits,
                     int tiles_per_split, int* cval, int slots)
{
    return launch_rows288<RouteI8, AblNone>(ctx, "knn_hamming_mfma_i8", Qe, Te, nullptr, nq, nq_pad, nt, splits, tiles_per_split,
                                            0u, cval, slots, nullptr, 0u, 0);
}

int launch_coarse_u8(pm_ctx* ctx, const void* Q8, const void* T8, const int* seeds, int nq, int nq_pad, int nt, int splits,
                     int tiles_per_split, int* cval, int slots, int group_rows, int form)
{
    return coarse_u8_dispatch<AblStamp>(ctx, Q8, T8, seeds, nq, nq_pad, nt, splits, tiles_per_split, cval, slots, group_rows, form);
}

int launch_coarse_f16s(pm_ctx* ctx, const _Float16* Qh, const _Float16* Th, const float* seeds, int nq, int nq_pad, int nt,
                       int splits, int tiles_per_split, unsigned keep_mask, float* cval, int slots)
{
    return launch_rows288<RouteF16S, AblNone>(ctx, "knn_l2_mfma_f16s", Qh, Th, seeds, nq, nq_pad, nt, splits, tiles_per_split,
                                              keep_mask, cval, slots, nullptr, 0u, 0);
}

}  // namespace pm_knn

extern "C" int pm_debug_knn_stamps(unsigned long long* out, int n_words)
{
    PM_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knn_stamps), sizeof(unsigned long long) * n_words, 0, hipMemcpyDeviceToHost));
    return PM_OK;
}
