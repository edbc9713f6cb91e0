// knn_coarse_ablation.hip — timing-only variants of the coarse kNN kernels (outputs are WRONG): which part of a tile
// costs what.  This translation unit replaces csrc/knn_coarse.hip in a diagnostic library built by
// tools/build_ablations.sh; it instantiates the same kernel bodies (csrc/knn_coarse_kernels.hpp) with another policy.
// It is never part of libpm_hip.so, and the product sources carry no switch for it.
//
//   -DABL_NO_EPI=1       drop the in-chain selection (accumulators kept live)
//   -DABL_NO_STAGE=1     do not stage the next train tile
//   -DABL_NO_BARRIER=1   no workgroup barrier per tile
//   -DABL_NO_LDSREAD=1   A fragments read once, not per chunk
#include "knn_coarse_kernels.hpp"

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

namespace pm_knn {

typedef Abl<ABL_NO_EPI != 0, ABL_NO_STAGE != 0, ABL_NO_BARRIER != 0, ABL_NO_LDSREAD != 0> Variant;

int launch_coarse_f32(pm_ctx* ctx, const float* dq, int nq, const float* dt, int nt, int dim, const float* tnorm,
                      int splits, int tiles_per_split, unsigned keep_mask, float* cval, int slots,
                      const unsigned long long* stats, unsigned epoch, int only_if_ineligible)
{
    return coarse_f32_dispatch<Variant>(ctx, dq, nq, dt, nt, dim, tnorm, splits, tiles_per_split, keep_mask, cval, slots,
                                        stats, epoch, only_if_ineligible);
}

int launch_coarse_f16(pm_ctx* ctx, const _Float16* Qh, const _Float16* Th, int nq, int nq_pad, int nt, int splits,
                      int tiles_per_split, unsigned keep_mask, float* cval, int slots,
                      const unsigned long long* stats, unsigned epoch, int mode, int dp)
{
    if (dp == 256)
        return launch_rows288<RouteF16T<256>, Variant>(ctx, "knn_l2_mfma_f16", Qh, Th, nullptr, nq, nq_pad, nt, splits, tiles_per_split,
                                                  keep_mask, cval, slots, stats, epoch, mode);
    return launch_rows288<RouteF16, Variant>(ctx, "knn_l2_mfma_f16", Qh, Th, nullptr, nq, nq_pad, nt, splits, tiles_per_split, keep_mask,
                                        cval, slots, stats, epoch, mode);
}

int launch_coarse_i8(pm_ctx* ctx, const void* Qe, const void* Te, int nq, int nq_pad, int nt, int splits,
                     int tiles_per_split, int* cval, int slots)
{
    return launch_rows288<RouteI8, Variant>(ctx, "knn_hamming_mfma_i8", Qe, Te, nullptr, nq, nq_pad, nt, splits, tiles_per_split,
                                            0u, cval, slots, nullptr, 0u, 0);
}

int launch_coarse_u8(pm_ctx* ctx, const void* Q8, const void* T8, const int* seeds, int nq, int nq_pad, int nt, int splits,
                     int tiles_per_split, int* cval, int slots, int group_rows, int form)
{
    return coarse_u8_dispatch<Variant>(ctx, Q8, T8, seeds, nq, nq_pad, nt, splits, tiles_per_split, cval, slots, group_rows, form);
}

int launch_coarse_f16s(pm_ctx* ctx, const _Float16* Qh, const _Float16* Th, const float* seeds, int nq, int nq_pad, int nt,
                       int splits, int tiles_per_split, unsigned keep_mask, float* cval, int slots)
{
    return launch_rows288<RouteF16S, Variant>(ctx, "knn_l2_mfma_f16s", Qh, Th, seeds, nq, nq_pad, nt, splits, tiles_per_split,
                                         keep_mask, cval, slots, nullptr, 0u, 0);
}

}  // namespace pm_knn
