// knn_coarse.hip — the product instantiation of the MFMA coarse passes (kernel bodies: knn_coarse_kernels.hpp).
// Only the AblNone policy is instantiated here, so libpm_hip.so carries no ablation code.
#include "knn_coarse_kernels.hpp"

namespace pm_knn {

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
    return launch_rows288<RouteI8, AblNone>(ctx, "knn_hamming_mfma_i8", Qe, Te, nullptr, nq, nq_pad, nt, splits, tiles_per_split, 0u,
                                   cval, slots, nullptr, 0u, 0);
}

int launch_coarse_u8(pm_ctx* ctx, const void* Q8, const void* T8, const int* seeds, int nq, int nq_pad, int nt, int splits,
                     int tiles_per_split, int* cval, int slots, int group_rows, int form)
{
    return coarse_u8_dispatch<AblNone>(ctx, Q8, T8, seeds, nq, nq_pad, nt, splits, tiles_per_split, cval, slots, group_rows, form);
}

int launch_coarse_f16s(pm_ctx* ctx, const _Float16* Qh, const _Float16* Th, const float* seeds, int nq, int nq_pad, int nt,
                       int splits, int tiles_per_split, unsigned keep_mask, float* cval, int slots)
{
    return launch_rows288<RouteF16S, AblNone>(ctx, "knn_l2_mfma_f16s", Qh, Th, seeds, nq, nq_pad, nt, splits, tiles_per_split,
                                         keep_mask, cval, slots, nullptr, 0u, 0);
}

}  // namespace pm_knn
