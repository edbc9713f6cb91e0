// ransac_fused.hip — the product instantiation of the one-launch RANSAC kernel (body: ransac_fused_kernels.hpp).
// Only the NoDiag policy is instantiated here: libpm_hip.so carries no stamp buffers and no diagnostic entry points.
#include "ransac_fused_kernels.hpp"

namespace pm_ransac {

int fused_launch(pm_ctx* ctx, const pm_points_view& v, const pm_ransac_params* p, int shard, pm_ransac_record* d_rec,
                 unsigned long long* d_key, double* d_F, uint8_t* d_mask, int mask_len, int* d_ninl, FinalOut** fo_out)
{
    return fused_launch_t<NoDiag>(ctx, v, p, shard, d_rec, d_key, d_F, d_mask, mask_len, d_ninl, fo_out);
}

}  // namespace pm_ransac
