// ransac_internal.hpp — entry points of ransac_fused.hip used by the C-ABI functions in ransac.hip (host-pointer
// and device-resident single-shard runs) and by the multi-GPU driver (mgpu.cpp).
#pragma once
#include "ransac_core.hpp"

namespace pm_ransac {

constexpr int RF_HB_MAX = 128;          // hypothesis ids per workgroup, at most
int fused_hb(const pm_ctx* ctx, long long nh);
size_t fused_scratch_bytes(const pm_ctx* ctx, const pm_ransac_params* p);
int fused_launch(pm_ctx* ctx, const pm_points_view& v, const pm_ransac_params* p, int shard, pm_ransac_record* d_rec,
                 unsigned long long* d_key, double* d_F, uint8_t* d_mask, int mask_len, int* d_ninl, FinalOut** fo_out);

}  // namespace pm_ransac
