// ransac_fused_stamps.hip — diagnostic launch of the one-launch RANSAC kernel with in-kernel shader-clock stamps
// (MI355X guide, "In-kernel stamps").  This translation unit replaces csrc/ransac_fused.hip in a library of its own
// (tools/build_stamps.sh); it instantiates the same kernel body (csrc/ransac_fused_kernels.hpp) with a stamping
// policy and exports the two read-back entry points tools/prof_ransac_stamps.py uses.  Never part of libpm_hip.so.
//
//   -DRF_SCALAR_FMA=1    experiment: unpacked v_fma_f32 scoring instead of v_pk_fma_f32 (same result bits)
#include "ransac_fused_kernels.hpp"

#ifndef RF_SCALAR_FMA
#define RF_SCALAR_FMA 0
#endif

// stamp buffers of their own that nothing else reads: [workgroup][phase]
__device__ unsigned long long g_rf_stamps[4096 * 12];
__device__ unsigned long long g_rf_solve_stamps[4096 * 8];

namespace pm_ransac {

struct StampDiag {
    static constexpr bool scalar_fma = RF_SCALAR_FMA != 0;
    static __device__ __forceinline__ void phase(int i)
    {
        if (threadIdx.x == 0 && blockIdx.x < 4096) {
            g_rf_stamps[blockIdx.x * 12 + i] = __builtin_amdgcn_s_memtime();
            if (i == 0 || i == 9) g_rf_stamps[blockIdx.x * 12 + 10 + (i ? 1 : 0)] = __builtin_amdgcn_s_memrealtime();
        }
    }
    static __device__ __forceinline__ void solve(int i)
    {
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_rf_solve_stamps[blockIdx.x * 8 + i] = __builtin_amdgcn_s_memtime();
    }
};

int fused_launch(pm_ctx* ctx, const pm_points_view& v, const pm_ransac_params* p, int shard, pm_ransac_record* d_rec,
                 unsigned long long* d_key, double* d_F, uint8_t* d_mask, int mask_len, int* d_ninl, FinalOut** fo_out)
{
    return fused_launch_t<StampDiag>(ctx, v, p, shard, d_rec, d_key, d_F, d_mask, mask_len, d_ninl, fo_out);
}

}  // namespace pm_ransac

extern "C" int pm_debug_rf_stamps(unsigned long long* out, int n_words)
{
    PM_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rf_stamps), sizeof(unsigned long long) * n_words, 0, hipMemcpyDeviceToHost));
    return PM_OK;
}

extern "C" int pm_debug_rf_solve_stamps(unsigned long long* out, int n_words)
{
    PM_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rf_solve_stamps), sizeof(unsigned long long) * n_words, 0, hipMemcpyDeviceToHost));
    return PM_OK;
}
