// Micro-benchmark: cycles per v_pk_fma_f32 / v_fma_f32 per SIMD on gfx950, by operand form and waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_fma_rate.hip -o /tmp/pk_fma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed)
{
    f32x2 a[8], b = {seed, seed + 1.f}, c = {seed * 0.5f, 0.25f};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f32x2{seed + i, seed - i};
    unsigned long long q = (static_cast<unsigned long long>(__float_as_uint(1.0001f)) << 32) | __float_as_uint(0.9999f);
    q = __builtin_amdgcn_readfirstlane(static_cast<int>(q)) | (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(q >> 32))) << 32);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                       // 3 VGPR pairs
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a[i]) : "v"(b), "v"(c));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(q), "v"(c));   // SGPR splat
            if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(a[i]) : "s"(q), "v"(c));   // SGPR twice
            if (MODE == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i][0]) : "v"(b[0]), "v"(c[0]));                 // scalar fma
            if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(q), "v"(c));                       // SGPR pair, no op_sel
            if (MODE == 6) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int wg_per_cu)
{
    const int iters = 2000, blocks = 256 * wg_per_cu;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    k<MODE><<<blocks, 256>>>(out, cyc, iters, 1.0f);
    k<MODE><<<blocks, 256>>>(out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += v;
    const double per_wave = sum / blocks / (iters * 8.0);          // cycles per instruction as one wave sees it
    printf("%-34s %d wave(s)/SIMD: %.2f cycles per instruction per wave = %.2f per SIMD\n", name, wg_per_cu, per_wave, per_wave / wg_per_cu);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_pk_fma_f32 3 x VGPR pair", w);
        run<1>("v_pk_fma_f32 VGPR, op_sel splat", w);
        run<2>("v_pk_fma_f32 SGPR splat + 2 VGPR", w);
        run<3>("v_pk_fma_f32 SGPR twice + 1 VGPR", w);
        run<5>("v_pk_fma_f32 SGPR pair, no op_sel", w);
        run<4>("v_fma_f32 3 x VGPR", w);
        run<6>("v_pk_mul_f32 2 x VGPR pair", w);
    }
    return 0;
}
