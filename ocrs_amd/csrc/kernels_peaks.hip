// Device micro-benchmarks behind ocrs_device_measure_peaks (bench.py reports them next to the
// nominal peaks of MI355X_MICROARCH.md; SURVEY.md §8d asks for the rates this box sustains):
//   * fp32 MFMA: register-only v_mfma_f32_32x32x2_f32 loop, four independent accumulators per
//     wave, two waves per SIMD on every CU, run long enough (>= 50 ms) to see the sustained clock;
//   * HBM: float4 grid-stride copy of a buffer much larger than L2 + Infinity Cache.
#include "common.hpp"
#include "kernels.hpp"

namespace ocrs {
namespace k {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
peak_mfma_f32_kernel(float* __restrict__ out, int iters) {
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    float a = (float)(threadIdx.x & 7) * 0.125f, b = 1.0f + (float)(threadIdx.x & 3) * 0.25f;
    for (int i = 0; i < iters; i++) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    if (s == 12345.678f) out[0] = s;  // keep the loop alive without a store on the hot path
}

__global__ void __launch_bounds__(256)
peak_copy_kernel(const f32x4v* __restrict__ src, f32x4v* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {  // four independent 16-byte loads in flight per lane
        const f32x4v a = __builtin_nontemporal_load(src + i);
        const f32x4v b = __builtin_nontemporal_load(src + i + stride);
        const f32x4v c = __builtin_nontemporal_load(src + i + 2 * stride);
        const f32x4v d = __builtin_nontemporal_load(src + i + 3 * stride);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride);
        __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

void measure_peaks(double* mfma_tflops, double* copy_gbps) {
    hipStream_t s;
    OCRS_HIP(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    OCRS_HIP(hipEventCreate(&e0));
    OCRS_HIP(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    int dev = 0;
    OCRS_HIP(hipGetDevice(&dev));
    OCRS_HIP(hipGetDeviceProperties(&prop, dev));
    float ms = 0.f;
    {
        float* d_out = nullptr;
        OCRS_HIP(hipMalloc(&d_out, 256));
        const int blocks = prop.multiProcessorCount * 2;  // 4 waves per block -> 2 waves per SIMD
        const int iters = 1 << 16;
        peak_mfma_f32_kernel<<<blocks, 256, 0, s>>>(d_out, 1 << 12);  // warm-up / clock ramp
        OCRS_HIP(hipEventRecord(e0, s));
        peak_mfma_f32_kernel<<<blocks, 256, 0, s>>>(d_out, iters);
        OCRS_HIP(hipEventRecord(e1, s));
        OCRS_HIP(hipEventSynchronize(e1));
        OCRS_HIP(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)blocks * 4.0 * iters * 4.0 * (2.0 * 32 * 32 * 2);
        *mfma_tflops = flops / (ms * 1e-3) / 1e12;
        OCRS_HIP(hipFree(d_out));
    }
    {
        const int64_t bytes = (int64_t)2 << 30;
        f32x4v *a = nullptr, *b = nullptr;
        OCRS_HIP(hipMalloc(&a, bytes));
        OCRS_HIP(hipMalloc(&b, bytes));
        OCRS_HIP(hipMemsetAsync(a, 0, bytes, s));
        const int64_t n = bytes / 16;
        const int blocks = prop.multiProcessorCount * 8;
        peak_copy_kernel<<<blocks, 256, 0, s>>>(a, b, n);
        OCRS_HIP(hipEventRecord(e0, s));
        const int reps = 8;
        for (int r = 0; r < reps; r++) peak_copy_kernel<<<blocks, 256, 0, s>>>(a, b, n);
        OCRS_HIP(hipEventRecord(e1, s));
        OCRS_HIP(hipEventSynchronize(e1));
        OCRS_HIP(hipEventElapsedTime(&ms, e0, e1));
        *copy_gbps = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
        OCRS_HIP(hipFree(a));
        OCRS_HIP(hipFree(b));
    }
    OCRS_HIP(hipEventDestroy(e0));
    OCRS_HIP(hipEventDestroy(e1));
    OCRS_HIP(hipStreamDestroy(s));
}

}  // namespace k
}  // namespace ocrs
