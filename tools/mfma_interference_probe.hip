// Stand-alone attempt to reproduce the interference of profiles/r5_relaxed_concurrency.txt outside the library: a dense MFMA kernel
// on one stream, a VALU kernel on another that checks every repetition of its arithmetic against its own first pass, per lane quarter.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/mfma_interference_probe.hip && /tmp/probe <aggressor 0..3> <victim 0..3>
// Result on MI355X (round 5): no mismatch in any combination — the effect needs more of the real kernels than this has.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: 32x32x16 bf16, 1: 16x16x32 bf16, 2: 32x32x2 f32
__global__ void __launch_bounds__(256) aggressor(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const float fa = 0.001f * lane, fb = 0.003f * lane;
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
        } else if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    for (int i = 0; i < 4; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WHAT>   // 0: IEEE division, 1: fma chain, 2: v_rcp_f32, 3: integer division
__global__ void __launch_bounds__(256) victim(unsigned* bad_by_quarter, int reps, const float* inputs) {
    const int tid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    unsigned bad = 0;
    float ref[8];
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            volatile float xin = inputs[(tid * 8 + j) & 65535];
            const float x = xin, y = inputs[(tid * 8 + j + 17) & 65535] + 1.5f;
            float v;
            if (WHAT == 0) v = x / y;
            else if (WHAT == 1) v = fmaf(x, y, 0.25f) * y + x;
            else if (WHAT == 2) v = __builtin_amdgcn_rcpf(y) * x;
            else v = (float)((int)(x * 1000.0f) / ((int)(y * 10.0f) | 1));
            if (r == 0) ref[j] = v;
            else bad += __float_as_uint(v) != __float_as_uint(ref[j]);
        }
    }
    if (bad) atomicAdd(&bad_by_quarter[lane >> 4], bad);
}

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0, what = argc > 2 ? atoi(argv[2]) : 0;
    hipStream_t sa, sv;
    (void)hipStreamCreate(&sa); (void)hipStreamCreate(&sv);
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    unsigned* bad; (void)hipMalloc(&bad, 16); (void)hipMemset(bad, 0, 16);
    std::vector<float> h(65536);
    for (int i = 0; i < 65536; i++) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f;
    float* in; (void)hipMalloc(&in, 65536 * 4); (void)hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    for (int round = 0; round < 4; round++) {   // two long-running aggressor blocks per CU, many short victim launches beside them
        if (kind == 0) aggressor<0><<<512, 256, 0, sa>>>(out, 400000);
        else if (kind == 1) aggressor<1><<<512, 256, 0, sa>>>(out, 800000);
        else if (kind == 2) aggressor<2><<<512, 256, 0, sa>>>(out, 200000);
        for (int k = 0; k < 200; k++) {
            if (what == 0) victim<0><<<512, 256, 0, sv>>>(bad, 200, in);
            else if (what == 1) victim<1><<<512, 256, 0, sv>>>(bad, 200, in);
            else if (what == 2) victim<2><<<512, 256, 0, sv>>>(bad, 200, in);
            else victim<3><<<512, 256, 0, sv>>>(bad, 200, in);
        }
    }
    (void)hipDeviceSynchronize();
    unsigned hb[4]; (void)hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
    printf("aggressor %d (0: 32x32x16 bf16, 1: 16x16x32 bf16, 2: 32x32x2 f32, 3: none), victim %d (0 div, 1 fma, 2 rcp, 3 int div): "
           "mismatching results by lane quarter %u %u %u %u\n", kind, what, hb[0], hb[1], hb[2], hb[3]);
    return 0;
}
