// Probe for the streaming detection blocks: VALU FMA issue rate (plain vs packed), the lane mapping
// and rounding of v_mfma_f32_4x4x1_16B_f32, and the DPP wave shifts.   hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o valu_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters, float w) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    const float x = out[threadIdx.x & 7];
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(w));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 acc = {a[i], a[i + 1]};
                f32x2 xx = {x, x};
                f32x2 ww = {w, w};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(xx), "v"(ww));
                a[i] = acc[0]; a[i + 1] = acc[1];
            }
        } else if constexpr (MODE == 2) {   // 4 independent 4x4x1 accumulators
            f32x4 c0 = {a[0], a[1], a[2], a[3]}, c1 = {a[4], a[5], a[6], a[7]}, c2 = {a[8], a[9], a[10], a[11]}, c3 = {a[12], a[13], a[14], a[15]};
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, w, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, w, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, w, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, w, c3, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) { a[i] = c0[i]; a[4 + i] = c1[i]; a[8 + i] = c2[i]; a[12 + i] = c3[i]; }
        } else if constexpr (MODE == 3) {   // DPP shift + fma
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[(i + 1) & 15]), 0x138, 0xF, 0xF, true));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(l), "s"(w));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void map_kernel(const float* a, const float* b, const float* c, float* d) {
    const int lane = threadIdx.x;
    f32x4 acc = {c[lane * 4], c[lane * 4 + 1], c[lane * 4 + 2], c[lane * 4 + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) d[lane * 4 + i] = acc[i];
}

__global__ void dpp_kernel(const int* in, int* shr, int* shl, int* swp) {
    const int v = in[threadIdx.x];
    shr[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, true);   // wave_shr:1
    shl[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, true);   // wave_shl:1
    swp[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
}

template <int MODE>
static int run_rate(const char* name, double flop_per_lane_iter) {
    float* d;
    const int blocks = 256 * 8, iters = 20000;
    CK(hipMalloc(&d, blocks * 256 * sizeof(float)));
    CK(hipMemset(d, 0, blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<MODE><<<blocks, 256>>>(d, 100, 1.0001f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    rate_kernel<MODE><<<blocks, 256>>>(d, iters, 1.0001f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)blocks * 256 * iters * flop_per_lane_iter;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
    CK(hipFree(d));
    return 0;
}

int main() {
    if (run_rate<0>("v_fma_f32 x16", 32)) return 1;
    if (run_rate<1>("v_pk_fma_f32 x8", 32)) return 1;
    if (run_rate<2>("v_mfma_f32_4x4x1 x4", 32)) return 1;
    if (run_rate<3>("dpp wave_shr + v_fma x16", 32)) return 1;
    // ---- 4x4x1 mapping and rounding
    std::vector<float> a(64), b(64), c(256), d(256);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) % 200001 - 100000) / 37000.0f; };
    for (auto& v : a) v = rnd();
    for (auto& v : b) v = rnd();
    for (auto& v : c) v = rnd();
    a[5] = 1e-39f; b[6] = 0.5f; c[6 * 4 + 1] = 3e-39f;   // a denormal product and addend: block 1, i = 1, j = 2
    float *da, *db, *dc, *dd;
    CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dc, 1024)); CK(hipMalloc(&dd, 1024));
    CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice));
    map_kernel<<<1, 64>>>(da, db, dc, dd);
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int i = 0; i < 4; i++) {
            const int blk = lane / 4;
            const float want = fmaf(a[blk * 4 + i], b[lane], c[lane * 4 + i]);   // D[i][j] in VGPR i of lane 4 blk + j
            if (memcmp(&want, &d[lane * 4 + i], 4) != 0) {
                if (bad < 8) printf("4x4x1 mismatch lane %d i %d: got %.9g want %.9g\n", lane, i, d[lane * 4 + i], want);
                bad++;
            }
        }
    printf("4x4x1: D[vgpr i][lane 4b+j] = fmaf(A[lane 4b+i], B[lane 4b+j], C) : %s (%d mismatches)\n", bad ? "NO" : "yes, bitwise", bad);
    // ---- DPP
    std::vector<int> in(64), o1(64), o2(64), o3(64);
    for (int i = 0; i < 64; i++) in[i] = 100 + i;
    int *di, *d1, *d2, *d3;
    CK(hipMalloc(&di, 256)); CK(hipMalloc(&d1, 256)); CK(hipMalloc(&d2, 256)); CK(hipMalloc(&d3, 256));
    CK(hipMemcpy(di, in.data(), 256, hipMemcpyHostToDevice));
    dpp_kernel<<<1, 64>>>(di, d1, d2, d3);
    CK(hipMemcpy(o1.data(), d1, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(o2.data(), d2, 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o3.data(), d3, 256, hipMemcpyDeviceToHost));
    int ok1 = 1, ok2 = 1, ok3 = 1;
    for (int i = 0; i < 64; i++) {
        ok1 &= o1[i] == (i > 0 ? in[i - 1] : 0);
        ok2 &= o2[i] == (i < 63 ? in[i + 1] : 0);
        ok3 &= o3[i] == in[i ^ 1];
    }
    printf("dpp wave_shr:1 gives lane-1 (0 at lane 0): %s [%d %d %d ... %d]\n", ok1 ? "yes" : "NO", o1[0], o1[1], o1[16], o1[63]);
    printf("dpp wave_shl:1 gives lane+1 (0 at lane 63): %s [%d %d ... %d %d]\n", ok2 ? "yes" : "NO", o2[0], o2[15], o2[62], o2[63]);
    printf("dpp quad_perm [1,0,3,2] gives lane^1: %s\n", ok3 ? "yes" : "NO");
    return 0;
}
