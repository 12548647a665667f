// Stand-alone reproducer of round 5's co-residency hazard from the TWO REAL KERNELS (DESIGN.md §4.4 "Concurrency"):
//   aggressor  gemm_split_kernel<NP>   (kernels_nn.hip; v_mfma_f32_32x32x16_bf16) — the GRU input projection of a 16-page
//              request: A [R = 45 056][K = 512] fp32, both directions' weights [512][768] cut into bf16 planes (split_weights)
//   victim     crop_lines_kernel       (kernels_lines.hip) — 77 line polygons of a 1024 x 1024 page, 64 rows each
// linked from the product's own sources (tools/build_hazard_repro.sh), launched on two streams of one process.  The victim is
// launched twice on the same inputs into two buffers and a compare kernel counts the words that differ: any difference is
// silent corruption (the kernel is deterministic: --aggressor none must and does report 0).
//   hazard_repro [--aggressor split3|split2|exact|none] [--seconds S] [--lines N] [--split-cus A]
//                [--victim-streams V]
// --split-cus A: aggressor on a stream confined to compute units [0, A), victims on the complementary units
// (hipExtStreamCreateWithCUMask).  --lines N: victim occupancy (64 * N workgroups per launch).  A build with
// -DOCRS_CROP_SETPRIO=n raises the victim's wave priority (s_setprio) — see build_hazard_repro.sh.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <map>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../ocrs_amd/csrc/common.hpp"
#include "../ocrs_amd/csrc/kernels.hpp"
#include "../ocrs_amd/csrc/split_mfma.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void compare_kernel(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* out /* [0] words, [1] first index + 1 */) {
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (a[i] != b[i]) { bad++; atomicMin(out + 1, (unsigned long long)i); }
    if (bad) atomicAdd(out, bad);
}

static hipStream_t make_stream(int lo, int hi, int total) {
    hipStream_t s = nullptr;
    if (lo == 0 && hi == total) { (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); return s; }
    std::vector<uint32_t> mask((size_t)(total + 31) / 32, 0u);
    for (int i = lo; i < hi; i++) mask[(size_t)i / 32] |= 1u << (i % 32);
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { fprintf(stderr, "CU mask refused\n"); exit(2); }
    return s;
}

int main(int argc, char** argv) {
    using namespace ocrs;
    std::string aggressor = "split3";
    double seconds = 10.0;
    int n_lines = 77, split_cus = 0, vstreams = 1, analyse = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--aggressor") aggressor = v;
        else if (k == "--seconds") seconds = atof(v.c_str());
        else if (k == "--lines") n_lines = atoi(v.c_str());
        else if (k == "--split-cus") split_cus = atoi(v.c_str());
        else if (k == "--victim-streams") vstreams = atoi(v.c_str());
        else if (k == "--analyse") analyse = atoi(v.c_str());   // N: for the first N differing twins, say WHERE every wrong word's value belongs
    }
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-0.5f, 0.5f);
    // ---- victim inputs: one page, n_lines horizontal line polygons (two words each: 8 vertices), heights 14..24, widths 500..1000
    const int H = 1024, W = 1024, out_h = 64;
    std::vector<float> page((size_t)H * W);
    for (float& x : page) x = U(rng);
    std::vector<k::LineDesc> desc;
    std::vector<int32_t> poly;
    int64_t out_floats = 0;
    for (int i = 0; i < n_lines; i++) {
        const int h = 14 + (int)(rng() % 11), w = 500 + (int)(rng() % 500), top = 4 + (int)((uint64_t)i * (H - 40) / n_lines), left = 5 + (int)(rng() % 16);
        const int mid = left + w / 2, sl = (int)(rng() % 3) - 1;   // second word one pixel up / down: edges that are not axis-parallel
        const int pts[8][2] = {{top, left}, {top, mid}, {top + sl, mid + 6}, {top + sl, left + w}, {top + h + sl, left + w}, {top + h + sl, mid + 6}, {top + h, mid}, {top + h, left}};
        k::LineDesc d{};
        d.page = 0; d.poly_off = (int32_t)(poly.size() / 2); d.poly_n = 8;
        d.top = top - 1; d.left = left; d.bh = h + 3; d.bw = w;
        d.resized_w = std::min(2400, std::max(10, 64 * w / h)); d.out_w = (d.resized_w + 49) / 50 * 50; d.out_off = out_floats;
        out_floats += (int64_t)out_h * d.out_w;
        for (auto& p : pts) { poly.push_back(p[0]); poly.push_back(p[1]); }
        desc.push_back(d);
    }
    float* d_page; const float** d_pages; int32_t *d_hw, *d_poly; k::LineDesc* d_desc;
    CK(hipMalloc(&d_page, page.size() * 4)); CK(hipMalloc(&d_pages, sizeof(float*))); CK(hipMalloc(&d_hw, 8));
    CK(hipMalloc(&d_poly, poly.size() * 4)); CK(hipMalloc(&d_desc, desc.size() * sizeof(k::LineDesc)));
    const int32_t hw[2] = {H, W};
    CK(hipMemcpy(d_page, page.data(), page.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pages, &d_page, sizeof d_page, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_hw, hw, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_poly, poly.data(), poly.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_desc, desc.data(), desc.size() * sizeof(k::LineDesc), hipMemcpyHostToDevice));
    // ---- aggressor inputs
    const int R = 45056, K = 512, N = 768;
    std::vector<float> hA((size_t)R * K), hB((size_t)2 * K * N), hbias((size_t)2 * N, 0.01f);
    for (float& x : hA) x = U(rng);
    for (float& x : hB) x = 0.1f * U(rng);
    std::vector<uint16_t> img, one;
    for (int z = 0; z < 2; z++) { k::split_weights(hB.data() + (size_t)z * K * N, K, N, N, &one); img.insert(img.end(), one.begin(), one.end()); }
    float *dA, *dB, *dbias, *dC; uint16_t* dimg;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dbias, hbias.size() * 4));
    CK(hipMalloc(&dC, (size_t)2 * R * N * 4)); CK(hipMalloc(&dimg, img.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dimg, img.data(), img.size() * 2, hipMemcpyHostToDevice));
    k::GemmDesc g{};
    g.A = dA; g.lda = K; g.B = dB; g.ldb = N; g.bias = dbias; g.C = dC; g.ldc = N; g.M = R; g.N = N; g.K = K; g.batch = 2;
    g.strideA = 0; g.strideB = (int64_t)K * N; g.strideBias = N; g.strideC = (int64_t)R * N;
    g.Bsplit = dimg; g.strideBsplit = (int64_t)one.size();
    Tuning tune = default_tuning();
    tune.v[OPT_NUMERICS] = aggressor == "split3" ? 1 : aggressor == "split2" ? 2 : 0;
    // the quiet run: what every word should be (and, the page being random floats, where else each VALUE occurs)
    std::vector<float> ref((size_t)out_floats);
    std::unordered_multimap<uint32_t, uint32_t> where;
    std::vector<int> line_of, row_of, col_of;
    if (analyse) {
        float* R; CK(hipMalloc(&R, out_floats * 4));
        k::crop_lines(d_pages, d_hw, d_desc, d_poly, n_lines, out_h, R, nullptr);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), R, out_floats * 4, hipMemcpyDeviceToHost));
        line_of.resize(out_floats); row_of.resize(out_floats); col_of.resize(out_floats);
        for (int l = 0; l < n_lines; l++)
            for (int r = 0; r < out_h; r++)
                for (int c = 0; c < desc[l].out_w; c++) {
                    const size_t i = (size_t)desc[l].out_off + (size_t)r * desc[l].out_w + c;
                    line_of[i] = l; row_of[i] = r; col_of[i] = c;
                    uint32_t b; memcpy(&b, &ref[i], 4);
                    if (ref[i] != -0.5f) where.emplace(b, (uint32_t)i);
                }
    }
    std::mutex amu;
    std::map<std::string, long> relation;
    long analysed = 0;
    hipStream_t sa = make_stream(0, split_cus ? split_cus : cus, cus);
    std::atomic<bool> stop{false};
    std::atomic<long> agg_launches{0};
    std::thread agg([&] {
        if (aggressor == "none") return;
        (void)hipSetDevice(0);
        TuningScope ts(&tune);
        hipEvent_t ev[4];
        for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        for (long i = 0; !stop.load(); i++) {
            if (i >= 4) (void)hipEventSynchronize(ev[i & 3]);   // at most four launches queued
            k::gemm(g, sa);
            (void)hipEventRecord(ev[i & 3], sa);
            agg_launches++;
        }
        (void)hipStreamSynchronize(sa);
    });
    // ---- victims: twin launches + compare, `vstreams` host threads / streams
    std::atomic<long> twins{0}, bad_twins{0}, bad_words{0};
    std::vector<std::thread> vt;
    for (int v = 0; v < vstreams; v++) vt.emplace_back([&, v] {
        (void)hipSetDevice(0);
        hipStream_t sv = make_stream(split_cus ? split_cus : 0, cus, cus);
        float *X, *Y; unsigned long long *d_out, h_out[2];
        (void)hipMalloc(&X, out_floats * 4); (void)hipMalloc(&Y, out_floats * 4); (void)hipMalloc(&d_out, 16);
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
        while (std::chrono::steady_clock::now() < t_end) {
            const unsigned long long init[2] = {0, ~0ull};
            (void)hipMemcpyAsync(d_out, init, 16, hipMemcpyHostToDevice, sv);
            k::crop_lines(d_pages, d_hw, d_desc, d_poly, n_lines, out_h, X, sv);
            k::crop_lines(d_pages, d_hw, d_desc, d_poly, n_lines, out_h, Y, sv);
            hipLaunchKernelGGL(compare_kernel, dim3(256), dim3(256), 0, sv, (const uint32_t*)X, (const uint32_t*)Y, (size_t)out_floats, d_out);
            (void)hipMemcpyAsync(h_out, d_out, 16, hipMemcpyDeviceToHost, sv);
            (void)hipStreamSynchronize(sv);
            twins++;
            if (h_out[0] && analyse) {
                std::lock_guard<std::mutex> lk(amu);
                if (analysed < analyse) {
                    analysed++;
                    std::vector<float> hx((size_t)out_floats), hy((size_t)out_floats);
                    (void)hipMemcpy(hx.data(), X, out_floats * 4, hipMemcpyDeviceToHost);
                    (void)hipMemcpy(hy.data(), Y, out_floats * 4, hipMemcpyDeviceToHost);
                    for (const std::vector<float>* o : {&hx, &hy})
                        for (size_t i = 0; i < (size_t)out_floats; i++) {
                            const float wv = (*o)[i];
                            if (memcmp(&wv, &ref[i], 4) == 0) continue;
                            char key[160];
                            uint32_t b; memcpy(&b, &wv, 4);
                            auto range = where.equal_range(b);
                            if (wv == -0.5f) snprintf(key, sizeof key, "wrong word is the FILL value (right one is a pixel), lanes %d-%d", (col_of[i] & 48), (col_of[i] & 48) + 15);
                            else if (range.first == range.second) snprintf(key, sizeof key, "wrong value occurs NOWHERE in the right output (right one %s), lanes %d-%d", ref[i] == -0.5f ? "is fill" : "is a pixel", (col_of[i] & 48), (col_of[i] & 48) + 15);
                            else {
                                size_t best = range.first->second;
                                for (auto it = range.first; it != range.second; ++it)
                                    if (line_of[it->second] == line_of[i] && (line_of[best] != line_of[i] || abs(row_of[it->second] - row_of[i]) < abs(row_of[best] - row_of[i]))) best = it->second;
                                snprintf(key, sizeof key, "wrong value is the right value of line %+d, row %+d, column %+d (mod 64: %+d)", line_of[best] - line_of[i], row_of[best] - row_of[i],
                                         col_of[best] - col_of[i], (col_of[best] - col_of[i]) % 64);
                            }
                            relation[key]++;
                        }
                }
            }
            if (h_out[0]) {
                bad_twins++; bad_words += (long)h_out[0];
                if (bad_twins.load() <= 8) fprintf(stderr, "twin %ld differs: %llu words, first at float %llu (stream %d)\n", twins.load(), h_out[0], h_out[1], v);
            }
        }
    });
    for (auto& t : vt) t.join();
    stop = true;
    agg.join();
    if (analyse) {
        std::vector<std::pair<long, std::string>> top;
        for (auto& kv : relation) top.emplace_back(kv.second, kv.first);
        std::sort(top.rbegin(), top.rend());
        fprintf(stderr, "where the wrong words of the first %ld differing twins come from (%zu kinds):\n", analysed, top.size());
        for (size_t i = 0; i < top.size() && i < 40; i++) fprintf(stderr, "  %8ld  %s\n", top[i].first, top[i].second.c_str());
    }
    printf("{\"aggressor\": \"%s\", \"split_cus\": %d, \"cus\": %d, \"victim_lines\": %d, \"victim_streams\": %d, \"seconds\": %.1f, "
           "\"aggressor_launches\": %ld, \"twin_launches\": %ld, \"twins_that_differ\": %ld, \"words_that_differ\": %ld}\n",
           aggressor.c_str(), split_cus, cus, n_lines, vstreams, seconds, agg_launches.load(), twins.load(), bad_twins.load(), bad_words.load());
    return 0;
}
