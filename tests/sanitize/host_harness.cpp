// Sanitizer harness for the HOST C++ of libocrs_amd (the reference gets data-race and memory safety
// from Rust's `Send + Sync` — detection.rs:67, recognition.rs:316; this code gets it from -fsanitize runs).
// TEST INFRASTRUCTURE.  tests/test_sanitizers.py builds it twice with clang++ — `-fsanitize=thread` and
// `-fsanitize=address,undefined` — from the product's own sources (no HIP object is linked: everything exercised here
// is host-only) and fails on any report.
//
//   host_harness coalescer                 the request coalescer (coalesce.hpp) through the body of ocrs_coalescer_selftest
//   host_harness shares                    the engine group's worker pool and fan-out (host_pool.hpp), incl. failing shares
//   host_harness layout <pages.bin> <thr>  find_text_lines (layout.cpp) over fuzz pages on the batch pool (for_pages),
//                                          every page also twice at once; prints an FNV-1a hash of all lines
//   host_harness beam                      ctc_beam_search (ctc_beam.cpp) against its textbook formulation on random matrices
//   host_harness text_items                text_item_rotated_rect (text_items.cpp) on random character boxes
//   host_harness jpeg <streams.bin>        the JPEG marker parser and Huffman decoders (jpeg_host.cpp) on valid, truncated and
//                                          corrupted streams: every one must come back as coefficients or as an ocrs::Error
#include "../../ocrs_amd/csrc/numa.hpp"
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "../../ocrs_amd/csrc/coalesce_selftest.hpp"
#include "../../ocrs_amd/csrc/engine.hpp"
#include "../../ocrs_amd/csrc/geometry.hpp"
#include "../../ocrs_amd/csrc/host_pool.hpp"
#include "../../ocrs_amd/csrc/jpeg.hpp"

using namespace ocrs;
using ocrs::geom::RotatedRect;

namespace ocrs {   // common.cpp is not linked (it owns HIP state); the two symbols the host sources refer to
void set_last_error(const std::string&) {}
const std::string& last_error() { static std::string s; return s; }
}  // namespace ocrs

static int check(bool ok, const char* what) {
    if (!ok) { fprintf(stderr, "FAILED: %s\n", what); return 1; }
    return 0;
}

static int run_coalescer() {
    int bad = 0;
    const int cases[][6] = {{1, 40, 2, 8, 200, 7}, {12, 30, 2, 8, 300, 7}, {24, 20, 1, 16, 100, 5}, {8, 25, 3, 4, 0, 0}};
    for (const auto& c : cases) {
        uint64_t out[5];
        coalescer_selftest(c[0], c[1], c[2], c[3], c[4], c[5], out);
        const uint64_t n = (uint64_t)c[0] * c[1];
        uint64_t expect_err = 0;
        for (uint64_t i = 1; i <= n; i++) expect_err += c[5] > 0 && i % c[5] == 0;
        bad += check(out[1] == n && out[3] == 0 && out[2] == expect_err && out[4] <= (uint64_t)c[3], "coalescer selftest counters");
        printf("coalescer threads=%d: %" PRIu64 " batches for %" PRIu64 " requests, %" PRIu64 " errors delivered, largest batch %" PRIu64 " pages\n",
               c[0], out[0], out[1], out[2], out[4]);
    }
    return bad;
}

static int run_shares() {
    int bad = 0;
    std::atomic<uint64_t> exits{0};
    {
        WorkerPool pool([&] { exits++; });
        constexpr int kCallers = 6, kRounds = 200, G = 8;
        std::atomic<uint64_t> ran{0}, thrown{0}, caught{0};
        std::vector<std::thread> callers;
        for (int c = 0; c < kCallers; c++)
            callers.emplace_back([&, c] {
                std::mt19937 rng(c * 7919 + 1);
                for (int r = 0; r < kRounds; r++) {
                    std::vector<char> work(G);
                    for (auto& w : work) w = (rng() % 3) != 0;
                    const int fail_m = (rng() % 5 == 0) ? (int)(rng() % G) : -1;
                    std::vector<int> touched(G, 0);
                    std::vector<std::exception_ptr> errs;
                    bool got = false;
                    try {
                        run_shares(pool, work, errs, [&](size_t m) {
                            touched[m]++;
                            ran++;
                            if ((int)m == fail_m) { thrown++; throw std::runtime_error("share failed"); }
                        });
                    } catch (const std::runtime_error&) {
                        got = true;
                        caught++;
                    }
                    for (int m = 0; m < G; m++)
                        if (touched[m] != (work[m] ? 1 : 0)) { fprintf(stderr, "share %d ran %d times\n", m, touched[m]); std::abort(); }
                    if (got != (fail_m >= 0 && work[fail_m])) { fprintf(stderr, "a failing share was not reported to its caller\n"); std::abort(); }
                }
            });
        for (auto& t : callers) t.join();
        bad += check(thrown == caught, "every failing share reached exactly its caller");
        bad += check(pool.threads() >= 1 && pool.threads() <= (size_t)kCallers * (G - 1), "pool size bounded by the shares in flight");
        printf("shares: %" PRIu64 " shares run on %zu kept threads, %" PRIu64 " failures delivered\n", ran.load(), pool.threads(), caught.load());
        exits = 0;
    }
    bad += check(exits.load() >= 1, "thread-exit hook ran");
    return bad;
}

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

static int run_layout(const char* path, int threads) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    uint32_t n_pages = 0;
    if (fread(&n_pages, 4, 1, f) != 1) return 1;
    std::vector<std::vector<RotatedRect>> pages(n_pages);
    for (auto& pg : pages) {
        uint32_t nw = 0;
        if (fread(&nw, 4, 1, f) != 1) return 1;
        std::vector<float> a((size_t)nw * 6);
        if (nw && fread(a.data(), 4, a.size(), f) != a.size()) return 1;
        for (uint32_t i = 0; i < nw; i++) pg.push_back(RotatedRect::from_array(a.data() + 6 * i));
    }
    fclose(f);
    auto digest = [](const std::vector<std::vector<RotatedRect>>& lines) {
        uint64_t h = 14695981039346656037ull;
        for (const auto& l : lines) {
            const uint64_t n = l.size();
            h = fnv(h, &n, sizeof n);
            for (const RotatedRect& r : l) { float a[6]; r.to_array(a); h = fnv(h, a, sizeof a); }
        }
        return h;
    };
    std::vector<uint64_t> first(n_pages), second(n_pages);
    // every page twice, 2 * n_pages tasks on the bounded pool: two threads may be inside the same page's analysis at once
    for_pages(2 * (size_t)n_pages, (size_t)threads, [&](size_t t) {
        const size_t p = t % n_pages;
        (t < n_pages ? first : second)[p] = digest(find_text_lines(pages[p]));
    });
    uint64_t h = 14695981039346656037ull;
    int bad = 0;
    for (uint32_t p = 0; p < n_pages; p++) {
        bad += first[p] != second[p];
        h = fnv(h, &first[p], 8);
    }
    printf("layout: %u pages on %d threads, hash %016" PRIx64 "\n", n_pages, threads, h);
    return check(bad == 0, "the same page analysed twice gave the same lines");
}

static int run_beam() {
    std::mt19937 rng(12345);
    std::normal_distribution<float> nd(0.f, 2.f);
    int bad = 0, cases = 0;
    for (int T : {1, 2, 7, 40}) {
        for (int C : {2, 5, 97}) {
            for (uint32_t width : {1u, 3u, 20u, 100u}) {
                std::vector<float> logp((size_t)T * C);
                for (int t = 0; t < T; t++) {
                    float m = -1e30f;
                    for (int c = 0; c < C; c++) { logp[(size_t)t * C + c] = nd(rng); m = std::max(m, logp[(size_t)t * C + c]); }
                    double s = 0;
                    for (int c = 0; c < C; c++) s += std::exp((double)logp[(size_t)t * C + c] - m);
                    for (int c = 0; c < C; c++) logp[(size_t)t * C + c] -= m + (float)std::log(s);
                    if (rng() % 4 == 0) logp[(size_t)t * C + rng() % C] = -INFINITY;   // masked label (allowed_chars)
                    if (rng() % 4 == 0 && C > 2) logp[(size_t)t * C + 1] = logp[(size_t)t * C + 2];   // an exact tie
                }
                const auto a = ctc_beam_search(logp.data(), T, C, C, width);
                const auto b = ctc_beam_search_reference(logp.data(), T, C, C, width);
                bool same = a.size() == b.size();
                for (size_t i = 0; same && i < a.size(); i++) same = a[i].label == b[i].label && a[i].pos == b[i].pos;
                bad += !same;
                cases++;
            }
        }
    }
    printf("beam: %d cases, %d differ from the textbook formulation\n", cases, bad);
    return check(bad == 0, "beam search equals its reference formulation");
}

static int run_text_items() {
    std::mt19937 rng(99);
    int made = 0;
    for (int it = 0; it < 2000; it++) {
        const int n = 1 + rng() % 40;
        std::vector<int32_t> tlbr;
        int x = (int)(rng() % 500) - 100;
        const int y = (int)(rng() % 500) - 100;
        for (int i = 0; i < n; i++) {
            const int w = rng() % 30, h = rng() % 30;   // degenerate (empty) boxes included
            const int dy = (int)(rng() % 7) - 3;
            tlbr.insert(tlbr.end(), {y + dy, x, y + dy + h, x + w});
            x += w + (int)(rng() % 5);
        }
        RotatedRect rr;
        if (text_item_rotated_rect(tlbr.data(), (size_t)n, &rr)) {
            float a[6];
            rr.to_array(a);
            for (float v : a)
                if (!(v == v)) return check(false, "NaN in a text item's rotated rect");
            made++;
        }
    }
    printf("text_items: %d rects\n", made);
    return check(made > 1500, "most random character runs have a rotated rect");
}

static int run_jpeg(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return 1;
    int decoded = 0, refused = 0;
    uint64_t h = 14695981039346656037ull;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t len = 0;
        if (fread(&len, 4, 1, f) != 1) return 1;
        std::vector<uint8_t> buf(len);   // exact size: an over-read is an ASan report
        if (len && fread(buf.data(), 1, len, f) != len) return 1;
        try {
            const jpeg::Coefficients c = jpeg::decode_coefficients(buf.data(), buf.size());
            decoded++;
            h = fnv(h, c.values.data(), c.values.size() * sizeof(int16_t));
            h = fnv(h, c.mask.data(), c.mask.size() * sizeof(uint64_t));
            if (c.offset.size() != c.mask.size() + 1 || c.offset.back() != c.values.size()) return check(false, "sparse layout inconsistent");
        } catch (const Error&) {
            refused++;
        }
    }
    fclose(f);
    printf("jpeg: %d decoded, %d refused, hash %016" PRIx64 "\n", decoded, refused, h);
    return check(decoded > 0, "at least the intact streams decode");
}

// numa.hpp: cpu-list parsing on hostile strings and BindScope from several threads at once (each binds to one of the CPUs the
// process may use and must find its own mask restored).
static int run_numa() {
    using namespace ocrs::numa;
    std::vector<int> v;
    const char* good[] = {"0-3,8,10-11\n", "", "7", " 1 , 2 ", "0-0"};
    const char* bad[] = {"3-1", "a", "1,,2", "1-", "70000", "-1", "1-2-3", ","};
    for (const char* g : good) if (!parse_cpulist(g, &v)) return check(false, "a good cpu list was refused");
    for (const char* b : bad) if (parse_cpulist(b, &v)) return check(false, "a malformed cpu list was accepted");
    if (parse_cpulist(nullptr, &v)) return check(false, "null list accepted");
    std::mt19937 rng(7);
    for (int i = 0; i < 20000; i++) {   // random bytes from the list alphabet: must never crash or loop
        std::string sx;
        const int n = rng() % 24;
        for (int k = 0; k < n; k++) sx.push_back("0123456789,- \n"[rng() % 15]);
        (void)parse_cpulist(sx.c_str(), &v);
        if (v.size() > 70000) return check(false, "implausible cpu count");
    }
    if (node_of_pci("0000:ff:1f.7", "/nonexistent") != -1 || cpus_of_node(0, &v, "/nonexistent")) return check(false, "missing sysfs must mean unknown");
    cpu_set_t mine;
    if (sched_getaffinity(0, sizeof mine, &mine) != 0) return check(true, "no affinity call here: nothing to test");
    std::vector<int> allowed;
    for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &mine)) allowed.push_back(c);
    std::atomic<int> wrong{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 8; t++)
        th.emplace_back([&, t] {
            for (int i = 0; i < 200; i++) {
                const int before = affinity_count();
                {
                    BindScope b({allowed[(t + i) % allowed.size()]});
                    if (b.bound() && affinity_count() != 1) wrong++;
                    BindScope none(std::vector<int>{});          // empty list: no binding
                    BindScope outside({CPU_SETSIZE - 1 == allowed.back() ? -5 : CPU_SETSIZE - 1});   // a CPU we may not use: no binding
                    if (none.bound() || outside.bound()) wrong++;
                }
                if (affinity_count() != before) wrong++;
            }
        });
    for (auto& x : th) x.join();
    printf("numa: %zu cpus allowed, %d wrong\n", allowed.size(), wrong.load());
    return check(wrong.load() == 0, "bind scopes restore the thread's mask");
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "";
    int bad = 0;
    if (mode == "coalescer") bad = run_coalescer();
    else if (mode == "shares") bad = run_shares();
    else if (mode == "layout" && argc > 3) bad = run_layout(argv[2], atoi(argv[3]));
    else if (mode == "beam") bad = run_beam();
    else if (mode == "text_items") bad = run_text_items();
    else if (mode == "numa") bad = run_numa();
    else if (mode == "jpeg" && argc > 2) bad = run_jpeg(argv[2]);
    else { fprintf(stderr, "usage: host_harness coalescer|shares|layout <pages.bin> <threads>|beam|text_items|numa|jpeg <streams.bin>\n"); return 2; }
    printf("%s: %s\n", mode.c_str(), bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
