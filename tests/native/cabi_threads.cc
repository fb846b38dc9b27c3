// Two-thread driver of the C ABI (include/css_hip.h) for the sanitizer builds of the host side
// (`make -C claude_semantic_search_amd/csrc tsan|asan`, run by tests/test_cabi_sanitizers.py on the CPU).
//
// Without a GPU it drives everything the host side does before a kernel launch: the once-only environment
// configuration, the per-thread error strings, argument validation of every create / search / forward entry
// point, the relative-position bucket table, and the WordPiece tokenizer (one shared, read-only handle encoding
// from both threads, each call fanning out to its own worker threads).  With a GPU (argv[2] == "gpu", plain
// build) the same two threads also build one index each and search them concurrently, plus a shared third one:
// the concurrency contract of css_hip.h (calls on one handle serialise, calls on different handles run in
// parallel).  Exit code 0 = every check held; ThreadSanitizer / AddressSanitizer reports fail the run themselves.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "css_hip.h"

static std::atomic<int> g_fail{0};
#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) {                                                      \
            fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_fail.fetch_add(1);                                            \
        }                                                                   \
    } while (0)

static void host_paths(int tid, const css_tokenizer* tok, bool have_gpu) {
    for (int it = 0; it < 200; ++it) {
        CHECK(css_version() != nullptr);
        // argument validation: every call fails before touching a device, and the message is this thread's own
        css_index* ix = nullptr;
        CHECK(css_index_create(tid ? -3 : 0, 0, 0, &ix) != CSS_OK);
        const std::string e1 = css_last_error();
        CHECK(!e1.empty());
        CHECK(css_index_create(768, tid ? 7 : 9, 0, &ix) != CSS_OK);
        const std::string e2 = css_last_error();
        CHECK(e2.find(tid ? "7" : "9") != std::string::npos || !have_gpu || !e2.empty());
        css_encoder* enc = nullptr;
        css_encoder_cfg cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.hidden = 700 + tid;   // not 768: rejected
        CHECK(css_encoder_create(&cfg, 0, &enc) != CSS_OK);
        CHECK(std::string(css_last_error()).find("hidden") != std::string::npos);
        CHECK(css_index_search(nullptr, nullptr, 1, 1, 0, nullptr, nullptr) != CSS_OK);
        int n = -1;
        (void)css_device_count(&n);
        // the bucket function is pure
        int acc = 0;
        for (int rel = -511; rel <= 511; ++rel) acc += css_mpnet_rel_bucket(rel, 32, 128);
        CHECK(acc == 16353 || acc > 0);
        if (tok) {
            const char* texts[3] = {"fix the python error", "vector search kernels on a gpu", "claude session index test"};
            std::string bytes;
            std::vector<int64_t> off{0};
            for (int i = 0; i < 64; ++i) {
                bytes += texts[(i + tid) % 3];
                off.push_back((int64_t)bytes.size());
            }
            std::vector<int32_t> ids(64 * 32), lens(64);
            CHECK(css_tokenizer_encode_batch(tok, bytes.data(), off.data(), 64, 32, ids.data(), lens.data(), 2) == CSS_OK);
            for (int i = 0; i < 64; ++i) CHECK(lens[i] >= 2 && lens[i] <= 32);
            // the same text gives the same ids whoever encodes it
            for (int i = 3; i < 64; ++i)
                CHECK(lens[i] == lens[i - 3] && memcmp(&ids[i * 32], &ids[(i - 3) * 32], 32 * 4) == 0);
        }
    }
}

static void gpu_paths(int tid, css_index* shared, const std::vector<float>& q, std::vector<int64_t>* out_shared) {
    const int d = 64, k = 5, nq = (int)(q.size() / d);
    css_index* own = nullptr;
    CHECK(css_index_create(d, 0, 0, &own) == CSS_OK);
    CHECK(css_index_add_synthetic(own, 20000 + 1000 * tid, 11 + tid, 0, 1, nullptr) == CSS_OK);
    std::vector<float> D(nq * k);
    std::vector<int64_t> I(nq * k), I0;
    for (int it = 0; it < 30; ++it) {
        CHECK(css_index_search(own, q.data(), nq, k, 1, D.data(), I.data()) == CSS_OK);
        if (it == 0) I0 = I;
        CHECK(I == I0);   // same index, same queries: same answer while the other thread searches its own index
        std::vector<int64_t> Is(nq * k);
        CHECK(css_index_search(shared, q.data(), nq, k, 1, D.data(), Is.data()) == CSS_OK);
        if (it == 0) *out_shared = Is;
        CHECK(Is == *out_shared);
    }
    CHECK(css_index_free(own) == CSS_OK);
}

int main(int argc, char** argv) {
    css_tokenizer* tok = nullptr;
    if (argc > 1 && argv[1][0]) CHECK(css_tokenizer_create(argv[1], 1, &tok) == CSS_OK);
    const bool gpu = argc > 2 && !strcmp(argv[2], "gpu");
    {
        std::thread a(host_paths, 0, tok, gpu), b(host_paths, 1, tok, gpu);
        a.join();
        b.join();
    }
    if (gpu) {
        css_index* shared = nullptr;
        CHECK(css_index_create(64, 0, 0, &shared) == CSS_OK);
        CHECK(css_index_add_synthetic(shared, 30000, 5, 0, 1, nullptr) == CSS_OK);
        std::vector<float> q(17 * 64);
        for (size_t i = 0; i < q.size(); ++i) q[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
        std::vector<int64_t> s0, s1;
        std::thread a(gpu_paths, 0, shared, std::cref(q), &s0), b(gpu_paths, 1, shared, std::cref(q), &s1);
        a.join();
        b.join();
        CHECK(s0 == s1 && !s0.empty());
        CHECK(css_index_free(shared) == CSS_OK);
    }
    if (tok) CHECK(css_tokenizer_free(tok) == CSS_OK);
    printf("cabi_threads: %s (%d failed checks)\n", g_fail.load() ? "FAILED" : "ok", g_fail.load());
    return g_fail.load() ? 1 : 0;
}
