// Native host threads against the C ABI (no interpreter lock between the calls): T threads, each issuing `per` single-query
// as_search calls back to back on ONE space -- what a C++ / Rust host of the library sees of the re-entrant search (gang scans:
// callers that arrive together share a pass over the items).  Built by tools/thread_bench.py into tools/probe/libthread_driver.so.
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>

#include "arrowspace_hip.h"

extern "C" double as_thread_driver(const as_space* sp, const as_graph* gr, const double* Q, int64_t nq, int64_t d, double tau, int nthreads, int per,
                                   int64_t topk, int64_t* out_first_idx, int64_t* out_errors) {
    std::vector<std::thread> th;
    std::vector<int64_t> errs(nthreads, 0);
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            std::vector<int64_t> idx(topk);
            std::vector<double> sc(topk);
            for (int i = 0; i < per; ++i) {
                const int64_t j = ((int64_t)t * 131 + i) % nq;
                int64_t len = 0;
                double lq = 0.0;
                const as_status s = as_search(sp, gr, Q + j * d, d, tau, idx.data(), sc.data(), &len, &lq);
                if (s != 0 || len != topk || (out_first_idx && idx[0] != out_first_idx[j])) errs[t] += 1;
            }
        });
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int64_t e = 0;
    for (int64_t v : errs) e += v;
    if (out_errors) *out_errors = e;
    return dt;
}
