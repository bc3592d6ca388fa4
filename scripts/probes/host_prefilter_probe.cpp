// host_prefilter_probe.cpp -- VERDICT r04 "next" 8: would a HOST-side pre-filter (drop the sample hashes whose two presence bits
// are clear before they cross PCIe) get the host-inclusive step below its 0.098 ms?  The step is PCIe-bound at 4.70 MB per
// 10^6-hash sample; ~84 % of the absent hashes would go.  What it costs the host is 10^6 probes per sample into the 162 MB
// presence filter (4 bits per distinct hash of the rs214-scale database).  This probe measures exactly that: a table of the
// filter's size with the filter's bit density, 10^6 sorted keys per sample drawn like FracMinHash hashes, T threads each
// taking a contiguous slice of the sorted sample (= a contiguous slice of the table: the same slice for every sample, so it
// can stay in that core complex's L3), eight rotating samples, the survivors written out compacted.
//   g++ -O3 -march=native -pthread scripts/probes/host_prefilter_probe.cpp -o /tmp/host_prefilter_probe && /tmp/host_prefilter_probe
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

static double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

int main(int argc, char** argv) {
    const uint64_t n_words = (162ull << 20) / 4;   // 162 MB of 32-bit filter words
    const uint64_t max_hash = 18446744073709552ull;  // scaled = 1000
    const uint64_t n_keys = 1000000, n_samples = 8;
    std::vector<uint32_t> filter(n_words);
    // 3.2e8 stored keys, two bits each out of 32 in their word: the density of the real filter (false positives ~16 %)
    {
        uint64_t s = 12345;
        for (uint64_t k = 0; k < 323000000ull; ++k) {
            s = mix(s + 0x9e3779b97f4a7c15ull);
            const uint64_t h = s % max_hash;
            const uint64_t w = (unsigned __int128)h * n_words / max_hash;
            const uint64_t m = mix(h);
            filter[w] |= (1u << (m & 31)) | (1u << ((m >> 5) & 31));
        }
    }
    std::vector<std::vector<uint64_t>> samples(n_samples);
    for (uint64_t q = 0; q < n_samples; ++q) {
        uint64_t s = 777 + q;
        samples[q].resize(n_keys);
        for (auto& h : samples[q]) { s = mix(s + 0x9e3779b97f4a7c15ull); h = s % max_hash; }
        std::sort(samples[q].begin(), samples[q].end());
    }
    std::vector<uint64_t> out(n_keys);
    const unsigned hw = std::thread::hardware_concurrency();
    printf("host threads available: %u; filter 162 MB; %llu keys per sample, %llu rotating samples\n", hw, (unsigned long long)n_keys, (unsigned long long)n_samples);
    for (unsigned T : {1u, 2u, 4u, 8u, 16u, 32u, 64u, 128u, 256u}) {
        if (T > hw) break;
        const int reps = T >= 16 ? 200 : (T >= 4 ? 40 : 10);
        std::atomic<uint64_t> kept{0};
        // every thread makes ALL passes over its own slice of the hash space (no barrier per sample: the throughput of a
        // pipeline that keeps T cores on this job); the time of the slowest thread / passes = ms per sample
        auto work = [&](unsigned t, int passes) {
            const uint64_t lo_h = max_hash / T * t, hi_h = t + 1 == T ? max_hash : max_hash / T * (t + 1);
            uint64_t k = 0;
            for (int q = 0; q < passes; ++q) {
                const std::vector<uint64_t>& S = samples[q % n_samples];
                const uint64_t a = std::lower_bound(S.begin(), S.end(), lo_h) - S.begin();
                const uint64_t b = std::lower_bound(S.begin(), S.end(), hi_h) - S.begin();
                uint64_t w_at = a;
                for (uint64_t i = a; i < b; ++i) {
                    if (i + 16 < b) __builtin_prefetch(&filter[(unsigned __int128)S[i + 16] * n_words / max_hash]);
                    const uint64_t h = S[i];
                    const uint64_t w = (unsigned __int128)h * n_words / max_hash;
                    const uint64_t m = mix(h);
                    const uint32_t need = (1u << (m & 31)) | (1u << ((m >> 5) & 31));
                    if ((filter[w] & need) == need) { out[w_at++] = h; ++k; }
                }
            }
            kept += k;
        };
        auto run = [&](int passes) {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; ++t) th.emplace_back(work, t, passes);
            for (auto& x : th) x.join();
        };
        run(8);  // warm
        kept = 0;
        const double t0 = now_ms();
        run(reps);
        const double ms = (now_ms() - t0) / reps;
        printf("T = %3u threads: %.4f ms per sample, %.1f %% of the hashes survive\n", T, ms, 100.0 * kept / ((double)reps * n_keys));
    }
    return 0;
}
