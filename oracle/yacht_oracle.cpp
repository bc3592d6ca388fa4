// yacht_oracle.cpp — CPU restatement of YACHT's hot path.  TEST INFRASTRUCTURE ONLY.
//
// Nothing under yacht_amd/ may import, link or call this file.  It exists so that tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg can check (and time a CPU figure
// beside) the HIP path.  Parity status: PINNED — see oracle/README.md: the train functions are
// checked against the genuine reference executable built into oracle/_ref/, the run functions
// against golden vectors produced by importing the reference's Python in the build container
// (tests/golden/make_golden.py), and both against the reference's own known-answer test.
//
// Each function restates, in its own code, the algorithm of the reference lines it cites
// (paths relative to the YACHT repository, v1.4.0).  It is C++ rather than plain C for one
// reason: the reference orders genomes with libstdc++'s unstable std::sort and ties can only be
// reproduced by calling the same routine (see oracle_train_select).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

typedef uint64_t u64;
typedef uint32_t u32;

extern "C" {

// ---------------------------------------------------------------------------------------------
// overlap[j] = |S ∩ R_j|.  The reference delegates this to `sourmash scripts multisearch`
// (src/yacht/hypothesis_recovery_src.py:93-113; third-party Rust plugin, not in the tree) and
// only uses "overlap > 0"; the count is the set-intersection size of the two hash sets.
// Restated as a sorted two-list intersection per reference (sketches are ascending, unique).
static u32 intersect_sorted(const u64* a, u64 na, const u64* b, u64 nb) {
    // a is the short list (a reference), b the long one (the sample): advance in b by
    // doubling steps, then bisect, never moving backwards.
    u32 c = 0;
    u64 lo = 0;
    for (u64 i = 0; i < na && lo < nb; ++i) {
        const u64 h = a[i];
        u64 step = 1, hi = lo;
        while (hi < nb && b[hi] < h) { lo = hi + 1; hi += step; step <<= 1; }
        if (hi > nb) hi = nb;
        while (lo < hi) {
            const u64 mid = lo + ((hi - lo) >> 1);
            if (b[mid] < h) lo = mid + 1; else hi = mid;
        }
        if (lo < nb && b[lo] == h) { ++c; ++lo; }
    }
    return c;
}

void oracle_overlap(const u64* values, const u64* offsets, u64 n_refs, const u64* sample, u64 n_sample,
                    u32* overlap, int threads) {
    if (threads < 1) threads = 1;
    auto work = [&](u64 j0, u64 j1) {
        for (u64 j = j0; j < j1; ++j)
            overlap[j] = intersect_sorted(values + offsets[j], offsets[j + 1] - offsets[j], sample, n_sample);
    };
    if (threads == 1 || n_refs < (u64)threads * 4) { work(0, n_refs); return; }
    // contiguous reference ranges cut by HASH count (the cost of a reference is its size, not 1): thread t ends at
    // the first reference boundary at or past t + 1 shares of the hashes
    std::vector<std::thread> pool;
    const u64 total = offsets[n_refs];
    u64 a = 0;
    for (int t = 0; t < threads && a < n_refs; ++t) {
        u64 b = n_refs;
        if (t + 1 < threads) {
            const u64 target = total / (u64)threads * (u64)(t + 1);
            b = (u64)(std::lower_bound(offsets + a, offsets + n_refs + 1, target) - offsets);
            b = std::min<u64>(std::max<u64>(b, a + 1), n_refs);
        }
        pool.emplace_back(work, a, b);
        a = b;
    }
    for (auto& th : pool) th.join();
}

// ---------------------------------------------------------------------------------------------
// get_exclusive_hashes (src/yacht/hypothesis_recovery_src.py:165-204), for the references with
// mask != 0 taken in index (= manifest) order:
//   :165-180  one pass over the subset: a hash seen once is "single", seen again -> "multiple";
//   :184-191  exclusive_j = hashes of R_j that are single;
//   :194-204  (|exclusive_j|, |exclusive_j ∩ sample|).
// Unmasked references get (0, 0).
void oracle_exclusive(const u64* values, const u64* offsets, u64 n_refs, const uint8_t* mask, const u64* sample,
                      u64 n_sample, u32* n_excl, u32* n_match) {
    std::unordered_map<u64, u32> seen;  // hash -> 1 (single) / 2 (multiple)
    u64 total = 0;
    for (u64 j = 0; j < n_refs; ++j)
        if (mask[j]) total += offsets[j + 1] - offsets[j];
    seen.reserve(total);
    for (u64 j = 0; j < n_refs; ++j) {
        if (!mask[j]) continue;
        for (u64 k = offsets[j]; k < offsets[j + 1]; ++k) {
            u32& s = seen[values[k]];
            if (s < 2) ++s;
        }
    }
    for (u64 j = 0; j < n_refs; ++j) {
        u32 e = 0, m = 0;
        if (mask[j]) {
            for (u64 k = offsets[j]; k < offsets[j + 1]; ++k) {
                const u64 h = values[k];
                if (seen[h] != 1) continue;
                ++e;
                if (std::binary_search(sample, sample + n_sample, h)) ++m;
            }
        }
        n_excl[j] = e;
        n_match[j] = m;
    }
}

// ---------------------------------------------------------------------------------------------
// Train core, counting part (src/cpp/main.cpp):
//   :215-246  inverted index hash -> [sketch ids], single thread, then drop hashes seen once;
//   :249-262  for every hash of sketch i and every id k in its list: M[i][k] += 1;
//   :274-308  keep (i, j), j != i, M > 0, both sketches non-empty, union > 0, and
//             NOT (1.0*M/|R_i| < C); j ascending inside i.
// Rows are split over `threads` exactly like :345-349 (contiguous chunks, the last takes the
// remainder).  Output: (i, j, M) triples sorted by (i, j) and the three statistics of :242-244.
// Returns the number of pairs; fills at most `cap` of them (call again with a larger cap).
u64 oracle_train_pairs(const u64* values, const u64* offsets, u64 n_refs, double c_thresh, int threads, u64 cap,
                       u32* pair_i, u32* pair_j, u32* pair_cnt, u64* stats /* [3] */) {
    if (threads < 1) threads = 1;
    std::unordered_map<u64, std::vector<int>> index;
    for (u64 i = 0; i < n_refs; ++i)
        for (u64 k = offsets[i]; k < offsets[i + 1]; ++k) index[values[k]].push_back((int)i);
    const u64 n_distinct = index.size();
    for (auto it = index.begin(); it != index.end();) {
        if (it->second.size() == 1) it = index.erase(it); else ++it;
    }
    if (stats) {
        stats[0] = n_distinct;
        stats[1] = n_distinct - index.size();
        stats[2] = index.size();
    }

    struct Triple { u32 i, j, c; };
    std::vector<std::vector<Triple>> found(threads);
    auto work = [&](int tid, u64 r0, u64 r1) {
        std::vector<int> row(n_refs, 0);
        for (u64 i = r0; i < r1; ++i) {
            std::fill(row.begin(), row.end(), 0);
            for (u64 k = offsets[i]; k < offsets[i + 1]; ++k) {
                auto it = index.find(values[k]);
                if (it == index.end()) continue;
                for (int o : it->second) ++row[o];
            }
            const u64 si = offsets[i + 1] - offsets[i];
            for (u64 j = 0; j < n_refs; ++j) {
                if (j == i || row[j] == 0) continue;
                const u64 sj = offsets[j + 1] - offsets[j];
                if (si == 0 || sj == 0) continue;
                if (si + sj - (u64)row[j] == 0) continue;
                const double c_ij = 1.0 * row[j] / si;
                if (c_ij < c_thresh) continue;
                found[tid].push_back({(u32)i, (u32)j, (u32)row[j]});
            }
        }
    };
    const u64 chunk = n_refs / threads;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        const u64 r0 = (u64)t * chunk;
        const u64 r1 = (t == threads - 1) ? n_refs : (u64)(t + 1) * chunk;
        pool.emplace_back(work, t, r0, r1);
    }
    for (auto& th : pool) th.join();
    u64 n = 0;
    for (int t = 0; t < threads; ++t)
        for (const Triple& x : found[t]) {
            if (n < cap) { pair_i[n] = x.i; pair_j[n] = x.j; pair_cnt[n] = x.c; }
            ++n;
        }
    return n;
}

// ---------------------------------------------------------------------------------------------
// do_yacht_train (src/cpp/main.cpp:371-407): std::sort of {id, size} by size only (unstable,
// initial order = input order), then walk: a genome is dropped when one of its listed
// neighbours is not dropped yet and is at least as large; otherwise it is selected.
// `pair_i/pair_j` = the kept pairs sorted by (i, j).  Returns the number selected.
u64 oracle_train_select(const u32* sizes, u64 n_refs, const u32* pair_i, const u32* pair_j, u64 n_pairs,
                        u32* selected) {
    std::vector<std::vector<int>> similars(n_refs);
    for (u64 k = 0; k < n_pairs; ++k) similars[pair_i[k]].push_back((int)pair_j[k]);
    std::vector<std::pair<int, int>> id_size(n_refs);
    for (u64 i = 0; i < n_refs; ++i) id_size[i] = {(int)i, (int)sizes[i]};
    std::sort(id_size.begin(), id_size.end(),
              [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second < b.second; });
    std::vector<bool> excluded(n_refs, false);
    u64 ns = 0;
    for (u64 t = 0; t < n_refs; ++t) {
        const int me = id_size[t].first, my_size = id_size[t].second;
        bool select = true;
        for (int other : similars[me]) {
            if (excluded[other]) continue;
            if ((int)sizes[other] >= my_size) { select = false; break; }
        }
        if (select) selected[ns++] = (u32)me; else excluded[me] = true;
    }
    return ns;
}

int oracle_hardware_threads(void) {
    const unsigned n = std::thread::hardware_concurrency();
    return n ? (int)n : 1;
}

}  // extern "C"
