// run_yacht_train_core — drop-in for the reference executable of the same name (src/cpp/main.cpp),
// a thin main() over libyacht_hip.so.
//
//   run_yacht_train_core [-t threads] [-p passes] [-c containment_threshold]
//                        file_list working_directory output_filename
//
// Same argv (defaults t=1, p=1, c=0.9; invalid value -> message on stderr, "Usage:" line, exit 1:
// main.cpp:142-184,430-436), same inputs (a text file of sketch paths; each sketch a sourmash JSON
// of which record 0 / signature 0 / "mins" is used: main.cpp:74-78) and the same outputs:
// <output_filename> with the selected paths in walk order and <working_directory>/<pass>_<tid3>.txt
// with the `i,j,jaccard,Cij,Cji` lines of each (pass, thread) row block (main.cpp:264-308,318-349).
// The intersections, the threshold filter and the selection are done by the library
// (yh_db_create / yh_pairwise / yh_train_select); this file only parses and prints.
#include "yh_sigread.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "yacht_hip.h"

namespace {

struct Args {
    std::string file_list, working_directory, output_filename;
    int threads = 1;
    int passes = 1;
    double c = 0.9;
};

void usage(const char* argv0) {
    std::cout << "Usage: " << argv0
              << " [--help] [--threads VAR] [--passes VAR] [--containment_threshold VAR] file_list working_directory "
                 "output_filename\n\n"
                 "Positional arguments:\n"
                 "  file_list                     file containing list of files to be processed\n"
                 "  working_directory             working directory (where temp files are generated)\n"
                 "  output_filename               output filename (where the reduced ref filenames will be written)\n\n"
                 "Optional arguments:\n"
                 "  -h, --help                    shows help message and exits\n"
                 "  -t, --threads                 number of threads [default: 1]\n"
                 "  -p, --passes                  number of passes [default: 1]\n"
                 "  -c, --containment_threshold   containment threshold [default: 0.9]\n";
}

// returns 0 ok, 1 error (message printed), 2 help shown
int parse_args(int argc, char** argv, Args& a) {
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i];
        auto need = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { std::cerr << "Too few arguments for '" << what << "'." << std::endl; return nullptr; }
            return argv[++i];
        };
        if (s == "-h" || s == "--help") { usage(argv[0]); return 2; }
        if (s == "-t" || s == "--threads") {
            const char* v = need("-t"); if (!v) return 1;
            char* end = nullptr; long x = strtol(v, &end, 10);
            if (!*v || *end) { std::cerr << "pattern '" << v << "' not found" << std::endl; return 1; }
            a.threads = (int)x;
        } else if (s == "-p" || s == "--passes") {
            const char* v = need("-p"); if (!v) return 1;
            char* end = nullptr; long x = strtol(v, &end, 10);
            if (!*v || *end) { std::cerr << "pattern '" << v << "' not found" << std::endl; return 1; }
            a.passes = (int)x;
        } else if (s == "-c" || s == "--containment_threshold") {
            const char* v = need("-c"); if (!v) return 1;
            char* end = nullptr; double x = strtod(v, &end);
            if (!*v || *end) { std::cerr << "pattern '" << v << "' not found" << std::endl; return 1; }
            a.c = x;
        } else if (s.size() > 1 && s[0] == '-' && !(s[1] >= '0' && s[1] <= '9') && s[1] != '.') {
            std::cerr << "Unknown argument: " << s << std::endl;
            return 1;
        } else {
            pos.push_back(s);
        }
    }
    if (pos.size() != 3) {
        std::cerr << (pos.size() < 3 ? "Too few arguments" : "Maximum number of positional arguments exceeded") << std::endl;
        return 1;
    }
    a.file_list = pos[0];
    a.working_directory = pos[1];
    a.output_filename = pos[2];
    if (a.threads < 1) { std::cerr << "number of threads must be at least 1" << std::endl; return 1; }
    if (a.passes < 1) { std::cerr << "number of passes must be at least 1" << std::endl; return 1; }
    if (a.c < 0.0 || a.c > 1.0) { std::cerr << "containment threshold must be between 0.0 and 1.0" << std::endl; return 1; }
    return 0;
}

// (the .sig reader: yh_sigread.h)
using yh_sig::read_mins;

long ms_since(std::chrono::high_resolution_clock::time_point t0) {
    return (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
}

}  // namespace

int main(int argc, char** argv) {
    Args a;
    const int pr = parse_args(argc, argv, a);
    if (pr == 2) return 0;
    if (pr != 0) { std::cout << "Usage: " << argv[0] << " -h" << std::endl; return 1; }

    std::cout << "Working with the following parameters:" << std::endl;
    std::cout << "**************************************" << std::endl << "*" << std::endl;
    std::cout << "*    file_list: " << a.file_list << std::endl;
    std::cout << "*    working_directory: " << a.working_directory << std::endl;
    std::cout << "*    output_filename: " << a.output_filename << std::endl;
    std::cout << "*    number_of_threads: " << a.threads << std::endl;
    std::cout << "*    num_of_passes: " << a.passes << std::endl;
    std::cout << "*    containment_threshold: " << a.c << std::endl;
    std::cout << "*" << std::endl << "**************************************" << std::endl;

    auto t0 = std::chrono::high_resolution_clock::now();
    std::cout << "Reading all sketches in filelist using all " << a.threads << " threads..." << std::endl;
    std::vector<std::string> names;
    {
        std::ifstream fl(a.file_list);
        if (!fl.is_open()) std::cerr << "Could not open the filelist: " << a.file_list << std::endl;
        std::string line;
        while (std::getline(fl, line)) names.push_back(line);
    }
    const uint64_t n = names.size();
    std::cout << "Total number of sketches to read: " << n << std::endl;
    std::vector<std::vector<uint64_t>> sketches(n);
    std::atomic<uint64_t> malformed{0};
    {
        std::vector<std::thread> pool;
        const uint64_t chunk = n / (uint64_t)a.threads;
        for (int t = 0; t < a.threads; ++t) {
            const uint64_t b = (uint64_t)t * chunk, e = (t == a.threads - 1) ? n : (uint64_t)(t + 1) * chunk;
            pool.emplace_back([&, b, e] {
                for (uint64_t i = b; i < e; ++i) {
                    int st = 0;
                    sketches[i] = read_mins(names[i], true, &st);
                    if (st == yh_sig::READ_MALFORMED) malformed.fetch_add(1);
                }
            });
        }
        for (auto& th : pool) th.join();
    }
    if (malformed.load()) {  // the reference's json::parse throws here and the process dies (main.cpp:73)
        std::cerr << "Error: " << malformed.load() << " signature file(s) could not be parsed" << std::endl;
        return 1;
    }
    std::cout << "All sketches read" << std::endl;
    std::vector<uint64_t> offsets(n + 1, 0);
    std::vector<uint64_t> empty_ids;
    for (uint64_t i = 0; i < n; ++i) {
        offsets[i + 1] = offsets[i] + sketches[i].size();
        if (sketches[i].empty()) empty_ids.push_back(i);
    }
    std::cout << "Number of empty sketches: " << empty_ids.size() << std::endl;
    if (!empty_ids.empty()) {
        std::cout << "Empty sketch ids: ";
        for (uint64_t i : empty_ids) std::cout << i << " ";
        std::cout << std::endl;
    }
    std::vector<uint64_t> values(offsets[n]);
    std::vector<uint32_t> sizes(n);
    for (uint64_t i = 0; i < n; ++i) {
        std::copy(sketches[i].begin(), sketches[i].end(), values.begin() + offsets[i]);
        sizes[i] = (uint32_t)sketches[i].size();
        std::vector<uint64_t>().swap(sketches[i]);
    }
    std::cout << "Time taken to read all sketches: " << ms_since(t0) << " milliseconds" << std::endl;

    t0 = std::chrono::high_resolution_clock::now();
    std::cout << "Building index from sketches..." << std::endl;
    yh_db* db = nullptr;
    int rc = yh_db_create(values.data(), offsets.data(), n, 0, YH_DB_PAIRWISE_ONLY, &db);
    if (rc != YH_OK) { std::cerr << "yh_db_create failed: " << yh_last_error() << std::endl; return 1; }
    uint64_t n_distinct = 0, n_single = 0, n_index = 0;
    yh_index_stats(db, &n_distinct, &n_single, &n_index);
    std::cout << "Total number of distinct hashes: " << n_distinct << std::endl;
    std::cout << "Total number of distinct hashes that appear in only one sketch: " << n_single << std::endl;
    std::cout << "Size of the index: " << n_index << std::endl;
    std::cout << "Time taken to build index: " << ms_since(t0) << " milliseconds" << std::endl;

    t0 = std::chrono::high_resolution_clock::now();
    std::cout << "Computing intersection matrix..." << std::endl;
    uint64_t n_pairs = 0;
    rc = yh_pairwise(db, a.c, 0, n, 0, nullptr, nullptr, nullptr, &n_pairs);
    std::vector<uint32_t> pi(std::max<uint64_t>(n_pairs, 1)), pj(pi.size()), pc(pi.size());
    if (rc == YH_OK) rc = yh_pairwise(db, a.c, 0, n, pi.size(), pi.data(), pj.data(), pc.data(), &n_pairs);
    if (rc != YH_OK) { std::cerr << "yh_pairwise failed: " << yh_last_error() << std::endl; yh_db_destroy(db); return 1; }
    yh_db_destroy(db);

    // comparison files: the reference's (pass, thread) row blocks
    const int64_t per_pass = (int64_t)std::ceil(1.0 * (double)n / a.passes);
    uint64_t k = 0;  // pairs are sorted by (i, j)
    for (int p = 0; p < a.passes; ++p) {
        const int64_t start = (int64_t)p * per_pass;
        const int64_t end = (p == a.passes - 1) ? (int64_t)n : (int64_t)(p + 1) * per_pass;
        const int64_t rows = end - start;
        const int64_t chunk = rows > 0 ? rows / a.threads : 0;
        for (int t = 0; t < a.threads; ++t) {
            const int64_t b = start + (int64_t)t * chunk;
            const int64_t e = (t == a.threads - 1) ? end : start + (int64_t)(t + 1) * chunk;
            std::string id = std::to_string(t);
            while (id.size() < 3) id = "0" + id;
            std::ofstream out(a.working_directory + "/" + std::to_string(p) + "_" + id + ".txt");
            while (k < n_pairs && (int64_t)pi[k] < b) ++k;
            for (; k < n_pairs && (int64_t)pi[k] < e; ++k) {
                const uint32_t i = pi[k], j = pj[k], m = pc[k];
                const double jaccard = 1.0 * m / ((size_t)sizes[i] + (size_t)sizes[j] - m);
                const double c_ij = 1.0 * m / (size_t)sizes[i];
                const double c_ji = 1.0 * m / (size_t)sizes[j];
                out << i << "," << j << "," << jaccard << "," << c_ij << "," << c_ji << std::endl;
            }
        }
        std::cout << "Pass " << p + 1 << "/" << a.passes << " done." << std::endl;
    }
    std::cout << "Time taken to compute intersection matrix: " << ms_since(t0) << " milliseconds" << std::endl;

    t0 = std::chrono::high_resolution_clock::now();
    std::cout << "Starting yacht train..." << std::endl;
    std::vector<uint32_t> selected(std::max<uint64_t>(n, 1));
    uint64_t n_sel = 0;
    rc = yh_train_select(sizes.data(), n, pi.data(), pj.data(), n_pairs, selected.data(), &n_sel);
    if (rc != YH_OK) { std::cerr << "yh_train_select failed: " << yh_last_error() << std::endl; return 1; }
    std::cout << "Writing to output file.." << std::endl;
    {
        std::ofstream out(a.output_filename);
        for (uint64_t s = 0; s < n_sel; ++s) out << names[selected[s]] << std::endl;
    }
    std::cout << "Time taken to do yacht train: " << ms_since(t0) << " milliseconds" << std::endl;
    return 0;
}
