// yh_sigread.hip -- host-side ingest for in-process callers: the "mins" of many .sig files, read and
// parsed by a pool of host threads (the stage in front of yh_db_create; replaces the reference's
// read_sketches / read_sketches_one_chunk, src/cpp/main.cpp:89-124).  No device code.
#include "yh_common.h"
#include "yh_sigread.h"

#include <string.h>

#include <atomic>
#include <thread>

struct yh_sig_batch {
    std::vector<std::vector<uint64_t>> mins;
    std::vector<uint8_t> status;  // yh_sig::READ_*
};

extern "C" {

int yh_sig_batch_read(const char* const* paths, uint64_t n_paths, int threads, yh_sig_batch** out) {
    if (!out || (n_paths && !paths)) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    if (threads < 1) { yh_set_error("number of threads must be at least 1"); return YH_ERR_INVALID_ARG; }
    yh_sig_batch* b = new (std::nothrow) yh_sig_batch;
    if (!b) { yh_set_error("out of host memory"); return YH_ERR_OOM; }
    b->mins.resize(n_paths);
    b->status.assign(n_paths, 0);
    std::atomic<uint64_t> next{0};
    std::atomic<bool> oom{false};
    auto work = [&]() {
        try {  // (an exception leaving a std::thread ends the host process)
            for (;;) {  // files vary in size: a shared cursor instead of the reference's fixed chunks
                const uint64_t i = next.fetch_add(16);
                if (i >= n_paths || oom.load()) break;
                for (uint64_t k = i; k < std::min<uint64_t>(i + 16, n_paths); ++k) {
                    int st = 0;
                    b->mins[k] = yh_sig::read_mins(paths[k] ? paths[k] : "", false, &st);
                    b->status[k] = (uint8_t)st;
                }
            }
        } catch (...) {
            oom.store(true);
        }
    };
    const int nt = (int)std::min<uint64_t>((uint64_t)threads, std::max<uint64_t>(n_paths / 16, 1));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (oom.load()) {
        delete b;
        yh_set_error("out of host memory while reading the signature files");
        return YH_ERR_OOM;
    }
    *out = b;
    return YH_OK;
}

int yh_sig_batch_status(const yh_sig_batch* b, uint8_t* status) {
    if (!b || (!status && !b->status.empty())) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    if (!b->status.empty()) memcpy(status, b->status.data(), b->status.size());
    return YH_OK;
}

int yh_sig_batch_sizes(const yh_sig_batch* b, uint64_t* offsets) {
    if (!b || !offsets) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    uint64_t acc = 0;
    offsets[0] = 0;
    for (size_t i = 0; i < b->mins.size(); ++i) { acc += b->mins[i].size(); offsets[i + 1] = acc; }
    return YH_OK;
}

int yh_sig_batch_values(const yh_sig_batch* b, uint64_t* values) {
    if (!b || (!values && !b->mins.empty())) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    uint64_t at = 0;
    for (const auto& m : b->mins) {
        if (!m.empty()) memcpy(values + at, m.data(), m.size() * sizeof(uint64_t));
        at += m.size();
    }
    return YH_OK;
}

int yh_sig_batch_destroy(yh_sig_batch* b) {
    delete b;
    return YH_OK;
}

}  // extern "C"
