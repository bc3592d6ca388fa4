// yh_sigread.hip -- host-side ingest for in-process callers: the "mins" of many .sig files, read and
// parsed by a pool of host threads (the stage in front of yh_db_create; replaces the reference's
// read_sketches / read_sketches_one_chunk, src/cpp/main.cpp:89-124).  No device code.
#include "yh_common.h"
#include "yh_sigread.h"

#include <string.h>

#include <atomic>
#include <thread>

#include <unistd.h>
#include <zlib.h>

struct yh_sig_batch {
    std::vector<std::vector<uint64_t>> mins;
    std::vector<uint8_t> status;  // yh_sig::READ_*
};

// ---- the two host passes of `yacht train` in front of the core, threaded ------------------------------
namespace {

bool read_whole(const char* path, std::string* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    out->clear();
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, got);
    fclose(f);
    return true;
}

// gzip member(s) -> text (zlib, window 15 + 16 = gzip wrapper); false on a corrupt stream
bool gunzip_buffer(const std::string& in, std::string* out) {
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) return false;
    out->clear();
    z.next_in = (Bytef*)in.data();
    z.avail_in = (uInt)in.size();
    char buf[1 << 16];
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        z.next_out = (Bytef*)buf;
        z.avail_out = sizeof buf;
        rc = inflate(&z, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&z); return false; }
        out->append(buf, sizeof buf - z.avail_out);
        if (rc == Z_STREAM_END && z.avail_in > 0) {  // a further member
            if (inflateReset(&z) != Z_OK) break;
            rc = Z_OK;
        } else if (rc == Z_OK && z.avail_in == 0 && z.avail_out != 0) break;  // truncated
    }
    inflateEnd(&z);
    return rc == Z_STREAM_END;
}

template <class F>
void for_each_threaded(uint64_t n, int threads, F&& body) {
    std::atomic<uint64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const uint64_t i = next.fetch_add(8);
            if (i >= n) break;
            for (uint64_t k = i; k < std::min<uint64_t>(i + 8, n); ++k) body(k);
        }
    };
    const int nt = (int)std::min<uint64_t>((uint64_t)std::max(threads, 1), std::max<uint64_t>(n / 8, 1));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}

}  // namespace


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

struct yh_sig_meta {
    std::vector<yh_sig::Meta> m;
    // yh_sig_meta_read_keep: what the train core would read from the same files (record 0, signature 0, no ksize check),
    // taken from the text while it is in memory -- yh_sig_meta_take_batch hands it on
    bool kept = false;
    std::vector<std::vector<uint64_t>> mins;
    std::vector<uint8_t> mins_status;
};

int yh_gunzip_files(const char* const* paths, uint64_t n_paths, int threads, uint8_t* status) {
    if (n_paths && (!paths || !status)) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    std::atomic<bool> oom{false};
    for_each_threaded(n_paths, threads, [&](uint64_t k) {
        status[k] = 1;
        try {
            const std::string path = paths[k] ? paths[k] : "";
            if (path.size() < 4 || path.compare(path.size() - 3, 3, ".gz") != 0) return;
            std::string packed, text;
            if (!read_whole(path.c_str(), &packed) || !gunzip_buffer(packed, &text)) return;
            const std::string dst = path.substr(0, path.size() - 3);
            FILE* f = fopen(dst.c_str(), "wb");
            if (!f) return;
            const bool ok = fwrite(text.data(), 1, text.size(), f) == text.size();
            if (fclose(f) != 0 || !ok) { (void)unlink(dst.c_str()); return; }
            (void)unlink(path.c_str());
            status[k] = 0;
        } catch (...) {
            oom.store(true);
        }
    });
    if (oom.load()) { yh_set_error("out of host memory while decompressing"); return YH_ERR_OOM; }
    return YH_OK;
}

static int sig_meta_read(const char* const* paths, uint64_t n_paths, int ksize, int threads, bool keep, yh_sig_meta** out) {
    if (!out || (n_paths && !paths)) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    yh_sig_meta* b = new (std::nothrow) yh_sig_meta;
    if (!b) { yh_set_error("out of host memory"); return YH_ERR_OOM; }
    b->m.resize(n_paths);
    b->kept = keep;
    if (keep) {
        b->mins.resize(n_paths);
        b->mins_status.assign(n_paths, (uint8_t)yh_sig::READ_CANNOT_OPEN);
    }
    std::atomic<bool> oom{false};
    for_each_threaded(n_paths, threads, [&](uint64_t k) {
        try {
            std::string text;
            if (!read_whole(paths[k] ? paths[k] : "", &text)) { b->m[k].status = yh_sig::META_CANNOT_OPEN; return; }
            if (keep) {  // (the train core reads the file as it is: a gzipped one does not parse there)
                int st = 0;
                b->mins[k] = yh_sig::mins_from_text(text, &st);
                b->mins_status[k] = (uint8_t)st;
            }
            if (text.size() >= 2 && (unsigned char)text[0] == 0x1f && (unsigned char)text[1] == 0x8b) {
                std::string plain;
                if (!gunzip_buffer(text, &plain)) { b->m[k].status = yh_sig::META_MALFORMED; return; }
                text.swap(plain);
            }
            b->m[k] = yh_sig::parse_meta(text, ksize);
        } catch (...) {
            oom.store(true);
        }
    });
    if (oom.load()) { delete b; yh_set_error("out of host memory while reading signature metadata"); return YH_ERR_OOM; }
    *out = b;
    return YH_OK;
}

int yh_sig_meta_read(const char* const* paths, uint64_t n_paths, int ksize, int threads, yh_sig_meta** out) {
    return sig_meta_read(paths, n_paths, ksize, threads, false, out);
}

int yh_sig_meta_read_keep(const char* const* paths, uint64_t n_paths, int ksize, int threads, yh_sig_meta** out) {
    return sig_meta_read(paths, n_paths, ksize, threads, true, out);
}

int yh_sig_meta_take_batch(yh_sig_meta* b, yh_sig_batch** out) {
    if (!b || !out) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    if (!b->kept) { yh_set_error("this metadata set was not read with yh_sig_meta_read_keep (or its sketches were taken already)"); return YH_ERR_UNSUPPORTED; }
    yh_sig_batch* r = new (std::nothrow) yh_sig_batch;
    if (!r) { yh_set_error("out of host memory"); return YH_ERR_OOM; }
    r->mins.swap(b->mins);
    r->status.swap(b->mins_status);
    b->kept = false;
    *out = r;
    return YH_OK;
}

int yh_sig_meta_get(const yh_sig_meta* b, uint8_t* status, uint64_t* n_hashes, uint64_t* scaled, double* mean_abundance,
                    uint8_t* has_abundance, char* md5 /* [n][33] */, uint64_t* name_offsets /* [n + 1] */) {
    if (!b || !status || !n_hashes || !scaled || !mean_abundance || !has_abundance || !md5 || !name_offsets) {
        yh_set_error("null argument");
        return YH_ERR_INVALID_ARG;
    }
    uint64_t at = 0;
    for (size_t i = 0; i < b->m.size(); ++i) {
        const yh_sig::Meta& x = b->m[i];
        status[i] = (uint8_t)x.status;
        n_hashes[i] = x.n_hashes;
        scaled[i] = x.scaled;
        mean_abundance[i] = x.mean_abundance;
        has_abundance[i] = x.has_abundance ? 1 : 0;
        memset(md5 + 33 * i, 0, 33);
        memcpy(md5 + 33 * i, x.md5.data(), std::min<size_t>(x.md5.size(), 32));
        name_offsets[i] = at;
        at += x.name.size();
    }
    name_offsets[b->m.size()] = at;
    return YH_OK;
}

int yh_sig_meta_names(const yh_sig_meta* b, char* names) {
    if (!b || !names) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    uint64_t at = 0;
    for (const auto& x : b->m) {
        memcpy(names + at, x.name.data(), x.name.size());
        at += x.name.size();
    }
    return YH_OK;
}

int yh_sig_meta_destroy(yh_sig_meta* b) {
    delete b;
    return YH_OK;
}

}  // extern "C"
