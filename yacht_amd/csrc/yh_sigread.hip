// yh_sigread.hip -- host-side ingest for in-process callers: the "mins" of many .sig files, read and
// parsed by a pool of host threads (the stage in front of yh_db_create; replaces the reference's
// read_sketches / read_sketches_one_chunk, src/cpp/main.cpp:89-124).  No device code.
#include "yh_common.h"
#include "yh_sigread.h"
#include "yh_pack.h"

#include <string.h>

#include <atomic>
#include <thread>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <time.h>
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
    std::vector<uint64_t> at(b->mins.size() + 1, 0);
    for (size_t i = 0; i < b->mins.size(); ++i) at[i + 1] = at[i] + b->mins[i].size();
    // (3.3e8 hashes at GTDB scale: 2.7 GB into fresh pages -- a few threads instead of one)
    const int threads = at.back() > (1u << 22) ? (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u) : 1;
    for_each_threaded(b->mins.size(), threads, [&](uint64_t i) {
        const auto& m = b->mins[i];
        if (!m.empty()) memcpy(values + at[i], m.data(), m.size() * sizeof(uint64_t));
    });
    return YH_OK;
}

// The parsed sketches as a PACKED CSR (yh_db_create_packed's input), straight from the files' own vectors: no CSR of 8 bytes per
// hash is made in between (round 6: `yacht train` uploads and writes the packed form).  Two-call sizing as yh_csr_pack;
// YH_ERR_UNSORTED when a file's mins are not strictly ascending (the caller then takes the CSR path, whose reader tolerates that).
int yh_sig_batch_pack(const yh_sig_batch* b, void* packed, uint64_t cap_bytes, uint64_t* packed_bytes, int threads) {
    if (!b || !packed_bytes) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    try {
        const size_t n = b->mins.size();
        std::vector<uint64_t> offsets(n + 1, 0);
        std::vector<const uint64_t*> parts(n, nullptr);
        for (size_t i = 0; i < n; ++i) {
            offsets[i + 1] = offsets[i] + b->mins[i].size();
            parts[i] = b->mins[i].data();
        }
        return yh_csr_pack_parts(reinterpret_cast<const u64* const*>(parts.data()), reinterpret_cast<const u64*>(offsets.data()), n, packed, cap_bytes,
                                 reinterpret_cast<u64*>(packed_bytes), threads);
    } catch (const std::bad_alloc&) {
        yh_set_error("yh_sig_batch_pack: out of host memory");
        return YH_ERR_OOM;
    }
}

int yh_sig_batch_destroy(yh_sig_batch* b) {
    delete b;
    return YH_OK;
}

struct yh_sig_meta {
    std::vector<yh_sig::Meta> m;
    std::vector<std::string> rel_paths;  // yh_zip_sig_ingest: where (under out_dir) every signature member was / would be written
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
        has_abundance[i] = (uint8_t)((x.has_abundance ? 1 : 0) | (x.is_first ? 2 : 0));
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

int yh_sig_meta_count(const yh_sig_meta* b, uint64_t* n) {
    if (!b || !n) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    *n = b->m.size();
    return YH_OK;
}

int yh_sig_meta_paths(const yh_sig_meta* b, uint64_t* path_offsets, char* paths) {
    if (!b || !path_offsets) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    uint64_t at = 0;
    for (size_t i = 0; i < b->m.size(); ++i) {
        const std::string& r = i < b->rel_paths.size() ? b->rel_paths[i] : std::string();
        path_offsets[i] = at;
        if (paths) memcpy(paths + at, r.data(), r.size());
        at += r.size();
    }
    path_offsets[b->m.size()] = at;
    return YH_OK;
}

// ---- a sourmash .zip database in ONE pass (`yacht train`: make_training_data_from_sketches.py:107-133, utils.py:201-221, :499-509) ----
namespace {

struct ZipEntry {
    std::string name;
    uint16_t method = 0;
    uint32_t crc = 0;
    uint64_t comp = 0, uncomp = 0, lho = 0;
};
inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const unsigned char* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

bool pread_all(int fd, void* buf, size_t n, uint64_t off) {
    char* q = (char*)buf;
    while (n) {
        const ssize_t got = pread(fd, q, n, (off_t)off);
        if (got <= 0) return false;
        q += got; off += (uint64_t)got; n -= (size_t)got;
    }
    return true;
}

// the central directory (zip64 too: a GTDB database has more than 65 535 members)
bool zip_directory(int fd, uint64_t fsize, std::vector<ZipEntry>* out, std::string* err) {
    const uint64_t tail = std::min<uint64_t>(fsize, 65557 + 20);
    std::vector<unsigned char> buf(tail);
    if (tail < 22 || !pread_all(fd, buf.data(), tail, fsize - tail)) { *err = "not a zip archive (too short)"; return false; }
    long at = -1;
    for (long i = (long)tail - 22; i >= 0; --i)
        if (rd32(&buf[i]) == 0x06054b50u) { at = i; break; }
    if (at < 0) { *err = "not a zip archive (no end-of-central-directory record)"; return false; }
    uint64_t n = rd16(&buf[at + 10]), cd_size = rd32(&buf[at + 12]), cd_off = rd32(&buf[at + 16]);
    const uint64_t eocd_pos = fsize - tail + (uint64_t)at;
    uint64_t cd_end = eocd_pos;  // where the central directory ends: right in front of the end record (or of its zip64 pair)
    if (n == 0xffffu || cd_size == 0xffffffffu || cd_off == 0xffffffffu) {
        if (at < 20 || rd32(&buf[at - 20]) != 0x07064b50u) { *err = "zip64 locator missing"; return false; }
        // the zip64 end record sits right in front of its locator (as Python's zipfile takes it: an archive with data put in
        // front of it has every stated offset shifted, this one too)
        if (eocd_pos < 20 + 56) { *err = "bad zip64 end-of-central-directory record"; return false; }
        const uint64_t z64 = eocd_pos - 20 - 56;
        unsigned char rec[56];
        if (!pread_all(fd, rec, 56, z64) || rd32(rec) != 0x06064b50u) { *err = "bad zip64 end-of-central-directory record"; return false; }
        n = rd64(rec + 32);
        cd_size = rd64(rec + 40);
        cd_off = rd64(rec + 48);
        cd_end = z64;
    }
    // every value above is the archive's own claim: check it against the file before anything is sized by it
    if (cd_size > cd_end || cd_off > cd_end - cd_size) { *err = "central directory outside the file"; return false; }
    if (n > cd_size / 46) { *err = "central directory too small for its entry count"; return false; }
    const uint64_t shift = cd_end - cd_size - cd_off;  // bytes put in front of the archive (a self-extracting stub): 0 normally
    cd_off += shift;
    std::vector<unsigned char> cd;
    try { cd.resize(cd_size); } catch (...) { *err = "out of host memory for the central directory"; return false; }
    if (cd_size && !pread_all(fd, cd.data(), cd_size, cd_off)) { *err = "cannot read the central directory"; return false; }
    out->clear();
    out->reserve(n);
    uint64_t p = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (p + 46 > cd_size || rd32(&cd[p]) != 0x02014b50u) { *err = "bad central directory entry"; return false; }
        ZipEntry e;
        e.method = rd16(&cd[p + 10]);
        e.crc = rd32(&cd[p + 16]);
        e.comp = rd32(&cd[p + 20]);
        e.uncomp = rd32(&cd[p + 24]);
        const uint16_t nlen = rd16(&cd[p + 28]), xlen = rd16(&cd[p + 30]), clen = rd16(&cd[p + 32]);
        e.lho = rd32(&cd[p + 42]);
        if (p + 46 + nlen + xlen + clen > cd_size) { *err = "bad central directory entry"; return false; }
        e.name.assign((const char*)&cd[p + 46], nlen);
        // zip64 extra field: the 8-byte forms of the fields that read 0xffffffff, in this order
        for (uint64_t x = p + 46 + nlen, xe = x + xlen; x + 4 <= xe;) {
            const uint16_t id = rd16(&cd[x]), sz = rd16(&cd[x + 2]);
            if (id == 0x0001u) {
                uint64_t q = x + 4;
                if (e.uncomp == 0xffffffffu && q + 8 <= xe) { e.uncomp = rd64(&cd[q]); q += 8; }
                if (e.comp == 0xffffffffu && q + 8 <= xe) { e.comp = rd64(&cd[q]); q += 8; }
                if (e.lho == 0xffffffffu && q + 8 <= xe) { e.lho = rd64(&cd[q]); q += 8; }
            }
            x += 4 + (uint64_t)sz;
        }
        if (e.lho > ~(uint64_t)0 - shift) { *err = "bad central directory entry"; return false; }
        e.lho += shift;
        out->push_back(std::move(e));
        p += 46 + (uint64_t)nlen + xlen + clen;
    }
    return true;
}

// A per-thread byte buffer that grows without being zero-filled (std::string::resize clears what inflate is about to
// overwrite: 8 GB of memset over a GTDB database).
struct RawBuf {
    char* p = nullptr;
    size_t cap = 0, n = 0;
    ~RawBuf() { free(p); }
    bool room(size_t want) {
        if (want <= cap) return true;
        size_t c = cap ? cap : (size_t)1 << 16;
        while (c < want) c *= 2;
        char* q = (char*)realloc(p, c);
        if (!q) return false;
        p = q;
        cap = c;
        return true;
    }
};
// libdeflate, when the machine has it (no header needed for three functions; zlib otherwise): 2-3 x zlib's inflate rate on
// whole buffers, which is what a zip member is.  Looked up once.
struct Deflate {
    void* (*alloc)(void) = nullptr;
    void (*release)(void*) = nullptr;
    int (*gzip)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    int (*raw)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    Deflate() {
        static const bool off = [] { const char* e = yh_tune_env("YH_NO_LIBDEFLATE"); return e && e[0] == '1'; }();
        if (off) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void* (*)(void))dlsym(h, "libdeflate_alloc_decompressor");
        release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        gzip = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_gzip_decompress");
        raw = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
        if (!alloc || !release || !gzip || !raw) alloc = nullptr;
    }
    bool ok() const { return alloc != nullptr; }
};
const Deflate& deflate_lib() { static const Deflate d; return d; }
// CRC-32 of a member's bytes (the central directory names it; Python's zipfile checks it on every read)
uint32_t crc32_of(const void* p, size_t n) {
    const Deflate& d = deflate_lib();
    if (d.ok() && d.crc) return d.crc(0u, p, n);
    uLong c = crc32(0L, Z_NULL, 0);
    const unsigned char* q = (const unsigned char*)p;
    while (n) {
        const uInt step = (uInt)std::min<size_t>(n, (size_t)1 << 30);
        c = crc32(c, q, step);
        q += step;
        n -= step;
    }
    return (uint32_t)c;
}
struct Decompressor {  // one per thread
    void* d = nullptr;
    Decompressor() { if (deflate_lib().ok()) d = deflate_lib().alloc(); }
    ~Decompressor() { if (d) deflate_lib().release(d); }
};

// ONE gzip member whose trailer names its size (what sourmash writes): a single inflate call straight into `out`.
// false: not that shape (several members, a size that does not fit, a corrupt stream) -- the caller takes the general route.
bool gunzip_oneshot(const char* in, size_t n_in, RawBuf* out) {
    if (n_in < 18 || (unsigned char)in[0] != 0x1f || (unsigned char)in[1] != 0x8b) return false;
    const uint32_t isize = rd32((const unsigned char*)in + n_in - 4);
    if (isize > (1u << 30)) return false;
    if (!out->room((size_t)isize + 1)) return false;
    static thread_local Decompressor dec;
    if (dec.d) {  // (LIBDEFLATE_SUCCESS = 0; anything else -- several members, a bad stream -- takes zlib's opinion below)
        size_t got = 0;
        if (deflate_lib().gzip(dec.d, in, n_in, out->p, (size_t)isize, &got) == 0 && got == isize) { out->n = isize; return true; }
    }
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) return false;
    z.next_in = (Bytef*)in;
    z.avail_in = (uInt)n_in;
    z.next_out = (Bytef*)out->p;
    z.avail_out = (uInt)isize + 1;
    const int rc = inflate(&z, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && z.avail_in == 0 && z.total_out == isize;
    inflateEnd(&z);
    out->n = ok ? isize : 0;
    return ok;
}

// the bytes of one member (stored or deflated), into a raw buffer
// (checked against the member's CRC-32 as the central directory states it: a corrupt member that still inflates is refused,
// as Python's zipfile refuses it -- ADVICE r04)
bool zip_member_raw(int fd, uint64_t fsize, const ZipEntry& e, RawBuf* scratch, RawBuf* out) {
    unsigned char lh[30];
    if (fsize < 30 || e.lho > fsize - 30 || !pread_all(fd, lh, 30, e.lho) || rd32(lh) != 0x04034b50u) return false;
    const uint64_t data = e.lho + 30 + rd16(lh + 26) + rd16(lh + 28);
    if (data > fsize || e.comp > fsize - data) return false;
    if (e.comp >= ((uint64_t)1 << 32) || e.uncomp >= ((uint64_t)1 << 32)) return false;  // (a 4 GB signature: the Python route's business)
    if (e.method == 0) {
        if (!out->room(e.comp + 1)) return false;
        out->n = e.comp;
        if (e.comp && !pread_all(fd, out->p, e.comp, data)) return false;
        return crc32_of(out->p, out->n) == e.crc;
    }
    if (e.method != 8) return false;
    if (!scratch->room(e.comp + 1) || !out->room(e.uncomp + 1)) return false;
    if (e.comp && !pread_all(fd, scratch->p, e.comp, data)) return false;
    static thread_local Decompressor dec;
    if (dec.d) {
        size_t got = 0;
        if (deflate_lib().raw(dec.d, scratch->p, e.comp, out->p, e.uncomp, &got) == 0 && got == e.uncomp) {
            out->n = e.uncomp;
            return crc32_of(out->p, out->n) == e.crc;
        }
    }
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) return false;
    z.next_in = (Bytef*)scratch->p;
    z.avail_in = (uInt)e.comp;
    z.next_out = (Bytef*)out->p;
    z.avail_out = (uInt)e.uncomp;
    const int rc = inflate(&z, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && z.avail_out == 0;
    inflateEnd(&z);
    out->n = ok ? e.uncomp : 0;
    return ok && crc32_of(out->p, out->n) == e.crc;
}

// a member's path must stay inside the working directory
bool safe_member_name(const std::string& n) {
    if (n.empty() || n[0] == '/' || n.find('\\') != std::string::npos) return false;
    size_t i = 0;
    while (i <= n.size()) {
        const size_t j = std::min(n.find('/', i), n.size());
        if (j - i == 2 && n[i] == '.' && n[i + 1] == '.') return false;
        i = j + 1;
    }
    return true;
}
bool ends_with(const std::string& s, const char* suf) {
    const size_t k = strlen(suf);
    return s.size() >= k && s.compare(s.size() - k, k, suf) == 0;
}
void mkdirs(const std::string& dir) {
    for (size_t i = 1; i <= dir.size(); ++i)
        if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0777);
}

}  // namespace

// ---- the members of the archive written to a directory IN THE BACKGROUND -----------------------------------------------
// `yacht train` leaves the unzipped signatures in its working directory (the reference's workflow expects them; this build
// never reads them back).  Creating 85 205 files in one directory is 2.8 s of a 5.9 s command -- the directory's lock, not
// CPU -- so the files are written by threads of their own WHILE the command does the rest: yh_zip_extract_start returns at
// once, yh_zip_extract_wait joins (and reports what failed).
struct yh_zip_job {
    std::thread main;
    std::atomic<int> failed{0};
    std::string error;
    uint64_t n_members = 0;
};

static void zip_extract_run_impl(yh_zip_job* job, std::string zip_path, std::string root, int threads);
// (nothing may leave a std::thread's function: an exception there is std::terminate -- the interpreter gone)
static void zip_extract_run(yh_zip_job* job, std::string zip_path, std::string root, int threads) {
    try {
        zip_extract_run_impl(job, zip_path, root, threads);
    } catch (const std::exception& ex) {
        job->error = zip_path + ": " + ex.what();
        job->failed.store(1);
    } catch (...) {
        job->error = zip_path + ": unexpected failure while extracting";
        job->failed.store(1);
    }
}
static void zip_extract_run_impl(yh_zip_job* job, std::string zip_path, std::string root, int threads) {
    const int fd = open(zip_path.c_str(), O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) { job->error = "cannot open " + zip_path; job->failed.store(1); if (fd >= 0) close(fd); return; }
    struct FdGuard { int fd; ~FdGuard() { close(fd); } } fd_guard{fd};
    const uint64_t fsize = (uint64_t)sb.st_size;
    std::vector<ZipEntry> dir;
    std::string err;
    if (!zip_directory(fd, fsize, &dir, &err)) { job->error = zip_path + ": " + err; job->failed.store(1); return; }
    for (const ZipEntry& e : dir)
        if (!safe_member_name(e.name)) { job->error = "archive member outside the working directory: " + e.name; job->failed.store(1); return; }
    job->n_members = dir.size();
    mkdirs(root);
    std::string last;
    for (const ZipEntry& e : dir) {
        const size_t slash = e.name.rfind('/');
        const std::string d = ends_with(e.name, "/") ? e.name.substr(0, e.name.size() - 1) : slash == std::string::npos ? std::string() : e.name.substr(0, slash);
        if (!d.empty() && d != last) { mkdirs(root + "/" + d); last = d; }
    }
    for_each_threaded(dir.size(), threads, [&](uint64_t i) {
        try {
            static thread_local RawBuf scratch, bytes, plain;
            const ZipEntry& e = dir[i];
            if (ends_with(e.name, "/")) return;
            if (!zip_member_raw(fd, fsize, e, &scratch, &bytes)) { job->failed.store(1); return; }
            std::string out_name = e.name;
            const char* text = bytes.p;
            size_t text_n = bytes.n;
            std::string general;
            if (ends_with(e.name, ".sig.gz")) {
                if (gunzip_oneshot(bytes.p, bytes.n, &plain)) { out_name.resize(out_name.size() - 3); text = plain.p; text_n = plain.n; }
                else if (gunzip_buffer(std::string(bytes.p, bytes.n), &general)) { out_name.resize(out_name.size() - 3); text = general.data(); text_n = general.size(); }
            }
            const std::string path = root + "/" + out_name;
            const int ofd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
            bool ok = ofd >= 0;
            for (size_t off = 0; ok && off < text_n;) {
                const ssize_t w = write(ofd, text + off, text_n - off);
                if (w <= 0) ok = false; else off += (size_t)w;
            }
            if (ofd >= 0 && close(ofd) != 0) ok = false;
            if (!ok) job->failed.store(1);
        } catch (...) {
            job->failed.store(1);
        }
    });
    if (job->failed.load() && job->error.empty()) job->error = zip_path + ": a member could not be read, inflated or written";
}

int yh_zip_extract_start(const char* zip_path, const char* out_dir, int threads, yh_zip_job** out) {
    if (!zip_path || !out_dir || !out) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    *out = nullptr;
    if (access(zip_path, R_OK) != 0) { yh_set_error("cannot open %s", zip_path); return YH_ERR_INVALID_ARG; }
    yh_zip_job* job = new (std::nothrow) yh_zip_job;
    if (!job) { yh_set_error("out of host memory"); return YH_ERR_OOM; }
    try {
        job->main = std::thread(zip_extract_run, job, std::string(zip_path), std::string(out_dir), std::max(threads, 1));
    } catch (...) {
        delete job;
        yh_set_error("cannot start a thread");
        return YH_ERR_OOM;
    }
    *out = job;
    return YH_OK;
}

int yh_zip_extract_wait(yh_zip_job* job, uint64_t* n_members) {
    if (!job) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    if (job->main.joinable()) job->main.join();
    const bool failed = job->failed.load() != 0;
    if (n_members) *n_members = job->n_members;
    if (failed) yh_set_error("%s", job->error.c_str());
    delete job;
    return failed ? YH_ERR_INVALID_ARG : YH_OK;
}

static int zip_sig_ingest_impl(const char* zip_path, const char* out_dir, int ksize, int threads, yh_sig_meta** out);
// (no exception crosses the C boundary: a corrupt archive that makes a vector throw is an error code, not std::terminate)
int yh_zip_sig_ingest(const char* zip_path, const char* out_dir, int ksize, int threads, yh_sig_meta** out) {
    if (!zip_path || !out) { yh_set_error("null argument"); return YH_ERR_INVALID_ARG; }
    *out = nullptr;
    try {
        return zip_sig_ingest_impl(zip_path, out_dir, ksize, threads, out);
    } catch (const std::bad_alloc&) {
        yh_set_error("out of host memory while reading %s", zip_path);
        return YH_ERR_OOM;
    } catch (const std::exception& ex) {
        yh_set_error("%s: %s", zip_path, ex.what());
        return YH_ERR_INVALID_ARG;
    } catch (...) {
        yh_set_error("%s: unexpected failure", zip_path);
        return YH_ERR_INVALID_ARG;
    }
}
static int zip_sig_ingest_impl(const char* zip_path, const char* out_dir, int ksize, int threads, yh_sig_meta** out) {
    const int fd = open(zip_path, O_RDONLY);
    if (fd < 0) { yh_set_error("cannot open %s", zip_path); return YH_ERR_INVALID_ARG; }
    struct FdGuard { int fd; ~FdGuard() { close(fd); } } fd_guard{fd};  // (closed on every way out, a thrown one included)
    struct stat sb;
    if (fstat(fd, &sb) != 0) { yh_set_error("cannot stat %s", zip_path); return YH_ERR_INVALID_ARG; }
    const uint64_t fsize = (uint64_t)sb.st_size;
    std::vector<ZipEntry> dir;
    std::string err;
    if (!zip_directory(fd, fsize, &dir, &err)) { yh_set_error("%s: %s", zip_path, err.c_str()); return YH_ERR_INVALID_ARG; }
    for (const ZipEntry& e : dir)
        if (!safe_member_name(e.name)) { yh_set_error("archive member outside the working directory: %s", e.name.c_str()); return YH_ERR_INVALID_ARG; }
    // the signature members, in central-directory order: "signatures/<x>.sig[.gz]"
    std::vector<uint64_t> sig_of(dir.size(), ~(uint64_t)0);
    uint64_t n_sig = 0;
    for (size_t i = 0; i < dir.size(); ++i) {
        const std::string& n = dir[i].name;
        if (n.compare(0, 11, "signatures/") == 0 && (ends_with(n, ".sig") || ends_with(n, ".sig.gz"))) sig_of[i] = n_sig++;
    }
    yh_sig_meta* b = new (std::nothrow) yh_sig_meta;
    if (!b) { yh_set_error("out of host memory"); return YH_ERR_OOM; }
    struct MetaGuard { yh_sig_meta* p; ~MetaGuard() { delete p; } } meta_guard{b};  // (released unless handed to the caller)
    b->m.resize(n_sig);
    b->rel_paths.resize(n_sig);
    b->kept = true;
    b->mins.resize(n_sig);
    b->mins_status.assign(n_sig, (uint8_t)yh_sig::READ_CANNOT_OPEN);
    const std::string root = out_dir ? std::string(out_dir) : std::string();
    if (out_dir) {  // every directory once, before the threads start (directory entries and the parents of the members)
        mkdirs(root);
        std::string last;
        for (const ZipEntry& e : dir) {
            const size_t slash = e.name.rfind('/');
            const std::string d = slash == std::string::npos ? std::string() : e.name.substr(0, slash);
            if (!d.empty() && d != last) { mkdirs(root + "/" + d); last = d; }
        }
    }
    std::atomic<bool> oom{false};
    std::atomic<int> io_failed{0};
    // (YH_TRACE_BUILD=1 behind the tuning gate: where the threads' time went, summed over the threads)
    static const bool trace = [] { const char* e = yh_tune_env("YH_TRACE_BUILD"); return e && e[0] == '1'; }();
    std::atomic<uint64_t> ns_read{0}, ns_gunzip{0}, ns_parse{0}, ns_write{0};
    auto now_ns = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; };
    const uint64_t t_begin = now_ns();
    for_each_threaded(dir.size(), threads, [&](uint64_t i) {
        try {
            static thread_local RawBuf scratch, bytes, plain;
            const ZipEntry& e = dir[i];
            if (ends_with(e.name, "/")) return;
            const uint64_t k = sig_of[i];
            if (k == ~(uint64_t)0 && !out_dir) return;  // (not a signature, nothing to write: not read at all)
            uint64_t t0 = trace ? now_ns() : 0;
            if (!zip_member_raw(fd, fsize, e, &scratch, &bytes)) {
                io_failed.store(1);
                if (k != ~(uint64_t)0) b->m[k].status = yh_sig::META_CANNOT_OPEN;
                return;
            }
            if (trace) { const uint64_t t1 = now_ns(); ns_read += t1 - t0; t0 = t1; }
            std::string out_name = e.name;
            const char* text = bytes.p;
            size_t text_n = bytes.n;
            std::string general;  // (only for streams the one-shot form does not take)
            if (ends_with(e.name, ".sig.gz")) {
                if (gunzip_oneshot(bytes.p, bytes.n, &plain)) {
                    out_name.resize(out_name.size() - 3);
                    text = plain.p;
                    text_n = plain.n;
                } else if (gunzip_buffer(std::string(bytes.p, bytes.n), &general)) {  // several members, an odd trailer
                    out_name.resize(out_name.size() - 3);
                    text = general.data();
                    text_n = general.size();
                }  // (one that does not inflate stays as it is)
            }
            if (trace) { const uint64_t t1 = now_ns(); ns_gunzip += t1 - t0; t0 = t1; }
            if (out_dir) {
                const std::string path = root + "/" + out_name;
                const int ofd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
                bool ok = ofd >= 0;
                for (size_t off = 0; ok && off < text_n;) {
                    const ssize_t w = write(ofd, text + off, text_n - off);
                    if (w <= 0) ok = false; else off += (size_t)w;
                }
                if (ofd >= 0 && close(ofd) != 0) ok = false;
                if (!ok) io_failed.store(1);
                if (trace) { const uint64_t t1 = now_ns(); ns_write += t1 - t0; t0 = t1; }
            }
            if (k == ~(uint64_t)0) return;
            b->rel_paths[k] = out_name;
            // one scan for both readers: the metadata of the signature of this k-mer size, and -- what the train core reads from
            // the file -- the mins of record 0 / signature 0, captured on the way when the whole file parses
            std::vector<uint64_t> first;
            bool have_first = false;
            b->m[k] = yh_sig::parse_meta(text, text_n, ksize, &first, &have_first);
            const int st_meta = b->m[k].status;
            if (have_first && (st_meta == yh_sig::META_OK || st_meta == yh_sig::META_NOT_ONE || st_meta == yh_sig::META_EMPTY)) {
                b->mins[k] = std::move(first);
                b->mins_status[k] = (uint8_t)yh_sig::READ_OK;
            } else {
                int st = 0;
                b->mins[k] = yh_sig::mins_from_text(text, text_n, &st);
                b->mins_status[k] = (uint8_t)st;
            }
            if (trace) ns_parse += now_ns() - t0;
        } catch (...) {
            oom.store(true);
        }
    });
    if (trace)
        fprintf(stderr, "[yh ingest] %zu members, %llu signatures, %d threads: wall %.3f s; thread-seconds: read+inflate %.2f, gunzip %.2f, write %.2f, parse+md5 %.2f\n",
                dir.size(), (unsigned long long)n_sig, threads, (now_ns() - t_begin) * 1e-9, ns_read.load() * 1e-9, ns_gunzip.load() * 1e-9,
                ns_write.load() * 1e-9, ns_parse.load() * 1e-9);
    if (oom.load()) { yh_set_error("out of host memory while reading %s", zip_path); return YH_ERR_OOM; }
    if (io_failed.load()) { yh_set_error("%s: a member could not be read, inflated or written", zip_path); return YH_ERR_INVALID_ARG; }
    meta_guard.p = nullptr;
    *out = b;
    return YH_OK;
}

}  // extern "C"
