// yh_sigread.h -- just enough JSON to reach [0]["signatures"][0]["mins"] of a sourmash .sig file: what the
// reference's train core reads (src/cpp/main.cpp:62-84, read_min_hashes: record 0, signature 0, "mins";
// ksize is NOT checked there; an unreadable file is an empty sketch).  Shared by the drop-in
// executable (train_core_main.cpp) and the library's batch reader (yh_sigread.hip).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace yh_sig {

struct Scanner {
    const char* p;
    const char* e;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool lit(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
    bool string(std::string* out) {
        ws();
        if (p >= e || *p != '"') return false;
        ++p;
        if (out) out->clear();
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) { if (out) out->push_back(p[1]); p += 2; }
            else { if (out) out->push_back(*p); ++p; }
        }
        if (p >= e) return false;
        ++p;
        return true;
    }
    bool skip() {  // any value
        ws();
        if (p >= e) return false;
        if (*p == '"') return string(nullptr);
        if (*p == '{') {
            ++p;
            if (lit('}')) return true;
            do { if (!string(nullptr) || !lit(':') || !skip()) return false; } while (lit(','));
            return lit('}');
        }
        if (*p == '[') {
            ++p;
            if (lit(']')) return true;
            do { if (!skip()) return false; } while (lit(','));
            return lit(']');
        }
        while (p < e && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\t' && *p != '\r') ++p;
        return true;
    }
    // positioned at an object: find `key`, leave the cursor on its value
    bool find_key(const char* key) {
        if (!lit('{')) return false;
        if (lit('}')) return false;
        std::string k;
        do {
            if (!string(&k) || !lit(':')) return false;
            if (k == key) return true;
            if (!skip()) return false;
        } while (lit(','));
        return false;
    }
};

// status (optional): READ_OK, READ_CANNOT_OPEN (the reference prints "Could not open the file!" and goes on with an
// empty sketch, main.cpp:66-71) or READ_MALFORMED (the reference's json::parse / operator[] throws and the program
// dies, main.cpp:73-81: callers must treat it as fatal).
enum { READ_OK = 0, READ_CANNOT_OPEN = 1, READ_MALFORMED = 2 };
// the mins of record 0, signature 0 of a signature file's text (what read_mins returns for the file)
inline std::vector<uint64_t> mins_from_text(const char* text_p, size_t text_n, int* status = nullptr) {
    std::vector<uint64_t> mins;
    if (status) *status = READ_OK;
    Scanner s{text_p, text_p + text_n};
    if (!s.lit('[') || !s.find_key("signatures") || !s.lit('[') || !s.find_key("mins") || !s.lit('[')) {
        if (status) *status = READ_MALFORMED;
        return mins;
    }
    if (s.lit(']')) return mins;
    mins.reserve(4096);
    do {
        s.ws();
        const char* q = s.p;
        uint64_t v = 0;
        while (q < s.e && *q >= '0' && *q <= '9') v = v * 10 + (uint64_t)(*q++ - '0');
        if (q == s.p) {  // not a number
            mins.clear();
            if (status) *status = READ_MALFORMED;
            return mins;
        }
        s.p = q;
        mins.push_back(v);
    } while (s.lit(','));
    if (!s.lit(']')) {  // the array never closes: a truncated file
        mins.clear();
        if (status) *status = READ_MALFORMED;
        return mins;
    }
    bool ascending = true;
    for (size_t i = 1; i < mins.size() && ascending; ++i) ascending = mins[i - 1] < mins[i];
    if (!ascending) {  // sourmash writes ascending unique mins; tolerate other writers
        std::sort(mins.begin(), mins.end());
        mins.erase(std::unique(mins.begin(), mins.end()), mins.end());
    }
    return mins;
}

inline std::vector<uint64_t> mins_from_text(const std::string& text, int* status = nullptr) {
    return mins_from_text(text.data(), text.size(), status);
}

inline std::vector<uint64_t> read_mins(const std::string& path, bool report = true, int* status = nullptr) {
    if (status) *status = READ_OK;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        if (report) std::cerr << "Could not open the file!" << std::endl;
        if (status) *status = READ_CANNOT_OPEN;
        return {};
    }
    std::string text;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long len = ftell(f);
        if (len > 0) text.resize((size_t)len);
        rewind(f);
    }
    size_t got = text.empty() ? 0 : fread(&text[0], 1, text.size(), f);
    if (got < text.size()) text.resize(got);
    for (char buf[1 << 16]; (got = fread(buf, 1, sizeof buf, f)) > 0;) text.append(buf, got);  // (unseekable input)
    fclose(f);
    return mins_from_text(text, status);
}

}  // namespace yh_sig

// ---- metadata of a signature file (the `yacht train` pass in front of the core: utils.py:89-110, :201-221) -----------
// What get_info_from_single_sig takes from sourmash for the ONE signature of a given k-mer size in a file: the
// record's name, the sketch's md5 (sourmash: md5 over str(ksize) followed by every hash in decimal, ascending),
// the mean abundance, the number of hashes and `scaled` (= round(2^64 / max_hash)).
namespace yh_sig {

struct Md5 {  // RFC 1321, streaming
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    uint64_t len = 0;
    unsigned char buf[64];
    size_t fill = 0;
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    // One 64-byte block, fully unrolled (round 6: the table-driven loop with a four-way branch per step ran at 316 MB/s -- 0.22 of the
    // 0.38 ms a 3 900-hash signature took to scan, md5 included; the ingest of 85 205 such members is CPU-bound on exactly that).
    void block(const unsigned char* p) {
        uint32_t m[16];
        memcpy(m, p, 64);  // (little-endian hosts: x86-64, the only target of this build)
        uint32_t A = a, B = b, C = c, D = d;
#define YH_MD5_F(x, y, z) ((z) ^ ((x) & ((y) ^ (z))))
#define YH_MD5_G(x, y, z) ((y) ^ ((z) & ((x) ^ (y))))
#define YH_MD5_H(x, y, z) ((x) ^ (y) ^ (z))
#define YH_MD5_I(x, y, z) ((y) ^ ((x) | ~(z)))
#define YH_MD5_STEP(f, w, x, y, z, data, k, sh) w += f(x, y, z) + (data) + (k); w = rol(w, sh) + x;
        YH_MD5_STEP(YH_MD5_F, A, B, C, D, m[0], 0xd76aa478u, 7)   YH_MD5_STEP(YH_MD5_F, D, A, B, C, m[1], 0xe8c7b756u, 12)
        YH_MD5_STEP(YH_MD5_F, C, D, A, B, m[2], 0x242070dbu, 17)  YH_MD5_STEP(YH_MD5_F, B, C, D, A, m[3], 0xc1bdceeeu, 22)
        YH_MD5_STEP(YH_MD5_F, A, B, C, D, m[4], 0xf57c0fafu, 7)   YH_MD5_STEP(YH_MD5_F, D, A, B, C, m[5], 0x4787c62au, 12)
        YH_MD5_STEP(YH_MD5_F, C, D, A, B, m[6], 0xa8304613u, 17)  YH_MD5_STEP(YH_MD5_F, B, C, D, A, m[7], 0xfd469501u, 22)
        YH_MD5_STEP(YH_MD5_F, A, B, C, D, m[8], 0x698098d8u, 7)   YH_MD5_STEP(YH_MD5_F, D, A, B, C, m[9], 0x8b44f7afu, 12)
        YH_MD5_STEP(YH_MD5_F, C, D, A, B, m[10], 0xffff5bb1u, 17) YH_MD5_STEP(YH_MD5_F, B, C, D, A, m[11], 0x895cd7beu, 22)
        YH_MD5_STEP(YH_MD5_F, A, B, C, D, m[12], 0x6b901122u, 7)  YH_MD5_STEP(YH_MD5_F, D, A, B, C, m[13], 0xfd987193u, 12)
        YH_MD5_STEP(YH_MD5_F, C, D, A, B, m[14], 0xa679438eu, 17) YH_MD5_STEP(YH_MD5_F, B, C, D, A, m[15], 0x49b40821u, 22)
        YH_MD5_STEP(YH_MD5_G, A, B, C, D, m[1], 0xf61e2562u, 5)   YH_MD5_STEP(YH_MD5_G, D, A, B, C, m[6], 0xc040b340u, 9)
        YH_MD5_STEP(YH_MD5_G, C, D, A, B, m[11], 0x265e5a51u, 14) YH_MD5_STEP(YH_MD5_G, B, C, D, A, m[0], 0xe9b6c7aau, 20)
        YH_MD5_STEP(YH_MD5_G, A, B, C, D, m[5], 0xd62f105du, 5)   YH_MD5_STEP(YH_MD5_G, D, A, B, C, m[10], 0x02441453u, 9)
        YH_MD5_STEP(YH_MD5_G, C, D, A, B, m[15], 0xd8a1e681u, 14) YH_MD5_STEP(YH_MD5_G, B, C, D, A, m[4], 0xe7d3fbc8u, 20)
        YH_MD5_STEP(YH_MD5_G, A, B, C, D, m[9], 0x21e1cde6u, 5)   YH_MD5_STEP(YH_MD5_G, D, A, B, C, m[14], 0xc33707d6u, 9)
        YH_MD5_STEP(YH_MD5_G, C, D, A, B, m[3], 0xf4d50d87u, 14)  YH_MD5_STEP(YH_MD5_G, B, C, D, A, m[8], 0x455a14edu, 20)
        YH_MD5_STEP(YH_MD5_G, A, B, C, D, m[13], 0xa9e3e905u, 5)  YH_MD5_STEP(YH_MD5_G, D, A, B, C, m[2], 0xfcefa3f8u, 9)
        YH_MD5_STEP(YH_MD5_G, C, D, A, B, m[7], 0x676f02d9u, 14)  YH_MD5_STEP(YH_MD5_G, B, C, D, A, m[12], 0x8d2a4c8au, 20)
        YH_MD5_STEP(YH_MD5_H, A, B, C, D, m[5], 0xfffa3942u, 4)   YH_MD5_STEP(YH_MD5_H, D, A, B, C, m[8], 0x8771f681u, 11)
        YH_MD5_STEP(YH_MD5_H, C, D, A, B, m[11], 0x6d9d6122u, 16) YH_MD5_STEP(YH_MD5_H, B, C, D, A, m[14], 0xfde5380cu, 23)
        YH_MD5_STEP(YH_MD5_H, A, B, C, D, m[1], 0xa4beea44u, 4)   YH_MD5_STEP(YH_MD5_H, D, A, B, C, m[4], 0x4bdecfa9u, 11)
        YH_MD5_STEP(YH_MD5_H, C, D, A, B, m[7], 0xf6bb4b60u, 16)  YH_MD5_STEP(YH_MD5_H, B, C, D, A, m[10], 0xbebfbc70u, 23)
        YH_MD5_STEP(YH_MD5_H, A, B, C, D, m[13], 0x289b7ec6u, 4)  YH_MD5_STEP(YH_MD5_H, D, A, B, C, m[0], 0xeaa127fau, 11)
        YH_MD5_STEP(YH_MD5_H, C, D, A, B, m[3], 0xd4ef3085u, 16)  YH_MD5_STEP(YH_MD5_H, B, C, D, A, m[6], 0x04881d05u, 23)
        YH_MD5_STEP(YH_MD5_H, A, B, C, D, m[9], 0xd9d4d039u, 4)   YH_MD5_STEP(YH_MD5_H, D, A, B, C, m[12], 0xe6db99e5u, 11)
        YH_MD5_STEP(YH_MD5_H, C, D, A, B, m[15], 0x1fa27cf8u, 16) YH_MD5_STEP(YH_MD5_H, B, C, D, A, m[2], 0xc4ac5665u, 23)
        YH_MD5_STEP(YH_MD5_I, A, B, C, D, m[0], 0xf4292244u, 6)   YH_MD5_STEP(YH_MD5_I, D, A, B, C, m[7], 0x432aff97u, 10)
        YH_MD5_STEP(YH_MD5_I, C, D, A, B, m[14], 0xab9423a7u, 15) YH_MD5_STEP(YH_MD5_I, B, C, D, A, m[5], 0xfc93a039u, 21)
        YH_MD5_STEP(YH_MD5_I, A, B, C, D, m[12], 0x655b59c3u, 6)  YH_MD5_STEP(YH_MD5_I, D, A, B, C, m[3], 0x8f0ccc92u, 10)
        YH_MD5_STEP(YH_MD5_I, C, D, A, B, m[10], 0xffeff47du, 15) YH_MD5_STEP(YH_MD5_I, B, C, D, A, m[1], 0x85845dd1u, 21)
        YH_MD5_STEP(YH_MD5_I, A, B, C, D, m[8], 0x6fa87e4fu, 6)   YH_MD5_STEP(YH_MD5_I, D, A, B, C, m[15], 0xfe2ce6e0u, 10)
        YH_MD5_STEP(YH_MD5_I, C, D, A, B, m[6], 0xa3014314u, 15)  YH_MD5_STEP(YH_MD5_I, B, C, D, A, m[13], 0x4e0811a1u, 21)
        YH_MD5_STEP(YH_MD5_I, A, B, C, D, m[4], 0xf7537e82u, 6)   YH_MD5_STEP(YH_MD5_I, D, A, B, C, m[11], 0xbd3af235u, 10)
        YH_MD5_STEP(YH_MD5_I, C, D, A, B, m[2], 0x2ad7d2bbu, 15)  YH_MD5_STEP(YH_MD5_I, B, C, D, A, m[9], 0xeb86d391u, 21)
#undef YH_MD5_STEP
#undef YH_MD5_F
#undef YH_MD5_G
#undef YH_MD5_H
#undef YH_MD5_I
        a += A; b += B; c += C; d += D;
    }
    void update(const char* p, size_t n) {
        len += n;
        if (fill == 0)
            while (n >= 64) { block((const unsigned char*)p); p += 64; n -= 64; }
        while (n) {
            const size_t k = std::min(n, sizeof buf - fill);
            memcpy(buf + fill, p, k);
            fill += k; p += k; n -= k;
            if (fill == 64) {
                block(buf);
                fill = 0;
                while (n >= 64) { block((const unsigned char*)p); p += 64; n -= 64; }  // (whole blocks straight from the caller's bytes)
            }
        }
    }
    std::string hex() {
        const uint64_t bits = len * 8;
        const char pad = (char)0x80;
        update(&pad, 1);
        const char zero = 0;
        while (fill != 56) update(&zero, 1);
        unsigned char l[8];
        for (int i = 0; i < 8; ++i) l[i] = (unsigned char)(bits >> (8 * i));
        update((const char*)l, 8);
        const uint32_t w[4] = {a, b, c, d};
        static const char* H = "0123456789abcdef";
        std::string out;
        for (int i = 0; i < 4; ++i)
            for (int k = 0; k < 4; ++k) { const unsigned v = (w[i] >> (8 * k)) & 0xff; out.push_back(H[v >> 4]); out.push_back(H[v & 15]); }
        return out;
    }
};

enum { META_OK = 0, META_CANNOT_OPEN = 1, META_MALFORMED = 2, META_NOT_ONE = 3, META_EMPTY = 4, META_NEEDS_GENERAL_READER = 5 };
struct Meta {
    int status = META_OK;
    int n_matching = 0;  // signatures of the requested k-mer size in the file
    std::string name, md5;
    double mean_abundance = 0.0;
    bool has_abundance = false;
    bool is_first = false;  // the signature is record 0 / signature 0 of its file: its mins are what the train core reads from it
    uint64_t n_hashes = 0, scaled = 0;
};

// a JSON string with the escapes sourmash's writer (Rust serde / Python json) produces, as UTF-8
inline bool json_string(Scanner& s, std::string* out) {
    s.ws();
    if (s.p >= s.e || *s.p != '"') return false;
    ++s.p;
    out->clear();
    auto put_utf8 = [&](uint32_t cp) {
        if (cp < 0x80) out->push_back((char)cp);
        else if (cp < 0x800) { out->push_back((char)(0xc0 | (cp >> 6))); out->push_back((char)(0x80 | (cp & 0x3f))); }
        else if (cp < 0x10000) { out->push_back((char)(0xe0 | (cp >> 12))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3f))); out->push_back((char)(0x80 | (cp & 0x3f))); }
        else { out->push_back((char)(0xf0 | (cp >> 18))); out->push_back((char)(0x80 | ((cp >> 12) & 0x3f))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3f))); out->push_back((char)(0x80 | (cp & 0x3f))); }
    };
    auto hex4 = [&](uint32_t* v) {
        if (s.e - s.p < 4) return false;
        uint32_t x = 0;
        for (int i = 0; i < 4; ++i) {
            const char ch = s.p[i];
            x = x * 16 + (uint32_t)(ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : ch >= 'A' && ch <= 'F' ? ch - 'A' + 10 : 99);
            if (x > 0xffffff) return false;
        }
        s.p += 4;
        *v = x;
        return true;
    };
    while (s.p < s.e && *s.p != '"') {
        if (*s.p != '\\') { out->push_back(*s.p++); continue; }
        if (++s.p >= s.e) return false;
        const char esc = *s.p++;
        switch (esc) {
            case 'n': out->push_back('\n'); break;
            case 't': out->push_back('\t'); break;
            case 'r': out->push_back('\r'); break;
            case 'b': out->push_back('\b'); break;
            case 'f': out->push_back('\f'); break;
            case 'u': {
                uint32_t cp;
                if (!hex4(&cp)) return false;
                if (cp >= 0xd800 && cp < 0xdc00 && s.e - s.p >= 6 && s.p[0] == '\\' && s.p[1] == 'u') {  // surrogate pair
                    s.p += 2;
                    uint32_t lo;
                    if (!hex4(&lo)) return false;
                    cp = 0x10000 + ((cp - 0xd800) << 10) + (lo - 0xdc00);
                }
                put_utf8(cp);
                break;
            }
            default: out->push_back(esc);  // \" \\ \/
        }
    }
    if (s.p >= s.e) return false;
    ++s.p;
    return true;
}

inline bool json_uint(Scanner& s, uint64_t* v) {
    s.ws();
    const char* q = s.p;
    uint64_t x = 0;
    while (q < s.e && *q >= '0' && *q <= '9') x = x * 10 + (uint64_t)(*q++ - '0');
    if (q == s.p) return false;
    if (q < s.e && (*q == '.' || *q == 'e' || *q == 'E')) return false;  // not an integer: leave it to the general reader
    s.p = q;
    *v = x;
    return true;
}

// `text` = the content of a .sig file.  The one signature of k-mer size `ksize` in it.
// first_mins / have_first (optional): the "mins" of record 0, signature 0 -- what the train core reads from the same file
// (mins_from_text) -- captured on the way when that array was there and strictly ascending; the caller may use it INSTEAD of a
// second scan only when the whole file parsed (status META_OK, META_NOT_ONE or META_EMPTY) and *have_first is set.
inline Meta parse_meta(const char* text_p, size_t text_n, int ksize, std::vector<uint64_t>* first_mins = nullptr, bool* have_first = nullptr) {
    Meta m;
    if (have_first) *have_first = false;
    uint64_t rec_idx = 0;
    Scanner s{text_p, text_p + text_n};
    auto bad = [&](int st) { m.status = st; return m; };
    if (!s.lit('[')) return bad(META_MALFORMED);
    if (s.lit(']')) return bad(META_NOT_ONE);
    do {  // records
        if (!s.lit('{')) return bad(META_MALFORMED);
        std::string rec_name;
        std::vector<Meta> found;
        uint64_t sig_idx = 0;
        if (!s.lit('}')) {
            std::string key;
            do {
                if (!s.string(&key) || !s.lit(':')) return bad(META_MALFORMED);
                if (key == "name") {
                    s.ws();
                    if (s.p < s.e && *s.p == '"') { if (!json_string(s, &rec_name)) return bad(META_MALFORMED); }
                    else if (!s.skip()) return bad(META_MALFORMED);
                } else if (key == "signatures") {
                    if (!s.lit('[')) return bad(META_MALFORMED);
                    if (!s.lit(']')) {
                        do {  // signatures
                            if (!s.lit('{')) return bad(META_MALFORMED);
                            uint64_t k = 0, max_hash = 0, n = 0, n_ab = 0;
                            bool have_k = false, have_ab = false, ascending = true, have_mins = false, md_streamed = false;
                            uint64_t ab_sum = 0;  // (exact; numpy's mean of an int64 array is this sum in float64 over the count)
                            uint64_t prev = 0;
                            Md5 md;
                            std::vector<uint64_t> mins;
                            if (!s.lit('}')) {
                                std::string sk;
                                do {
                                    if (!s.string(&sk) || !s.lit(':')) return bad(META_MALFORMED);
                                    if (sk == "ksize") { if (!json_uint(s, &k)) return bad(META_NEEDS_GENERAL_READER); have_k = true; }
                                    else if (sk == "max_hash") { if (!json_uint(s, &max_hash)) return bad(META_NEEDS_GENERAL_READER); }
                                    else if (sk == "mins") {
                                        have_mins = true;
                                        if (!s.lit('[')) return bad(META_MALFORMED);
                                        // the sketch's md5 is over str(ksize) + every hash in decimal: when the k-mer size is known
                                        // by now (sourmash writes it first) the digits go into the md5 straight from the text
                                        md_streamed = have_k && (int)k == ksize;
                                        if (md_streamed) {
                                            char tmp[24];
                                            const int len = snprintf(tmp, sizeof tmp, "%llu", (unsigned long long)k);
                                            md.update(tmp, (size_t)len);
                                        }
                                        if (!s.lit(']')) {
                                            do {
                                                uint64_t v;
                                                s.ws();
                                                const char* t0 = s.p;
                                                if (!json_uint(s, &v)) return bad(META_MALFORMED);
                                                if (md_streamed) {
                                                    if (s.p - t0 > 1 && *t0 == '0') md_streamed = false;  // (not str(int): formatted below)
                                                    else md.update(t0, (size_t)(s.p - t0));
                                                }
                                                if (n && v <= prev) ascending = false;
                                                prev = v;
                                                mins.push_back(v);
                                                ++n;
                                            } while (s.lit(','));
                                            if (!s.lit(']')) return bad(META_MALFORMED);
                                        }
                                    } else if (sk == "abundances") {
                                        s.ws();
                                        if (s.p < s.e && *s.p == '[') {
                                            ++s.p;
                                            have_ab = true;
                                            if (!s.lit(']')) {
                                                do {
                                                    uint64_t v;
                                                    if (!json_uint(s, &v)) return bad(META_NEEDS_GENERAL_READER);
                                                    ab_sum += v;
                                                    ++n_ab;
                                                } while (s.lit(','));
                                                if (!s.lit(']')) return bad(META_MALFORMED);
                                            }
                                        } else if (!s.skip()) return bad(META_MALFORMED);  // null
                                    } else if (!s.skip()) return bad(META_MALFORMED);
                                } while (s.lit(','));
                                if (!s.lit('}')) return bad(META_MALFORMED);
                            }
                            if (!have_k) return bad(META_MALFORMED);
                            if (first_mins && have_first && rec_idx == 0 && sig_idx == 0 && have_mins && ascending) {
                                *first_mins = mins;  // (a copy: the md5 below still walks them)
                                *have_first = true;
                            }
                            ++sig_idx;
                            if ((int)k == ksize) {
                                if (!ascending || (have_ab && n_ab != n)) return bad(META_NEEDS_GENERAL_READER);  // (re-ordered / de-duplicated there)
                                Meta one;
                                one.is_first = rec_idx == 0 && sig_idx == 1;  // (sig_idx counts this signature already)
                                one.n_hashes = n;
                                one.has_abundance = have_ab;
                                one.mean_abundance = (have_ab && n) ? (double)ab_sum / (double)n : 0.0;
                                one.scaled = max_hash ? (uint64_t)__builtin_nearbyintl(18446744073709551616.0L / (long double)max_hash) : 0;  // sourmash: round(2^64 / max_hash)
                                if (!md_streamed) {  // (ksize behind mins in the object, or a number not written as str(int))
                                    md = Md5();
                                    char tmp[24];
                                    int len = snprintf(tmp, sizeof tmp, "%llu", (unsigned long long)k);
                                    md.update(tmp, (size_t)len);
                                    for (uint64_t v : mins) {
                                        char* q = tmp + sizeof tmp;
                                        do { *--q = (char)('0' + v % 10); v /= 10; } while (v);
                                        md.update(q, (size_t)(tmp + sizeof tmp - q));
                                    }
                                }
                                one.md5 = md.hex();
                                found.push_back(std::move(one));
                            }
                        } while (s.lit(','));
                        if (!s.lit(']')) return bad(META_MALFORMED);
                    }
                } else if (!s.skip()) return bad(META_MALFORMED);
            } while (s.lit(','));
            if (!s.lit('}')) return bad(META_MALFORMED);
        }
        ++rec_idx;
        for (Meta& f : found) {  // ("name" may come behind "signatures" in the record)
            ++m.n_matching;
            if (m.n_matching == 1) {
                const int keep = m.n_matching;
                m = std::move(f);
                m.n_matching = keep;
                m.name = rec_name;
            }
        }
    } while (s.lit(','));
    if (!s.lit(']')) return bad(META_MALFORMED);
    if (m.n_matching != 1) m.status = META_NOT_ONE;
    else if (m.n_hashes == 0) m.status = META_EMPTY;
    return m;
}
inline Meta parse_meta(const std::string& text, int ksize) { return parse_meta(text.data(), text.size(), ksize); }

}  // namespace yh_sig
