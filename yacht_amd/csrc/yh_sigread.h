// yh_sigread.h -- just enough JSON to reach [0]["signatures"][0]["mins"] of a sourmash .sig file: what the
// reference's train core reads (src/cpp/main.cpp:62-84, read_min_hashes: record 0, signature 0, "mins";
// ksize is NOT checked there; an unreadable file is an empty sketch).  Shared by the drop-in
// executable (train_core_main.cpp) and the library's batch reader (yh_sigread.hip).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
inline std::vector<uint64_t> read_mins(const std::string& path, bool report = true, int* status = nullptr) {
    std::vector<uint64_t> mins;
    if (status) *status = READ_OK;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        if (report) std::cerr << "Could not open the file!" << std::endl;
        if (status) *status = READ_CANNOT_OPEN;
        return mins;
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
    Scanner s{text.data(), text.data() + text.size()};
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

}  // namespace yh_sig
