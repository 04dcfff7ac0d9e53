// obs_io.hpp - host-side observation ingestion (no GPU involved).  Included by imcoal_fwd.hip only.
//
//  * read_observation_file: the reference's text format (whitespace-separated decimal symbols, written by
//    scripts/prepare-alignments.py:92-105 and read at src/IMCoalHMM/hmm.py:13-14) or this library's packed
//    cache ("IMCOBS1\n" magic; 2 bits per column when nsym <= 4, 1 byte up to 256 symbols, 2 bytes beyond);
//    templated on the symbol type: uint8_t for alphabets up to 256, uint16_t beyond (the ILS quartet alphabet of
//    scripts/prepare-alignments.py:186-190 has 257 symbols);
//  * encode_pairwise: the pairwise symbol rule of scripts/prepare-alignments.py:99-105
//    (2 = either base not in ACGT, 0 = equal, 1 = different; case-insensitive);
//  * write_cache.
#pragma once

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace imc {

static const char kCacheMagic[8] = {'I', 'M', 'C', 'O', 'B', 'S', '1', '\n'};

struct IoResult {
    int code = 0;          // 0 ok, -2 symbol, -5 io, -1 arg (values of include/imcoal_fwd.h)
    std::string msg;
};

inline IoResult io_fail(int code, const std::string &m)
{
    IoResult r;
    r.code = code;
    r.msg = m;
    return r;
}

template <typename T>
inline IoResult read_observation_file(const char *path, int nsym, std::vector<T> &sym)
{
    if (nsym > (sizeof(T) == 1 ? 256 : 65536)) return io_fail(-1, "alphabet too large for this symbol type");
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return io_fail(-5, std::string("cannot open ") + path + ": " + std::strerror(errno));
    sym.clear();
    char head[8];
    const size_t nh = std::fread(head, 1, 8, fp);
    if (nh == 8 && std::memcmp(head, kCacheMagic, 8) == 0) {
        // ---- packed cache ----
        uint32_t fs = 0, bits = 0;
        uint64_t L = 0;
        if (std::fread(&fs, 4, 1, fp) != 1 || std::fread(&bits, 4, 1, fp) != 1 || std::fread(&L, 8, 1, fp) != 1) {
            std::fclose(fp);
            return io_fail(-5, std::string("truncated cache header in ") + path);
        }
        if ((bits != 2 && bits != 8 && bits != 16) || fs < 1 || fs > 65536 || (bits == 2 && fs > 4) || (bits == 8 && fs > 256)) {
            std::fclose(fp);
            return io_fail(-5, std::string("bad cache header in ") + path);
        }
        if ((int)fs > nsym) {
            std::fclose(fp);
            return io_fail(-2, "cache holds an alphabet of " + std::to_string(fs) + " symbols, more than nsym");
        }
        sym.resize(L);
        if (bits == 8 || bits == 16) {
            const size_t w = bits / 8;
            std::vector<uint8_t> raw(L * w);
            if (L && std::fread(raw.data(), 1, raw.size(), fp) != raw.size()) { std::fclose(fp); return io_fail(-5, std::string("truncated cache ") + path); }
            for (uint64_t t = 0; t < L; ++t) sym[t] = (T)(w == 1 ? raw[t] : (uint16_t)(raw[2 * t] | raw[2 * t + 1] << 8));
        } else {
            std::vector<uint8_t> pk((L + 3) / 4);
            if (!pk.empty() && std::fread(pk.data(), 1, pk.size(), fp) != pk.size()) { std::fclose(fp); return io_fail(-5, std::string("truncated cache ") + path); }
            for (uint64_t t = 0; t < L; ++t) sym[t] = (pk[t >> 2] >> (2 * (t & 3))) & 3u;
        }
        std::fclose(fp);
        for (uint64_t t = 0; t < L; ++t)
            if (sym[t] >= nsym) return io_fail(-2, "symbol " + std::to_string(sym[t]) + " at column " + std::to_string(t) + " >= nsym");
        return IoResult();
    }
    // ---- text ----
    std::vector<char> buf(1 << 22);
    long cur = -1;   // token being accumulated, -1 = none
    size_t n = nh;
    std::memcpy(buf.data(), head, nh);
    IoResult rc;
    while (n > 0) {
        for (size_t i = 0; i < n; ++i) {
            const unsigned char ch = (unsigned char)buf[i];
            if (ch >= '0' && ch <= '9') {
                cur = (cur < 0 ? 0 : cur) * 10 + (ch - '0');
                if (cur > 1000000) cur = 1000000;
            } else if (ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r' || ch == '\f' || ch == '\v') {
                if (cur >= 0) {
                    if (cur >= nsym) {
                        std::fclose(fp);
                        return io_fail(-2, "symbol " + std::to_string(cur) + " at column " + std::to_string(sym.size()) + " >= nsym");
                    }
                    sym.push_back((T)cur);
                    cur = -1;
                }
            } else {
                std::fclose(fp);
                return io_fail(-5, std::string("unexpected character in ") + path);   // int() would raise ValueError (hmm.py:14)
            }
        }
        n = std::fread(buf.data(), 1, buf.size(), fp);
    }
    std::fclose(fp);
    if (cur >= 0) {
        if (cur >= nsym) return io_fail(-2, "symbol " + std::to_string(cur) + " >= nsym");
        sym.push_back((T)cur);
    }
    return IoResult();
}

template <typename T>
inline IoResult write_cache(const char *path, const T *sym, uint64_t L, int nsym)
{
    for (uint64_t t = 0; t < L; ++t)
        if (sym[t] >= nsym) return io_fail(-2, "symbol " + std::to_string(sym[t]) + " at column " + std::to_string(t) + " >= nsym");
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return io_fail(-5, std::string("cannot create ") + path + ": " + std::strerror(errno));
    const uint32_t fs = (uint32_t)nsym, bits = nsym <= 4 ? 2u : nsym <= 256 ? 8u : 16u;
    bool ok = std::fwrite(kCacheMagic, 1, 8, fp) == 8 && std::fwrite(&fs, 4, 1, fp) == 1 && std::fwrite(&bits, 4, 1, fp) == 1 &&
              std::fwrite(&L, 8, 1, fp) == 1;
    if (ok && bits >= 8 && L) {
        std::vector<uint8_t> raw(L * (bits / 8));
        for (uint64_t t = 0; t < L; ++t) {
            if (bits == 8) raw[t] = (uint8_t)sym[t];
            else { raw[2 * t] = (uint8_t)(sym[t] & 0xff); raw[2 * t + 1] = (uint8_t)((uint32_t)sym[t] >> 8); }
        }
        ok = std::fwrite(raw.data(), 1, raw.size(), fp) == raw.size();
    }
    if (ok && bits == 2) {
        std::vector<uint8_t> pk((L + 3) / 4, 0);
        for (uint64_t t = 0; t < L; ++t) pk[t >> 2] |= (uint8_t)(sym[t] << (2 * (t & 3)));
        if (!pk.empty()) ok = std::fwrite(pk.data(), 1, pk.size(), fp) == pk.size();
    }
    ok = (std::fclose(fp) == 0) && ok;
    return ok ? IoResult() : io_fail(-5, std::string("write failed: ") + path);
}

inline void encode_pairwise(const char *s1, const char *s2, size_t L, uint8_t *out)
{
    bool clean[256] = {false};
    for (const char *p = "ACGTacgt"; *p; ++p) clean[(unsigned char)*p] = true;
    for (size_t t = 0; t < L; ++t) {
        const unsigned char a = (unsigned char)s1[t], b = (unsigned char)s2[t];
        if (!clean[a] || !clean[b]) out[t] = 2;
        else out[t] = ((a & 0xdf) == (b & 0xdf)) ? 0 : 1;   // ASCII upper-casing of letters
    }
}

}  // namespace imc
