// pair_dict.hpp - host-side observation compression for the MI355X forward engine.
//
// Role in the reference: ziphmm.preprocess_raw_observations(obs, NSYM) called once per alignment
// file at src/IMCoalHMM/hmm.py:16 (-> new_obs, sym2pair, new_nsyms).  Same idea (zipHMM: replace
// recurring adjacent symbol pairs by new symbols so that one N x N operator per new symbol covers
// many columns), re-designed for the GPU kernels' constraints:
//   * the dictionary is an ORDERED merge list and the stream is kept at several "levels"
//     (alphabet sizes kLevels[]): level A uses the first A-S merges.  The LDS-table kernels (N <= 40)
//     need every token's N x N operator resident in one CU's LDS, so they stop at 44..128 tokens;
//     the large-N kernels keep their table in global memory, where one more token costs one more
//     small GEMM per evaluation and saves a GEMM per occurrence, so they profit from thousands;
//   * tokens below 256 are trained one merge at a time (classic most-frequent-pair replacement) and
//     their streams are bytes; beyond that, merges are found in ROUNDS of 64..256 pairs (one count
//     of all adjacent pairs, the most frequent become tokens, one left-to-right replacement pass),
//     which keeps training and encoding at 72 passes for 16384 tokens; those streams are 16-bit;
//   * one dictionary per process and alphabet is trained on the first sufficiently long chunk and
//     reused for every later chunk, so that all chunks of a likelihood share one operator table;
//   * position 0 of a chunk is never merged (it is consumed by pi .* E[:,o_0]).
// The result is exact whatever the parse: a token's operator is the ordered product of its two
// halves' operators.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace imc {

typedef uint16_t tok_t;

// alphabet sizes at which the token stream is kept; dense enough below 256 that some level sits
// close under any LDS limit (N=20: 44 tokens = 155 KB of operators, N=10: 96 tokens)
constexpr int kLevels[] = {8,   12,  16,   24,   32,   44,   64,   96,   128,  192,   256,  384,
                           512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384};
constexpr int kNumLevels = 23;
constexpr int kByteAlphabet = 256;   // levels up to here: one merge per pass, byte streams
constexpr int kMaxAlphabet = 16384;
constexpr int kMaxRawAlphabet = 4096;   // largest raw alphabet S (symbols beyond 255 travel as 16-bit values)
// first token id found in rounds: 256 for byte-sized raw alphabets, the raw alphabet itself beyond
constexpr int wide_base(int nsym) { return nsym > kByteAlphabet ? nsym : kByteAlphabet; }
// merges per round beyond kByteAlphabet: 64 up to 1024 tokens, 256 from there (every wide level is a round boundary)
constexpr int round_size(int first_token) { return first_token < 1024 ? 64 : 256; }

struct PairDict {
    int nsym = 0;                    // raw alphabet size S
    int alphabet = 0;                // S + number of merges
    std::vector<tok_t> left, right;  // for z >= S: token z = left[z] followed by right[z]
    std::vector<uint32_t> span;      // raw columns covered by a token (saturating)
    uint64_t id = 0;

    void add(int l, int r)
    {
        left.push_back((tok_t)l);
        right.push_back((tok_t)r);
        const uint64_t sp = (uint64_t)span[l] + span[r];
        span.push_back(sp > 0xffffffffull ? 0xffffffffu : (uint32_t)sp);
        ++alphabet;
    }
};

// Replace non-overlapping occurrences of (a,b) by z, left to right, in place.  Returns new length.
template <typename T>
inline size_t replace_pair(T *seq, size_t n, T a, T b, T z)
{
    size_t w = 0, t = 0;
    while (t < n) {
        if (seq[t] == a && t + 1 < n && seq[t + 1] == b) { seq[w++] = z; t += 2; }
        else { seq[w++] = seq[t++]; }
    }
    return w;
}

// One round: every adjacent pair found in `keys` ((a << 16 | b), token ids in `ids`) becomes its token,
// scanning left to right (so overlapping candidates resolve deterministically).
inline size_t replace_round(tok_t *seq, size_t n, const std::vector<uint32_t> &keys, const std::vector<tok_t> &ids)
{
    size_t cap = 16;   // open-addressing table, at least 4x oversized
    while (cap < keys.size() * 4) cap <<= 1;
    std::vector<uint32_t> hk(cap, 0xffffffffu);
    std::vector<tok_t> hv(cap, 0);
    auto slot = [&](uint32_t k) { return (size_t)((k * 2654435761u) >> 7) & (cap - 1); };
    for (size_t q = 0; q < keys.size(); ++q) {
        size_t s = slot(keys[q]);
        while (hk[s] != 0xffffffffu) s = (s + 1) & (cap - 1);
        hk[s] = keys[q];
        hv[s] = ids[q];
    }
    size_t w = 0, t = 0;
    while (t < n) {
        if (t + 1 < n) {
            const uint32_t k = (uint32_t)seq[t] << 16 | seq[t + 1];
            size_t s = slot(k);
            while (hk[s] != 0xffffffffu && hk[s] != k) s = (s + 1) & (cap - 1);
            if (hk[s] == k) { seq[w++] = hv[s]; t += 2; continue; }
        }
        seq[w++] = seq[t++];
    }
    return w;
}

// Phase 1: train the first merges (alphabet <= kByteAlphabet) on `seq`, a scratch copy of (a prefix of)
// the chunk without its first column.  Stops when the best pair occurs fewer than `min_count` times.
// max_depth > 0: a pair is only merged while its token's depth in the dictionary DAG (1 + the deeper child's) stays
// within it - the operator table is built one depth per launch, and the last depths hold a handful of very long tokens.
inline std::vector<int> dict_depths(const PairDict &d)
{
    std::vector<int> depth((size_t)d.alphabet, 0);
    for (int z = d.nsym; z < d.alphabet; ++z) depth[z] = 1 + std::max(depth[d.left[z]], depth[d.right[z]]);
    return depth;
}

inline void train_dict(PairDict &d, int nsym, std::vector<uint8_t> seq, size_t min_count, int max_depth = 0)
{
    d.nsym = nsym;
    d.alphabet = nsym;
    d.left.assign(nsym, 0);
    d.right.assign(nsym, 0);
    d.span.assign(nsym, 1);
    std::vector<int> depth((size_t)nsym, 0);
    size_t n = seq.size();
    std::vector<uint64_t> count((size_t)kByteAlphabet * kByteAlphabet);
    while (d.alphabet < kByteAlphabet && n >= 2) {
        const int A = d.alphabet;
        std::fill(count.begin(), count.begin() + (size_t)A * A, 0);
        // plain adjacent-pair counts (runs a,a,a over-count (a,a); it only ranks candidates)
        for (size_t t = 0; t + 1 < n; ++t) count[(size_t)seq[t] * A + seq[t + 1]]++;
        uint64_t best = 0;
        int ba = -1, bb = -1;
        for (int a = 0; a < A; ++a)
            for (int b = 0; b < A; ++b)
                if (count[(size_t)a * A + b] > best && (max_depth <= 0 || std::max(depth[a], depth[b]) < max_depth)) {
                    best = count[(size_t)a * A + b]; ba = a; bb = b;
                }
        if (ba < 0 || best < min_count) break;
        const int z = d.alphabet;
        d.add(ba, bb);
        depth.push_back(1 + std::max(depth[ba], depth[bb]));
        n = replace_pair<uint8_t>(seq.data(), n, (uint8_t)ba, (uint8_t)bb, (uint8_t)z);
    }
}

// LSD radix sort of 32-bit keys, 16 bits per pass.
inline void radix_sort_u32(std::vector<uint32_t> &keys, std::vector<uint32_t> &scratch)
{
    scratch.resize(keys.size());
    std::vector<size_t> count(65537);
    for (int pass = 0; pass < 2; ++pass) {
        const int shift = 16 * pass;
        std::fill(count.begin(), count.end(), 0);
        for (uint32_t k : keys) count[((k >> shift) & 0xffffu) + 1]++;
        for (size_t i = 1; i < count.size(); ++i) count[i] += count[i - 1];
        for (uint32_t k : keys) scratch[count[(k >> shift) & 0xffffu]++] = k;
        keys.swap(scratch);
    }
}

// A dictionary with no merges yet (raw alphabets beyond 256 symbols skip phase 1: their symbols are not bytes).
inline void init_dict(PairDict &d, int nsym)
{
    d.nsym = nsym;
    d.alphabet = nsym;
    d.left.assign(nsym, 0);
    d.right.assign(nsym, 0);
    d.span.assign(nsym, 1);
}

// Phase 2: extend a full byte dictionary (or the bare raw alphabet when that has more than 256 symbols) in rounds on
// `seq`, the byte-level token stream (raw stream) of the training chunk, first column removed.  A pair needs
// `min_count` occurrences to become a token.
inline void train_dict_wide(PairDict &d, std::vector<tok_t> seq, size_t min_count, int max_depth = 0)
{
    if (d.alphabet < wide_base(d.nsym)) return;
    std::vector<int> depth = dict_depths(d);
    size_t n = seq.size();
    std::vector<uint32_t> keys, scratch;
    while (d.alphabet < kMaxAlphabet && n >= 2) {
        const size_t round = (size_t)round_size(d.alphabet);
        keys.resize(n - 1);
        for (size_t t = 0; t + 1 < n; ++t) keys[t] = (uint32_t)seq[t] << 16 | seq[t + 1];
        radix_sort_u32(keys, scratch);
        std::vector<std::pair<uint64_t, uint32_t>> cand;   // (count, key)
        for (size_t i = 0; i < keys.size();) {
            size_t j = i;
            while (j < keys.size() && keys[j] == keys[i]) ++j;
            if (j - i >= min_count && (max_depth <= 0 || std::max(depth[keys[i] >> 16], depth[keys[i] & 0xffffu]) < max_depth))
                cand.push_back({(uint64_t)(j - i), keys[i]});
            i = j;
        }
        if (cand.empty()) break;
        std::sort(cand.begin(), cand.end(),
                  [](const std::pair<uint64_t, uint32_t> &x, const std::pair<uint64_t, uint32_t> &y) {
                      return x.first != y.first ? x.first > y.first : x.second < y.second;
                  });
        const size_t take = std::min<size_t>({cand.size(), round, (size_t)(kMaxAlphabet - d.alphabet)});
        std::vector<uint32_t> rk;
        std::vector<tok_t> ri;
        for (size_t q = 0; q < take; ++q) {
            rk.push_back(cand[q].second);
            ri.push_back((tok_t)d.alphabet);
            d.add((int)(cand[q].second >> 16), (int)(cand[q].second & 0xffffu));
            depth.push_back(1 + std::max(depth[cand[q].second >> 16], depth[cand[q].second & 0xffffu]));
        }
        n = replace_round(seq.data(), n, rk, ri);
        if (take < round) break;   // a short round is the last one, so that round boundaries stay where round_size puts them
    }
}

struct EncodedLevels {
    // level l: the token stream (including the raw first column at index 0) using the first
    // min(kLevels[l], dict.alphabet) tokens - bytes[l] for alphabets <= 256, wide[l] beyond.
    std::vector<uint8_t> bytes[kNumLevels];
    std::vector<tok_t> wide[kNumLevels];
    int alphabet[kNumLevels];
    size_t length[kNumLevels];
    bool is_wide[kNumLevels];
};

// Byte-level part of the encoding (also prepares the training stream of phase 2); returns the deepest
// byte stream.  `out` may be null.
inline std::vector<uint8_t> encode_bytes(const PairDict &d, const uint8_t *obs, size_t L, EncodedLevels *out)
{
    std::vector<uint8_t> seq(obs, obs + L);
    size_t n = L;
    int lvl = 0;
    auto snapshot = [&](int A) {
        while (out && lvl < kNumLevels && kLevels[lvl] <= A && kLevels[lvl] <= kByteAlphabet) {
            out->bytes[lvl].assign(seq.begin(), seq.begin() + n);
            out->alphabet[lvl] = std::max(d.nsym, std::min(kLevels[lvl], d.alphabet));
            out->length[lvl] = n;
            out->is_wide[lvl] = false;
            ++lvl;
        }
    };
    snapshot(d.nsym);   // levels not larger than the raw alphabet are the raw stream
    const int top = std::min(d.alphabet, kByteAlphabet);
    for (int z = d.nsym; z < top; ++z) {
        if (n > 2)
            n = 1 + replace_pair<uint8_t>(seq.data() + 1, n - 1, (uint8_t)d.left[z], (uint8_t)d.right[z], (uint8_t)z);
        snapshot(z + 1);
    }
    // byte levels beyond the trained alphabet: same as the final byte stream
    while (out && lvl < kNumLevels && kLevels[lvl] <= kByteAlphabet) {
        out->bytes[lvl].assign(seq.begin(), seq.begin() + n);
        out->alphabet[lvl] = std::max(d.nsym, top);
        out->length[lvl] = n;
        out->is_wide[lvl] = false;
        ++lvl;
    }
    seq.resize(n);
    return seq;
}

// Encode a chunk with a fixed dictionary: apply the merges in order, snapshot at every level.  The raw symbols come
// as bytes (`obs`, alphabets up to 256) or as 16-bit values (`obs16`, larger alphabets: no byte levels exist then).
inline void encode_levels(const PairDict &d, const uint8_t *obs, const tok_t *obs16, size_t L, EncodedLevels &out)
{
    std::vector<uint8_t> bytes;
    std::vector<tok_t> seq;
    int lvl = 0;
    if (obs16) {
        for (; lvl < kNumLevels && kLevels[lvl] <= kByteAlphabet; ++lvl) {   // not larger than the raw alphabet: the raw stream
            out.alphabet[lvl] = d.nsym;
            out.length[lvl] = L;
            out.is_wide[lvl] = true;
        }
        seq.assign(obs16, obs16 + L);
    } else {
        bytes = encode_bytes(d, obs, L, &out);
        while (lvl < kNumLevels && kLevels[lvl] <= kByteAlphabet) ++lvl;
        seq.assign(bytes.begin(), bytes.end());
    }
    size_t n = seq.size();
    for (int z0 = wide_base(d.nsym); z0 < d.alphabet; z0 += round_size(z0)) {
        const int z1 = std::min(z0 + round_size(z0), d.alphabet);
        std::vector<uint32_t> rk;
        std::vector<tok_t> ri;
        for (int z = z0; z < z1; ++z) {
            rk.push_back((uint32_t)d.left[z] << 16 | d.right[z]);
            ri.push_back((tok_t)z);
        }
        if (n > 2) n = 1 + replace_round(seq.data() + 1, n - 1, rk, ri);
        while (lvl < kNumLevels && kLevels[lvl] <= z1) {
            // the stream now holds every token below z1 (with 256-symbol-or-smaller raw alphabets the rounds end
            // exactly on the level sizes; beyond, a level is the first round end that reaches it)
            out.wide[lvl].assign(seq.begin(), seq.begin() + n);
            out.alphabet[lvl] = z1;
            out.length[lvl] = n;
            out.is_wide[lvl] = true;
            ++lvl;
        }
    }
    // levels beyond the trained alphabet: same as the deepest stream (wide only if the dictionary is)
    for (; lvl < kNumLevels; ++lvl) {
        if (d.alphabet > kByteAlphabet) out.wide[lvl].assign(seq.begin(), seq.begin() + n);
        else out.bytes[lvl] = bytes;
        out.alphabet[lvl] = std::max(d.nsym, d.alphabet);
        out.length[lvl] = n;
        out.is_wide[lvl] = d.alphabet > kByteAlphabet;
    }
}

}  // namespace imc
