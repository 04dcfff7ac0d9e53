// pair_dict.hpp - host-side observation compression for the MI355X forward engine.
//
// Role in the reference: ziphmm.preprocess_raw_observations(obs, NSYM) called once per alignment
// file at src/IMCoalHMM/hmm.py:16 (-> new_obs, sym2pair, new_nsyms).  Same idea (zipHMM: replace
// recurring adjacent symbol pairs by new symbols so that one N x N operator per new symbol covers
// many columns), re-designed for the GPU kernel's constraints:
//   * the dictionary must be small enough that every token's N x N operator stays resident in
//     one CU's LDS, so it is built as an ORDERED merge list and the stream is kept at several
//     "levels" (alphabet sizes 8/12/16/24/32/44/64/96/128/192/256): level A uses the first A-S merges.  The forward
//     call picks the level that minimises its cost model among those whose table fits LDS for the model's N
//     (the large-N kernels keep their table in global memory and may use all 256 tokens);
//   * one dictionary per process and alphabet is trained on the first sufficiently long chunk and
//     reused for every later chunk, so that all chunks of a likelihood share one operator table;
//   * position 0 of a chunk is never merged (it is consumed by pi .* E[:,o_0]).
// The result is exact: a token's operator is the ordered product of its two halves' operators.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace imc {

// alphabet sizes at which the token stream is kept; dense enough that some level sits close under any LDS limit
// (N=20: 44 tokens = 155 KB of operators, N=10: 96 tokens)
constexpr int kLevels[] = {8, 12, 16, 24, 32, 44, 64, 96, 128, 192, 256};
constexpr int kNumLevels = 11;
constexpr int kMaxAlphabet = 256;   // token ids are bytes

struct PairDict {
    int nsym = 0;                  // raw alphabet size S
    int alphabet = 0;              // S + number of merges
    uint8_t left[kMaxAlphabet];    // for z >= S: token z = left[z] followed by right[z]
    uint8_t right[kMaxAlphabet];
    uint32_t span[kMaxAlphabet];   // raw columns covered by a token (saturating)
    uint64_t id = 0;
};

// Replace non-overlapping occurrences of (a,b) by z, left to right, in place.  Returns new length.
inline size_t replace_pair(uint8_t *seq, size_t n, uint8_t a, uint8_t b, uint8_t z)
{
    size_t w = 0, t = 0;
    while (t < n) {
        if (seq[t] == a && t + 1 < n && seq[t + 1] == b) { seq[w++] = z; t += 2; }
        else { seq[w++] = seq[t++]; }
    }
    return w;
}

// Train the ordered merge list on `seq` (a scratch copy of the chunk without its first column).
// Stops at kMaxAlphabet tokens or when the best pair occurs fewer than `min_count` times.
inline void train_dict(PairDict &d, int nsym, std::vector<uint8_t> seq, size_t min_count)
{
    d.nsym = nsym;
    d.alphabet = nsym;
    for (int s = 0; s < nsym; ++s) { d.left[s] = d.right[s] = 0; d.span[s] = 1; }
    size_t n = seq.size();
    std::vector<uint64_t> count((size_t)kMaxAlphabet * kMaxAlphabet);
    while (d.alphabet < kMaxAlphabet && n >= 2) {
        const int A = d.alphabet;
        std::fill(count.begin(), count.begin() + (size_t)A * A, 0);
        // plain adjacent-pair counts (runs a,a,a over-count (a,a); it only ranks candidates)
        for (size_t t = 0; t + 1 < n; ++t) count[(size_t)seq[t] * A + seq[t + 1]]++;
        uint64_t best = 0;
        int ba = -1, bb = -1;
        for (int a = 0; a < A; ++a)
            for (int b = 0; b < A; ++b)
                if (count[(size_t)a * A + b] > best) { best = count[(size_t)a * A + b]; ba = a; bb = b; }
        if (ba < 0 || best < min_count) break;
        const int z = d.alphabet++;
        d.left[z] = (uint8_t)ba;
        d.right[z] = (uint8_t)bb;
        const uint64_t sp = (uint64_t)d.span[ba] + d.span[bb];
        d.span[z] = sp > 0xffffffffull ? 0xffffffffu : (uint32_t)sp;
        n = replace_pair(seq.data(), n, (uint8_t)ba, (uint8_t)bb, (uint8_t)z);
    }
}

struct EncodedLevels {
    // streams[l] = token stream (including the raw first column at index 0) using the first
    // min(kLevels[l], dict.alphabet) tokens; empty vector when identical to the previous level.
    std::vector<uint8_t> streams[kNumLevels];
    int alphabet[kNumLevels];
};

// Encode a chunk with a fixed dictionary: apply the merges in order, snapshot at every level.
inline void encode_levels(const PairDict &d, const uint8_t *obs, size_t L, EncodedLevels &out)
{
    std::vector<uint8_t> seq(obs, obs + L);
    size_t n = L;
    int lvl = 0;
    auto snapshot = [&](int A) {
        while (lvl < kNumLevels && kLevels[lvl] <= A) {
            out.streams[lvl].assign(seq.begin(), seq.begin() + n);
            out.alphabet[lvl] = std::max(d.nsym, std::min(kLevels[lvl], d.alphabet));
            ++lvl;
        }
    };
    snapshot(d.nsym);   // levels not larger than the raw alphabet are the raw stream
    for (int z = d.nsym; z < d.alphabet; ++z) {
        if (n > 2) n = 1 + replace_pair(seq.data() + 1, n - 1, d.left[z], d.right[z], (uint8_t)z);
        snapshot(z + 1);
    }
    // levels beyond the trained alphabet: same as the final stream
    while (lvl < kNumLevels) {
        out.streams[lvl].assign(seq.begin(), seq.begin() + n);
        out.alphabet[lvl] = d.alphabet;
        ++lvl;
    }
}

}  // namespace imc
