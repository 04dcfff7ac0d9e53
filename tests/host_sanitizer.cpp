#include "../imcoalhmm_amd/csrc/pair_dict.hpp"
#include "../imcoalhmm_amd/csrc/obs_io.hpp"
#include "../imcoalhmm_amd/csrc/model_host.hpp"
#include <random>
#include <cstdio>
// Trains dictionaries (byte phase and, on the long low-entropy streams, the 16-bit round phase), encodes every
// level, decodes it back, and round-trips the packed cache format.  Built with -fsanitize=address,undefined.
// Host-side model construction (model_host.hpp): expm over every Pade degree and the scaled branch, and the (pi, T)
// recursion on random piecewise systems with changing state spaces, projections and repeated pieces.  The class structure
// is random, so the joint matrix need not sum to one - the routine then returns its error text; what is checked here is
// that nothing reads or writes out of bounds on the way.
static int model_host_checks(std::mt19937 &rng)
{
    std::vector<double> work;
    for (int n : {1, 2, 4, 15, 33}) {
        for (double scale : {1e-4, 0.1, 0.5, 1.5, 3.0, 6.0, 50.0, 2000.0}) {
            std::vector<double> Q((size_t)n * n), out((size_t)n * n);
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
                for (int j = 0; j < n; ++j)
                    if (j != i) { Q[(size_t)i * n + j] = (rng() % 1000) * 1e-3 * scale / n; s += Q[(size_t)i * n + j]; }
                Q[(size_t)i * n + i] = -s;
            }
            if (!imc_model::expm(Q.data(), out.data(), n, work)) { std::printf("expm failed\n"); return 1; }
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
                for (int j = 0; j < n; ++j) s += out[(size_t)i * n + j];
                if (std::fabs(s - 1.0) > 1e-9) { std::printf("expm of a generator: row sum %.3e off (n %d scale %g)\n", s - 1.0, n, scale); return 1; }
            }
        }
    }
    for (int rep = 0; rep < 40; ++rep) {
        const int n = 1 + rep % 9;
        std::vector<int32_t> space_size(n), cls_off(1, 0), cls_idx, piece_q(std::max(0, n - 1)), piece_proj(std::max(0, n - 1), -1);
        const int sa = 3 + rng() % 6, sb = 2 + rng() % 5;               // two state spaces; the switch happens once
        const int sw = n > 1 ? (int)(rng() % n) : 0;
        for (int i = 0; i < n; ++i) space_size[i] = i <= sw ? sa : sb;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) {                               // random (possibly empty, possibly overlapping) classes
                for (int z = 0; z < space_size[i]; ++z)
                    if (rng() % 3 == (unsigned)k || (k == 0 && z == 0)) cls_idx.push_back(z);
                cls_off.push_back((int32_t)cls_idx.size());
            }
        std::vector<int32_t> q_size = {sa, sb}, proj_off = {0};
        std::vector<double> proj((size_t)sa * sb, 0.0);
        for (int r = 0; r < sa; ++r) proj[(size_t)r * sb + rng() % sb] = 1.0;
        for (int i = 0; i + 1 < n; ++i) {
            piece_q[i] = space_size[i] == sa ? 0 : 1;
            if (space_size[i] != space_size[i + 1]) piece_proj[i] = 0;
        }
        std::vector<double> Q((size_t)sa * sa + (size_t)sb * sb), dt(std::max(0, n - 1)), start(space_size[0], 0.0), pi(n), T((size_t)n * n);
        for (int which = 0; which < 2; ++which) {
            const int m = which ? sb : sa;
            double *M = Q.data() + (which ? (size_t)sa * sa : 0);
            for (int i = 0; i < m; ++i) {
                double s = 0.0;
                for (int j = 0; j < m; ++j)
                    if (j != i) { M[(size_t)i * m + j] = (rng() % 100) * 0.05; s += M[(size_t)i * m + j]; }
                M[(size_t)i * m + i] = -s;
            }
        }
        for (size_t i = 0; i < dt.size(); ++i) dt[i] = (rng() % 3 == 0 && i > 0) ? dt[i - 1] : (1 + rng() % 50) * 0.01;   // repeated pieces
        start[cls_idx[0]] = 1.0;                                        // (state 0 is always in interval 0's B class)
        std::vector<int32_t> q_off = {0, sa * sa};
        imc_model::Structure st{n, space_size.data(), cls_off.data(), cls_idx.data(), piece_q.data(), piece_proj.data(), 2, q_size.data(),
                                q_off.data(), sa * sa + sb * sb, proj_off.data(), proj.data()};
        (void)imc_model::transitions_one(st, Q.data(), dt.data(), start.data(), pi.data(), T.data());     // (result or error text)
    }
    return 0;
}

int main() {
    std::mt19937 rng(1);
    if (model_host_checks(rng)) return 1;
    int wide_levels = 0;
    for (int rep = 0; rep < 22; ++rep) {
        const int nsym = 2 + rep % 5;
        const bool big = rep >= 20;                       // two long streams that fill the byte dictionary
        const size_t L = big ? 600000 : (rep % 4 == 0) ? 3 : 5000 + 3777 * rep;
        std::vector<uint8_t> obs(L);
        for (auto &x : obs) x = (rng() % 100 < (big ? 97 : 80)) ? 0 : rng() % nsym;
        imc::PairDict d;
        const int max_depth = rep % 3 == 1 ? 3 + rep % 4 : 0;      // every third stream with a cap on the token depth
        imc::train_dict(d, nsym, std::vector<uint8_t>(obs.begin() + 1, obs.end()), 4, max_depth);
        if (d.alphabet >= imc::kByteAlphabet) {
            const std::vector<uint8_t> b = imc::encode_bytes(d, obs.data(), L, nullptr);
            imc::train_dict_wide(d, std::vector<imc::tok_t>(b.begin() + 1, b.end()), 3, max_depth);
        }
        if ((int)d.left.size() != d.alphabet || (int)d.right.size() != d.alphabet) { std::printf("dictionary size\n"); return 1; }
        {
            const std::vector<int> depth = imc::dict_depths(d);
            for (int z = 0; z < d.alphabet; ++z) {
                if (z < nsym ? depth[z] != 0 : (depth[z] < 1 || d.left[z] >= z || d.right[z] >= z)) { std::printf("depth / order\n"); return 1; }
                if (max_depth > 0 && depth[z] > max_depth) { std::printf("depth cap %d exceeded: %d\n", max_depth, depth[z]); return 1; }
            }
        }
        imc::EncodedLevels enc;
        imc::encode_levels(d, obs.data(), nullptr, L, enc);
        // decode each level and compare
        for (int l = 0; l < imc::kNumLevels; ++l) {
            std::vector<uint8_t> out;
            std::vector<int> stack;
            if (enc.alphabet[l] > d.alphabet || enc.alphabet[l] < nsym) { std::printf("level alphabet\n"); return 1; }
            if (enc.length[l] != (enc.is_wide[l] ? enc.wide[l].size() : enc.bytes[l].size())) { std::printf("level length\n"); return 1; }
            if (l > 0 && enc.length[l] > enc.length[l - 1]) { std::printf("levels must not grow\n"); return 1; }
            wide_levels += enc.is_wide[l];
            for (size_t q = 0; q < enc.length[l]; ++q) {
                const int t = enc.is_wide[l] ? (int)enc.wide[l][q] : (int)enc.bytes[l][q];
                if (t >= enc.alphabet[l]) { std::printf("token out of alphabet\n"); return 1; }
                stack.assign(1, t);
                while (!stack.empty()) {
                    const int z = stack.back(); stack.pop_back();
                    if (z < nsym) out.push_back((uint8_t)z);
                    else { stack.push_back(d.right[z]); stack.push_back(d.left[z]); }
                }
            }
            if (out != obs) { std::printf("MISMATCH rep %d level %d\n", rep, l); return 1; }
        }
        const char *path = "imc_sanitizer_tmp.imc";
        auto r = imc::write_cache(path, obs.data(), L, nsym);
        std::vector<uint8_t> back;
        auto r2 = imc::read_observation_file(path, nsym, back);
        if (r.code || r2.code || back != obs) { std::printf("cache mismatch\n"); return 1; }
    }
    if (!wide_levels) { std::printf("the 16-bit dictionary phase never ran\n"); return 1; }
    // Raw alphabets of 65, 200, 256 (bytes) and 257, 1000 (16-bit symbols, no byte phase): train, encode every level,
    // decode it back; 16-bit cache round trip.
    int merged_levels = 0;
    for (int nsym : {65, 200, 256, 257, 1000}) {
        const size_t L = 400000;
        std::vector<imc::tok_t> obs(L);
        for (auto &x : obs) x = (imc::tok_t)((rng() % 100 < 90) ? (rng() % 3) * (nsym / 3) : rng() % nsym);
        obs[7] = (imc::tok_t)(nsym - 1);
        std::vector<uint8_t> obs8(obs.begin(), obs.end());
        const bool wide_raw = nsym > imc::kByteAlphabet;
        imc::PairDict d;
        if (wide_raw) {
            imc::init_dict(d, nsym);
            imc::train_dict_wide(d, std::vector<imc::tok_t>(obs.begin() + 1, obs.end()), 3);
        } else {
            imc::train_dict(d, nsym, std::vector<uint8_t>(obs8.begin() + 1, obs8.end()), 4);
            if (d.alphabet >= imc::kByteAlphabet) {
                const std::vector<uint8_t> b = imc::encode_bytes(d, obs8.data(), L, nullptr);
                imc::train_dict_wide(d, std::vector<imc::tok_t>(b.begin() + 1, b.end()), 3);
            }
        }
        if (d.alphabet <= nsym) { std::printf("no merges at nsym %d\n", nsym); return 1; }
        imc::EncodedLevels enc;
        imc::encode_levels(d, wide_raw ? nullptr : obs8.data(), wide_raw ? obs.data() : nullptr, L, enc);
        for (int l = 0; l < imc::kNumLevels; ++l) {
            if (enc.alphabet[l] > d.alphabet || enc.alphabet[l] < nsym) { std::printf("level alphabet (nsym %d level %d)\n", nsym, l); return 1; }
            if (enc.alphabet[l] <= nsym) continue;                       // the raw stream itself
            ++merged_levels;
            std::vector<imc::tok_t> out;
            std::vector<int> stack;
            if (l > 0 && enc.length[l] > enc.length[l - 1]) { std::printf("levels must not grow\n"); return 1; }
            for (size_t q = 0; q < enc.length[l]; ++q) {
                const int t = enc.is_wide[l] ? (int)enc.wide[l][q] : (int)enc.bytes[l][q];
                if (t >= enc.alphabet[l]) { std::printf("token out of alphabet (nsym %d level %d)\n", nsym, l); return 1; }
                stack.assign(1, t);
                while (!stack.empty()) {
                    const int z = stack.back(); stack.pop_back();
                    if (z < nsym) out.push_back((imc::tok_t)z);
                    else { stack.push_back(d.right[z]); stack.push_back(d.left[z]); }
                }
            }
            if (out != obs) { std::printf("MISMATCH nsym %d level %d\n", nsym, l); return 1; }
        }
        const char *path = "imc_sanitizer_tmp16.imc";
        std::vector<imc::tok_t> back;
        auto w = imc::write_cache(path, obs.data(), L, nsym);
        auto rr = imc::read_observation_file(path, nsym, back);
        if (w.code || rr.code || back != obs) { std::printf("16-bit cache mismatch\n"); return 1; }
        std::remove(path);
    }
    if (merged_levels < 10) { std::printf("large alphabets produced too few merged levels\n"); return 1; }
    std::printf("sanitizer run ok\n");
    return 0;
}
