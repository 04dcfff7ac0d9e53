#include "../imcoalhmm_amd/csrc/pair_dict.hpp"
#include "../imcoalhmm_amd/csrc/obs_io.hpp"
#include <random>
#include <cstdio>
int main() {
    std::mt19937 rng(1);
    for (int rep = 0; rep < 20; ++rep) {
        const int nsym = 2 + rep % 5;
        const size_t L = (rep % 4 == 0) ? 3 : 5000 + 3777 * rep;
        std::vector<uint8_t> obs(L);
        for (auto &x : obs) x = (rng() % 10 < 8) ? 0 : rng() % nsym;
        imc::PairDict d;
        imc::train_dict(d, nsym, std::vector<uint8_t>(obs.begin() + 1, obs.end()), 4);
        imc::EncodedLevels enc;
        imc::encode_levels(d, obs.data(), L, enc);
        // decode each level and compare
        for (int l = 0; l < imc::kNumLevels; ++l) {
            std::vector<uint8_t> out;
            std::vector<uint8_t> stack;
            for (uint8_t t : enc.streams[l]) {
                stack.assign(1, t);
                while (!stack.empty()) {
                    uint8_t z = stack.back(); stack.pop_back();
                    if (z < nsym) out.push_back(z);
                    else { stack.push_back(d.right[z]); stack.push_back(d.left[z]); }
                }
            }
            if (out != obs) { std::printf("MISMATCH rep %d level %d\n", rep, l); return 1; }
            for (uint8_t t : enc.streams[l]) if (t >= enc.alphabet[l]) { std::printf("token out of alphabet\n"); return 1; }
        }
        const char *path = "imc_sanitizer_tmp.imc";
        auto r = imc::write_cache(path, obs.data(), L, nsym);
        std::vector<uint8_t> back;
        auto r2 = imc::read_observation_file(path, nsym, back);
        if (r.code || r2.code || back != obs) { std::printf("cache mismatch\n"); return 1; }
    }
    std::printf("sanitizer run ok\n");
    return 0;
}
