// Randomised host-side check of the layout builders, meant to be compiled with -fsanitize=address,undefined
// (tests/test_layout_fuzz.py): ragged rows, empty rows, repeated tids, rows too long for a tile, merge on/off.
#include "../emsar_amd/csrc/layout_tiled.hpp"
#include "../emsar_amd/csrc/sets.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
int main() {
    std::mt19937 rng(1);
    for (int trial = 0; trial < 40; trial++) {
        int n_tx = 50 + rng() % 5000;
        int n_rows = rng() % 20000;
        std::vector<uint64_t> rp(1, 0);
        std::vector<int32_t> ci;
        for (int r = 0; r < n_rows; r++) {
            int k = rng() % 100 < 50 ? 1 : (rng() % 100 < 2 ? 700 + rng() % 600 : 1 + rng() % 40);
            if (rng() % 50 == 0) k = 0;
            int t0 = rng() % n_tx;
            for (int j = 0; j < k; j++) ci.push_back(rng() % 10 == 0 ? (int)(rng() % n_tx) : std::min(n_tx - 1, t0 + j % 64));
            rp.push_back(ci.size());
        }
        if (trial % 4 == 1) setenv("EMSAR_HIP_FRAG_ROWS", "3072", 1);          // many independently tiled fragments, several threads
        else if (trial % 4 == 2) setenv("EMSAR_HIP_FRAG_ROWS", "7000", 1);
        else unsetenv("EMSAR_HIP_FRAG_ROWS");
        for (int merge = 0; merge < 2; merge++) {
            emsar::TiledLayout L;
            int rc = emsar::build_tiled(n_rows, n_tx, rp.data(), ci.data(), L, merge);
            int ck = rc ? -99 : emsar::check_tiled(L, rp.data(), ci.data());
            if (rc || ck) { printf("FAIL trial %d merge %d rc %d ck %d\n", trial, merge, rc, ck); return 1; }
        }
        emsar::WindowedLayout W;
        int rc = emsar::build_windowed(n_rows, n_tx, rp.data(), ci.data(), 256 << (trial % 5), 4096, W);
        if (rc || emsar::check_windowed(W, rp.data(), ci.data())) { printf("FAIL windowed %d\n", trial); return 1; }
    }
    // set-resident records: sparse family-like matrices (many small sets), weights with zeros, a few huge rows
    for (int trial = 0; trial < 60; trial++) {
        int n_tx = 20 + rng() % 6000;
        int n_rows = rng() % 30000;
        int fam = 2 + rng() % (trial % 3 == 0 ? 400 : 12);
        std::vector<uint64_t> rp(1, 0);
        std::vector<int32_t> ci, w;
        for (int r = 0; r < n_rows; r++) {
            int k = rng() % 100 < 40 ? 1 : 1 + rng() % std::min(fam, 30);
            if (rng() % 60 == 0) k = 0;
            if (trial % 7 == 0 && rng() % 2000 == 0) k = 3000;
            int base = (int)(rng() % n_tx) / fam * fam;
            for (int j = 0; j < k; j++) ci.push_back(rng() % 400 == 0 ? (int)(rng() % n_tx) : std::min(n_tx - 1, base + (int)(rng() % fam)));
            rp.push_back(ci.size());
            w.push_back(rng() % 3 == 0 ? 0 : 1 + rng() % 50);
        }
        emsar::ResidentSets S;
        const int32_t *wp = trial % 5 == 4 ? nullptr : w.data();
        emsar::build_sets(n_rows, n_tx, rp.data(), ci.data(), wp, S);
        int ck = emsar::check_sets(n_rows, n_tx, rp.data(), ci.data(), wp, S);
        if (ck) { printf("FAIL sets trial %d ck %d\n", trial, ck); return 1; }
        if (S.n_closed_tids + S.n_resident_tids + S.n_streamed_tids != n_tx) { printf("FAIL sets census %d\n", trial); return 1; }
    }
    printf("ok\n");
    return 0;
}
