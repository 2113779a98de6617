// Randomised host-side check of the layout builders, meant to be compiled with -fsanitize=address,undefined
// (tests/test_layout_fuzz.py): ragged rows, empty rows, repeated tids, rows too long for a tile, merge on/off.
#include "../emsar_amd/csrc/layout_tiled.hpp"
#include "../emsar_amd/csrc/sets.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
template <class T> static bool same_pod(const std::vector<T> &a, const std::vector<T> &b) {
    return a.size() == b.size() && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
}
static bool same_layout(const emsar::TiledLayout &a, const emsar::TiledLayout &b) {
    return same_pod(a.slices, b.slices) && same_pod(a.groups, b.groups) && same_pod(a.chunks, b.chunks) &&
           a.single_row == b.single_row && a.single_tid == b.single_tid && a.slot_row == b.slot_row && a.fwd == b.fwd && a.bwd == b.bwd &&
           a.coo == b.coo && a.far_tid == b.far_tid && a.far_blk_tid == b.far_blk_tid && a.far_ptr == b.far_ptr && a.far_src == b.far_src && a.pair_row == b.pair_row && a.pair_tid == b.pair_tid &&
           a.left_ptr == b.left_ptr && a.left_col == b.left_col && a.left_row == b.left_row && a.mem_ptr == b.mem_ptr && a.mem_row == b.mem_row;
}
int main(int argc, char **argv) {
    std::mt19937 rng(1);
    const int n_layout = argc > 1 ? atoi(argv[1]) : 40, n_sets = argc > 2 ? atoi(argv[2]) : 60;
    for (int trial = 0; trial < n_layout; trial++) {
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
        if (trial % 3 == 0) setenv("EMSAR_HIP_CHUNKS", trial % 2 ? "3" : "40", 1); else unsetenv("EMSAR_HIP_CHUNKS");
        if (trial % 5 == 4) setenv("EMSAR_HIP_FAR_EXPORT", "0", 1); else unsetenv("EMSAR_HIP_FAR_EXPORT");
        if (trial % 4 == 1) setenv("EMSAR_HIP_FRAG_ROWS", "3072", 1);          // many independently tiled fragments, several threads
        else if (trial % 4 == 2) setenv("EMSAR_HIP_FRAG_ROWS", "7000", 1);
        else unsetenv("EMSAR_HIP_FRAG_ROWS");
        for (int merge = 0; merge < 2; merge++) {
            emsar::TiledLayout L, L1;
            setenv("EMSAR_HOST_THREADS", trial % 2 ? "5" : "16", 1);             // classification, both sorts and the fragments on several threads
            int rc = emsar::build_tiled(n_rows, n_tx, rp.data(), ci.data(), L, merge);
            int ck = rc ? -99 : emsar::check_tiled(L, rp.data(), ci.data());
            if (rc || ck) { printf("FAIL trial %d merge %d rc %d ck %d\n", trial, merge, rc, ck); return 1; }
            setenv("EMSAR_HOST_THREADS", "1", 1);                                // the layout must not depend on the thread count
            rc = emsar::build_tiled(n_rows, n_tx, rp.data(), ci.data(), L1, merge);
            if (rc || !same_layout(L, L1)) { printf("FAIL trial %d merge %d: layout depends on the thread count\n", trial, merge); return 1; }
            unsetenv("EMSAR_HOST_THREADS");
        }
    }
    // set-resident records: sparse family-like matrices (many small sets), weights with zeros, a few huge rows
    for (int trial = 0; trial < n_sets; trial++) {
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
