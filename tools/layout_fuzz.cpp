// Randomised host-side check of the layout builders, meant to be compiled with -fsanitize=address,undefined
// (tests/test_layout_fuzz.py): ragged rows, empty rows, repeated tids, rows too long for a tile, merge on/off.
#include "../emsar_amd/csrc/layout_tiled.hpp"
#include "../emsar_amd/csrc/sets.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
static bool same_layout(const emsar::TiledLayout &a, const emsar::TiledLayout &b) {
    return a.tiles.size() == b.tiles.size() && (a.tiles.empty() || memcmp(a.tiles.data(), b.tiles.data(), a.tiles.size() * sizeof(emsar::Tile)) == 0) &&
           a.single_row == b.single_row && a.single_tid == b.single_tid && a.slot_row == b.slot_row && a.fwd == b.fwd && a.bwd == b.bwd &&
           a.coo == b.coo && a.far_tid == b.far_tid && a.left_ptr == b.left_ptr && a.left_col == b.left_col && a.left_row == b.left_row &&
           a.mem_ptr == b.mem_ptr && a.mem_row == b.mem_row;
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
        if (trial % 4 == 1) setenv("EMSAR_HIP_FRAG_ROWS", "3072", 1);          // many independently tiled fragments, several threads
        else if (trial % 4 == 2) setenv("EMSAR_HIP_FRAG_ROWS", "7000", 1);
        else unsetenv("EMSAR_HIP_FRAG_ROWS");
        for (int merge = 0; merge < 2; merge++) {
            emsar::TiledLayout L, L1;
            setenv("EMSAR_HOST_THREADS", trial % 2 ? "5" : "16", 1);             // classification, both sorts and the fragments on several threads
            int rc = emsar::build_tiled(n_rows, n_tx, rp.data(), ci.data(), L, merge);
            int ck = rc ? -99 : emsar::check_tiled(L, rp.data(), ci.data());
            if (rc || ck) { printf("FAIL trial %d merge %d rc %d ck %d\n", trial, merge, rc, ck); return 1; }
            {                                                                    // what the unit kernel reads first, derived from the layout
                emsar::UnitTables U;
                emsar::build_unit_tables(L, U);
                const int cu = emsar::check_unit_tables(L, U);
                if (cu) { printf("FAIL trial %d merge %d unit tables %d\n", trial, merge, cu); return 1; }
            }
            if (!L.tiles.empty()) {                                              // a descriptor that points past its arrays must be caught on the host
                size_t last = 0;
                for (size_t i = 0; i < L.tiles.size(); i++) if (L.tiles[i].fwd_off > L.tiles[last].fwd_off) last = i;
                emsar::TiledLayout B = L;
                B.tiles[last].k[B.tiles[last].n_slices - 1] += 1;                  // one forward column more than was stored
                emsar::TiledLayout C = L;
                C.tiles[trial % C.tiles.size()].row_base += emsar::kTileSliceRows * 4;   // row slots of another tile, or past the end
                emsar::TiledLayout D = L;
                D.tiles[trial % D.tiles.size()].near_n = 0; D.tiles[trial % D.tiles.size()].far_n = 0;   // ids beyond the dictionary
                if (emsar::check_tiled_extents(B) == 0 || emsar::check_tiled_extents(C) == 0 || emsar::check_tiled_extents(D) == 0) {
                    printf("FAIL trial %d merge %d: a bad descriptor passed the extent check\n", trial, merge); return 1;
                }
            }
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
        if (S.n_closed_tids + S.n_resident_tids + S.n_streamed_tids + S.CL.n_tids != n_tx) { printf("FAIL sets census %d\n", trial); return 1; }
    }
    printf("ok\n");
    return 0;
}
