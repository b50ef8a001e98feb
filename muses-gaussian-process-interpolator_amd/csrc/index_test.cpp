// CPU unit test of the pure index functions (tile_index.h) - built by tests/test_index_functions.py with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// Exhaustive over small shapes: every decode is a bijection onto the tiles it should enumerate, nothing reads or
// writes out of bounds, no signed overflow.  Exit code 0 and "INDEX-OK" on success.
#include "tile_index.h"
#include <stdio.h>
#include <set>
#include <utility>

#define CHECK(cond, ...) do { if (!(cond)) { printf("FAILED %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); return 1; } } while (0)

int main() {
    // lower-triangle linear index
    for (int64_t idx = 0; idx < 200000; ++idx) {
        int i, j;
        tri_decode(idx, i, j);
        CHECK(j >= 0 && j <= i && (int64_t)i * (i + 1) / 2 + j == idx, "tri_decode(%lld) = (%d, %d)", (long long)idx, i, j);
    }
    {   // far out: the sqrt seed must be corrected by the two loops
        int i, j;
        const int64_t big = (int64_t)40000 * 40001 / 2 + 12345;
        tri_decode(big, i, j);
        CHECK(i == 40000 && j == 12345, "tri_decode at 8e8: (%d, %d)", i, j);
    }
    // banded tile order, full rectangle and lower triangle: a bijection for every shape
    for (int tm = 1; tm <= 40; ++tm)
        for (int tn = 1; tn <= 20; ++tn) {
            std::set<std::pair<int, int>> seen;
            for (int64_t idx = 0; idx < (int64_t)tm * tn; ++idx) {
                int ti, tj;
                tile_decode<false>(idx, tm, tn, ti, tj);
                CHECK(ti >= 0 && ti < tm && tj >= 0 && tj < tn, "tile_decode<full>(%lld; %d x %d) = (%d, %d)", (long long)idx, tm, tn, ti, tj);
                CHECK(seen.insert({ti, tj}).second, "tile_decode<full> repeats (%d, %d)", ti, tj);
            }
        }
    for (int tm = 1; tm <= 70; ++tm) {
        std::set<std::pair<int, int>> seen;
        const int64_t nt = (int64_t)tm * (tm + 1) / 2;
        for (int64_t idx = 0; idx < nt; ++idx) {
            int ti, tj;
            tile_decode<true>(idx, tm, tm, ti, tj);
            CHECK(ti >= 0 && ti < tm && tj >= 0 && tj <= ti, "tile_decode<lower>(%lld; %d) = (%d, %d)", (long long)idx, tm, ti, tj);
            CHECK(seen.insert({ti, tj}).second, "tile_decode<lower> repeats (%d, %d) at tm = %d", ti, tj, tm);
        }
    }
    {   // the size of the roofline run: 256 x 256 tiles of 128
        const int tm = 256;
        const int64_t nt = (int64_t)tm * (tm + 1) / 2;
        int64_t sum = 0;
        for (int64_t idx = 0; idx < nt; ++idx) {
            int ti, tj;
            tile_decode<true>(idx, tm, tm, ti, tj);
            CHECK(tj <= ti && ti < tm, "tile_decode<lower> at 256: (%d, %d)", ti, tj);
            sum += (int64_t)ti * tm + tj;
        }
        int64_t want = 0;
        for (int i = 0; i < tm; ++i) for (int j = 0; j <= i; ++j) want += (int64_t)i * tm + j;
        CHECK(sum == want, "tile_decode<lower> at 256 is not a permutation of the triangle");
    }
    // XCD remap: a permutation of [0, ntiles)
    for (int64_t nt = 1; nt <= 300; ++nt) {
        std::set<int64_t> seen;
        for (int64_t b = 0; b < nt; ++b) {
            const int64_t m = gemm_xcd_remap(b, nt);
            CHECK(m >= 0 && m < nt && seen.insert(m).second, "gemm_xcd_remap(%lld, %lld) = %lld", (long long)b, (long long)nt, (long long)m);
        }
    }
    // block-cyclic staircase: the enumerated tiles are in range, distinct, and cover EVERY tile on/below the global diagonal
    for (int P = 1; P <= 3; ++P) for (int Q = 1; Q <= 4; ++Q) for (int p = 0; p < P; ++p) for (int q = 0; q < Q; ++q)
        for (int tpb = 1; tpb <= 4; tpb *= 2) for (int bi0 = 0; bi0 <= 2; ++bi0) for (int bj0 = 0; bj0 <= 2; ++bj0)
            for (int mb = 1; mb <= 5; mb += 2) for (int nbk = 1; nbk <= 5; nbk += 2) {
                const int tm = mb * tpb, tn = nbk * tpb;
                gemm_stair st;
                const int64_t total = gemm_stair_fill(st, tm, tn, P, p, Q, q, bi0, bj0, tpb);
                CHECK(total >= 0 && st.prefix[st.nbands] == total, "staircase total");
                std::set<std::pair<int, int>> live;
                for (int64_t b = 0; b < total; ++b) {
                    int ti, tj;
                    const bool ok = gemm_stair_decode(st, b, tm, ti, tj);
                    CHECK(ti >= 0 && ti < tm && tj >= 0 && tj < tn, "staircase decode out of range (%d, %d) of %d x %d", ti, tj, tm, tn);
                    if (ok) CHECK(live.insert({ti, tj}).second, "staircase repeats (%d, %d)", ti, tj);
                }
                for (int ti = 0; ti < tm; ++ti) for (int tj = 0; tj < tn; ++tj) {
                    const int64_t gr = (int64_t)(p + P * (bi0 + ti / tpb)) * tpb + ti % tpb, gc = (int64_t)(q + Q * (bj0 + tj / tpb)) * tpb + tj % tpb;
                    CHECK((gr >= gc) == (live.count({ti, tj}) == 1), "staircase: tile (%d, %d) global (%lld, %lld) P%d Q%d p%d q%d", ti, tj,
                          (long long)gr, (long long)gc, P, Q, p, q);
                }
            }
    // merged staircase (the driver's trailing update with the next super-panel's block columns first): region 1 = the enumerated
    // tiles of the first cw tile columns, the two regions together exactly the plain staircase's live tiles, live1 = region 1's
    for (int P = 1; P <= 3; ++P) for (int Q = 1; Q <= 4; ++Q) for (int p = 0; p < P; ++p) for (int q = 0; q < Q; ++q)
        for (int tpb = 1; tpb <= 4; tpb *= 2) for (int bi0 = 0; bi0 <= 2; ++bi0) for (int bj0 = 0; bj0 <= 1; ++bj0)
            for (int mb = 1; mb <= 5; mb += 2) for (int nbk = 2; nbk <= 6; nbk += 2) for (int fc = 1; fc < nbk && fc <= 2; ++fc) {
                const int tm = mb * tpb, tn = nbk * tpb, cw = fc * tpb;
                gemm_stair st, plain;
                int64_t live1 = -1;
                const int64_t total = gemm_stair_fill_merged(st, tm, tn, P, p, Q, q, bi0, bj0, tpb, cw, &live1);
                const int64_t total_plain = gemm_stair_fill(plain, tm, tn, P, p, Q, q, bi0, bj0, tpb);
                CHECK(total == total_plain && st.mg_ncol == st.prefix[st.nbands] && st.mg_ncol + st.prefix2[st.nbands] == total, "merged staircase totals");
                std::set<std::pair<int, int>> live, want;
                int64_t counted = 0;
                for (int64_t b = 0; b < total; ++b) {
                    int ti, tj;
                    const bool ok = gemm_stair_merged_decode(st, b, total, tm, ti, tj);
                    CHECK(ti >= 0 && ti < tm && tj >= 0 && tj < tn, "merged staircase decode out of range (%d, %d) of %d x %d", ti, tj, tm, tn);
                    CHECK((tj < cw) == (b < st.mg_ncol), "merged staircase: tile (%d, %d) in the wrong region (cw %d)", ti, tj, cw);
                    if (ok) {
                        CHECK(live.insert({ti, tj}).second, "merged staircase repeats (%d, %d)", ti, tj);
                        if (b < st.mg_ncol) ++counted;
                    }
                }
                for (int64_t b = 0; b < total_plain; ++b) {
                    int ti, tj;
                    if (gemm_stair_decode(plain, b, tm, ti, tj)) want.insert({ti, tj});
                }
                CHECK(live == want, "merged staircase covers other tiles than the plain one (P%d Q%d p%d q%d)", P, Q, p, q);
                CHECK(counted == live1, "merged staircase: %lld live tiles in the first region, the host counted %lld", (long long)counted, (long long)live1);
            }
    // merged trailing update: the first region is exactly the first cw tile columns, the two regions together the whole triangle
    for (int tm = 2; tm <= 45; ++tm)
        for (int cw = 1; cw <= 8 && cw < tm; ++cw) {
            const int64_t nt = (int64_t)tm * (tm + 1) / 2, ncol = merged_column_tiles(tm, cw);
            std::set<std::pair<int, int>> seen;
            for (int64_t b = 0; b < nt; ++b) {
                int ti, tj;
                merged_decode(b, tm, cw, ncol, nt, ti, tj);
                CHECK(ti >= 0 && ti < tm && tj >= 0 && tj <= ti, "merged_decode(%lld; %d, %d) = (%d, %d)", (long long)b, tm, cw, ti, tj);
                CHECK((tj < cw) == (b < ncol), "merged_decode: tile (%d, %d) in the wrong region (tm %d, cw %d)", ti, tj, tm, cw);
                CHECK(seen.insert({ti, tj}).second, "merged_decode repeats (%d, %d)", ti, tj);
            }
        }
    // update-role tiles of the panel step: column block by column block, lower part
    for (int nrows1 = 1; nrows1 <= 40; ++nrows1)
        for (int rest = 0; rest <= 7 && rest <= nrows1; ++rest) {
            int total = 0;
            for (int tj = 0; tj < rest; ++tj) total += nrows1 - tj;
            std::set<std::pair<int, int>> seen;
            for (int t = 0; t < total; ++t) {
                int ti, tj;
                panel_rest_decode(t, nrows1, ti, tj);
                CHECK(tj >= 0 && tj < rest && ti >= tj && ti < nrows1, "panel_rest_decode(%d; %d, %d) = (%d, %d)", t, nrows1, rest, ti, tj);
                CHECK(seen.insert({ti, tj}).second, "panel_rest_decode repeats");
            }
        }
    // buffer pool: the smallest block that fits within an eighth of slack, never a smaller one
    {
        int dummy[8];
        std::vector<mgpi_pool_block> blocks = {{&dummy[0], 1000}, {&dummy[1], 4096}, {&dummy[2], 1100}, {&dummy[3], 1 << 20}, {&dummy[4], 1050}};
        CHECK(mgpi_pool_pick(blocks, 1000) == 0, "exact fit");
        CHECK(mgpi_pool_pick(blocks, 1001) == 4, "smallest block above the request");
        CHECK(mgpi_pool_pick(blocks, 1060) == 2, "next one up");
        CHECK(mgpi_pool_pick(blocks, 2000) == -1, "4096 wastes more than an eighth of 2000");
        CHECK(mgpi_pool_pick(blocks, 3700) == 1, "4096 is within an eighth of 3700");
        CHECK(mgpi_pool_pick(blocks, (size_t)2 << 20) == -1, "nothing large enough");
        CHECK(mgpi_pool_pick(std::vector<mgpi_pool_block>(), 8) == -1, "empty pool");
    }
    printf("INDEX-OK\n");
    return 0;
}
