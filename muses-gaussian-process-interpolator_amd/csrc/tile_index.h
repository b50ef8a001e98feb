// Pure index arithmetic of the tile kernels: which tile a workgroup owns, which tiles of a block-cyclic local matrix are
// live, which pooled buffer a request gets.  No HIP in here on purpose: oracle-free CPU unit tests build this header with
// gcc -fsanitize=address,undefined (tests/test_index_functions.py, csrc/index_test.cpp) - SURVEY.md section 5, sanitizers.
#pragma once
#include <stdint.h>
#include <math.h>
#include <stddef.h>
#include <vector>

#ifdef __HIPCC__
#define MGPI_HD __host__ __device__ __forceinline__
#else
#define MGPI_HD inline
#endif

// linear lower-triangle tile index -> (i, j), j <= i
MGPI_HD void tri_decode(int64_t idx, int& ti, int& tj) {
    int i = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
    while ((int64_t)(i + 1) * (i + 2) / 2 <= idx) ++i;
    while ((int64_t)i * (i + 1) / 2 > idx) --i;
    ti = i;
    tj = (int)(idx - (int64_t)i * (i + 1) / 2);
}

// Tile order for L2 reuse.  The tile list is cut into bands of GEMM_BAND tile rows; inside a band the order is
// column-major, so 64 consecutive tiles (= what one XCD has in flight: 32 CUs x 2 workgroups) form an 8 x 8 block
// that shares 8 A slabs and 8 B slabs through that XCD's 4 MiB L2.  With plain row-major order every tile
// pulled its own B slab from beyond L2 (measured: 54 % L2 hit rate, 2.5x the algorithmic HBM bytes).
#ifndef GEMM_BAND
#define GEMM_BAND 8      // 4 and 16 measured: see DESIGN.md (traffic per tile, isolated rate)
#endif
template <bool LOWER>
MGPI_HD void tile_decode(int64_t idx, int tm, int tn, int& ti, int& tj) {
    constexpr int S = GEMM_BAND;
    if (!LOWER) {
        const int64_t per_band = (int64_t)S * tn;
        const int band = (int)(idx / per_band);
        const int within = (int)(idx - band * per_band);
        const int rows = (tm - band * S < S) ? (tm - band * S) : S;
        tj = within / rows;
        ti = band * S + within - tj * rows;
        return;
    }
    // lower triangle: band b holds rows [bS, bS + rows): rows * bS tiles left of the diagonal block (column-major)
    // plus the rows x rows lower triangle on the diagonal
    const int64_t full = (int64_t)(tm / S);                       // number of complete bands
    auto cum = [](int64_t b) { return (int64_t)S * S * b * (b - 1) / 2 + (int64_t)S * (S + 1) / 2 * b; };
    int64_t band;
    if (idx >= cum(full)) {
        band = full;
    } else {
        band = (int64_t)((-0.5 * S + sqrt(0.25 * S * S + 2.0 * S * S * (double)idx)) / (double)(S * S));
        while (band + 1 <= full && cum(band + 1) <= idx) ++band;
        while (cum(band) > idx) --band;
    }
    const int rows = (tm - (int)band * S < S) ? (tm - (int)band * S) : S;
    const int64_t within = idx - cum(band);
    const int64_t left = (int64_t)rows * band * S;
    if (within < left) {
        tj = (int)(within / rows);
        ti = (int)(band * S + within - (int64_t)tj * rows);
    } else {
        int ii, jj;
        tri_decode(within - left, ii, jj);
        ti = (int)(band * S) + ii;
        tj = (int)(band * S) + jj;
    }
}

// block-cyclic "staircase": which tiles of a rank's LOCAL trailing matrix lie on or below the GLOBAL diagonal.
// Local tile (ti, tj) belongs to global block row p + P*(bi0 + ti/tpb) and block column q + Q*(bj0 + tj/tpb).
// Only the (almost entirely) live part is enumerated: per band of GEMM_BAND tile rows the columns up to the last one
// that band's bottom row reaches; prefix[b] = number of enumerated tiles before band b.  (Enumerating the full
// rectangle and letting dead tiles exit left the XCD that got the top bands idle: 1.9x slower.)
#define GEMM_STAIR_BANDS 64
struct gemm_stair {
    int on, P, p, Q, q, bi0, bj0, tpb;
    int nbands;
    int prefix[GEMM_STAIR_BANDS + 1];
    int prefix2[GEMM_STAIR_BANDS + 1];   // merged staircase: prefix = the first mg_cw tile columns (region 1), prefix2 = the rest
    int zero_c;          // the accumulators start from 0 instead of the C tile: C = -A B (rides along in this by-value argument)
    // merged trailing update (chol.hip): the first mg_ncol workgroups own the tiles of the first mg_cw tile columns - the next
    // panel's block column - and each bumps *mg_counter once its tile is stored
    int mg_cw, mg_ncol;
    unsigned* mg_counter;
};


// host side of the staircase: per band of GEMM_BAND tile rows the columns its bottom row reaches; returns the number of
// enumerated tiles (0 = nothing to do, -1 = too many bands)
inline int64_t gemm_stair_fill(gemm_stair& st, int tm, int tn, int P, int p, int Q, int q, int bi0, int bj0, int tpb) {
    st.on = 1; st.P = P; st.p = p; st.Q = Q; st.q = q; st.bi0 = bi0; st.bj0 = bj0; st.tpb = tpb;
    st.nbands = (tm + GEMM_BAND - 1) / GEMM_BAND;
    if (st.nbands > GEMM_STAIR_BANDS) return -1;
    int64_t total = 0;
    for (int b = 0; b < st.nbands; ++b) {
        st.prefix[b] = (int)total;
        const int rows = (tm - b * GEMM_BAND < GEMM_BAND) ? (tm - b * GEMM_BAND) : GEMM_BAND;
        const int last = b * GEMM_BAND + rows - 1;
        const int64_t gr = (int64_t)(p + P * (bi0 + last / tpb)) * tpb + last % tpb;      // global tile row of the band's bottom row
        int cmax = 0;                                                                        // columns it reaches
        while (cmax < tn && (int64_t)(q + Q * (bj0 + cmax / tpb)) * tpb + cmax % tpb <= gr) ++cmax;
        total += (int64_t)rows * cmax;
    }
    st.prefix[st.nbands] = (int)total;
    return total;
}

// XCD-aware order: workgroups b, b + 8, .. share an XCD; each XCD gets a contiguous run of the tile list
MGPI_HD int64_t gemm_xcd_remap(int64_t bid, int64_t ntiles) {
    const int64_t q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, loc = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Merged staircase (the multi-GPU driver's trailing update with the next super-panel's local block columns first): the same
// enumeration cut at tile column cw - region 1 = the columns [0, cw) of every band, region 2 = the rest - each region dealt
// to the XCDs on its own, region 1 dispatched (hence finished) first.  Returns the total number of enumerated tiles, sets
// st.mg_ncol (enumerated tiles of region 1) and *live1 (those of them on/below the global diagonal: what the counter reaches).
inline int64_t gemm_stair_fill_merged(gemm_stair& st, int tm, int tn, int P, int p, int Q, int q, int bi0, int bj0, int tpb, int cw,
                                      int64_t* live1) {
    st.on = 1; st.P = P; st.p = p; st.Q = Q; st.q = q; st.bi0 = bi0; st.bj0 = bj0; st.tpb = tpb;
    st.nbands = (tm + GEMM_BAND - 1) / GEMM_BAND;
    if (st.nbands > GEMM_STAIR_BANDS) return -1;
    int64_t n1 = 0, n2 = 0, live = 0;
    for (int b = 0; b < st.nbands; ++b) {
        st.prefix[b] = (int)n1;
        st.prefix2[b] = (int)n2;
        const int rows = (tm - b * GEMM_BAND < GEMM_BAND) ? (tm - b * GEMM_BAND) : GEMM_BAND;
        const int last = b * GEMM_BAND + rows - 1;
        const int64_t gr = (int64_t)(p + P * (bi0 + last / tpb)) * tpb + last % tpb;
        int cmax = 0;
        while (cmax < tn && (int64_t)(q + Q * (bj0 + cmax / tpb)) * tpb + cmax % tpb <= gr) ++cmax;
        const int c1 = cmax < cw ? cmax : cw;
        n1 += (int64_t)rows * c1;
        n2 += (int64_t)rows * (cmax - c1);
        for (int r = 0; r < rows; ++r) {
            const int ti = b * GEMM_BAND + r;
            const int I = p + P * (bi0 + ti / tpb);
            for (int tj = 0; tj < c1; ++tj) {
                const int J = q + Q * (bj0 + tj / tpb);
                if (!(I < J || (I == J && (ti % tpb) < (tj % tpb)))) ++live;
            }
        }
    }
    st.prefix[st.nbands] = (int)n1;
    st.prefix2[st.nbands] = (int)n2;
    st.mg_cw = cw;
    st.mg_ncol = (int)n1;
    if (live1) *live1 = live;
    return n1 + n2;
}

MGPI_HD bool gemm_stair_merged_decode(const gemm_stair& st, int64_t bid, int64_t ntiles, int tiles_m, int& ti, int& tj) {
    const bool first = bid < st.mg_ncol;
    const int* pre = first ? st.prefix : st.prefix2;
    const int64_t idx = first ? gemm_xcd_remap(bid, st.mg_ncol) : gemm_xcd_remap(bid - st.mg_ncol, ntiles - st.mg_ncol);
    int band = 0;
    while (band + 1 < st.nbands && pre[band + 1] <= idx) ++band;
    const int within = (int)idx - pre[band];
    const int rows = (tiles_m - band * GEMM_BAND < GEMM_BAND) ? (tiles_m - band * GEMM_BAND) : GEMM_BAND;
    tj = within / rows + (first ? 0 : st.mg_cw);
    ti = band * GEMM_BAND + within % rows;
    const int I = st.p + st.P * (st.bi0 + ti / st.tpb), J = st.q + st.Q * (st.bj0 + tj / st.tpb);
    return !(I < J || (I == J && (ti % st.tpb) < (tj % st.tpb)));
}

// staircase decode of one workgroup: false = the tile lies above the global diagonal (or outside) and is skipped
MGPI_HD bool gemm_stair_decode(const gemm_stair& st, int64_t bid, int tiles_m, int& ti, int& tj) {
    int band = 0;
    while (band + 1 < st.nbands && st.prefix[band + 1] <= bid) ++band;
    const int within = (int)bid - st.prefix[band];
    const int rows = (tiles_m - band * GEMM_BAND < GEMM_BAND) ? (tiles_m - band * GEMM_BAND) : GEMM_BAND;
    tj = within / rows;
    ti = band * GEMM_BAND + within - tj * rows;
    const int I = st.p + st.P * (st.bi0 + ti / st.tpb), J = st.q + st.Q * (st.bj0 + tj / st.tpb);
    return !(I < J || (I == J && (ti % st.tpb) < (tj % st.tpb)));
}

// Merged trailing update: the lower triangle of a tm x tm tile grid in TWO regions, the first cw tile columns first.
//   workgroups [0, ncol): the tiles (i, j), j < cw, j <= i - the top cw x cw triangle, then the rows below it row by row
//   workgroups [ncol, ntiles): the lower triangle of the remaining (tm - cw) tile rows / columns in the banded order above
// Each region gets its own XCD split: workgroups b, b + 8, .. share an XCD whatever the region, so a region's list is
// dealt to the XCDs in contiguous runs.  The first region is dispatched (hence finished) first.
inline int64_t merged_column_tiles(int tm, int cw) { return (int64_t)cw * (cw + 1) / 2 + (int64_t)(tm - cw) * cw; }
MGPI_HD void merged_decode(int64_t bid, int tm, int cw, int64_t ncol, int64_t ntiles, int& ti, int& tj) {
    if (bid < ncol) {
        int64_t idx = gemm_xcd_remap(bid, ncol);
        const int64_t tri = (int64_t)cw * (cw + 1) / 2;
        if (idx < tri) {
            tri_decode(idx, ti, tj);
        } else {
            idx -= tri;
            ti = cw + (int)(idx / cw);
            tj = (int)(idx % cw);
        }
    } else {
        tile_decode<true>(gemm_xcd_remap(bid - ncol, ntiles - ncol), tm - cw, tm - cw, ti, tj);
        ti += cw;
        tj += cw;
    }
}

// update-role workgroup t of panel_step_kernel (chol.hip) -> tile (ti, tj), ti >= tj, column blocks tj < rest_cols,
// row blocks ti < nrows1: tiles are listed column block by column block, nrows1 - tj of them in column block tj
MGPI_HD void panel_rest_decode(int t, int nrows1, int& ti, int& tj) {
    tj = 0;
    while (t >= nrows1 - tj) { t -= nrows1 - tj; ++tj; }
    ti = tj + t;
}

// buffer pool: index of the pooled block a request of `bytes` gets - the smallest block that fits and wastes at most
// an eighth of the request - or -1
struct mgpi_pool_block { void* ptr; size_t bytes; };
inline int mgpi_pool_pick(const std::vector<mgpi_pool_block>& blocks, size_t bytes) {
    int best = -1;
    for (int i = 0; i < (int)blocks.size(); ++i) {
        const size_t b = blocks[i].bytes;
        if (b >= bytes && b - bytes <= bytes / 8 && (best < 0 || b < blocks[best].bytes)) best = i;
    }
    return best;
}
