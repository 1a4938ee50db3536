// lines.h -- line starts straight from the census (index.hip's list form), without the expanded record index.
// The census leaves, per 16 KiB tile of the stream, the in-tile offsets of its newlines in stream order (u16, IDX_LIST_CAP per tile) and --
// after its closing scan -- the number of newlines in front of every tile.  Line j + 1 starts behind newline j (0-based rank), so
//   line_start[j + 1] = tile * 16384 - mis + list[tile][j - offs[tile]] + 1      for the tile with offs[tile] <= j < offs[tile + 1];
// uq_index_lines writes these out for every line (8 B a line written, read again by every consumer).  The queued pack kernels and the QNAME
// sample take them from the lists instead: a place (tile, slot) found once per pack tile by a binary search over offs[] (cv_locate), the
// lanes of the tile walk on from there (cv_line_start).  Both forms name the same bytes: tests/test_gpu_pack.py compares them.
#pragma once
#include "common.h"

constexpr uint64_t CV_TILE = 16384;            // index.hip: IDX_TILE, IDX_LIST_CAP
constexpr uint32_t CV_LIST_CAP = 1024;

struct CensusView {
    const uint16_t* list;                      // [nb][CV_LIST_CAP]
    const uint32_t* offs;                      // [nb] newlines in front of tile t (exclusive scan of the census's counts)
    uint64_t nb;
    uint32_t mis;                              // the buffer's distance from its 16-byte-aligned base: tile t starts at stream position t * CV_TILE - mis
};

// place of newline `rank` (0 <= rank < nlines): its tile and ITS SLOT + 1 in the tile's list; rank -1 (the start of the stream): (0, 0)
__device__ __forceinline__ void cv_locate(const CensusView& cv, int64_t rank, uint32_t& T, uint32_t& kk) {
    if (rank < 0) { T = 0; kk = 0; return; }
    uint64_t lo = 0, hi = cv.nb;               // largest t with offs[t] <= rank (offs[0] = 0)
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)cv.offs[mid] <= (uint64_t)rank) lo = mid; else hi = mid;
    }
    // tiles without a newline share their successor's count: the search ends on the LAST tile with offs <= rank, the one that holds it
    T = (uint32_t)lo; kk = (uint32_t)((uint64_t)rank - cv.offs[lo]) + 1u;
}

// start of the line behind newline (place + d): the walk goes on into the following tiles where the place's own list ends.  `nlines` closes
// the last tile's list.  ok = false: beyond the last newline.
__device__ __forceinline__ uint64_t cv_line_start(const CensusView& cv, uint64_t nlines, uint32_t T, uint32_t kk, uint32_t d, bool& ok) {
    ok = true;
    int64_t idx = (int64_t)kk - 1 + (int64_t)d;
    if (idx < 0) return 0;                     // (0, 0) + 0: the stream's first byte
    uint64_t t = T;
    uint64_t lo = cv.offs[t];
    while (t < cv.nb) {
        const uint64_t end = t + 1 < cv.nb ? (uint64_t)cv.offs[t + 1] : nlines;
        const uint64_t c = end - lo;
        if ((uint64_t)idx < c) return t * CV_TILE - cv.mis + cv.list[t * CV_LIST_CAP + (uint64_t)idx] + 1;
        idx -= (int64_t)c; lo = end; ++t;
    }
    ok = false;
    return 0;
}
