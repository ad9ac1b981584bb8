// csrc/zstd_chain.h -- scalar "chain walk" of the Zstd fast match finder: resolves every sequence that one batch of 64
// speculative probes holds (reference: AOCL_ZSTD_compressBlock_fast_noDict_generic, algos/zstd/lib/compress/zstd_fast.c:427-706).
//
// Plain scalar C++ on purpose: the kernel (zstd_kernels.hip) runs it on wave-uniform values (SGPRs, v_readlane for the
// per-lane inputs) and tests/zstd_chain_model.cpp runs the very same code on the CPU inside a lane-by-lane model of the
// batch, where it is checked sequence by sequence against the oracle's sequential parser.
//
// Vocabulary.  A batch probes the 64 consecutive positions pos0 .. pos0+63 ("lanes").  A *search* starts at lane S (the
// end of the previous match) and visits its lanes in pairs: at every even probe k (k - S even) the reference first tests
// the repcode at k+2 (:519-542), then the table candidate of probe k, then the candidate of probe k+1 (:546-598).
//   M event  lane j holds a table candidate whose 4 bytes match (okmask bit j; only lanes < lim are trustworthy).
//   R event  even probe k whose bytes at k+2 repeat at distance rep1.
// Byte equality at a fixed distance is kept as a 64-bit map over the batch's lanes (bit i <-> src[pos0+i] ==
// src[pos0+i-dist], bits outside the known range are 0): for the two repcodes the batch starts with, the map is one
// ballot over a coalesced byte load; after an M event the new rep1 is that match's offset and its map is the equality
// mask of the lane's own 32-byte probe / candidate windows - the same bytes the forward extension compared - known up
// to 28 bytes past the probe.
// The walk stops as soon as anything is not decidable from registers:
//   ZC_SPECIAL   the next event is known (ev, el) but needs the one-at-a-time path (long extension or catch-up,
//                immediate rep2 match, table writes beyond the trustworthy lanes);
//   ZC_CONTINUE  no event below lim: the search goes on in the next batch, `lim - S` probes already examined;
//   ZC_RESTART   a fresh batch has to start at lane S (S is the start of a search; never returned before at least
//                one sequence was resolved, so the parse always advances);
//   ZC_REFRESH   the equality map of the current rep1 (taken from a match's 32-byte window) does not reach far enough to
//                decide the search that starts at lane S: the caller loads fresh full-width maps for (rep1, rep2) and calls
//                walk() again with S0 = S and the commit mask so far - one load round instead of a whole new batch.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define ZC_FN __device__ __forceinline__
#else
#define ZC_FN static inline
#endif

namespace zchain {

ZC_FN uint64_t shr64(uint64_t m, int s) { return s < 64 ? m >> s : 0; }     // s >= 0
ZC_FN int ctz64(uint64_t v) { return v ? (int)__builtin_ctzll(v) : 64; }
// "4 equal bytes start here"
ZC_FN uint64_t quad(uint64_t eq) { const uint64_t a = eq & (eq >> 1); return a & (a >> 2); }

// equality map of lane j's match offset: eqb bit i <-> byte (pos0 + j - 4 + i) equals its candidate byte; known for
// lanes [j, min(j + 28, 64))
ZC_FN uint64_t map_from_window(uint32_t eqb, int j)
{
    const uint64_t e = eqb & ~0xFu;
    return j >= 4 ? e << (j - 4) : e >> (4 - j);
}

// per-lane record of a table candidate, packed for one readlane: candidate position (17 bits) | forward length beyond
// the 4 match bytes, 0..24 (24 = runs out of the window) << 17 | catch-up bytes 0..4 (7 = needs the exact path) << 22
ZC_FN uint32_t pack_info(uint32_t c, uint32_t fw, uint32_t bk4) { return c | (fw << 17) | (bk4 << 22); }
ZC_FN uint64_t pack_info_wide(uint32_t c, uint32_t fw, uint32_t bk4) { return ((uint64_t)c << 32) | (fw << 17) | (bk4 << 22); }

enum { ZC_RESTART = 0, ZC_SPECIAL = 1, ZC_CONTINUE = 2, ZC_REFRESH = 3 };

struct WalkOut {
    int outcome;
    int S;              // lane where the unfinished search starts
    int R;              // ZC_CONTINUE: lanes [S, R) of the unfinished search hold no event
    int ev, el;         // ZC_SPECIAL: 1 = table match at lane el, 2 = repcode found by even probe el
    int32_t arel;       // anchor - pos0
    uint32_t rep1, rep2;
    uint64_t commit;    // lanes whose table write (position + 2) the reference has executed
    int ntail, tail[3]; // ZC_RESTART: the LAST ntail of tail[0..2] (lane indices, may exceed 63) are table writes to perform one
                        // by one, in order, after the commit
    bool post;          // ZC_RESTART: the immediate rep2 test at S (:690) is still to be made, with loads
    int why;            // ZC_SPECIAL, statistics only: 1 long extension, 2 long catch-up / candidate at the chunk start,
                        // 4 table writes beyond the batch, 16 immediate rep2 match
};

ZC_FN uint64_t lane_range(int a, int b)   // bits [a, b), 0 <= a < 64, b <= 64
{
    if (b <= a) return 0;
    const uint64_t hi = b >= 64 ? ~0ull : ((1ull << b) - 1);
    return hi & ~((1ull << a) - 1);
}
ZC_FN uint64_t lane_bit(int i) { return (i >= 0 && i < 64) ? 1ull << i : 0; }

// bits [a, b) for 0 <= a <= 63, a < b <= 64 (two shifts instead of a guarded one: b - 1 <= 63)
ZC_FN uint64_t span(int a, int b) { return (~0ull << a) & ~((~0ull << (b - 1)) << 1); }

// okmask: lanes below lim with a matching table candidate.  eq1 / eq2: equality maps at distance rep1 / rep2 over all 64
// lanes (eq2 is ignored while rep2 == 0).  info(j), eqb(j): per-lane inputs of lane j;
// emit(litLength, offBase, matchLength, anchorPosition) stores one sequence.
//
// Written for the scalar unit: every shift amount is kept below 64 by construction (S <= 63 inside the loop, event lanes
// below lim <= 64) instead of being guarded, the loop carries only what the next sequence needs, and whatever describes HOW
// the chain ended (the three pending table writes, the special event) is assembled once, after the loop.
template <class Info, class Eqb, class Emit>
ZC_FN WalkOut walk(const uint64_t okmask, const int lim, const uint32_t pos0, int32_t arel, uint32_t rep1, uint32_t rep2,
                   uint64_t eq1, const uint64_t eq2, Info info, Eqb eqb, Emit emit, const int S0 = 0, const uint64_t commit0 = 0,
                   const uint32_t lowPos = 0)
{
    int S = S0;
    uint64_t commit = commit0;
    uint64_t q1 = quad(eq1), q2 = quad(eq2);
    int end1 = 64, end2 = 64;                            // the maps are known for lanes below end
    int outcome = ZC_RESTART, R = 0, ev = 0, el = 0, why = 0, E = 0;
    bool fin = false, post = false;                      // the chain ended with a sequence whose table writes / rep2 test are pending
    while (S < lim) {                                    // (S <= 63 in the body)
        const uint64_t okS = okmask >> S;
        const uint64_t t = ((q1 >> 2) >> S) & 0x5555555555555555ull;                 // R events of this search, relative to S (S + 2 may be 65: two shifts)
        const int jm = S + ctz64(okS);                                                // first M event, >= 64: none
        const int kr = S + ctz64(t);                                                  // first R event
        if (t != 0 && kr <= jm) {
            if (kr >= lim) { commit |= span(S, lim); R = lim; outcome = ZC_CONTINUE; break; }
            // ---- repcode found by even probe k at k + 2 (:519-542) ----
            const int k = kr;
            const uint64_t e1k = eq1 >> (k + 1);                                      // bit 0 <-> lane k + 1 = q - 1 (:533)
            const uint32_t back1 = (uint32_t)e1k & 1u;
            const int f = ctz64(~(e1k >> 5));                                         // equality run from q + 4
            E = k + 6 + f;
            const bool imm = rep2 != 0 && E + 4 <= end2;                              // :690 immediate rep2 test decidable here
            why = (E < end1 ? 0 : 1) | ((imm && ((q2 >> (E & 63)) & 1)) ? 16 : 0);
            commit |= span(S, k + 1);
            ev = 2; el = k;
            if (why) { outcome = ZC_SPECIAL; break; }
            emit((uint32_t)(k + 2 - (int)back1 - arel), 1u, back1 + 4u + (uint32_t)f, (uint32_t)((int32_t)pos0 + arel));
            arel = E; S = E;
            post = rep2 != 0 && !imm;                                                 // else by two loads after the walk
        } else {
            const int U = t ? 1000 : end1 - 5;           // even probes below U have a known repcode test (k + 6 <= end1)
            if (!(okS != 0 && jm < U)) {
                R = lim < U ? lim : U;                   // no event in [S, R)
                if (R < lim && S > 0) { outcome = end1 < 64 ? ZC_REFRESH : ZC_RESTART; break; }    // fresh maps / a fresh batch see further
                commit |= span(S, R); outcome = ZC_CONTINUE; break;                   // (R > S: lim > S, and U > S or S == 0 ... see below)
            }
            // ---- table match at lane j (:546-598, :651-669) ----
            const int j = jm;
            const auto in = info(j);
            // a 64-bit record carries a full-width candidate position in its upper half (frames: positions are offsets into
            // the frame, the catch-up stops at lowPos, the prefix start); the 32-bit record of the chunk kernels holds 17 bits
            uint32_t c, cl;
            if constexpr (sizeof(in) == 8) { c = (uint32_t)((uint64_t)in >> 32); cl = c - lowPos; }
            else { c = (uint32_t)in & 0x1FFFFu; cl = c; }
            const uint32_t fw = ((uint32_t)in >> 17) & 31u, bk4 = ((uint32_t)in >> 22) & 7u;
            const uint32_t room = (uint32_t)(j - arel);
            const uint32_t maxback = room < cl ? room : cl;
            const uint32_t back = bk4 < maxback ? bk4 : maxback;
            E = j + 4 + (int)fw;
            const bool imm = E + 4 <= end1;                                           // rep2 becomes the old rep1
            why = (fw < 24 ? 0 : 1) | (bk4 != 7 && !(bk4 == 4 && maxback > 4) ? 0 : 2) | (E - 2 >= lim && j + 2 >= 64 ? 4 : 0)
                | ((imm && ((q1 >> (E & 63)) & 1)) ? 16 : 0);
            commit |= span(S, j + 1);
            ev = 1; el = j;
            if (why) { outcome = ZC_SPECIAL; break; }
            const uint32_t off = pos0 + (uint32_t)j - c;
            emit((uint32_t)(j - (int)back - arel), off + 3u, 4u + back + fw, (uint32_t)((int32_t)pos0 + arel));
            rep2 = rep1; rep1 = off;
            arel = E; S = E;
            post = !imm;
            q2 = q1; end2 = end1;
            eq1 = (uint64_t)(eqb(j) >> 4) << j;                                       // = map_from_window(eqb(j), j)
            q1 = quad(eq1); end1 = j + 28 < 64 ? j + 28 : 64;
        }
        // table writes at el+1, el+2 and E-2 (:537, :680-681): lanes below lim join the batch commit; from the first one at or
        // beyond lim on they follow one by one, in order, and the chain ends with this sequence - as it does when the immediate
        // rep2 test (:690) could not be decided from the maps
        if (E - 2 >= lim || post) { fin = true; break; }
        commit |= (3ull << (el + 1)) | (1ull << (E - 2));
    }
    WalkOut o;
    o.outcome = outcome; o.S = S; o.R = R; o.ev = 0; o.el = 0; o.why = 0; o.arel = arel; o.rep1 = rep1; o.rep2 = rep2;
    o.ntail = 0; o.tail[0] = o.tail[1] = o.tail[2] = 0; o.post = false;
    if (outcome == ZC_SPECIAL) { o.ev = ev; o.el = el; o.why = why; }
    if (fin) {
        const int w1 = el + 1, w2 = el + 2, w3 = E - 2;
        o.tail[0] = w1; o.tail[1] = w2; o.tail[2] = w3;
        o.ntail = (w1 >= lim) + (w2 >= lim) + (w3 >= lim);
        if (w1 < lim) commit |= lane_bit(w1);
        if (w2 < lim) commit |= lane_bit(w2);
        if (w3 < lim) commit |= lane_bit(w3);
        o.post = post;
    }
    o.commit = commit;
    return o;
}

// 4-bit mask of the zero bytes of x
ZC_FN uint32_t zero_bytes(uint32_t x)
{
    uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
    t = (~t & 0x80808080u) >> 7;
    return (t | (t >> 7) | (t >> 14) | (t >> 21)) & 15u;
}

}  // namespace zchain
