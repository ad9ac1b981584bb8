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

// per-lane record of a table candidate, packed for one readlane: match offset (position - candidate, 17 bits; the 64-bit
// record of the frame kernels carries it in its upper half) | forward length beyond the 4 match bytes, 0..24 (24 = runs out
// of the window) << 17 | catch-up code << 22: 0..4 bytes in front of the match are equal and the 5th is not (or the
// candidate has no 5th: it sits 4 bytes behind the prefix start), 5 = four are equal and the candidate has more room (the
// exact path counts on when the anchor allows more than four).  Everything that depends on the lane alone - a run that
// leaves the window (fw = 24), a candidate within 4 bytes of the prefix start, table writes that would fall behind lane 63 -
// is NOT in the record: such lanes are named by the `hard` mask and always take the exact path.
ZC_FN uint32_t pack_info(uint32_t off, uint32_t fw, uint32_t bk) { return off | (fw << 17) | (bk << 22); }
ZC_FN uint64_t pack_info_wide(uint32_t off, uint32_t fw, uint32_t bk) { return ((uint64_t)off << 32) | (fw << 17) | (bk << 22); }
// catch-up code of a candidate at distance `cl` from the prefix start whose 4 bytes in front compare as bk4 = 0..4
ZC_FN uint32_t back_code(uint32_t bk4, uint32_t cl) { return bk4 + ((bk4 == 4 && cl > 4) ? 1u : 0u); }

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
    int why;            // ZC_SPECIAL, statistics only: 1 long extension / lane of the hard mask, 2 catch-up beyond four bytes,
                        // 16 immediate rep2 match
};

ZC_FN uint64_t lane_bit(int i) { return (i >= 0 && i < 64) ? 1ull << i : 0; }

// bits [a, b) for 0 <= a <= 63, a < b <= 64 (two shifts instead of a guarded one: b - 1 <= 63)
ZC_FN uint64_t span(uint32_t a, uint32_t b) { return (~0ull << a) & ~((~0ull << (b - 1)) << 1); }
ZC_FN uint32_t ctz64u(uint64_t v) { return v ? (uint32_t)__builtin_ctzll(v) : 64u; }

// okmask: lanes below lim with a matching table candidate; hard: those of them that always take the exact path.  eq1 / eq2:
// equality maps at distance rep1 / rep2 over all 64 lanes (eq2 is ignored while rep2 == 0).  info(j), eqb(j): per-lane inputs
// of lane j; emit(litLength, offBase, matchLength) stores one sequence (they follow each other without gaps: a sequence's
// literals start where the one before ended).
//
// Written for the scalar unit.  Every shift amount is kept below 64 by construction (S <= 63 inside the loop, event lanes
// below lim <= 64) instead of being guarded; conditions are 0 / 1 integers combined with & and | (a C++ bool that the
// compiler cannot prove uniform becomes a 64-bit lane mask and a vector compare); both kinds of event meet in ONE emit and
// ONE end-of-sequence test; the loop carries only what the next sequence needs, and whatever describes HOW the chain ended
// (the span a broken-off search commits, the three pending table writes, the special event) is assembled once, after it.
template <class Info, class Eqb, class Emit>
ZC_FN WalkOut walk(const uint64_t okmask, const uint64_t hard, const int lim_, const uint32_t pos0, int32_t arel_, uint32_t rep1, uint32_t rep2,
                   uint64_t eq1, const uint64_t eq2, Info info, Eqb eqb, Emit emit, const int S0 = 0, const uint64_t commit0 = 0)
{
    (void)pos0;
    const uint64_t kEven = 0x5555555555555555ull;
    const uint32_t lim = (uint32_t)lim_;
    uint32_t S = (uint32_t)S0, arel = (uint32_t)arel_;   // (arel: two's complement of a possibly negative lane; only differences are used)
    uint64_t commit = commit0;
    uint64_t q1 = quad(eq1), q2 = quad(eq2);
    uint32_t end1 = 64, end2 = 64;                       // the maps are known for lanes below end
    // how the loop was left: 0 ran out of lanes, 1 no event in [S, R), 2 undecidable without fresh maps / a fresh batch,
    // 3 special event, 4 the sequence just emitted ends the chain
    uint32_t how = 0, R = 0, el = 0, E = 0, isM = 0, post = 0, why = 0, Sev = 0;
    while (S < lim) {                                    // (S <= 63 in the body)
        const uint64_t okS = okmask >> S;
        const uint64_t t = ((q1 >> 2) >> S) & kEven;     // R events of this search, relative to S (S + 2 may be 65: two shifts)
        const uint32_t dm = ctz64u(okS), dr = ctz64u(t); // distance of the first M / R event, 64: none
        uint32_t ll, offBase, ml, special;
        Sev = S;
        if (dr <= (dm < 63u ? dm : 63u)) {
            // ---- repcode found by even probe k at k + 2 (:519-542) ----
            const uint32_t k = S + dr;
            if (k >= lim) { R = lim; how = 1; break; }
            const uint64_t e1k = eq1 >> (k + 1);                                      // bit 0 <-> lane k + 1 = q - 1 (:533)
            const uint32_t back1 = (uint32_t)e1k & 1u;
            const uint32_t f = ctz64u(~(e1k >> 5));                                   // equality run from q + 4
            E = k + 6 + f;
            const uint32_t has2 = rep2 != 0 ? 1u : 0u;
            const uint32_t imm = has2 & (E + 4 <= end2 ? 1u : 0u);                    // :690 immediate rep2 test decidable here
            const uint32_t hit = imm & (uint32_t)(q2 >> (E & 63u));
            special = (E >= end1 ? 1u : 0u) | (hit & 1u);
            el = k; isM = 0;
            ll = k + 2 - back1 - arel; offBase = 1u; ml = back1 + 4u + f;
            post = has2 & (imm ^ 1u);                                                 // else by two loads after the walk
            if (special) { why = (E >= end1 ? 1u : 0u) | ((hit & 1u) << 4); how = 3; break; }
        } else {
            // no repcode event in front of the first table match; without any (t == 0) the even probes' repcode tests are
            // known below end1 - 5 only (k + 6 <= end1)
            const uint32_t j = S + dm;
            if (dm == 64u || (t == 0 && j + 5 >= end1)) {
                R = end1 - 5 < lim ? end1 - 5 : lim;                                  // (t != 0 here means dm < dr: an event)
                how = (R < lim && S > 0) ? 2u : 1u;                                   // fresh maps / a fresh batch see further
                break;
            }
            // ---- table match at lane j (:546-598, :651-669) ----
            const auto in = info(j);
            uint32_t off;
            if constexpr (sizeof(in) == 8) off = (uint32_t)((uint64_t)in >> 32); else off = (uint32_t)in & 0x1FFFFu;
            const uint32_t fw = ((uint32_t)in >> 17) & 31u, bk = ((uint32_t)in >> 22) & 7u;
            const uint32_t room = j - arel;
            const uint32_t back = bk < room ? bk : room;                              // 5: more than four bytes may be equal
            E = j + 4 + fw;
            const uint32_t imm = E + 4 <= end1 ? 1u : 0u;                             // rep2 becomes the old rep1
            const uint32_t hit = imm & (uint32_t)(q1 >> (E & 63u));
            const uint32_t hardj = (uint32_t)(hard >> j) & 1u;
            special = hardj | (back == 5u ? 1u : 0u) | (hit & 1u);
            el = j; isM = 1;
            ll = j - back - arel; offBase = off + 3u; ml = 4u + back + fw;
            post = imm ^ 1u;
            if (special) { why = hardj | (back == 5u ? 2u : 0u) | ((hit & 1u) << 4); how = 3; break; }
            rep2 = rep1; rep1 = off;
            q2 = q1; end2 = end1;
            eq1 = (uint64_t)(eqb(j) >> 4) << j;                                       // = map_from_window(eqb(j), j)
            q1 = quad(eq1); end1 = j + 28 < 64 ? j + 28 : 64;
        }
        emit(ll, offBase, ml);
        arel = E; S = E;
        // table writes at el+1, el+2 and E-2 (:537, :680-681): lanes below lim join the batch commit; from the first one at or
        // beyond lim on they follow one by one, in order, and the chain ends with this sequence - as it does when the immediate
        // rep2 test (:690) could not be decided from the maps
        if ((E - 2 >= lim ? 1u : 0u) | post) { how = 4; break; }
        commit |= span(Sev, el + 3) | (1ull << (E - 2));                              // the search's lanes, el+1, el+2, E-2 (el + 2 <= E - 2 < lim)
    }
    WalkOut o;
    o.outcome = ZC_RESTART; o.S = (int)S; o.R = (int)R; o.ev = 0; o.el = 0; o.why = 0; o.arel = (int32_t)arel; o.rep1 = rep1; o.rep2 = rep2;
    o.ntail = 0; o.tail[0] = o.tail[1] = o.tail[2] = 0; o.post = false;
    if (how == 1) { commit |= span(S, R); o.outcome = ZC_CONTINUE; }                  // (R > S: lim > S, and end1 - 5 > S or S == 0)
    else if (how == 2) o.outcome = end1 < 64 ? ZC_REFRESH : ZC_RESTART;
    else if (how == 3) { commit |= span(S, el + 1); o.outcome = ZC_SPECIAL; o.ev = isM ? 1 : 2; o.el = (int)el; o.why = (int)why; }
    else if (how == 4) {
        const int w1 = (int)el + 1, w2 = (int)el + 2, w3 = (int)E - 2, l = (int)lim;
        commit |= span(Sev, el + 1);
        o.tail[0] = w1; o.tail[1] = w2; o.tail[2] = w3;
        o.ntail = (w1 >= l) + (w2 >= l) + (w3 >= l);
        if (w1 < l) commit |= lane_bit(w1);
        if (w2 < l) commit |= lane_bit(w2);
        if (w3 < l) commit |= lane_bit(w3);
        o.post = post != 0;
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

// 8-bit mask of the zero bytes of the pair (x0, x1): bit k <-> byte k of x0, bit 4 + k <-> byte k of x1.  The zero flags
// (bit 7 of every zero byte) of x0 are moved into bit 3 of their bytes next to x1's in bit 7, and one multiplication gathers
// the eight of them in the product's top byte: flag 8k+3 and flag 8k+7 both travel 21-7k bits up (to 24+k and 28+k), and no two
// partial products share a bit, so there are no carries.
ZC_FN uint32_t zero_bytes8(uint32_t x0, uint32_t x1)
{
    const uint32_t t0 = ((x0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x0, t1 = ((x1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x1;
    const uint32_t f = ((~t0 >> 4) & 0x08080808u) | (~t1 & 0x80808080u);
    return (f * 0x00204081u) >> 24;
}

}  // namespace zchain
