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
// Byte equality at a fixed distance is kept as a 128-bit map (bit i <-> src[pos0+i] == src[pos0+i-dist], bits outside
// the known range are 0): for the two repcodes the batch starts with, the maps come from two coalesced loads; after an
// M event the new rep1 is that match's offset and its map is the equality mask of the lane's own 32-byte probe /
// candidate windows - the same bytes the forward extension compared - known up to 28 bytes past the probe.
// The walk stops as soon as anything is not decidable from registers:
//   ZC_SPECIAL   the next event is known (ev, el) but needs the one-at-a-time path (long extension or catch-up,
//                immediate rep2 match, table writes beyond the trustworthy lanes);
//   ZC_CONTINUE  no event below lim: the search goes on in the next batch, `lim - S` probes already examined;
//   ZC_RESTART   a fresh batch has to start at lane S (S is the start of a search; never returned before at least
//                one sequence was resolved, so the parse always advances).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define ZC_FN __device__ __forceinline__
#else
#define ZC_FN static inline
#endif

namespace zchain {

struct M128 { uint64_t lo, hi; };

ZC_FN M128 shr(M128 m, uint32_t s)
{
    M128 r;
    if (s == 0) return m;
    if (s < 64) { r.lo = (m.lo >> s) | (m.hi << (64 - s)); r.hi = m.hi >> s; }
    else if (s < 128) { r.lo = m.hi >> (s - 64); r.hi = 0; }
    else { r.lo = 0; r.hi = 0; }
    return r;
}
ZC_FN M128 shl(M128 m, uint32_t s)
{
    M128 r;
    if (s == 0) return m;
    if (s < 64) { r.hi = (m.hi << s) | (m.lo >> (64 - s)); r.lo = m.lo << s; }
    else if (s < 128) { r.hi = m.lo << (s - 64); r.lo = 0; }
    else { r.lo = 0; r.hi = 0; }
    return r;
}
ZC_FN M128 band(M128 a, M128 b) { M128 r; r.lo = a.lo & b.lo; r.hi = a.hi & b.hi; return r; }
ZC_FN uint64_t bits_at(M128 m, uint32_t i) { return shr(m, i).lo; }
ZC_FN int ctz64(uint64_t v) { return v ? (int)__builtin_ctzll(v) : 64; }

// equality map at one distance: eq = byte equality, q = "4 equal bytes start here", known for indices < end
struct Desc { M128 eq, q; int32_t end; };

ZC_FN Desc make_desc(M128 eq, int32_t end)
{
    Desc d; d.eq = eq;
    const M128 a = band(eq, shr(eq, 1));
    d.q = band(a, shr(a, 2));
    d.end = end;
    return d;
}
// map of the batch's initial repcodes: B bit j <-> byte pos0+1+j is equal (lanes 0..63), Bh bit j <-> byte pos0+65+j (lanes 0..27)
constexpr int32_t kRepEnd = 93;
ZC_FN Desc desc_from_ballots(uint64_t B, uint64_t Bh)
{
    M128 eq; eq.lo = B << 1; eq.hi = (B >> 63) | ((Bh & 0xFFFFFFFull) << 1);
    return make_desc(eq, kRepEnd);
}
// map of the offset of the match found by lane j: eqb bit i <-> byte (pos0 + j - 4 + i) equals its candidate byte
ZC_FN Desc desc_from_window(uint32_t eqb, int j)
{
    M128 eq; eq.lo = eqb & ~0xFull; eq.hi = 0;
    eq = j >= 4 ? shl(eq, (uint32_t)(j - 4)) : shr(eq, (uint32_t)(4 - j));
    return make_desc(eq, j + 28);
}

// per-lane record of a table candidate, packed for one readlane: candidate position (17 bits) | forward length beyond
// the 4 match bytes, 0..24 (24 = runs out of the window) << 17 | catch-up bytes 0..4 (7 = needs the exact path) << 22
ZC_FN uint32_t pack_info(uint32_t c, uint32_t fw, uint32_t bk4) { return c | (fw << 17) | (bk4 << 22); }

enum { ZC_RESTART = 0, ZC_SPECIAL = 1, ZC_CONTINUE = 2 };

struct WalkOut {
    int outcome;
    int S;              // lane where the unfinished search starts
    int ev, el;         // ZC_SPECIAL: 1 = table match at lane el, 2 = repcode found by even probe el
    int32_t arel;       // anchor - pos0
    uint32_t rep1, rep2;
    uint64_t commit;    // lanes whose table write (position + 2) the reference has executed
    int ntail, tail[3]; // ZC_RESTART: the LAST ntail of tail[0..2] (lane indices, may exceed 63) are table writes to perform one
                        // by one, in order, after the commit
    bool post;          // ZC_RESTART: the immediate rep2 test at S (:690) is still to be made, with loads
    int why;            // ZC_SPECIAL, statistics only: 1 long extension, 2 long catch-up / candidate at the chunk start,
                        // 4 table writes beyond lim, 8 immediate rep2 test not decidable, 16 immediate rep2 match
};

ZC_FN uint64_t lane_range(int a, int b)   // bits [a, b), 0 <= a, b <= 64
{
    if (b <= a) return 0;
    const uint64_t hi = b >= 64 ? ~0ull : ((1ull << b) - 1);
    return hi & ~((1ull << a) - 1);       // a < 64 here
}
ZC_FN uint64_t lane_bit(int i) { return (i >= 0 && i < 64) ? 1ull << i : 0; }

// the last sequence of a chain: its three table writes w1 <= w2 <= w3; those below lim join the commit mask, the others
// (a suffix, because the lanes are ordered) are left to the caller
ZC_FN void finish(WalkOut& o, uint64_t& commit, int w1, int w2, int w3, int lim, bool post)
{
    o.tail[0] = w1; o.tail[1] = w2; o.tail[2] = w3;
    o.ntail = (w1 >= lim) + (w2 >= lim) + (w3 >= lim);
    if (w1 < lim) commit |= lane_bit(w1);
    if (w2 < lim) commit |= lane_bit(w2);
    if (w3 < lim) commit |= lane_bit(w3);
    o.post = post;
    o.outcome = ZC_RESTART;
}

// info(j), eqb(j): per-lane inputs of lane j; emit(litLength, offBase, matchLength, anchorPosition) stores one sequence.
template <class Info, class Eqb, class Emit>
ZC_FN WalkOut walk(const uint64_t okmask, const int lim, const uint32_t pos0, int32_t arel, uint32_t rep1, uint32_t rep2,
                   Desc D1, Desc D2, Info info, Eqb eqb, Emit emit)
{
    WalkOut o;
    int S = 0;
    uint64_t commit = 0;
    o.ev = 0; o.el = 0; o.why = 0; o.ntail = 0; o.post = false; o.tail[0] = o.tail[1] = o.tail[2] = 0;
    for (;;) {
        if (S >= lim) { o.outcome = ZC_RESTART; break; }
        const uint64_t mm = okmask & (~0ull << S);
        const int jm = ctz64(mm);
        const uint64_t t = bits_at(D1.q, (uint32_t)S + 2) & 0x5555555555555555ull;
        const int kr = t ? S + (int)__builtin_ctzll(t) : 1000;
        const int KL = D1.end - 6;                       // even probes <= KL have a known repcode test
        int ev;
        if (kr <= jm) ev = kr < lim ? 2 : 0;             // (kr found: every earlier repcode test is known negative)
        else if (jm < 64) { if (!t && jm > KL) { o.outcome = ZC_RESTART; break; } ev = 1; }
        else ev = 0;
        if (ev == 0) {
            if (!t && lim - 1 > KL) { o.outcome = ZC_RESTART; break; }
            commit |= lane_range(S, lim);
            o.outcome = ZC_CONTINUE; break;
        }
        if (ev == 2) {
            const int k = kr, q = k + 2;
            const uint32_t back1 = (uint32_t)(bits_at(D1.eq, (uint32_t)q - 1) & 1);          // :533
            const int f = ctz64(~bits_at(D1.eq, (uint32_t)q + 4));
            const int E = q + 4 + f;
            // table writes at k+1, k+2 and E-2 (:537, :680-681): lanes below lim join the batch commit; from the first one
            // at or beyond lim on they follow one by one (`tail`, in order) and the chain ends with this sequence
            const bool tails = E - 2 >= lim;
            int why = (E < D1.end && f < 64 ? 0 : 1) | (tails && k + 2 >= 64 ? 4 : 0);   // f == 64: bits_at holds 64 bits
            bool post = false;                                                                    // :690 immediate rep2 test
            if (!why && rep2) {
                if (E + 4 <= D2.end) why = (bits_at(D2.q, (uint32_t)E) & 1) ? 16 : 0;
                else post = true;                                                                 // by two loads after the walk
            }
            commit |= lane_range(S, k + 1);
            if (why) { o.outcome = ZC_SPECIAL; o.ev = 2; o.el = k; o.why = why; break; }
            emit((uint32_t)(q - (int)back1 - arel), 1u, back1 + 4u + (uint32_t)f, (uint32_t)((int32_t)pos0 + arel));
            arel = E; S = E;
            if (tails || post) { finish(o, commit, k + 1, k + 2, E - 2, lim, post); break; }
            commit |= lane_bit(k + 1) | lane_bit(k + 2) | lane_bit(E - 2);
        } else {
            const int j = jm;
            const uint32_t in = info(j);
            const uint32_t c = in & 0x1FFFFu, fw = (in >> 17) & 31u, bk4 = (in >> 22) & 7u;
            const uint32_t room = (uint32_t)(j - arel);
            const uint32_t maxback = room < c ? room : c;
            const uint32_t back = bk4 < maxback ? bk4 : maxback;
            const int E = j + 4 + (int)fw;
            const bool tails = E - 2 >= lim;
            int why = (fw < 24 ? 0 : 1) | (bk4 != 7 && !(bk4 == 4 && maxback > 4) ? 0 : 2) | (tails && j + 2 >= 64 ? 4 : 0);
            bool post = false;                                                                    // rep2 becomes the old rep1
            if (!why) {
                if (E + 4 <= D1.end) why = (bits_at(D1.q, (uint32_t)E) & 1) ? 16 : 0;
                else post = true;
            }
            commit |= lane_range(S, j + 1);
            if (why) { o.outcome = ZC_SPECIAL; o.ev = 1; o.el = j; o.why = why; break; }
            const uint32_t off = pos0 + (uint32_t)j - c;
            emit((uint32_t)(j - (int)back - arel), off + 3u, 4u + back + fw, (uint32_t)((int32_t)pos0 + arel));
            rep2 = rep1; rep1 = off;
            arel = E; S = E;
            if (tails || post) { finish(o, commit, j + 1, j + 2, E - 2, lim, post); break; }
            commit |= lane_bit(j + 1) | lane_bit(j + 2) | lane_bit(E - 2);
            D2 = D1; D1 = desc_from_window(eqb(j), j);
        }
    }
    o.S = S; o.arel = arel; o.rep1 = rep1; o.rep2 = rep2; o.commit = commit;
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
