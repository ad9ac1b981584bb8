// csrc/zstd_entropy.hip.h -- Zstandard entropy-stage building blocks for the gfx950 kernels: table construction
// and bit packing on a 64-lane wavefront.
//
// The tables the reference builds with serial loops over symbols (/root/reference algos/zstd/lib/compress/
// fse_compress.c, huf_compress.c) are built here in the shape the wave offers: ONE SYMBOL PER LANE for the FSE
// alphabets (<= 53 symbols) and four per lane for the byte alphabet, so that
//   * sums / maxima over the alphabet are wave reductions, "first symbol with ..." is ballot + ctz;
//   * everything the reference accumulates while it walks the alphabet (remaining probability, cumulative state
//     counts, bit positions of a header) is a prefix sum over lanes;
//   * "the k-th state of symbol s" / "the k-th symbol with this code length" are ranks taken from lane masks;
//   * the sort of the byte alphabet is a rank sort (every symbol counts the keys in front of it).
// What stays serial is what is serial by definition: the two-queue Huffman merge (scalar, operands in lane registers),
// the length-limit repair loop, and the quicksort of a count bucket whose order for EQUAL counts is defined by the
// reference's partition scheme (one bucket per lane, all buckets at once).  Results - normalised counts, table
// descriptions, state tables, code lengths and code values - are bit-identical to the reference's; the cited lines
// say which rule each piece implements.
#pragma once
#include "common.hip.h"

namespace aoclgpu {
namespace zs {

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64; typedef int16_t s16;
// Every pointer of the entropy stage carries its address space (round 6): the tables and scratch areas are LDS, the sequence store,
// the literal buffer and the staged block are global memory.  As generic pointers (until round 5) every access of this stage
// was a flat_* instruction - an aperture check per access, and a wait on BOTH counters at every use.
#define ZLDS __attribute__((address_space(3)))
#define ZGLB __attribute__((address_space(1)))
typedef ZLDS u32 l32; typedef ZLDS u16 l16; typedef ZLDS u8 l8; typedef ZLDS int li32;
typedef ZGLB u32 g32; typedef ZGLB u16 g16; typedef ZGLB u8 g8;
__device__ __forceinline__ void lds_or(l32* p, u32 v) { __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_add(l32* p, u32 v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ int hb(u32 v) { return 31 - __builtin_clz(v); }          // ZSTD_highbit32

// ---- format constants (zstd_internal.h:100-169; RFC 8878 tables) ----
__device__ const u8 kLLbits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
__device__ const u8 kMLbits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                    1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
__device__ const s16 kLLnorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
__device__ const s16 kMLnorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                     1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
__device__ const s16 kOFnorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };
__device__ const u32 kLLbase[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,0x80,0x100,0x200,0x400,
                                     0x800,0x1000,0x2000,0x4000,0x8000,0x10000 };
__device__ const u32 kMLbase[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,
                                     35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003 };

__device__ __forceinline__ u32 ll_code(u32 v)
{   // ZSTD_LLcode (zstd_compress_internal.h:527-540)
    if (v > 63) return (u32)hb(v) + 19;
    if (v < 16) return v;
    if (v < 24) return 16 + ((v - 16) >> 1);
    if (v < 32) return 20 + ((v - 24) >> 2);
    if (v < 48) return 22 + ((v - 32) >> 3);
    return 24;
}
__device__ __forceinline__ u32 ml_code(u32 v)
{   // ZSTD_MLcode (zstd_compress_internal.h:545-556), v = matchLength - 3
    if (v > 127) return (u32)hb(v) + 36;
    if (v < 32) return v;
    if (v < 40) return 32 + ((v - 32) >> 1);
    if (v < 48) return 36 + ((v - 40) >> 2);
    if (v < 64) return 38 + ((v - 48) >> 3);
    if (v < 96) return 39 + ((v - 48) >> 4);
    return 42;
}

// ------------------------------------------------------------------------------------------------
// wave primitives
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 wsum(u32 v) { for (int d = 32; d; d >>= 1) v += (u32)__shfl_xor((int)v, d); return v; }
__device__ __forceinline__ u32 wmaxu(u32 v) { for (int d = 32; d; d >>= 1) { const u32 t = (u32)__shfl_xor((int)v, d); v = v > t ? v : t; } return v; }
__device__ __forceinline__ u32 wscan(u32 v, int lane)      // inclusive prefix sum over lanes
{
    for (int d = 1; d < 64; d <<= 1) { const u32 t = (u32)__shfl_up((int)v, d); if (lane >= d) v += t; }
    return v;
}
__device__ __forceinline__ u64 wscan64(u64 v, int lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        const u32 lo = (u32)__shfl_up((int)(u32)v, d), hi = (u32)__shfl_up((int)(u32)(v >> 32), d);
        if (lane >= d) v += ((u64)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ u32 wmaxscan(u32 v, int lane)   // inclusive prefix maximum over lanes
{
    for (int d = 1; d < 64; d <<= 1) { const u32 t = (u32)__shfl_up((int)v, d); if (lane >= d && t > v) v = t; }
    return v;
}
// number of set bits of `m` in lanes below this one
__device__ __forceinline__ u32 below(u64 m) { return (u32)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }
__device__ __forceinline__ int top_lane(u64 m) { return 63 - (int)__builtin_clzll(m | 1); }
// lanes that hold the same 6-bit value as this lane (among lanes with `on`)
__device__ __forceinline__ u64 same_value6(u32 v, bool on)
{
    u64 m = ballot(on);
#pragma unroll
    for (int b = 0; b < 6; ++b) { const u64 k = ballot((v >> b) & 1); m &= ((v >> b) & 1) ? k : ~k; }
    return m;
}

// Bits of one lane each, concatenated in lane order into `out` (global memory), little-endian bit order; returns the
// number of bytes (the last one zero-padded).  tile: LDS scratch of at least (total bits / 32 + 3) words.
__device__ u32 wave_emit_bits(g8* out, u64 v, u32 nb, l32* tile, u32 tileWords, int lane)
{
    const u32 incl = wscan(nb, lane), total = lane_get(incl, 63);
    const u32 words = (total + 31) / 32 + 2;
    for (u32 i = (u32)lane; i < words && i < tileWords; i += kWave) tile[i] = 0;
    wave_fence();
    if (nb) {
        if (nb < 64) v &= (1ull << nb) - 1;
        const u32 bit = incl - nb, w = bit >> 5, sh = bit & 31;
        const u64 a = v << sh;
        const u32 a2 = sh ? (u32)(v >> (64 - sh)) : 0;
        if ((u32)a) lds_or(&tile[w], (u32)a);
        if ((u32)(a >> 32)) lds_or(&tile[w + 1], (u32)(a >> 32));
        if (a2) lds_or(&tile[w + 2], a2);
    }
    wave_fence();
    const u32 nbytes = (total + 7) >> 3;
    const l8* tb = (const l8*)tile;
    for (u32 i = (u32)lane; i < nbytes; i += kWave) out[i] = tb[i];
    wave_fence();
    return nbytes;
}

// ------------------------------------------------------------------------------------------------
// FSE tables, one symbol per lane (fse_compress.c)
// ------------------------------------------------------------------------------------------------
struct FseCT {            // FSE_CTable as plain arrays (tableLog <= 9 on this path)
    int tableLog;
    u16 stateTable[512];
    int deltaNbBits[56];
    int deltaFindState[56];
};
struct FseScratch {       // LDS scratch of the table builders
    u8 cell[512];         // state -> symbol
    u8 occ[512];          // spread: k-th occurrence -> symbol
    u16 next[64];         // next free state-table slot per symbol
    u32 bits[40];         // header bits
};

__device__ __forceinline__ u32 fse_min_log(u32 srcSize, u32 maxSym)
{   // fse_compress.c:346-354
    const u32 a = (u32)hb(srcSize) + 1, b = (u32)hb(maxSym) + 2;
    return a < b ? a : b;
}
__device__ __forceinline__ u32 fse_optimal_log(u32 maxLog, u32 srcSize, u32 maxSym, u32 minus)
{   // fse_compress.c:356-373
    const u32 maxBitsSrc = (u32)hb(srcSize - 1) - minus;
    u32 t = maxLog; const u32 minBits = fse_min_log(srcSize, maxSym);
    if (maxBitsSrc < t) t = maxBitsSrc;
    if (minBits > t) t = minBits;
    if (t < 5) t = 5;
    if (t > 12) t = 12;
    return t;
}

// Secondary normalisation, used when the correction of the primary one would eat more than half of the largest
// probability (FSE_normalizeM2, fse_compress.c:376-461).  0 ok, -1 error.
__device__ int fse_renormalize_wave(int& norm, const u32 c, u32 total, const u32 maxSym, const u32 tableLog, const int lowProb, int lane)
{
    const bool valid = (u32)lane <= maxSym, live = valid && c != 0;
    const u32 lowThreshold = total >> tableLog;
    u32 lowOne = (u32)(((u64)total * 3) >> (tableLog + 1));
    bool open = false;                                       // probability still to be assigned
    {
        const bool isLow = live && c <= lowThreshold, isOne = live && !isLow && c <= lowOne;
        norm = !live ? 0 : (isLow ? lowProb : (isOne ? 1 : 0));
        open = live && !isLow && !isOne;
    }
    u32 distributed = wsum((live && !open) ? 1u : 0u);
    total -= wsum((live && !open) ? c : 0u);
    u32 toDist = (1u << tableLog) - distributed;
    if (toDist == 0) return 0;
    if (total / toDist > lowOne) {                           // a second round with a higher bar for "1"
        lowOne = (u32)(((u64)total * 3) / (toDist * 2));
        const bool now = open && c <= lowOne;
        if (now) { norm = 1; open = false; }
        distributed += wsum(now ? 1u : 0u);
        total -= wsum(now ? c : 0u);
        toDist = (1u << tableLog) - distributed;
    }
    if (distributed == maxSym + 1) {                         // everything is small: the most frequent symbol takes the rest
        const u32 cmax = wmaxu(valid ? c : 0u);
        const int who = first_lane(ballot(valid && c == cmax && cmax > 0));
        if (lane == (who < 64 ? who : 0)) norm += (int)toDist;
        return 0;
    }
    if (total == 0) {                                        // only 1s and lows are left to grow: round robin over the positive ones
        const u64 pos = ballot(norm > 0);
        const u32 p = (u32)__builtin_popcountll(pos);
        if (p == 0) return -1;
        if ((pos >> lane) & 1) norm += (int)(toDist / p + (below(pos) < toDist % p ? 1u : 0u));
        return 0;
    }
    {   // the open symbols share toDist in proportion to their counts: positions on a 2^vStepLog grid, prefix sum over lanes
        const u32 vStepLog = 62 - tableLog;
        const u64 mid = (1ull << (vStepLog - 1)) - 1;
        const u64 rStep = (((1ull << vStepLog) * toDist) + mid) / total;
        const u64 mine = open ? (u64)c * rStep : 0;
        const u64 end = mid + wscan64(mine, lane), start = end - mine;
        const u32 w = (u32)(end >> vStepLog) - (u32)(start >> vStepLog);
        if (ballot(open && w < 1)) return -1;
        if (open) norm = (int)w;
    }
    return 0;
}

// Normalised counts for a table of 2^tableLog states (FSE_normalizeCount, fse_compress.c:464-526).  c = this lane's
// count (0 beyond maxSym).  Returns tableLog, 0 for a single-symbol distribution, -1 on error; norm = this lane's result.
__device__ int fse_normalize_wave(int& norm, const u32 c, const u32 total, const u32 maxSym, const u32 tableLog, const bool useLowProb, int lane)
{
    norm = 0;
    if (tableLog < 5 || tableLog > 12 || tableLog < fse_min_log(total, maxSym)) return -1;
    const bool valid = (u32)lane <= maxSym;
    if (ballot(valid && c == total)) return 0;
    const int lowProb = useLowProb ? -1 : 1;
    const u32 scale = 62 - tableLog;
    const u64 step = (1ull << 62) / total, vStep = 1ull << (scale - 20);
    const u32 lowThreshold = total >> tableLog;
    const bool live = valid && c != 0, isLow = live && c <= lowThreshold, big = live && !isLow;
    u32 proba = 0;
    if (big) {
        const u64 x = (u64)c * step;
        proba = (u32)(x >> scale);
        if (proba < 8) {                                     // rounding thresholds of the small probabilities (rtbTable :467)
            const u32 rtb = proba == 0 ? 0u : (proba == 1 ? 473195u : (proba == 2 ? 504333u : (proba == 3 ? 520860u :
                            (proba == 4 ? 550000u : (proba == 5 ? 700000u : (proba == 6 ? 750000u : 830000u))))));
            proba += (x - ((u64)proba << scale)) > vStep * rtb ? 1u : 0u;
        }
    }
    norm = isLow ? lowProb : (int)proba;
    const int still = (1 << tableLog) - (int)wsum(isLow ? 1u : proba);
    const u32 pmax = wmaxu(proba);
    const int fl = first_lane(ballot(big && proba == pmax));   // the first symbol with the largest probability
    const int largest = fl < 64 ? fl : 0;
    const int nl = (int)lane_get((u32)norm, largest);
    if (-still >= (nl >> 1)) return fse_renormalize_wave(norm, c, total, maxSym, tableLog, lowProb, lane) ? -1 : (int)tableLog;
    if (lane == largest) norm += still;
    return (int)tableLog;
}

// Table description (FSE_writeNCount, fse_compress.c:236-329): 4 bits of tableLog, then per symbol a value whose width
// follows the probability still unassigned, zero runs as repeat codes.  What the reference tracks while walking the
// alphabet - `remaining`, the width - is a prefix sum over lanes here, every lane forms its own bits (<= 53), and the bit
// positions are another prefix sum.  Returns bytes written to out (global), 0 on error.
__device__ u32 fse_write_ncount_wave(g8* out, const int norm, const u32 maxSym, const u32 tableLog, ZLDS FseScratch* sc, int lane)
{
    const int tableSize = 1 << tableLog;
    const bool valid = (u32)lane <= maxSym;
    const u32 a = valid ? (u32)(norm < 0 ? -norm : norm) : 0u;
    const u32 incl = wscan(a, lane);
    const int remBefore = tableSize + 1 - (int)(incl - a), remAfter = remBefore - (int)a;
    const u64 nz = ballot(valid && norm != 0);
    const bool prevZero = lane > 0 && !((nz >> (lane - 1)) & 1);
    // a symbol gets its own value unless it continues a run of zeros (the run's first zero does); nothing follows once
    // the whole probability is assigned
    const bool emit = valid && remBefore > 1 && (norm != 0 || !prevZero);
    if (ballot(emit && remAfter < 1)) return 0;
    const u64 em = ballot(emit);
    if (!em || (int)lane_get((u32)remAfter, top_lane(em)) != 1) return 0;
    const int thr = remBefore >= tableSize ? tableSize : (1 << hb((u32)remBefore));
    const u32 width = (u32)hb((u32)thr) + 1;
    const int mx = (2 * thr - 1) - remBefore;
    int cnt = norm + 1;
    if (cnt >= thr) cnt += mx;
    u64 v = 0; u32 nb = 0;
    if (emit) {
        if (lane == 0) { v = tableLog - 5; nb = 4; }
        if (norm != 0 && prevZero) {                         // zeros since the run's first one: 24 at a time, 3 at a time, rest
            const u64 before = nz & ((1ull << lane) - 1);
            const int prevNz = before ? top_lane(before) : -1;
            const u32 r = (u32)(lane - 1 - prevNz) - 1u;
            const u32 rem = r % 24, ones = 16 * (r / 24) + 2 * (rem / 3);
            v |= (((1ull << ones) - 1) | ((u64)(rem % 3) << ones)) << nb;
            nb += ones + 2;
        }
        v |= (u64)(u32)cnt << nb;
        nb += width - (cnt < mx ? 1u : 0u);
    }
    return wave_emit_bits(out, v, nb, sc->bits, 40, lane);
}

// State -> symbol map shared by the encoding and the decoding tables (FSE_buildCTable_wksp fse_compress.c:68-110 and
// ZSTD_buildFSETable zstd_decompress_block.c:497-560 spread identically).  The reference steps through the table;
// occurrence k of the spread lands on the k-th visited state that is not reserved for a low-probability symbol, so the
// lanes take the visit numbers 64 at a time: reserved states are skipped by a ballot count, the symbol of occurrence k
// comes from a prefix maximum over "first occurrence" marks.  put(state, symbol) stores the result; occ = 2^tableLog
// bytes of LDS scratch.
template <class Occ, class Put>
__device__ __forceinline__ void fse_spread_wave(Occ* occ, const int norm, const u32 maxSym, const u32 tableLog, int lane, Put put)
{
    const u32 tableSize = 1u << tableLog, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    const bool valid = (u32)lane <= maxSym;
    const u64 lowm = ballot(valid && norm == -1);
    const u32 high = tableSize - 1 - (u32)__builtin_popcountll(lowm);
    // low-probability symbols own the top states, in symbol order from the top down
    if ((lowm >> lane) & 1) put(tableSize - 1 - below(lowm), (u32)lane);
    // occurrence -> symbol: mark every spread symbol's first occurrence, prefix maximum (symbols ascend with k)
    const u32 sp = (valid && norm > 0) ? (u32)norm : 0u;
    const u32 scum = wscan(sp, lane) - sp;
    for (u32 i = (u32)lane; i < tableSize; i += kWave) occ[i] = 0;
    wave_fence();
    if (sp) occ[scum] = (u8)(lane + 1);
    wave_fence();
    u32 carry = 0;
    for (u32 k0 = 0; k0 < tableSize; k0 += kWave) {
        const u32 k = k0 + (u32)lane;
        u32 m = k < tableSize ? (u32)occ[k] : 0u;
        m = wmaxscan(m, lane);
        m = m > carry ? m : carry;
        if (k < tableSize) occ[k] = (u8)m;
        carry = lane_get(m, 63);
    }
    wave_fence();
    // visit j of the walk is state (j * step) & mask; visits of reserved states do not consume an occurrence
    u32 skipped = 0;
    for (u32 j0 = 0; j0 < tableSize; j0 += kWave) {
        const u32 j = j0 + (u32)lane, p = (j * step) & mask;
        const bool in = j < tableSize, skip = in && p > high;
        const u64 sk = ballot(skip);
        if (in && !skip) put(p, (u32)occ[j - skipped - below(sk)] - 1u);
        skipped += (u32)__builtin_popcountll(sk);
    }
    wave_fence();
}

// Encoding table (FSE_buildCTable_wksp, fse_compress.c:68-214): spread, then a symbol's states are numbered in
// increasing state order: rank among the lanes holding the same symbol + the symbol's running base.
__device__ void fse_build_ctable_wave(ZLDS FseCT* ct, const int norm, const u32 maxSym, const u32 tableLog, ZLDS FseScratch* sc, int lane)
{
    const u32 tableSize = 1u << tableLog;
    const bool valid = (u32)lane <= maxSym;
    const u32 w = !valid ? 0u : (norm == -1 ? 1u : (norm > 0 ? (u32)norm : 0u));       // states the symbol owns
    const u32 cum = wscan(w, lane) - w;                                               // first of them in the state table
    if (lane == 0) ct->tableLog = (int)tableLog;
    if (valid) {
        sc->next[lane] = (u16)cum;
        if (norm == 0) { ct->deltaNbBits[lane] = (int)(((tableLog + 1) << 16) - tableSize); ct->deltaFindState[lane] = 0; }
        else if (w == 1) { ct->deltaNbBits[lane] = (int)((tableLog << 16) - tableSize); ct->deltaFindState[lane] = (int)cum - 1; }
        else {
            const u32 maxBitsOut = tableLog - (u32)hb(w - 1), minStatePlus = w << maxBitsOut;
            ct->deltaNbBits[lane] = (int)((maxBitsOut << 16) - minStatePlus);
            ct->deltaFindState[lane] = (int)cum - (int)w;
        }
    }
    fse_spread_wave(sc->occ, norm, maxSym, tableLog, lane, [&](u32 state, u32 sym) { sc->cell[state] = (u8)sym; });
    // state table: the states of a symbol in increasing order (tableU16[cumul[s]++] = tableSize + u, :118-121)
    for (u32 u0 = 0; u0 < tableSize; u0 += kWave) {
        const u32 u = u0 + (u32)lane;
        const bool in = u < tableSize;
        const u32 sym = in ? (u32)sc->cell[u] : 0u;
        const u64 same = same_value6(sym, in);
        if (in) {
            const u32 base = sc->next[sym];
            ct->stateTable[base + below(same)] = (u16)(tableSize + u);
            if (lane == top_lane(same)) sc->next[sym] = (u16)(base + (u32)__builtin_popcountll(same));
        }
        wave_fence();
    }
}
__device__ __forceinline__ void fse_build_rle(ZLDS FseCT* ct, u32 sym)
{   // fse_compress.c:530-548
    ct->tableLog = 0; ct->stateTable[0] = 0; ct->stateTable[1] = 0;
    ct->deltaNbBits[sym] = 0; ct->deltaFindState[sym] = 0;
}
__device__ __forceinline__ u32 fse_init_state(const ZLDS FseCT* ct, u32 sym)
{   // FSE_initCState2 (fse.h:451-463)
    const u32 nb = (u32)(ct->deltaNbBits[sym] + (1 << 15)) >> 16;
    const u32 v = (nb << 16) - (u32)ct->deltaNbBits[sym];
    return ct->stateTable[(int)(v >> nb) + ct->deltaFindState[sym]];
}

// ------------------------------------------------------------------------------------------------
// Huffman code of the literals (huf_compress.c), byte alphabet: symbol 64*t + lane lives in lane `lane`, slot t
// ------------------------------------------------------------------------------------------------
struct HufCT { u8 nbBits[256]; u16 val[256]; u32 tableLog; };
struct HufScratch {
    u32 key[256];         // sort keys, then the sorted leaves: count << 8 | symbol
    u32 tmp[256];
    u8  len[256];         // code length per sorted leaf
    u8  weight[256];
    u32 cnt13[16];
    FseCT wct;            // FSE table of the weights
    FseScratch fs;
    u32 wbits[2][128];    // weight chain output: value << 8 | bits
    short todo[12][64];   // per-bucket work lists of the tie-order sort
};

#ifdef ZSTD_PROF            /* experiment build: sub-stages of huf_build_wave (tools/zstd_entropy_prof.py) */
__device__ unsigned long long g_zstd_hprof[8];
#define HSTAGE(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) atomicAdd(&g_zstd_hprof[k], t_ - hprev); hprev = t_; } while (0)
#else
#define HSTAGE(k) do { } while (0)
#endif
constexpr u32 kHufCutoff = 165;      // RANK_POSITION_DISTINCT_COUNT_CUTOFF = 158 + highbit32(158) (huf_compress.c:552)
// bucket of the reference's sort (huf_compress.c:560-570): one per count below the cutoff, one per power of two above
__device__ __forceinline__ u32 huf_bucket(u32 c) { return c < kHufCutoff ? c : (u32)hb(c) + 158; }

// The reference orders equal counts inside a power-of-two bucket by what its quicksort happens to do (partition around
// the last element; a range handed on as a whole is insertion-sorted when it has at most 8 elements, the rest of a
// partitioned range keeps being partitioned - huf_compress.c:572-622).  That order is part of the visible result (equal
// counts can get different code lengths), so such a bucket is sorted by the same scheme: a[lo..hi] are keys
// (count << 8 | symbol) in symbol order, compared by count only.  Runs on one lane per bucket, all buckets at once;
// `todo` is that lane's list of ranges still to do (ranges are disjoint, so their order is free).
__device__ void huf_bucket_sort(l32* a, int lo0, int hi0, ZLDS short* todo /* 64 */)
{
    auto insertion = [&](int lo, int hi) {                   // stable, descending by count
        for (int i = lo + 1; i <= hi; i++) {
            const u32 key = a[i]; int j = i - 1;
            while (j >= lo && (a[j] >> 8) < (key >> 8)) { a[j + 1] = a[j]; j--; }
            a[j + 1] = key;
        }
    };
    int sp = 0;
    todo[sp++] = (short)lo0; todo[sp++] = (short)hi0;
    while (sp > 0) {
        int hi = todo[--sp], lo = todo[--sp];
        if (hi - lo < 8) { insertion(lo, hi); continue; }
        while (lo < hi) {
            const u32 pivot = a[hi] >> 8; int i = lo - 1;    // counts above the pivot's to the front
            for (int j = lo; j < hi; j++) if ((a[j] >> 8) > pivot) { i++; const u32 t = a[i]; a[i] = a[j]; a[j] = t; }
            { const u32 t = a[i + 1]; a[i + 1] = a[hi]; a[hi] = t; }
            const int mid = i + 1;
            int cl, ch;                                      // the smaller side is handed on, the other one stays in this loop
            if (mid - lo < hi - mid) { cl = lo; ch = mid - 1; lo = mid + 1; }
            else { cl = mid + 1; ch = hi; hi = mid - 1; }
            if (ch - cl < 8) { if (ch > cl) insertion(cl, ch); }
            else if (sp < 62) { todo[sp++] = (short)cl; todo[sp++] = (short)ch; }
        }
    }
}

// Code lengths and code values of the literals (HUF_buildCTable_wksp, huf_compress.c:758-794, with HUF_sort :624-668,
// HUF_buildTree :685-725, HUF_setMaxHeight :378-495).  hist = counts of symbols 0..maxSym (LDS).  Returns the table log.
__device__ u32 huf_build_wave(ZLDS HufCT* ct_, const l32* hist_, const u32 maxSym_, const u32 maxNbBits_, ZLDS HufScratch* sc_, int lane)
{
    // (a call: the arguments arrive in vector registers and count as divergent - made scalar again, common.hip.h uni_ptr)
    ZLDS HufCT* ct = uni_ptr(ct_); const l32* hist = uni_ptr(hist_); const u32 maxSym = uni(maxSym_), maxNbBits = uni(maxNbBits_); ZLDS HufScratch* sc = uni_ptr(sc_);
    // ---- 1. sort: count descending; equal counts in symbol order (the reference's bucket fill order), except inside a
    //         power-of-two bucket of more than 8 symbols, where the quicksort's own order for ties applies ----
#ifdef ZSTD_PROF
    unsigned long long hprev = __builtin_amdgcn_s_memtime();
#endif
    u32 c[4], key[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const u32 s = 64u * t + (u32)lane;
        c[t] = s <= maxSym ? hist[s] : 0u;
        key[t] = (c[t] << 8) | (255u - s);                   // larger key = earlier in the order
        sc->key[s] = key[t];
    }
    wave_fence();
    u32 rank[4] = {0, 0, 0, 0};
    const u32 n1 = maxSym + 1;
    for (u32 j = 0; j < n1; ++j) {                           // rank sort: every symbol counts the keys in front of it
        const u32 kj = sc->key[j];                           // (one LDS broadcast per key)
#pragma unroll
        for (int t = 0; t < 4; ++t) rank[t] += kj > key[t] ? 1u : 0u;
    }
    wave_fence();
    HSTAGE(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) { const u32 s = 64u * t + (u32)lane; if (s <= maxSym) sc->tmp[rank[t]] = (c[t] << 8) | s; }
    wave_fence();
    const u32 nonNull = wsum((c[0] ? 1u : 0u) + (c[1] ? 1u : 0u) + (c[2] ? 1u : 0u) + (c[3] ? 1u : 0u));   // leaves 0..nonNull-1
    {   // power-of-two buckets: lane b owns bucket hb = 7 + b (counts 165.. fall into hb >= 7); sorted[] is count-descending,
        // so a bucket is one contiguous range.  The reference's sorting loop starts one bucket early (its bucket n holds the
        // symbols of index n - 1, huf_compress.c:649-664): the symbols whose count is exactly 164 are quicksorted as well -
        // all ties, so nine or more of them come out permuted.  Lane 11 owns that bucket.
        const u32 hbv = 7u + (u32)lane;
        int lo = -1, hi = -2;
        bool ties = false;
        if (lane < 12) {
            const u32 cmin = lane == 11 ? kHufCutoff - 1 : (hbv == 7 ? kHufCutoff : (1u << hbv));
            const u32 cmax = lane == 11 ? kHufCutoff - 1 : (2u << hbv) - 1;
            // entries with count in [cmin, cmax]: binary search on the descending array
            int a = 0, b = (int)nonNull;                     // first index with count <= cmax
            while (a < b) { const int m = (a + b) >> 1; if ((sc->tmp[m] >> 8) > cmax) a = m + 1; else b = m; }
            lo = a; b = (int)nonNull;                        // first index with count < cmin
            while (a < b) { const int m = (a + b) >> 1; if ((sc->tmp[m] >> 8) >= cmin) a = m + 1; else b = m; }
            hi = a - 1;
            if (hi - lo >= 8) for (int i = lo; i < hi; ++i) ties |= (sc->tmp[i] >> 8) == (sc->tmp[i + 1] >> 8);
        }
        HSTAGE(6);
        // back to the bucket's fill order (symbol ascending), then the reference's scheme.  The fill order needs no sort: a
        // symbol's place in its bucket is the number of smaller symbols with a count in the bucket's range - ballots over the
        // four symbol slots (until round 6 the owning lane insertion-sorted its bucket in LDS: 37 % of this function).
        for (u64 tm = ballot(lane < 12 && ties); tm; tm &= tm - 1) {
            const int bk = (int)__builtin_ctzll(tm);
            const u32 bmin = bk == 11 ? kHufCutoff - 1 : (bk == 0 ? kHufCutoff : (1u << (7 + bk)));
            const u32 bmax = bk == 11 ? kHufCutoff - 1 : (2u << (7 + bk)) - 1;
            const u32 blo = lane_get((u32)lo, bk);
            u32 before = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool in = c[t] >= bmin && c[t] <= bmax;            // (c is 0 beyond maxSym and for absent symbols)
                const u64 m = ballot(in);
                if (in) sc->tmp[blo + before + below(m)] = (c[t] << 8) | (64u * t + (u32)lane);
                before += (u32)__builtin_popcountll(m);
            }
        }
        wave_fence();
        HSTAGE(7);
        if (lane < 12 && ties) huf_bucket_sort(sc->tmp, lo, hi, sc->todo[lane]);
    }
    wave_fence();
    // sorted leaves, four per lane: leaf i = 64*t + lane
    u32 leafCnt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const u32 i = 64u * t + (u32)lane; leafCnt[t] = i < nonNull ? sc->tmp[i] >> 8 : 0u; }

    HSTAGE(1);
    // ---- 2. tree: two-queue merge.  Serial by definition; runs as scalar code with the queues in lane registers
    //         (leaf i in lane i & 63 of leafCnt[i >> 6], internal node k likewise in intCnt) ----
    //         Round 6: the queues live in LDS (leaf counts = the sorted array, internal counts in the weight chains' scratch,
    //         parent links in the sort keys' place, all dead here) and the loop is scalar code around uniform-address reads -
    //         the next leaf's count is requested when a leaf is taken, an internal node's count was written a node ago at
    //         least.  (Until then the queues were lane registers: every access a four-way register select around a readlane
    //         or four compare-and-select writes, ~670 cycles per pick.)
    u32 leafPar[4] = {0, 0, 0, 0}, intPar[4] = {0, 0, 0, 0};
    const int nInt = (int)nonNull - 1;                       // internal nodes, created in non-decreasing count order
    {
        l16* lpar = (l16*)sc->key; l16* ipar = lpar + 256; l32* icnt = (l32*)sc->wbits; l32* dummy = (l32*)sc->todo;
        // (stores of a uniform value: lane 0 writes the slot, the other lanes a scratch word of their own - no exec change)
        auto put16 = [&](l16* a, int i, u32 v) { l16* q = lane == 0 ? a + i : (l16*)(dummy + lane); *q = (u16)v; };
        auto put32 = [&](l32* a, int i, u32 v) { l32* q = lane == 0 ? a + i : dummy + lane; *q = v; };
        int li = (int)nonNull - 1, ii = 0;                   // smallest unused leaf, oldest unused internal node
        u32 lc = uni(sc->tmp[li] >> 8), ic = 0;
        u32 lcn = li > 0 ? sc->tmp[li - 1] >> 8 : 0u;        // the leaf behind it, on its way
        for (int k = 0; k < nInt; ++k) {
            u32 sum = 0;
#pragma unroll
            for (int pick = 0; pick < 2; ++pick) {           // the smaller of the two queue heads; a leaf only if strictly smaller
                const bool leaf = li >= 0 && (ii >= k || lc < ic);     // (huf_compress.c:707-708)
                if (leaf) {
                    sum += lc; put16(lpar, li, (u32)k); li--;
                    lc = uni(lcn); lcn = li > 0 ? sc->tmp[li - 1] >> 8 : 0u;
                }
                else { sum += ic; put16(ipar, ii, (u32)k); ii++; ic = ii < k ? uni(icnt[ii]) : 0u; }
            }
            put32(icnt, k, sum);
            if (ii == k) ic = sum;                           // the node just made is the queue's head when the queue was empty
        }
        wave_fence();
#pragma unroll
        for (int t = 0; t < 4; ++t) { leafPar[t] = lpar[64 * t + lane]; intPar[t] = ipar[64 * t + lane]; }
    }
    auto rd = [&](const u32 (&r)[4], int i) -> u32 {
        const int t = i >> 6, l = i & 63;
        return t == 0 ? lane_get(r[0], l) : (t == 1 ? lane_get(r[1], l) : (t == 2 ? lane_get(r[2], l) : lane_get(r[3], l)));
    };
    HSTAGE(2);
    // depth of every internal node below the root (node nInt-1): pointer jumping over the parent links
    u32 dep[4], par[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int k = 64 * t + lane; const bool root = k >= nInt - 1; dep[t] = root ? 0u : 1u; par[t] = root ? (u32)(nInt > 0 ? nInt - 1 : 0) : intPar[t]; }
    auto gather = [&](const u32 (&r)[4], u32 idx) -> u32 {
        const u32 l = idx & 63, t = idx >> 6;
        const u32 a0 = (u32)__shfl((int)r[0], (int)l), a1 = (u32)__shfl((int)r[1], (int)l), a2 = (u32)__shfl((int)r[2], (int)l), a3 = (u32)__shfl((int)r[3], (int)l);
        return t == 0 ? a0 : (t == 1 ? a1 : (t == 2 ? a2 : a3));
    };
    for (int round = 0; round < 8; ++round) {
        u32 nd[4], np[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { nd[t] = dep[t] + gather(dep, par[t]); np[t] = gather(par, par[t]); }
#pragma unroll
        for (int t = 0; t < 4; ++t) { dep[t] = nd[t]; par[t] = np[t]; }
    }
    u32 len[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const u32 i = 64u * t + (u32)lane; len[t] = i < nonNull ? gather(dep, leafPar[t]) + 1u : 0u; }
    if (nonNull == 1) len[0] = lane == 0 ? 1u : len[0];

    HSTAGE(3);
    // ---- 3. length limit (HUF_setMaxHeight).  Leaves are in count-descending = length-ascending order ----
    u32 bits;
    {
        // the reference takes the length of the last (rarest) leaf as the largest one and cuts from the end while lengths
        // exceed the target (:384-405)
        const u32 largest = rd(len, (int)nonNull - 1);
        bits = largest;
        if (largest > maxNbBits) {
            const u32 target = maxNbBits;
            int n0 = -1;                                     // last leaf that is already short enough
#pragma unroll
            for (int t = 0; t < 4; ++t) { const u64 m = ballot(64u * t + (u32)lane < nonNull && len[t] <= target); if (m) n0 = 64 * t + top_lane(m); }
            // too-long codes are cut to the target; the debt (in units of the deepest level) is paid back below
            u32 cost = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = 64 * t + lane;
                if (i > n0 && (u32)i < nonNull) { cost += (1u << (largest - target)) - (1u << (largest - len[t])); len[t] = target; }
            }
            int totalCost = (int)(wsum(cost) >> (largest - target));
#pragma unroll
            for (int t = 0; t < 4; ++t) { const u32 i = 64u * t + (u32)lane; if (i < nonNull) { sc->len[i] = (u8)len[t]; } }
            wave_fence();
            if (lane == 0) {
                // rankLast[r]: last leaf whose length is target - r.  Each step lengthens the cheapest such leaf that pays
                // off the largest affordable power of two (:430-476), then hands back any overpayment (:478-492).
                const u32 none = 0xFFFFFFFFu;
                u32 rankLast[14];
                for (int r = 0; r < 14; ++r) rankLast[r] = none;
                int n = (int)nonNull - 1;
                while (n >= 0 && sc->len[n] == target) n--;
                {   u32 cur = target;
                    for (int pos = n; pos >= 0; pos--) { const u32 l = sc->len[pos]; if (l >= cur) continue; cur = l; rankLast[target - cur] = (u32)pos; }
                }
                while (totalCost > 0) {
                    u32 dec = (u32)hb((u32)totalCost) + 1;
                    for (; dec > 1; dec--) {
                        const u32 hp = rankLast[dec], lp = rankLast[dec - 1];
                        if (hp == none) continue;
                        if (lp == none) break;
                        if ((sc->tmp[hp] >> 8) <= 2 * (sc->tmp[lp] >> 8)) break;
                    }
                    while (dec <= 12 && rankLast[dec] == none) dec++;
                    totalCost -= 1 << (dec - 1);
                    sc->len[rankLast[dec]]++;
                    if (rankLast[dec - 1] == none) rankLast[dec - 1] = rankLast[dec];
                    if (rankLast[dec] == 0) rankLast[dec] = none;
                    else { rankLast[dec]--; if (sc->len[rankLast[dec]] != target - dec) rankLast[dec] = none; }
                }
                while (totalCost < 0) {
                    if (rankLast[1] == none) {
                        while (n >= 0 && sc->len[n] == target) n--;
                        sc->len[n + 1]--; rankLast[1] = (u32)(n + 1); totalCost++;
                        continue;
                    }
                    sc->len[rankLast[1] + 1]--; rankLast[1]++; totalCost++;
                }
            }
            wave_fence();
#pragma unroll
            for (int t = 0; t < 4; ++t) { const u32 i = 64u * t + (u32)lane; len[t] = i < nonNull ? (u32)sc->len[i] : 0u; }
            bits = target;
        }
    }
    HSTAGE(4);
    // ---- 4. canonical values: symbols of one length are numbered in symbol order, the lengths' first values come from
    //         the per-length counts, longest first (:776-790) ----
    for (u32 s = (u32)lane; s < 256; s += kWave) ct->nbBits[s] = 0;
    wave_fence();
#pragma unroll
    for (int t = 0; t < 4; ++t) { const u32 i = 64u * t + (u32)lane; if (i < nonNull) ct->nbBits[sc->tmp[i] & 0xFF] = (u8)len[t]; }
    wave_fence();
    u32 sl[4];                                               // code length by symbol (slot t = symbols 64t..64t+63)
#pragma unroll
    for (int t = 0; t < 4; ++t) sl[t] = ct->nbBits[64u * t + (u32)lane];
    u32 val[4] = {0, 0, 0, 0};
    {
        u32 first = 0;                                       // first value of the current length
        for (u32 b = bits; b > 0; --b) {
            u32 before = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u64 m = ballot(sl[t] == b);
                if (sl[t] == b) val[t] = first + before + below(m);
                before += (u32)__builtin_popcountll(m);
            }
            first = (first + before) >> 1;
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) ct->val[64u * t + (u32)lane] = (u16)val[t];
    if (lane == 0) ct->tableLog = bits;
    wave_fence();
    HSTAGE(5);
    return bits;
}

// The code lengths as the frame stores them (HUF_writeCTable_wksp, huf_compress.c:256-295): weights, FSE-compressed when
// that pays (HUF_compressWeights :173-213), else packed two per byte.  Returns bytes written to op (global), 0 on failure.
__device__ u32 huf_write_table_wave(g8* op_, const ZLDS HufCT* ct_, const u32 maxSym_, const u32 huffLog_, ZLDS HufScratch* sc_, int lane)
{
    g8* op = uni_ptr(op_); const ZLDS HufCT* ct = uni_ptr(ct_); const u32 maxSym = uni(maxSym_), huffLog = uni(huffLog_); ZLDS HufScratch* sc = uni_ptr(sc_);   // (a call: see huf_build_wave)
    // weight of symbol s < maxSym (the last symbol's is implied)
    u32 wsym[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const u32 s = 64u * t + (u32)lane;
        const u32 l = s < maxSym ? (u32)ct->nbBits[s] : 0u;
        wsym[t] = l ? huffLog + 1 - l : 0u;
        if (s < maxSym) sc->weight[s] = (u8)wsym[t];
    }
    if (lane == 0) sc->weight[maxSym] = 0;
    wave_fence();
    u32 h = 0;
    const u32 n = maxSym;
    if (n > 1) {                                             // FSE over the 13 possible weights
        u32 cw = 0;                                          // count of weight `lane`
        for (u32 v = 0; v <= 12; ++v) {
            const u32 k = (u32)__builtin_popcountll(ballot(wsym[0] == v && (u32)lane < n)) + (u32)__builtin_popcountll(ballot(wsym[1] == v && 64u + (u32)lane < n)) +
                          (u32)__builtin_popcountll(ballot(wsym[2] == v && 128u + (u32)lane < n)) + (u32)__builtin_popcountll(ballot(wsym[3] == v && 192u + (u32)lane < n));
            if ((u32)lane == v) cw = k;
        }
        const u64 present = ballot(cw != 0);
        const u32 maxW = (u32)top_lane(present), maxCount = wmaxu(cw);
        if (maxCount == n) h = 1;                            // one weight only: not compressible this way
        else if (maxCount != 1) {
            const u32 tableLog = fse_optimal_log(6, n, maxW, 2);
            int norm = 0;
            if (fse_normalize_wave(norm, cw, n, maxW, tableLog, false, lane) >= 0) {
                const u32 hd = fse_write_ncount_wave(op + 1, norm, maxW, tableLog, &sc->fs, lane);
                if (hd) {
                    fse_build_ctable_wave(&sc->wct, norm, maxW, tableLog, &sc->fs, lane);
                    wave_fence();
                    // FSE_compress_usingCTable (fse_compress.c:558-615) on n > 2 weights, last to first, two interleaved
                    // states: lane k runs the chain of the weights with index parity k, then all lanes place the bits
                    if (n > 2) {
                        const ZLDS FseCT* t = &sc->wct;
                        if (lane < 2) {
                            u32 st = 0; bool started = false;
                            for (int i = (int)n - 1; i >= 0; --i) if ((i & 1) == lane) {
                                const u32 sym = sc->weight[i];
                                if (!started) { st = fse_init_state(t, sym); started = true; sc->wbits[lane][i >> 1] = 0; }
                                else {
                                    const u32 nb = (st + (u32)t->deltaNbBits[sym]) >> 16;
                                    sc->wbits[lane][i >> 1] = ((st & ((1u << nb) - 1)) << 8) | nb;
                                    st = t->stateTable[(int)(st >> nb) + t->deltaFindState[sym]];
                                }
                            }
                            sc->cnt13[lane] = st;
                        }
                        wave_fence();
                        // emission order: weight n-1 first; then the two final states (state 1, state 0), then the end mark
                        g8* dst = op + 1 + hd;
                        u32 done = 0, carryV = 0, carryN = 0;          // bits (< 8) waiting for the next round
                        for (u32 e0 = 0; e0 < n + 1; e0 += kWave) {
                            const u32 e = e0 + (u32)lane;
                            u64 v = 0; u32 nb = 0;
                            if (e < n) { const u32 i = n - 1 - e; const u32 x = sc->wbits[i & 1][i >> 1]; v = x >> 8; nb = x & 0xFF; }
                            else if (e == n) {
                                const u32 tl = (u32)t->tableLog;
                                v = (u64)(sc->cnt13[1] & ((1u << tl) - 1)) | ((u64)(sc->cnt13[0] & ((1u << tl) - 1)) << tl) | (1ull << (2 * tl));
                                nb = 2 * tl + 1;
                            }
                            if (lane == 0) { v = (v << carryN) | carryV; nb += carryN; }
                            const u32 tot = lane_get(wscan(nb, lane), 63);
                            const u32 by = wave_emit_bits(dst + done, v, nb, sc->fs.bits, 40, lane);
                            const bool lastRound = e0 + kWave >= n + 1;
                            if (lastRound) done += by;
                            else { done += tot >> 3; carryN = tot & 7; carryV = carryN ? ((u32)((const l8*)sc->fs.bits)[tot >> 3]) : 0u; }
                        }
                        const u32 cs = done;
                        if (cs) h = hd + cs;
                    }
                }
            }
        }
    }
    if (h > 1 && h < maxSym / 2) { if (lane == 0) op[0] = (u8)h; wave_fence(); return h + 1; }
    if (maxSym > 128) return 0;
    if (lane == 0) op[0] = (u8)(128 + (maxSym - 1));
    for (u32 k = (u32)lane; 2 * k < maxSym; k += kWave) op[k + 1] = (u8)((sc->weight[2 * k] << 4) + sc->weight[2 * k + 1]);
    wave_fence();
    return ((maxSym + 1) / 2) + 1;
}

// ------------------------------------------------------------------------------------------------
// wave-parallel bit packer: every lane contributes (value, nbits <= 128 via two 64-bit halves) per round,
// in lane order; the packer keeps a running bit cursor, ORs the pieces into a zeroed LDS tile and streams
// completed DWORDS to global memory (round 6; until then the tile left byte by byte, one store instruction
// per 64 bytes); up to 31 bits wait in the tile's first word for the next round.
// ------------------------------------------------------------------------------------------------
struct WavePacker {
    l32* tile;          // LDS, kTileWords dwords, zeroed
    g8* out;            // global destination of this bitstream
    u32 outPos;         // bytes already flushed (a multiple of 4 until close)
    u32 carryBits;      // 0..31 bits pending in tile[0]
    static constexpr u32 kTileWords = 64 * 4 + 2;   // 64 lanes x 128 bits + carry

    __device__ __forceinline__ void init(l32* ldsTile, g8* dst, int lane)
    {
        tile = ldsTile; out = dst; outPos = 0; carryBits = 0;
        for (u32 i = (u32)lane; i < kTileWords; i += kWave) tile[i] = 0;
        wave_fence();
    }
    // all 64 lanes call; lo/hi = this lane's bits (LSB first), nb = 0..128
    __device__ __forceinline__ void push(u64 lo, u64 hi, u32 nb, int lane)
    {
        const u32 incl = wave_scan_incl(nb);
        const u32 total = lane_get(incl, 63);
        u32 bit = carryBits + incl - nb;
        if (nb) {
            const u32 nlo = nb < 64 ? nb : 64;
            if (nlo < 64) lo &= (1ull << nlo) - 1;
            // place up to 64 bits at `bit`
            u32 w = bit >> 5; const u32 sh = bit & 31;
            const u64 a = lo << sh;
            const u32 a2 = sh ? (u32)(lo >> (64 - sh)) : 0;
            if ((u32)a) lds_or(&tile[w], (u32)a);
            if ((u32)(a >> 32)) lds_or(&tile[w + 1], (u32)(a >> 32));
            if (a2) lds_or(&tile[w + 2], a2);
            if (nb > 64) {
                const u32 nhi = nb - 64;
                if (nhi < 64) hi &= (1ull << nhi) - 1;
                bit += 64; w = bit >> 5;
                const u64 b = hi << sh;
                const u32 b2 = sh ? (u32)(hi >> (64 - sh)) : 0;
                if ((u32)b) lds_or(&tile[w], (u32)b);
                if ((u32)(b >> 32)) lds_or(&tile[w + 1], (u32)(b >> 32));
                if (b2) lds_or(&tile[w + 2], b2);
            }
        }
        wave_fence();
        const u32 endBit = carryBits + total;
        const u32 nwords = endBit >> 5;
        u32 carry = 0;
        for (u32 i = (u32)lane; i <= nwords; i += kWave) {                // (nwords <= 257: at most five trips)
            const u32 v = tile[i];
            if (i < nwords) st32(out + outPos + 4u * i, v); else carry = v;
            tile[i] = 0;
        }
        const u32 rem = endBit & 31;
        const int cl = (int)(nwords & 63u);                               // the lane that read the partial word
        carry = lane_get(carry, cl);
        wave_fence();
        if (lane == 0 && rem) tile[0] = carry;
        wave_fence();
        outPos += 4u * nwords; carryBits = rem;
    }
    // end mark + final partial word; returns the stream size in bytes
    __device__ __forceinline__ u32 close(int lane)
    {
        push(lane == 0 ? 1ull : 0ull, 0, lane == 0 ? 1u : 0u, lane);
        const u32 nb = (carryBits + 7) >> 3;                              // 0..4 bytes left in tile[0]
        if ((u32)lane < nb) out[outPos + lane] = ((const l8*)tile)[lane];
        outPos += nb;
        wave_fence();
        if (lane == 0) tile[0] = 0;
        carryBits = 0;
        wave_fence();
        return outPos;
    }
};

}  // namespace zs
}  // namespace aoclgpu
