// csrc/zstd_entropy.hip.h -- Zstandard entropy-stage building blocks for the gfx950 kernels.
//
// Serial table construction (a few hundred symbols; runs on lane 0 with its scratch in LDS) and the
// wave-parallel bit packers.  Each function restates the reference routine named next to it
// (/root/reference algos/zstd/lib/compress/{huf_compress.c,fse_compress.c,zstd_compress_sequences.c}); the
// table contents, header bytes and bitstreams are byte-identical to the reference's.
#pragma once
#include "common.hip.h"

namespace aoclgpu {
namespace zs {

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64; typedef int16_t s16;

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
// serial little bit writer (lane 0 only): headers, FSE-compressed Huffman weights
// ------------------------------------------------------------------------------------------------
struct BitW {
    u8* p; u64 acc; int nb; u32 pos;
    __device__ void init(u8* dst) { p = dst; acc = 0; nb = 0; pos = 0; }
    __device__ void add(u64 v, int n)
    {
        if (n == 0) return;
        v &= (1ull << n) - 1;
        acc |= v << nb; nb += n;
        while (nb >= 8) { p[pos++] = (u8)acc; acc >>= 8; nb -= 8; }
    }
    __device__ u32 close() { add(1, 1); if (nb > 0) { p[pos++] = (u8)acc; acc = 0; nb = 0; } return pos; }
};

// ------------------------------------------------------------------------------------------------
// FSE (fse_compress.c)
// ------------------------------------------------------------------------------------------------
struct FseCT {            // FSE_CTable as plain arrays (tableLog <= 9 on this path)
    int tableLog;
    u16 stateTable[512];
    int deltaNbBits[56];
    int deltaFindState[56];
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

// FSE_normalizeM2 (fse_compress.c:376-461): 0 ok, -1 error
__device__ int fse_normalize_m2(s16* norm, u32 tableLog, const u32* count, u32 total, u32 maxSym, s16 lowProb)
{
    const s16 NYA = -2;
    u32 s, distributed = 0, toDist;
    const u32 lowThreshold = total >> tableLog;
    u32 lowOne = (u32)(((u64)total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProb; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = NYA;
    }
    toDist = (1u << tableLog) - distributed;
    if (toDist == 0) return 0;
    if ((total / toDist) > lowOne) {
        lowOne = (u32)(((u64)total * 3) / (toDist * 2));
        for (s = 0; s <= maxSym; s++)
            if (norm[s] == NYA && count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; }
        toDist = (1u << tableLog) - distributed;
    }
    if (distributed == maxSym + 1) {
        u32 maxV = 0, maxC = 0;
        for (s = 0; s <= maxSym; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] += (s16)toDist;
        return 0;
    }
    if (total == 0) {
        for (s = 0; toDist > 0; s = (s + 1) % (maxSym + 1)) if (norm[s] > 0) { toDist--; norm[s]++; }
        return 0;
    }
    {   const u64 vStepLog = 62 - tableLog, mid = (1ull << (vStepLog - 1)) - 1;
        const u64 rStep = ((((u64)1 << vStepLog) * toDist) + mid) / total;
        u64 tmpTotal = mid;
        for (s = 0; s <= maxSym; s++) if (norm[s] == NYA) {
            const u64 end = tmpTotal + ((u64)count[s] * rStep);
            const u32 w = (u32)(end >> vStepLog) - (u32)(tmpTotal >> vStepLog);
            if (w < 1) return -1;
            norm[s] = (s16)w; tmpTotal = end;
        }
    }
    return 0;
}

// FSE_normalizeCount (fse_compress.c:464-526): tableLog on success, 0 = rle special case, -1 error
__device__ int fse_normalize(s16* norm, u32 tableLog, const u32* count, u32 total, u32 maxSym, int useLowProb)
{
    const u32 rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    if (tableLog < 5 || tableLog > 12) return -1;
    if (tableLog < fse_min_log(total, maxSym)) return -1;
    const s16 lowProb = useLowProb ? -1 : 1;
    const u64 scale = 62 - tableLog, step = ((u64)1 << 62) / total, vStep = 1ull << (scale - 20);
    int still = 1 << tableLog;
    u32 s, largest = 0; s16 largestP = 0;
    const u32 lowThreshold = total >> tableLog;
    for (s = 0; s <= maxSym; s++) {
        if (count[s] == total) return 0;
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProb; still--; }
        else {
            s16 proba = (s16)(((u64)count[s] * step) >> scale);
            if (proba < 8) {
                const u64 rest = vStep * rtb[proba];
                proba += ((u64)count[s] * step) - ((u64)proba << scale) > rest;
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba; still -= proba;
        }
    }
    if (-still >= (norm[largest] >> 1)) {
        if (fse_normalize_m2(norm, tableLog, count, total, maxSym, lowProb)) return -1;
    } else norm[largest] += (s16)still;
    return (int)tableLog;
}

// FSE_writeNCount (fse_compress.c:236-329): bytes written, 0 on error
__device__ u32 fse_write_ncount(u8* out, const s16* norm, u32 maxSym, u32 tableLog)
{
    u8* const ostart = out;
    const int tableSize = 1 << tableLog;
    int nbBits = (int)tableLog + 1, remaining = tableSize + 1, threshold = tableSize, bitCount = 0, previousIs0 = 0;
    u32 bitStream = 0, symbol = 0;
    const u32 alphabetSize = maxSym + 1;
    bitStream += (tableLog - 5) << bitCount; bitCount += 4;
    while (symbol < alphabetSize && remaining > 1) {
        if (previousIs0) {
            u32 start = symbol;
            while (symbol < alphabetSize && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) {
                start += 24; bitStream += 0xFFFFu << bitCount;
                out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16;
            }
            while (symbol >= start + 3) { start += 3; bitStream += 3u << bitCount; bitCount += 2; }
            bitStream += (symbol - start) << bitCount; bitCount += 2;
            if (bitCount > 16) { out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
        }
        {   int count = norm[symbol++];
            const int max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (u32)count << bitCount;
            bitCount += nbBits; bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1) return 0;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) { out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
    }
    if (remaining != 1) return 0;
    out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (u32)(out - ostart);
}

// FSE_buildCTable_wksp (fse_compress.c:68-214).  `scratch` holds tableSize bytes + (maxSym+2) u16.
__device__ void fse_build_ctable(FseCT* ct, const s16* norm, u32 maxSym, u32 tableLog, u8* tableSymbol, u16* cumul)
{
    const u32 tableSize = 1u << tableLog, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    u32 high = tableSize - 1, u, s, pos = 0;
    ct->tableLog = (int)tableLog;
    cumul[0] = 0;
    for (u = 1; u <= maxSym + 1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tableSymbol[high--] = (u8)(u - 1); }
        else cumul[u] = cumul[u - 1] + (u16)norm[u - 1];
    }
    cumul[maxSym + 1] = (u16)(tableSize + 1);
    for (s = 0; s <= maxSym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            tableSymbol[pos] = (u8)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (u = 0; u < tableSize; u++) { const u8 sym = tableSymbol[u]; ct->stateTable[cumul[sym]++] = (u16)(tableSize + u); }
    u32 total = 0;
    for (s = 0; s <= maxSym; s++) {
        if (norm[s] == 0) { ct->deltaNbBits[s] = (int)(((tableLog + 1) << 16) - (1u << tableLog)); ct->deltaFindState[s] = 0; }
        else if (norm[s] == -1 || norm[s] == 1) {
            ct->deltaNbBits[s] = (int)((tableLog << 16) - (1u << tableLog));
            ct->deltaFindState[s] = (int)(total - 1); total++;
        } else {
            const u32 maxBitsOut = tableLog - (u32)hb((u32)norm[s] - 1), minStatePlus = (u32)norm[s] << maxBitsOut;
            ct->deltaNbBits[s] = (int)((maxBitsOut << 16) - minStatePlus);
            ct->deltaFindState[s] = (int)(total - (u32)norm[s]); total += (u32)norm[s];
        }
    }
}
__device__ __forceinline__ void fse_build_rle(FseCT* ct, u32 sym)
{   // fse_compress.c:530-548
    ct->tableLog = 0; ct->stateTable[0] = 0; ct->stateTable[1] = 0;
    ct->deltaNbBits[sym] = 0; ct->deltaFindState[sym] = 0;
}
__device__ __forceinline__ u32 fse_init_state(const FseCT* ct, u32 sym)
{   // FSE_initCState2 (fse.h:451-463)
    const u32 nb = (u32)(ct->deltaNbBits[sym] + (1 << 15)) >> 16;
    const u32 v = (nb << 16) - (u32)ct->deltaNbBits[sym];
    return ct->stateTable[(int)(v >> nb) + ct->deltaFindState[sym]];
}

// ------------------------------------------------------------------------------------------------
// Huffman (huf_compress.c)
// ------------------------------------------------------------------------------------------------
struct HNode { u32 count; u16 parent; u8 byte; u8 nbBits; };
struct HufCT { u8 nbBits[256]; u16 val[256]; u32 tableLog; };

constexpr u32 kHufCutoff = 165;      // RANK_POSITION_DISTINCT_COUNT_CUTOFF = 158 + highbit32(158)
__device__ __forceinline__ u32 huf_rank_index(u32 c) { return c < kHufCutoff ? c : (u32)hb(c) + 158; }

__device__ void huf_insertion_sort(HNode* a, int low, int high)
{
    const int size = high - low + 1; a += low;
    for (int i = 1; i < size; i++) {
        const HNode key = a[i]; int j = i - 1;
        while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; }
        a[j + 1] = key;
    }
}
__device__ int huf_partition(HNode* a, int low, int high)
{
    const u32 pivot = a[high].count; int i = low - 1;
    for (int j = low; j < high; j++) if (a[j].count > pivot) { i++; const HNode t = a[i]; a[i] = a[j]; a[j] = t; }
    { const HNode t = a[i + 1]; a[i + 1] = a[high]; a[high] = t; }
    return i + 1;
}
// HUF_simpleQuickSort (huf_compress.c:603-622) with an explicit stack: every recursive call of the reference is
// one stack frame here; frames cover disjoint ranges, so the processing order does not change the result.
__device__ void huf_quicksort(HNode* a, int low0, int high0, short* stack /* >= 128 shorts */)
{
    int sp = 0;
    stack[sp++] = (short)low0; stack[sp++] = (short)high0;
    while (sp > 0) {
        int high = stack[--sp], low = stack[--sp];
        if (high - low < 8) { huf_insertion_sort(a, low, high); continue; }
        while (low < high) {
            const int idx = huf_partition(a, low, high);
            int cl, ch;                                  // the side the reference recurses into
            if (idx - low < high - idx) { cl = low; ch = idx - 1; low = idx + 1; }
            else { cl = idx + 1; ch = high; high = idx - 1; }
            if (ch - cl < 8) { if (ch > cl) huf_insertion_sort(a, cl, ch); }     // small call: sort it now, no frame
            else { stack[sp++] = (short)cl; stack[sp++] = (short)ch; }           // <= n/8 pending frames
        }
    }
}
struct RankPos { u16 base, curr; };
// HUF_sort (huf_compress.c:624-668)
__device__ void huf_sort(HNode* node, const u32* count, u32 maxSym, RankPos* rp /* 192 */, short* stack)
{
    u32 n; const u32 n1 = maxSym + 1;
    for (n = 0; n < 192; n++) { rp[n].base = 0; rp[n].curr = 0; }
    for (n = 0; n < n1; n++) rp[huf_rank_index(count[n])].base++;
    for (n = 191; n > 0; n--) { rp[n - 1].base += rp[n].base; rp[n - 1].curr = rp[n - 1].base; }
    for (n = 0; n < n1; n++) {
        const u32 c = count[n], r = huf_rank_index(c) + 1, pos = rp[r].curr++;
        node[pos].count = c; node[pos].byte = (u8)n;
    }
    for (n = kHufCutoff; n < 191; n++) {
        const int size = rp[n].curr - rp[n].base;
        if (size > 1) huf_quicksort(node + rp[n].base, 0, size - 1, stack);
    }
}
// HUF_buildTree (huf_compress.c:685-725); node[-1] must be addressable
__device__ int huf_build_tree(HNode* node, u32 maxSym)
{
    const int START = 256;
    int nonNull = (int)maxSym, lowS, lowN, nodeNb = START, n, root;
    while (node[nonNull].count == 0) nonNull--;
    lowS = nonNull; root = nodeNb + lowS - 1; lowN = nodeNb;
    node[nodeNb].count = node[lowS].count + node[lowS - 1].count;
    node[lowS].parent = node[lowS - 1].parent = (u16)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= root; n++) node[n].count = 1u << 30;
    node[-1].count = 1u << 31;
    while (nodeNb <= root) {
        const int n1 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        const int n2 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        node[nodeNb].count = node[n1].count + node[n2].count;
        node[n1].parent = node[n2].parent = (u16)nodeNb;
        nodeNb++;
    }
    node[root].nbBits = 0;
    for (n = root - 1; n >= START; n--) node[n].nbBits = (u8)(node[node[n].parent].nbBits + 1);
    for (n = 0; n <= nonNull; n++) node[n].nbBits = (u8)(node[node[n].parent].nbBits + 1);
    return nonNull;
}
// HUF_setMaxHeight (huf_compress.c:378-495)
__device__ u32 huf_set_max_height(HNode* node, u32 lastNonNull, u32 target)
{
    const u32 largest = node[lastNonNull].nbBits;
    if (largest <= target) return largest;
    int totalCost = 0, n = (int)lastNonNull;
    const u32 baseCost = 1u << (largest - target);
    while (node[n].nbBits > target) {
        totalCost += (int)(baseCost - (1u << (largest - node[n].nbBits)));
        node[n].nbBits = (u8)target; n--;
    }
    while (node[n].nbBits == target) --n;
    totalCost >>= (largest - target);
    const u32 noSym = 0xF0F0F0F0u;
    u32 rankLast[14]; u32 cur = target;
    for (int p = 0; p < 14; p++) rankLast[p] = noSym;
    for (int pos = n; pos >= 0; pos--) {
        if (node[pos].nbBits >= cur) continue;
        cur = node[pos].nbBits; rankLast[target - cur] = (u32)pos;
    }
    while (totalCost > 0) {
        u32 dec = (u32)hb((u32)totalCost) + 1;
        for (; dec > 1; dec--) {
            const u32 highPos = rankLast[dec], lowPos = rankLast[dec - 1];
            if (highPos == noSym) continue;
            if (lowPos == noSym) break;
            if (node[highPos].count <= 2 * node[lowPos].count) break;
        }
        while (dec <= 12 && rankLast[dec] == noSym) dec++;
        totalCost -= 1 << (dec - 1);
        node[rankLast[dec]].nbBits++;
        if (rankLast[dec - 1] == noSym) rankLast[dec - 1] = rankLast[dec];
        if (rankLast[dec] == 0) rankLast[dec] = noSym;
        else { rankLast[dec]--; if (node[rankLast[dec]].nbBits != target - dec) rankLast[dec] = noSym; }
    }
    while (totalCost < 0) {
        if (rankLast[1] == noSym) {
            while (node[n].nbBits == target) n--;
            node[n + 1].nbBits--; rankLast[1] = (u32)(n + 1); totalCost++;
            continue;
        }
        node[rankLast[1] + 1].nbBits--; rankLast[1]++; totalCost++;
    }
    return target;
}
// HUF_buildCTable_wksp (huf_compress.c:758-794); tbl = 513 nodes (tbl[0] is the barrier entry)
__device__ u32 huf_build(HufCT* ct, const u32* count, u32 maxSym, u32 maxNbBits, HNode* tbl, RankPos* rp, short* stack)
{
    HNode* const node = tbl + 1;
    u16 nbPerRank[13], valPerRank[13];
    for (int i = 0; i < 513; i++) { tbl[i].count = 0; tbl[i].parent = 0; tbl[i].byte = 0; tbl[i].nbBits = 0; }
    for (int i = 0; i < 13; i++) { nbPerRank[i] = 0; valPerRank[i] = 0; }
    huf_sort(node, count, maxSym, rp, stack);
    const int nonNull = huf_build_tree(node, maxSym);
    const u32 bits = huf_set_max_height(node, (u32)nonNull, maxNbBits);
    for (int n = 0; n <= nonNull; n++) nbPerRank[node[n].nbBits]++;
    {   u16 min = 0;
        for (int n = (int)bits; n > 0; n--) { valPerRank[n] = min; min = (u16)(min + nbPerRank[n]); min >>= 1; }
    }
    for (int n = 0; n < 256; n++) { ct->nbBits[n] = 0; ct->val[n] = 0; }
    for (int n = 0; n <= (int)maxSym; n++) ct->nbBits[node[n].byte] = node[n].nbBits;
    for (int n = 0; n <= (int)maxSym; n++) ct->val[n] = ct->nbBits[n] ? valPerRank[ct->nbBits[n]]++ : 0;
    ct->tableLog = bits;
    return bits;
}

// FSE_compress_usingCTable on the Huffman weights (fse_compress.c:558-615): two interleaved states
__device__ u32 fse_compress_weights(u8* dst, const u8* w, u32 n, const FseCT* ct)
{
    BitW b; u32 st[2] = {0, 0}; int init[2] = {0, 0};
    if (n <= 2) return 0;
    b.init(dst);
    for (u32 i = n; i-- > 0;) {
        const int k = (int)(i & 1);
        const u32 sym = w[i];
        if (!init[k]) { st[k] = fse_init_state(ct, sym); init[k] = 1; }
        else {
            const u32 nb = (st[k] + (u32)ct->deltaNbBits[sym]) >> 16;
            b.add(st[k], (int)nb);
            st[k] = ct->stateTable[(int)(st[k] >> nb) + ct->deltaFindState[sym]];
        }
    }
    b.add(st[1], ct->tableLog);
    b.add(st[0], ct->tableLog);
    return b.close();
}

// scratch needed by the literal-table writer
struct HufWriteScratch { u8 weight[256]; u32 count[13]; s16 norm[13]; FseCT ct; u8 tableSymbol[64]; u16 cumul[16]; };

// HUF_compressWeights (huf_compress.c:173-213)
__device__ u32 huf_compress_weights(u8* dst, const u8* w, u32 n, HufWriteScratch* sc)
{
    u32 maxSym = 0, maxCount = 0, i;
    if (n <= 1) return 0;
    for (i = 0; i < 13; i++) sc->count[i] = 0;
    for (i = 0; i < n; i++) sc->count[w[i]]++;
    for (i = 0; i <= 12; i++) { if (sc->count[i]) maxSym = i; if (sc->count[i] > maxCount) maxCount = sc->count[i]; }
    if (maxCount == n) return 1;
    if (maxCount == 1) return 0;
    const u32 tableLog = fse_optimal_log(6, n, maxSym, 2);
    if (fse_normalize(sc->norm, tableLog, sc->count, n, maxSym, 0) < 0) return 0;
    const u32 h = fse_write_ncount(dst, sc->norm, maxSym, tableLog);
    if (!h) return 0;
    fse_build_ctable(&sc->ct, sc->norm, maxSym, tableLog, sc->tableSymbol, sc->cumul);
    const u32 c = fse_compress_weights(dst + h, w, n, &sc->ct);
    if (!c) return 0;
    return h + c;
}
// HUF_writeCTable_wksp (huf_compress.c:256-295); 0 on failure
__device__ u32 huf_write_ctable(u8* op, const HufCT* ct, u32 maxSym, u32 huffLog, HufWriteScratch* sc)
{
    u32 n;
    for (n = 0; n < maxSym; n++) sc->weight[n] = ct->nbBits[n] ? (u8)(huffLog + 1 - ct->nbBits[n]) : 0;
    {   const u32 h = huf_compress_weights(op + 1, sc->weight, maxSym, sc);
        if (h > 1 && h < maxSym / 2) { op[0] = (u8)h; return h + 1; }
    }
    if (maxSym > 128) return 0;
    op[0] = (u8)(128 + (maxSym - 1));
    sc->weight[maxSym] = 0;
    for (n = 0; n < maxSym; n += 2) op[(n / 2) + 1] = (u8)((sc->weight[n] << 4) + sc->weight[n + 1]);
    return ((maxSym + 1) / 2) + 1;
}

// ------------------------------------------------------------------------------------------------
// wave-parallel bit packer: every lane contributes (value, nbits <= 128 via two 64-bit halves) per round,
// in lane order; the packer keeps a running bit cursor, ORs the pieces into a zeroed LDS tile and streams
// completed bytes to global memory.
// ------------------------------------------------------------------------------------------------
struct WavePacker {
    u32* tile;          // LDS, kTileWords dwords, zeroed
    u8* out;            // global destination of this bitstream
    u32 outPos;         // bytes already flushed
    u32 carryBits;      // 0..7 bits pending in tile[0]'s low byte
    static constexpr u32 kTileWords = 64 * 4 + 2;   // 64 lanes x 128 bits + carry

    __device__ void init(u32* ldsTile, u8* dst, int lane)
    {
        tile = ldsTile; out = dst; outPos = 0; carryBits = 0;
        for (u32 i = (u32)lane; i < kTileWords; i += kWave) tile[i] = 0;
        wave_fence();
    }
    // all lanes call; lo/hi = this lane's bits (LSB first), nb = 0..128
    __device__ void push(u64 lo, u64 hi, u32 nb, int lane)
    {
        // exclusive prefix sum of nb across lanes
        u32 incl = nb;
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        const u32 total = lane_get(incl, 63);
        u32 bit = carryBits + incl - nb;
        if (nb) {
            const u32 nlo = nb < 64 ? nb : 64;
            if (nlo < 64) lo &= (1ull << nlo) - 1;
            // place up to 64 bits at `bit`
            u32 w = bit >> 5; const u32 sh = bit & 31;
            const u64 a = lo << sh;
            const u32 a2 = sh ? (u32)(lo >> (64 - sh)) : 0;
            if ((u32)a) atomicOr(&tile[w], (u32)a);
            if ((u32)(a >> 32)) atomicOr(&tile[w + 1], (u32)(a >> 32));
            if (a2) atomicOr(&tile[w + 2], a2);
            if (nb > 64) {
                const u32 nhi = nb - 64;
                if (nhi < 64) hi &= (1ull << nhi) - 1;
                bit += 64; w = bit >> 5;
                const u64 b = hi << sh;
                const u32 b2 = sh ? (u32)(hi >> (64 - sh)) : 0;
                if ((u32)b) atomicOr(&tile[w], (u32)b);
                if ((u32)(b >> 32)) atomicOr(&tile[w + 1], (u32)(b >> 32));
                if (b2) atomicOr(&tile[w + 2], b2);
            }
        }
        wave_fence();
        const u32 endBit = carryBits + total;
        const u32 nbytes = endBit >> 3;
        const u8* tb = reinterpret_cast<const u8*>(tile);
        for (u32 i = (u32)lane; i < nbytes; i += kWave) out[outPos + i] = tb[i];
        wave_fence();
        const u32 rem = endBit & 7;
        const u32 carry = rem ? (u32)tb[nbytes] : 0;
        wave_fence();
        const u32 usedWords = (endBit + 31) / 32 + 1;
        for (u32 i = (u32)lane; i < usedWords && i < kTileWords; i += kWave) tile[i] = 0;
        wave_fence();
        if (lane == 0 && rem) tile[0] = carry;
        wave_fence();
        outPos += nbytes; carryBits = rem;
    }
    // end mark + final partial byte; returns the stream size in bytes
    __device__ u32 close(int lane)
    {
        push(lane == 0 ? 1ull : 0ull, 0, lane == 0 ? 1u : 0u, lane);
        if (carryBits) { if (lane == 0) out[outPos] = (u8)tile[0]; outPos++; }
        wave_fence();
        if (lane == 0) tile[0] = 0;
        carryBits = 0;
        wave_fence();
        return outPos;
    }
};

}  // namespace zs
}  // namespace aoclgpu
