/* oracle/zstd_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's Zstandard level-1 one-shot path for inputs of at most 128 KiB
 * (= one block per frame, the GPU engine's chunk contract):
 *
 *   ZSTD_compress(dst,cap,src,n,1) (algos/zstd/lib/compress/zstd_compress.c:5534-5560)
 *     parameters   ZSTD_getCParams_internal :7295-7331 + clevels.h:76-108 + ZSTD_adjustCParams_internal :1539-1633
 *     frame        ZSTD_writeFrameHeader :4586-4632, ZSTD_compress_frameChunk :4504-4583, ZSTD_writeEpilogue
 *     block        ZSTD_compressBlock_internal :4333-4395, ZSTD_buildSeqStore :3259-3418
 *     match finder AOCL_ZSTD_compressBlock_fast_noDict_generic  zstd_fast.c:427-706  (mls 5 or 6, step 2)
 *     entropy      ZSTD_entropyCompressSeqStore(_internal) zstd_compress.c:2910-3064
 *                  literals  zstd_compress_literals.c:39-235, huf_compress.c (sort :624, tree :685,
 *                            setMaxHeight :378, CTable :735, writeCTable :256-295, weights :173-213,
 *                            1X/4X streams :1051-1211, HUF_compress_internal :1326-1434), hist.c
 *                  sequences zstd_compress.c:2715-2902, zstd_compress_sequences.c:156-382,
 *                            fse_compress.c (optimalTableLog :370, normalizeCount :464, writeNCount :236-329,
 *                            buildCTable :68-214, compress_usingCTable :558-615), fse.h:451-480, bitstream.h
 *   ZSTD_decompress (algos/zstd/lib/decompress/zstd_decompress.c:1252, zstd_decompress_block.c) -- restated from
 *     the frame format (blocks raw/RLE/compressed, Huffman literals, FSE sequences, repeat offsets).
 *
 * The output of orc_zstd_compress is byte-identical to the reference's ZSTD_compress(level 1) for every
 * n <= 128 KiB (pinned by tests/test_oracle_vs_ref.py).  Destination capacity must be >= ZSTD_compressBound(n).
 */
#include "oracle.h"
#include <string.h>
#include <stdlib.h>

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64;
typedef int16_t s16;

static u32 z_rd32(const u8* p) { u32 v; memcpy(&v, p, 4); return v; }
static u64 z_rd64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
static int z_hb(u32 v) { int r = 0; while (v >>= 1) r++; return r; }   /* ZSTD_highbit32 (v != 0) */

size_t orc_zstd_compress_bound(size_t n)
{   /* zstd.h:473 ZSTD_COMPRESSBOUND */
    return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
}

/* ------------------------------------------------------------------------------------------
 * bit writer: LSB-first, forward growing (bitstream.h:153-240); close adds the 1-bit end mark
 * ------------------------------------------------------------------------------------------ */
typedef struct { u8* p; u64 acc; int nb; size_t pos; } bw_t;
static void bw_init(bw_t* b, u8* dst) { b->p = dst; b->acc = 0; b->nb = 0; b->pos = 0; }
static void bw_add(bw_t* b, u64 v, int n)
{
    if (n == 0) return;
    v &= (n >= 64) ? ~0ull : ((1ull << n) - 1);
    b->acc |= v << b->nb; b->nb += n;
    while (b->nb >= 8) { b->p[b->pos++] = (u8)b->acc; b->acc >>= 8; b->nb -= 8; }
}
static size_t bw_close(bw_t* b)
{
    bw_add(b, 1, 1);
    if (b->nb > 0) { b->p[b->pos++] = (u8)b->acc; b->acc = 0; b->nb = 0; }
    return b->pos;
}

/* ------------------------------------------------------------------------------------------
 * FSE compression tables (fse_compress.c)
 * ------------------------------------------------------------------------------------------ */
enum { FSE_MINLOG = 5, FSE_MAXLOG = 12, FSE_MAXSYM = 255 };
typedef struct {
    int tableLog;
    u16 stateTable[1 << 9];           /* big enough for every table this path builds (log <= 9) */
    int deltaNbBits[64]; int deltaFindState[64];
} fse_ct;

static u32 fse_min_log(size_t srcSize, u32 maxSym)
{   /* fse_compress.c:346-354 */
    u32 a = (u32)z_hb((u32)srcSize) + 1, b = (u32)z_hb(maxSym) + 2;
    return a < b ? a : b;
}
static u32 fse_optimal_log(u32 maxLog, size_t srcSize, u32 maxSym, u32 minus)
{   /* fse_compress.c:356-373 */
    u32 maxBitsSrc = (u32)z_hb((u32)(srcSize - 1)) - minus;
    u32 t = maxLog, minBits = fse_min_log(srcSize, maxSym);
    if (t == 0) t = 11;
    if (maxBitsSrc < t) t = maxBitsSrc;
    if (minBits > t) t = minBits;
    if (t < FSE_MINLOG) t = FSE_MINLOG;
    if (t > FSE_MAXLOG) t = FSE_MAXLOG;
    return t;
}

/* returns 0 on success, -1 on failure (fse_compress.c:376-461) */
static int fse_normalize_m2(s16* norm, u32 tableLog, const u32* count, size_t total, u32 maxSym, s16 lowProb)
{
    const s16 NYA = -2;
    u32 s, distributed = 0, toDist;
    const u32 lowThreshold = (u32)(total >> tableLog);
    u32 lowOne = (u32)((total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProb; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = NYA;
    }
    toDist = (1u << tableLog) - distributed;
    if (toDist == 0) return 0;
    if ((total / toDist) > lowOne) {
        lowOne = (u32)((total * 3) / (toDist * 2));
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
        const u64 rStep = ((((u64)1 << vStepLog) * toDist) + mid) / (u32)total;
        u64 tmpTotal = mid;
        for (s = 0; s <= maxSym; s++) if (norm[s] == NYA) {
            const u64 end = tmpTotal + (count[s] * rStep);
            const u32 w = (u32)(end >> vStepLog) - (u32)(tmpTotal >> vStepLog);
            if (w < 1) return -1;
            norm[s] = (s16)w; tmpTotal = end;
        }
    }
    return 0;
}

/* fse_compress.c:464-526.  returns tableLog, 0 for the rle special case, -1 on error */
static int fse_normalize(s16* norm, u32 tableLog, const u32* count, size_t total, u32 maxSym, int useLowProb)
{
    static const u32 rtb[] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    if (tableLog == 0) tableLog = 11;
    if (tableLog < FSE_MINLOG || tableLog > FSE_MAXLOG) return -1;
    if (tableLog < fse_min_log(total, maxSym)) return -1;
    {   const s16 lowProb = useLowProb ? -1 : 1;
        const u64 scale = 62 - tableLog, step = ((u64)1 << 62) / (u32)total, vStep = 1ull << (scale - 20);
        int still = 1 << tableLog;
        u32 s, largest = 0; s16 largestP = 0;
        const u32 lowThreshold = (u32)(total >> tableLog);
        for (s = 0; s <= maxSym; s++) {
            if (count[s] == total) return 0;
            if (count[s] == 0) { norm[s] = 0; continue; }
            if (count[s] <= lowThreshold) { norm[s] = lowProb; still--; }
            else {
                s16 proba = (s16)((count[s] * step) >> scale);
                if (proba < 8) {
                    const u64 rest = vStep * rtb[proba];
                    proba += (count[s] * step) - ((u64)proba << scale) > rest;
                }
                if (proba > largestP) { largestP = proba; largest = s; }
                norm[s] = proba; still -= proba;
            }
        }
        if (-still >= (norm[largest] >> 1)) {
            if (fse_normalize_m2(norm, tableLog, count, total, maxSym, lowProb)) return -1;
        } else norm[largest] += (s16)still;
    }
    return (int)tableLog;
}

/* fse_compress.c:236-329; returns bytes written (the reference also stores one spill byte past it) */
static size_t fse_write_ncount(u8* out, const s16* norm, u32 maxSym, u32 tableLog)
{
    u8* const ostart = out;
    const int tableSize = 1 << tableLog;
    int nbBits = (int)tableLog + 1, remaining = tableSize + 1, threshold = tableSize, bitCount = 0, previousIs0 = 0;
    u32 bitStream = 0, symbol = 0;
    const u32 alphabetSize = maxSym + 1;
    bitStream += (tableLog - FSE_MINLOG) << bitCount; bitCount += 4;
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
    return (size_t)(out - ostart);
}

/* fse_compress.c:68-214 */
static void fse_build_ctable(fse_ct* ct, const s16* norm, u32 maxSym, u32 tableLog)
{
    const u32 tableSize = 1u << tableLog, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    u16 cumul[FSE_MAXSYM + 2];
    u8 tableSymbol[1 << 9];
    u32 high = tableSize - 1, u, s, pos = 0;
    ct->tableLog = (int)tableLog;
    cumul[0] = 0;
    for (u = 1; u <= maxSym + 1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tableSymbol[high--] = (u8)(u - 1); }
        else cumul[u] = cumul[u - 1] + (u16)norm[u - 1];
    }
    cumul[maxSym + 1] = (u16)(tableSize + 1);
    /* spread: both reference code paths (with / without low-probability symbols) visit the cells in this order */
    for (s = 0; s <= maxSym; s++) {
        int i;
        for (i = 0; i < norm[s]; i++) {
            tableSymbol[pos] = (u8)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (u = 0; u < tableSize; u++) { const u8 sym = tableSymbol[u]; ct->stateTable[cumul[sym]++] = (u16)(tableSize + u); }
    {   u32 total = 0;
        for (s = 0; s <= maxSym; s++) {
            if (norm[s] == 0) { ct->deltaNbBits[s] = (int)(((tableLog + 1) << 16) - (1u << tableLog)); ct->deltaFindState[s] = 0; }
            else if (norm[s] == -1 || norm[s] == 1) {
                ct->deltaNbBits[s] = (int)((tableLog << 16) - (1u << tableLog));
                ct->deltaFindState[s] = (int)(total - 1); total++;
            } else {
                const u32 maxBitsOut = tableLog - (u32)z_hb((u32)norm[s] - 1), minStatePlus = (u32)norm[s] << maxBitsOut;
                ct->deltaNbBits[s] = (int)((maxBitsOut << 16) - minStatePlus);
                ct->deltaFindState[s] = (int)(total - (u32)norm[s]); total += (u32)norm[s];
            }
        }
    }
}
static void fse_build_rle(fse_ct* ct, u32 sym)
{   /* fse_compress.c:530-548 */
    ct->tableLog = 0; ct->stateTable[0] = 0; ct->stateTable[1] = 0;
    ct->deltaNbBits[sym] = 0; ct->deltaFindState[sym] = 0;
}
static u32 fse_init_state(const fse_ct* ct, u32 sym)
{   /* fse.h:451-463 FSE_initCState2 */
    const u32 nb = (u32)(ct->deltaNbBits[sym] + (1 << 15)) >> 16;
    const u32 v = (nb << 16) - (u32)ct->deltaNbBits[sym];
    return ct->stateTable[(int)(v >> nb) + ct->deltaFindState[sym]];
}
static u32 fse_encode(bw_t* b, const fse_ct* ct, u32 state, u32 sym)
{   /* fse.h:465-472 */
    const u32 nb = (state + (u32)ct->deltaNbBits[sym]) >> 16;
    bw_add(b, state, (int)nb);
    return ct->stateTable[(int)(state >> nb) + ct->deltaFindState[sym]];
}

/* ------------------------------------------------------------------------------------------
 * Huffman (huf_compress.c)
 * ------------------------------------------------------------------------------------------ */
typedef struct { u32 count; u16 parent; u8 byte; u8 nbBits; } hnode;
typedef struct { u8 nbBits[256]; u16 val[256]; u32 tableLog; } huf_ct;

/* huf_compress.c:552-563.  RANK_POSITION_DISTINCT_COUNT_CUTOFF = 158 + highbit32(158) = 165 (the reference's
 * own comment says 166): counts < 165 get one bucket each, larger counts share log2 buckets. */
enum { HUF_CUTOFF = 165 };
static u32 huf_rank_index(u32 c) { return c < HUF_CUTOFF ? c : (u32)z_hb(c) + 158; }

static void huf_insertion_sort(hnode* a, int low, int high)
{
    int i; const int size = high - low + 1; a += low;
    for (i = 1; i < size; i++) {
        const hnode key = a[i]; int j = i - 1;
        while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; }
        a[j + 1] = key;
    }
}
static int huf_partition(hnode* a, int low, int high)
{
    const u32 pivot = a[high].count; int i = low - 1, j;
    for (j = low; j < high; j++) if (a[j].count > pivot) { hnode t; i++; t = a[i]; a[i] = a[j]; a[j] = t; }
    { hnode t = a[i + 1]; a[i + 1] = a[high]; a[high] = t; }
    return i + 1;
}
static void huf_quicksort(hnode* a, int low, int high)
{   /* huf_compress.c:603-622 */
    if (high - low < 8) { huf_insertion_sort(a, low, high); return; }
    while (low < high) {
        const int idx = huf_partition(a, low, high);
        if (idx - low < high - idx) { huf_quicksort(a, low, idx - 1); low = idx + 1; }
        else { huf_quicksort(a, idx + 1, high); high = idx - 1; }
    }
}
static void huf_sort(hnode* node, const u32* count, u32 maxSym)
{   /* huf_compress.c:624-668 */
    struct { u16 base, curr; } rp[192];
    u32 n; const u32 n1 = maxSym + 1;
    memset(rp, 0, sizeof(rp));
    for (n = 0; n < n1; n++) rp[huf_rank_index(count[n])].base++;
    for (n = 191; n > 0; n--) { rp[n - 1].base += rp[n].base; rp[n - 1].curr = rp[n - 1].base; }
    for (n = 0; n < n1; n++) {
        const u32 c = count[n], r = huf_rank_index(c) + 1, pos = rp[r].curr++;
        node[pos].count = c; node[pos].byte = (u8)n;
    }
    for (n = HUF_CUTOFF; n < 191; n++) {
        const int size = rp[n].curr - rp[n].base;
        if (size > 1) huf_quicksort(node + rp[n].base, 0, size - 1);
    }
}
static int huf_build_tree(hnode* node /* node[-1] valid */, u32 maxSym)
{   /* huf_compress.c:685-725 */
    enum { START = 256 };
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
static u32 huf_set_max_height(hnode* node, u32 lastNonNull, u32 target)
{   /* huf_compress.c:378-495 */
    const u32 largest = node[lastNonNull].nbBits;
    if (largest <= target) return largest;
    {   int totalCost = 0, n = (int)lastNonNull;
        const u32 baseCost = 1u << (largest - target);
        while (node[n].nbBits > target) {
            totalCost += (int)(baseCost - (1u << (largest - node[n].nbBits)));
            node[n].nbBits = (u8)target; n--;
        }
        while (node[n].nbBits == target) --n;
        totalCost >>= (largest - target);
        {   const u32 noSym = 0xF0F0F0F0u;
            u32 rankLast[14]; int pos; u32 cur = target;
            for (pos = 0; pos < 14; pos++) rankLast[pos] = noSym;
            for (pos = n; pos >= 0; pos--) {
                if (node[pos].nbBits >= cur) continue;
                cur = node[pos].nbBits; rankLast[target - cur] = (u32)pos;
            }
            while (totalCost > 0) {
                u32 dec = (u32)z_hb((u32)totalCost) + 1;
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
        }
    }
    return target;
}
/* HUF_buildCTable_wksp (huf_compress.c:758-794); returns maxNbBits */
static u32 huf_build(huf_ct* ct, const u32* count, u32 maxSym, u32 maxNbBits)
{
    hnode tbl[513]; hnode* const node = tbl + 1;
    int nonNull, n; u32 bits;
    u16 nbPerRank[13] = {0}, valPerRank[13] = {0};
    memset(tbl, 0, sizeof(tbl));
    huf_sort(node, count, maxSym);
    nonNull = huf_build_tree(node, maxSym);
    bits = huf_set_max_height(node, (u32)nonNull, maxNbBits);
    for (n = 0; n <= nonNull; n++) nbPerRank[node[n].nbBits]++;
    {   u16 min = 0;
        for (n = (int)bits; n > 0; n--) { valPerRank[n] = min; min = (u16)(min + nbPerRank[n]); min >>= 1; }
    }
    memset(ct->nbBits, 0, sizeof(ct->nbBits)); memset(ct->val, 0, sizeof(ct->val));
    for (n = 0; n <= (int)maxSym; n++) ct->nbBits[node[n].byte] = node[n].nbBits;
    for (n = 0; n <= (int)maxSym; n++) ct->val[n] = ct->nbBits[n] ? valPerRank[ct->nbBits[n]]++ : 0;
    ct->tableLog = bits;
    return bits;
}

/* FSE_compress_usingCTable for the Huffman weights (fse_compress.c:558-615): two interleaved states,
 * symbol i handled by state (i even ? 1 : 2), processed from the last symbol to the first */
static size_t fse_compress_weights(u8* dst, const u8* w, size_t n, const fse_ct* ct)
{
    bw_t b; u32 st[2]; int init[2] = {0, 0}; size_t i;
    if (n <= 2) return 0;
    bw_init(&b, dst);
    for (i = n; i-- > 0;) {
        const int k = (int)(i & 1);      /* 0 -> CState1, 1 -> CState2 */
        if (!init[k]) { st[k] = fse_init_state(ct, w[i]); init[k] = 1; }
        else st[k] = fse_encode(&b, ct, st[k], w[i]);
    }
    bw_add(&b, st[1], ct->tableLog);     /* FSE_flushCState(CState2) then (CState1) */
    bw_add(&b, st[0], ct->tableLog);
    return bw_close(&b);
}
/* HUF_compressWeights (huf_compress.c:173-213): 0 = not compressible, 1 = rle */
static size_t huf_compress_weights(u8* dst, const u8* w, size_t n)
{
    u32 count[13] = {0}, maxSym = 0, maxCount = 0, i, tableLog; s16 norm[13]; fse_ct ct; size_t h, c;
    if (n <= 1) return 0;
    for (i = 0; i < n; i++) count[w[i]]++;
    for (i = 0; i <= 12; i++) { if (count[i]) maxSym = i; if (count[i] > maxCount) maxCount = count[i]; }
    if (maxCount == n) return 1;
    if (maxCount == 1) return 0;
    tableLog = fse_optimal_log(6, n, maxSym, 2);
    if (fse_normalize(norm, tableLog, count, n, maxSym, 0) < 0) return 0;
    h = fse_write_ncount(dst, norm, maxSym, tableLog);
    if (!h) return 0;
    fse_build_ctable(&ct, norm, maxSym, tableLog);
    c = fse_compress_weights(dst + h, w, n, &ct);
    if (!c) return 0;
    return h + c;
}
/* HUF_writeCTable_wksp (huf_compress.c:256-295); returns 0 on failure */
static size_t huf_write_ctable(u8* op, const huf_ct* ct, u32 maxSym, u32 huffLog)
{
    u8 weight[256]; u32 n;
    for (n = 0; n < maxSym; n++) weight[n] = ct->nbBits[n] ? (u8)(huffLog + 1 - ct->nbBits[n]) : 0;
    {   const size_t h = huf_compress_weights(op + 1, weight, maxSym);
        if (h > 1 && h < maxSym / 2) { op[0] = (u8)h; return h + 1; }
    }
    if (maxSym > 128) return 0;
    op[0] = (u8)(128 + (maxSym - 1));
    weight[maxSym] = 0;
    for (n = 0; n < maxSym; n += 2) op[(n / 2) + 1] = (u8)((weight[n] << 4) + weight[n + 1]);
    return ((maxSym + 1) / 2) + 1;
}
/* HUF_compress1X_usingCTable (huf_compress.c:1051-1116): symbols last -> first, then the end mark */
static size_t huf_encode_1x(u8* dst, const u8* src, size_t n, const huf_ct* ct)
{
    bw_t b; size_t i;
    bw_init(&b, dst);
    for (i = n; i-- > 0;) bw_add(&b, ct->val[src[i]], ct->nbBits[src[i]]);
    return bw_close(&b);
}
static size_t huf_encode_4x(u8* dst, const u8* src, size_t n, const huf_ct* ct)
{   /* huf_compress.c:1163-1211 */
    const size_t seg = (n + 3) / 4; u8* op = dst + 6; size_t c; int i;
    if (n < 12) return 0;
    for (i = 0; i < 3; i++) {
        c = huf_encode_1x(op, src + (size_t)i * seg, seg, ct);
        if (c == 0 || c > 65535) return 0;
        dst[2 * i] = (u8)c; dst[2 * i + 1] = (u8)(c >> 8); op += c;
    }
    c = huf_encode_1x(op, src + 3 * seg, n - 3 * seg, ct);
    if (c == 0 || c > 65535) return 0;
    op += c;
    return (size_t)(op - dst);
}

/* HUF_compress_internal (huf_compress.c:1326-1434): 0 = not compressible, 1 = rle.  `old` holds the table of the last
 * confirmed block (all-zero lengths before the first) and `*repeat` its state (0 = HUF_repeat_none, 1 = HUF_repeat_check);
 * when a new table is written it replaces `old` and `*repeat` becomes 0, when the old one is reused `*repeat` stays 1 and
 * the output has no table description. */
static u32 huf_estimate(const huf_ct* ct, const u32* count, u32 maxSym)
{   /* HUF_estimateCompressedSize :809-818 */
    size_t nbBits = 0; u32 s2;
    for (s2 = 0; s2 <= maxSym; s2++) nbBits += (size_t)ct->nbBits[s2] * count[s2];
    return (u32)(nbBits >> 3);
}
static size_t huf_encode(u8* dst, const u8* src, size_t n, int single, const huf_ct* ct, size_t h)
{   /* HUF_compressCTable_internal :1216-1240 with `h` header bytes already at dst */
    const size_t c = single ? huf_encode_1x(dst + h, src, n, ct) : huf_encode_4x(dst + h, src, n, ct);
    if (!c) return 0;
    if (h + c >= n - 1) return 0;
    return h + c;
}
static size_t huf_compress(u8* dst, const u8* src, size_t n, int single, int suspect, int preferRepeat, huf_ct* old, int* repeat)
{
    u32 count[256], maxSym = 255, largest = 0, i, huffLog; huf_ct ct; size_t h;
    if (!n) return 0;
    if (suspect && n >= 4096 * 10) {                                  /* :1361-1375 */
        u32 cnt[256], lb = 0, le = 0;
        memset(cnt, 0, sizeof(cnt)); for (i = 0; i < 4096; i++) cnt[src[i]]++;
        for (i = 0; i < 256; i++) if (cnt[i] > lb) lb = cnt[i];
        memset(cnt, 0, sizeof(cnt)); for (i = 0; i < 4096; i++) cnt[src[n - 4096 + i]]++;
        for (i = 0; i < 256; i++) if (cnt[i] > le) le = cnt[i];
        if (lb + le <= ((2 * 4096) >> 7) + 4) return 0;
    }
    memset(count, 0, sizeof(count));
    for (i = 0; i < n; i++) count[src[i]]++;
    maxSym = 0;
    for (i = 0; i < 256; i++) { if (count[i]) maxSym = i; if (count[i] > largest) largest = count[i]; }
    if (largest == n) { dst[0] = src[0]; return 1; }
    if (largest <= (n >> 7) + 4) return 0;
    if (*repeat) {                                                    /* HUF_validateCTable :796-807, :1386-1391 */
        int bad = 0;
        for (i = 0; i <= maxSym; i++) bad |= (count[i] != 0) & (old->nbBits[i] == 0);
        if (bad) *repeat = 0;
    }
    if (preferRepeat && *repeat) return huf_encode(dst, src, n, single, old, 0);   /* :1392-1397 */
    huffLog = fse_optimal_log(11, n, maxSym, 1);                      /* HUF_optimalTableLog :1268-1283 */
    huffLog = huf_build(&ct, count, maxSym, huffLog);
    h = huf_write_ctable(dst, &ct, maxSym, huffLog);
    if (!h) return 0;                                                 /* error -> raw literals */
    if (*repeat) {                                                    /* :1419-1426 */
        const u32 oldSize = huf_estimate(old, count, maxSym), newSize = huf_estimate(&ct, count, maxSym);
        if (oldSize <= h + newSize || h + 12 >= n) return huf_encode(dst, src, n, single, old, 0);
    }
    if (h + 12 >= n) return 0;
    *repeat = 0; *old = ct;                                           /* :1429-1432 */
    return huf_encode(dst, src, n, single, &ct, h);
}

/* ZSTD_compressLiterals for a first block (zstd_compress_literals.c:129-235) */
static size_t lit_raw(u8* dst, const u8* src, size_t n, u32 type)
{
    const u32 fl = 1 + (n > 31) + (n > 4095);
    if (fl == 1) dst[0] = (u8)(type + (n << 3));
    else if (fl == 2) { const u32 v = type + (1 << 2) + ((u32)n << 4); dst[0] = (u8)v; dst[1] = (u8)(v >> 8); }
    else { const u32 v = type + (3 << 2) + ((u32)n << 4); dst[0] = (u8)v; dst[1] = (u8)(v >> 8); dst[2] = (u8)(v >> 16); }
    if (type == 0) { memcpy(dst + fl, src, n); return fl + n; }
    dst[fl] = src[0]; return fl + 1;
}
/* Huffman state carried from block to block of one frame (ZSTD_hufCTables_t, zstd_compress_internal.h:70-73) */
typedef struct { huf_ct ct; int repeat; int disabled; } huf_state;   /* disabled: ZSTD_literalsCompressionIsDisabled */
static size_t compress_literals(u8* dst, const u8* src, size_t n, int suspect, const huf_state* prev, huf_state* next)
{
    const size_t lh = 3 + (n >= 1024) + (n >= 16384);
    const int single = n < 256;
    int repeat = prev->repeat; u32 hType;
    size_t c;
    *next = *prev;                                                    /* :144 */
    if (prev->disabled) return lit_raw(dst, src, n, 0);               /* :146-147 */
    if (n < 64) return lit_raw(dst, src, n, 0);                       /* ZSTD_minLiteralsToCompress: 8 << 3 */
    c = huf_compress(dst + lh, src, n, single, suspect, n <= 1024, &next->ct, &repeat);   /* preferRepeat: strategy < lazy, :164 */
    hType = repeat ? 3 : 2;                                           /* set_repeat : set_compressed, :185-189 */
    if (c == 0 || c >= n - ((n >> 6) + 2)) { *next = *prev; return lit_raw(dst, src, n, 0); }   /* ZSTD_minGain */
    if (c == 1) { *next = *prev; return lit_raw(dst, src, n, 1); }    /* n >= 64 >= 8: rle */
    if (hType == 2) next->repeat = 1;                                 /* a new table: HUF_repeat_check, :213-216 */
    if (lh == 3) { const u32 v = hType + ((u32)(!single) << 2) + ((u32)n << 4) + ((u32)c << 14); dst[0] = (u8)v; dst[1] = (u8)(v >> 8); dst[2] = (u8)(v >> 16); }
    else if (lh == 4) { const u32 v = hType + (2 << 2) + ((u32)n << 4) + ((u32)c << 18); memcpy(dst, &v, 4); }
    else { const u32 v = hType + (3 << 2) + ((u32)n << 4) + ((u32)c << 22); memcpy(dst, &v, 4); dst[4] = (u8)(c >> 10); }
    return lh + c;
}

/* ------------------------------------------------------------------------------------------
 * sequences
 * ------------------------------------------------------------------------------------------ */
typedef struct { u32 offBase; u32 litLength; u32 matchLength; } zseq;   /* full-width values */

static const u8 kLLbits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
static const u8 kMLbits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
static const s16 kLLnorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
static const s16 kMLnorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                 1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
static const s16 kOFnorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

static u32 ll_code(u32 v)
{   /* zstd_compress_internal.h:527-540 */
    if (v > 63) return (u32)z_hb(v) + 19;
    if (v < 16) return v;
    if (v < 24) return 16 + ((v - 16) >> 1);
    if (v < 32) return 20 + ((v - 24) >> 2);
    if (v < 48) return 22 + ((v - 32) >> 3);
    return 24;
}
static u32 ml_code(u32 v)
{   /* zstd_compress_internal.h:545-556 (v = matchLength - 3) */
    if (v > 127) return (u32)z_hb(v) + 36;
    if (v < 32) return v;
    if (v < 40) return 32 + ((v - 32) >> 1);
    if (v < 48) return 36 + ((v - 40) >> 2);
    if (v < 64) return 38 + ((v - 48) >> 3);
    if (v < 96) return 39 + ((v - 48) >> 4);    /* 64..79 -> 40, 80..95 -> 41 */
    return 42;
}

/* ZSTD_selectEncodingType for strategy ZSTD_fast, first block (zstd_compress_sequences.c:156-204): 0 basic,1 rle,2 compressed */
static u32 g_strategy = 1;      /* ZSTD_strategy of the frame being written: 1 fast, 2 double-fast (set by orc_zstd_compress_cparams) */
static int select_type(u32 mostFrequent, size_t nbSeq, u32 defaultNormLog, int defaultAllowed)
{
    if (mostFrequent == nbSeq) return (defaultAllowed && nbSeq <= 2) ? 0 : 1;
    if (defaultAllowed) {
        const size_t dynMin = (((size_t)1 << defaultNormLog) * (10 - g_strategy)) >> 3;   /* mult = 10 - strategy (:184-186) */
        if (nbSeq < dynMin || mostFrequent < (nbSeq >> (defaultNormLog - 1))) return 0;
    }
    return 2;
}

/* one of the three symbol streams: histogram -> mode -> table description -> CTable (ZSTD_buildCTable :242-288) */
static int build_stream(u8** op, fse_ct* ct, const u8* codes, size_t nbSeq, u32 maxCode, u32 fseLog,
                        const s16* defNorm, u32 defLog, u32 defMax, int* lastCountSize)
{
    u32 count[64], max = 0, most = 0, i; int type, defaultAllowed;
    memset(count, 0, sizeof(count));
    for (i = 0; i < nbSeq; i++) count[codes[i]]++;
    for (i = 0; i <= maxCode; i++) { if (count[i]) max = i; if (count[i] > most) most = count[i]; }
    defaultAllowed = (defMax == 28) ? (max <= 28) : 1;                /* zstd_compress.c:2847-2848 */
    type = select_type(most, nbSeq, defLog, defaultAllowed);
    if (type == 1) { fse_build_rle(ct, max); *(*op)++ = codes[0]; }
    else if (type == 0) fse_build_ctable(ct, defNorm, defMax, defLog);
    else {
        s16 norm[64]; size_t nbSeq1 = nbSeq, h;
        const u32 tableLog = fse_optimal_log(fseLog, nbSeq, max, 2);
        if (count[codes[nbSeq - 1]] > 1) { count[codes[nbSeq - 1]]--; nbSeq1--; }
        if (fse_normalize(norm, tableLog, count, nbSeq1, max, nbSeq1 >= 2048) < 0) return -1;
        h = fse_write_ncount(*op, norm, max, tableLog);
        if (!h) return -1;
        fse_build_ctable(ct, norm, max, tableLog);
        *op += h; *lastCountSize = (int)h;
    }
    return type;
}

/* ZSTD_entropyCompressSeqStore_internal (zstd_compress.c:2910-3027); returns block content size, 0 = emit raw */
static size_t entropy_compress(u8* dst, const u8* lits, size_t litSize, const zseq* seqs, size_t nbSeq, const huf_state* prev, huf_state* next)
{
    u8* op = dst;
    const int suspect = (nbSeq == 0) || (litSize / nbSeq >= 20);
    op += compress_literals(op, lits, litSize, suspect, prev, next);
    if (nbSeq < 128) *op++ = (u8)nbSeq;
    else if (nbSeq < 0x7F00) { op[0] = (u8)((nbSeq >> 8) + 0x80); op[1] = (u8)nbSeq; op += 2; }
    else { op[0] = 0xFF; op[1] = (u8)(nbSeq - 0x7F00); op[2] = (u8)((nbSeq - 0x7F00) >> 8); op += 3; }
    if (nbSeq == 0) return (size_t)(op - dst);
    {   u8* llc = (u8*)malloc(nbSeq * 3); u8* ofc = llc + nbSeq; u8* mlc = ofc + nbSeq;
        fse_ct ctLL, ctOF, ctML; int tLL, tOF, tML, lastCount = 0; size_t i, bs; u8* seqHead; bw_t b;
        u32 sLL, sOF, sML;
        for (i = 0; i < nbSeq; i++) {                                 /* ZSTD_seqToCodes :2715-2741 */
            llc[i] = (u8)(seqs[i].litLength > 0xFFFF ? 35 : ll_code(seqs[i].litLength));
            ofc[i] = (u8)z_hb(seqs[i].offBase);
            mlc[i] = (u8)(seqs[i].matchLength - 3 > 0xFFFF ? 52 : ml_code(seqs[i].matchLength - 3));
        }
        seqHead = op++;
        tLL = build_stream(&op, &ctLL, llc, nbSeq, 35, 9, kLLnorm, 6, 35, &lastCount);
        tOF = tLL < 0 ? -1 : build_stream(&op, &ctOF, ofc, nbSeq, 31, 8, kOFnorm, 5, 28, &lastCount);
        tML = tOF < 0 ? -1 : build_stream(&op, &ctML, mlc, nbSeq, 52, 9, kMLnorm, 6, 52, &lastCount);
        if (tML < 0) { free(llc); return 0; }
        /* lastCount = size of the LAST table description written in compressed mode (zstd_compress.c:2836,2866,2894) */
        *seqHead = (u8)((tLL << 6) + (tOF << 4) + (tML << 2));
        /* ZSTD_encodeSequences_body (zstd_compress_sequences.c:290-382), 64-bit, no long offsets */
        bw_init(&b, op);
        sML = fse_init_state(&ctML, mlc[nbSeq - 1]); sOF = fse_init_state(&ctOF, ofc[nbSeq - 1]); sLL = fse_init_state(&ctLL, llc[nbSeq - 1]);
        bw_add(&b, seqs[nbSeq - 1].litLength, kLLbits[llc[nbSeq - 1]]);
        bw_add(&b, seqs[nbSeq - 1].matchLength - 3, kMLbits[mlc[nbSeq - 1]]);
        bw_add(&b, seqs[nbSeq - 1].offBase, ofc[nbSeq - 1]);
        for (i = nbSeq - 1; i-- > 0;) {
            sOF = fse_encode(&b, &ctOF, sOF, ofc[i]);
            sML = fse_encode(&b, &ctML, sML, mlc[i]);
            sLL = fse_encode(&b, &ctLL, sLL, llc[i]);
            bw_add(&b, seqs[i].litLength, kLLbits[llc[i]]);
            bw_add(&b, seqs[i].matchLength - 3, kMLbits[mlc[i]]);
            bw_add(&b, seqs[i].offBase, ofc[i]);
        }
        bw_add(&b, sML, ctML.tableLog); bw_add(&b, sOF, ctOF.tableLog); bw_add(&b, sLL, ctLL.tableLog);
        bs = bw_close(&b);
        free(llc);
        op += bs;
        if (lastCount && (size_t)lastCount + bs < 4) return 0;        /* zstd_compress.c:3017-3023 */
    }
    return (size_t)(op - dst);
}

/* ------------------------------------------------------------------------------------------
 * match finder: AOCL_ZSTD_compressBlock_fast_noDict_generic (zstd_fast.c:427-706), fresh context:
 * table indices are position + 2 (ZSTD_WINDOW_START_INDEX), so 0 means "empty" (< prefixStartIndex)
 * ------------------------------------------------------------------------------------------ */
static u32 z_hash(const u8* p, u32 hlog, u32 mls)
{
    const u64 v = z_rd64(p);
    if (mls == 4) return ((u32)v * 2654435761u) >> (32 - hlog);                   /* ZSTD_hash4 (zstd_compress_internal.h:836-841) */
    if (mls == 5) return (u32)(((v << 24) * 889523592379ull) >> (64 - hlog));
    if (mls == 6) return (u32)(((v << 16) * 227718039650203ull) >> (64 - hlog));
    return (u32)(((v << 8) * 58295818150454627ull) >> (64 - hlog));               /* ZSTD_hash7 (:848-850) */
}
static u32 z_count(const u8* s, u32 a, u32 b, u32 end)
{   /* ZSTD_count: common prefix of s[a..end) and s[b..] */
    u32 n = 0; while (a + n < end && s[a + n] == s[b + n]) n++; return n;
}
#define PUT(pos) table[z_hash(src + (pos), hlog, mls)] = (pos) + 2

/* window state of one frame (ZSTD_window_t, zstd_compress_internal.h:178-185): indices are position + 2 */
typedef struct { u32 dictLimit, lowLimit, maxDist; } zwin;
/* ZSTD_getLowestPrefixIndex (zstd_compress_internal.h:1305-1320), no dictionary */
static u32 lowest_prefix(const zwin* w, u32 curr) { return (curr - w->dictLimit > w->maxDist) ? curr - w->maxDist : w->dictLimit; }

/* One block [bs, be) of a frame whose bytes start at src[0]: `table` and `rep` carry over from the previous block.
 * Returns the number of sequences; positions are offsets into src, table entries position + 2. */
/* stepSize of the fast parser (zstd_fast.c:440): targetLength + !targetLength + 1 - 2 for levels >= -1, 1 - level below
 * (set by orc_zstd_compress_frame_cp for the frame it writes) */
static u32 g_fast_step = 2;
static size_t fast_block_at(const u8* src, u32 bs, u32 be, u32* table, u32 hlog, u32 mls, const zwin* w, u32 rep[2],
                            zseq* seqs, u8* lits, size_t* litSizeOut)
{
    const u32 n = be;                                                 /* iend */
    const u32 prefixIdx = lowest_prefix(w, be + 2), prefixPos = prefixIdx - 2;     /* :443-446 */
    size_t nbSeq = 0, litSize = 0;
    u32 anchor = bs, ip0 = bs + (bs == prefixPos), rep1 = rep[0], rep2 = rep[1], saved1 = 0, saved2 = 0;
    {   const u32 curr = ip0 + 2, maxRep = curr - lowest_prefix(w, curr);          /* :481-489 */
        if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
        if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }
    }
    if (n < 16 + 4) goto cleanup;                                     /* ilimit = n - 16 must leave room for ip3 */
    {   const u32 ilimit = n - 16;
        for (;;) {                                                    /* _start */
            u32 step = g_fast_step, nextStep = ip0 + 128, ip1 = ip0 + 1, ip2 = ip0 + step, ip3 = ip2 + 1;
            u32 hash0, hash1, idx, current0 = 0, match0 = 0, offcode = 0, mLength = 0;
            int found = 0;
            if (ip3 >= ilimit) goto cleanup;
            hash0 = z_hash(src + ip0, hlog, mls); hash1 = z_hash(src + ip1, hlog, mls);
            idx = table[hash0];
            do {
                const u32 rval = (ip2 >= rep1) ? z_rd32(src + ip2 - rep1) : 0;
                current0 = ip0; table[hash0] = ip0 + 2;
                if (rep1 > 0 && z_rd32(src + ip2) == rval) {          /* repcode at ip2 (:529-542) */
                    u32 back;
                    ip0 = ip2; match0 = ip0 - rep1;
                    back = (src[ip0 - 1] == src[match0 - 1]);
                    ip0 -= back; match0 -= back; offcode = 1; mLength = back + 4;
                    table[hash1] = ip1 + 2;
                    found = 2; break;
                }
                if (idx >= prefixIdx && z_rd32(src + idx - 2) == z_rd32(src + ip0)) { table[hash1] = ip1 + 2; found = 1; break; }
                idx = table[hash1];
                hash0 = hash1; hash1 = z_hash(src + ip2, hlog, mls);
                ip0 = ip1; ip1 = ip2; ip2 = ip3;
                current0 = ip0; table[hash0] = ip0 + 2;
                if (idx >= prefixIdx && z_rd32(src + idx - 2) == z_rd32(src + ip0)) {
                    if (step <= 4) table[hash1] = ip1 + 2;
                    found = 1; break;
                }
                idx = table[hash1];
                hash0 = hash1; hash1 = z_hash(src + ip2, hlog, mls);
                ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
                if (ip2 >= nextStep) { step++; nextStep += 128; }
            } while (ip3 < ilimit);
            if (!found) goto cleanup;
            if (found == 1) {                                         /* _offset (:651-665) */
                match0 = idx - 2;
                rep2 = rep1; rep1 = ip0 - match0; offcode = rep1 + 3; mLength = 4;
                while (ip0 > anchor && match0 > prefixPos && src[ip0 - 1] == src[match0 - 1]) { ip0--; match0--; mLength++; }
            }
            mLength += z_count(src, ip0 + mLength, match0 + mLength, n);      /* _match */
            seqs[nbSeq].litLength = ip0 - anchor; seqs[nbSeq].offBase = offcode; seqs[nbSeq].matchLength = mLength; nbSeq++;
            memcpy(lits + litSize, src + anchor, ip0 - anchor); litSize += ip0 - anchor;
            ip0 += mLength; anchor = ip0;
            if (ip0 <= ilimit) {
                PUT(current0 + 2);
                PUT(ip0 - 2);
                if (rep2 > 0) {
                    while (ip0 <= ilimit && z_rd32(src + ip0) == z_rd32(src + ip0 - rep2)) {
                        const u32 rLength = z_count(src, ip0 + 4, ip0 + 4 - rep2, n) + 4;
                        const u32 t = rep2; rep2 = rep1; rep1 = t;
                        PUT(ip0);
                        ip0 += rLength;
                        seqs[nbSeq].litLength = 0; seqs[nbSeq].offBase = 1; seqs[nbSeq].matchLength = rLength; nbSeq++;
                        anchor = ip0;
                    }
                }
            }
        }
    }
cleanup:
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;            /* :626-630 */
    rep[0] = rep1 ? rep1 : saved1;
    rep[1] = rep2 ? rep2 : saved2;
    memcpy(lits + litSize, src + anchor, n - anchor); litSize += n - anchor;     /* ZSTD_storeLastLiterals */
    *litSizeOut = litSize;
    return nbSeq;
}
/* the only block of a fresh context: rep {1,4,8}, where 4 > maxRep = 1 is parked (:484-489) */
static size_t fast_block(const u8* src, u32 n, u32 hlog, u32 mls, zseq* seqs, u8* lits, size_t* litSizeOut)
{
    u32* table = (u32*)calloc((size_t)1 << hlog, sizeof(u32));
    const zwin w = { 2, 2, 1u << 31 };
    u32 rep[2] = { 1, 4 };
    const size_t nbSeq = fast_block_at(src, 0, n, table, hlog, mls, &w, rep, seqs, lits, litSizeOut);
    free(table);
    return nbSeq;
}

/* ------------------------------------------------------------------------------------------
 * match finder of levels 3-4: AOCL_ZSTD_compressBlock_doubleFast_noDict_generic (zstd_double_fast.c:332-573, the variant
 * the default build dispatches to; AOCL_ZSTD_SEARCH_SKIP_OPT_DFAST_FAST is off, so it is the upstream parse :99-305 with
 * ilimit = iend - 8 - 19, :351).  Two tables: `L` hashes 8 bytes (hashLog bits), `S` hashes minMatch bytes (chainLog
 * bits); fresh context: indices are position + 2, a candidate is usable when its index is > 2 (:438, :456).
 * ------------------------------------------------------------------------------------------ */
static u32 z_hash8(const u8* p, u32 hlog) { return (u32)((z_rd64(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - hlog)); }   /* ZSTD_hash8 */

/* One block [bs, be) of a frame: the tables and `rep` carry over from the previous block (positions are offsets into src,
 * table entries position + 2; a candidate is usable when its index is above the prefix start, :438, :456) */
static size_t dfast_block_at(const u8* src, u32 bs, u32 be, u32* L, u32* S, u32 hlogL, u32 hlogS, u32 mls, const zwin* w, u32 rep[2],
                             zseq* seqs, u8* lits, size_t* litSizeOut)
{
    const u32 n = be;
    const u32 prefixIdx = lowest_prefix(w, be + 2), prefixPos = prefixIdx - 2;     /* :347-348 */
    size_t nbSeq = 0, litSize = 0;
    u32 anchor = bs, ip = bs + (bs == prefixPos), off1 = rep[0], off2 = rep[1], saved1 = 0, saved2 = 0;
    const int64_t ilimit = (int64_t)n - 8 - 19;
    {   const u32 curr0 = ip + 2, maxRep = curr0 - lowest_prefix(w, curr0);        /* :393-399 */
        if (off2 > maxRep) { saved2 = off2; off2 = 0; }
        if (off1 > maxRep) { saved1 = off1; off1 = 0; }
    }
#define DF_STORE(ll_, ob_, ml_) do { seqs[nbSeq].litLength = (ll_); seqs[nbSeq].offBase = (ob_); seqs[nbSeq].matchLength = (ml_); nbSeq++; \
        memcpy(lits + litSize, src + anchor, (ll_)); litSize += (ll_); } while (0)
    for (;;) {                                                        /* one iteration per sequence (:402) */
        u32 step = 1, nextStep = ip + 256, ip1 = ip + 1;
        u32 hl0, hl1 = 0, idxl0, idxl1 = 0, idxs0 = 0, curr = 0, mLength, offset, m;
        if ((int64_t)ip1 > ilimit) goto cleanup;
        hl0 = z_hash8(src + ip, hlogL); idxl0 = L[hl0];
        do {                                                          /* one iteration per position (:419) */
            const u32 hs0 = z_hash(src + ip, hlogS, mls);
            idxs0 = S[hs0];
            curr = ip;                                                /* (position; the tables hold curr + 2) */
            L[hl0] = S[hs0] = curr + 2;
            if (off1 > 0 && z_rd32(src + ip + 1 - off1) == z_rd32(src + ip + 1)) {      /* repcode at ip+1 (:425-431) */
                mLength = z_count(src, ip + 1 + 4, ip + 1 + 4 - off1, n) + 4;
                ip++;
                DF_STORE(ip - anchor, 1, mLength);
                goto match_stored;
            }
            hl1 = z_hash8(src + ip1, hlogL);
            if (idxl0 > prefixIdx && z_rd64(src + idxl0 - 2) == z_rd64(src + ip)) {     /* long match at ip (:436-449) */
                m = idxl0 - 2;
                mLength = z_count(src, ip + 8, m + 8, n) + 8;
                offset = ip - m;
                while (ip > anchor && m > prefixPos && src[ip - 1] == src[m - 1]) { ip--; m--; mLength++; }
                goto match_found;
            }
            idxl1 = L[hl1];
            if (idxs0 > prefixIdx && z_rd32(src + idxs0 - 2) == z_rd32(src + ip)) goto search_next_long;   /* short match (:454-460) */
            if (ip1 >= nextStep) { step++; nextStep += 256; }
            ip = ip1; ip1 += step;
            hl0 = hl1; idxl0 = idxl1;
        } while ((int64_t)ip1 <= ilimit);
        goto cleanup;
search_next_long:
        if (idxl1 > prefixIdx && z_rd64(src + idxl1 - 2) == z_rd64(src + ip1)) {        /* a long match one position on wins (:495-504) */
            m = idxl1 - 2;
            ip = ip1;
            mLength = z_count(src, ip + 8, m + 8, n) + 8;
            offset = ip - m;
            while (ip > anchor && m > prefixPos && src[ip - 1] == src[m - 1]) { ip--; m--; mLength++; }
            goto match_found;
        }
        m = idxs0 - 2;
        mLength = z_count(src, ip + 4, m + 4, n) + 4;
        offset = ip - m;
        while (ip > anchor && m > prefixPos && src[ip - 1] == src[m - 1]) { ip--; m--; mLength++; }
match_found:
        off2 = off1; off1 = offset;
        if (step < 4) L[hl1] = ip1 + 2;                               /* :516-526 */
        DF_STORE(ip - anchor, offset + 3, mLength);
match_stored:
        ip += mLength; anchor = ip;
        if ((int64_t)ip <= ilimit) {
            const u32 ins = curr + 2;                                 /* complementary insertions (:550-554) */
            L[z_hash8(src + ins, hlogL)] = ins + 2;
            L[z_hash8(src + ip - 2, hlogL)] = ip - 2 + 2;
            S[z_hash(src + ins, hlogS, mls)] = ins + 2;
            S[z_hash(src + ip - 1, hlogS, mls)] = ip - 1 + 2;
            while ((int64_t)ip <= ilimit && off2 > 0 && z_rd32(src + ip) == z_rd32(src + ip - off2)) {   /* :558-572 */
                const u32 rLength = z_count(src, ip + 4, ip + 4 - off2, n) + 4;
                const u32 t = off2; off2 = off1; off1 = t;
                S[z_hash(src + ip, hlogS, mls)] = ip + 2;
                L[z_hash8(src + ip, hlogL)] = ip + 2;
                DF_STORE(0, 1, rLength);
                ip += rLength; anchor = ip;
            }
        }
    }
cleanup:
#undef DF_STORE
    saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;            /* :478-486 */
    rep[0] = off1 ? off1 : saved1;
    rep[1] = off2 ? off2 : saved2;
    memcpy(lits + litSize, src + anchor, n - anchor); litSize += n - anchor;
    *litSizeOut = litSize;
    return nbSeq;
}
static size_t dfast_block(const u8* src, u32 n, u32 hlogL, u32 hlogS, u32 mls, zseq* seqs, u8* lits, size_t* litSizeOut)
{
    u32* L = (u32*)calloc((size_t)1 << hlogL, sizeof(u32));
    u32* S = (u32*)calloc((size_t)1 << hlogS, sizeof(u32));
    const zwin w = { 2, 2, 1u << 31 };
    u32 rep[2] = { 1, 4 };                                            /* rep {1,4,8}: 4 > maxRep = 1 is parked (:393-399) */
    const size_t nbSeq = dfast_block_at(src, 0, n, L, S, hlogL, hlogS, mls, &w, rep, seqs, lits, litSizeOut);
    free(L); free(S);
    return nbSeq;
}

/* cParams for levels 1 and 2 and a known source size <= 128 KiB (clevels.h:80-82,107-108 + ZSTD_adjustCParams_internal
 * zstd_compress.c:1575-1595): both levels use the "fast" strategy with step 2 there; level 2 has a 2^15 table and minMatch
 * 5 (4 for inputs of at most 16 KiB) */
static void level_params(int level, size_t n, u32* hlog, u32* mls)
{
    const u32 srcLog = (n < 64) ? 6 : (u32)z_hb((u32)n - 1) + 1;
    if (n <= 16384) { const u32 w = srcLog < 14 ? srcLog : 14; *hlog = (15 < w + 1) ? 15 : w + 1; *mls = level == 2 ? 4 : 5; }
    else { const u32 w = srcLog < 17 ? srcLog : 17, hl = level == 2 ? 15 : 13; *hlog = (hl < w + 1) ? hl : w + 1; *mls = level == 2 ? 5 : 6; }
}
static void level1_params(size_t n, u32* hlog, u32* mls) { level_params(1, n, hlog, mls); }

size_t orc_zstd_compress_level(const u8* src, size_t n, u8* dst, size_t dstCap, int level);
size_t orc_zstd_compress(const u8* src, size_t n, u8* dst, size_t dstCap) { return orc_zstd_compress_level(src, n, dst, dstCap, 1); }

/* One single-block frame (n <= 128 KiB) with explicit compression parameters, as ZSTD_compress_advanced makes it:
 * strategy 1 = fast (hashLog, minMatch; chainLog unused), 2 = double-fast (hashLog = long table, chainLog = short table,
 * minMatch = bytes hashed into the short table).  Returns the frame size, 0 on failure. */
size_t orc_zstd_compress_cparams(const u8* src, size_t n, u8* dst, size_t dstCap, int strategy, u32 hashLog, u32 chainLog, u32 minMatch)
{
    u8* op = dst;
    if (n > (128u << 10) || dstCap < orc_zstd_compress_bound(n) || dstCap < 9) return 0;
    if (strategy != 1 && strategy != 2) return 0;
    /* frame header: magic, FHD (single segment, FCS size code), FCS (zstd_compress.c:4586-4632) */
    op[0] = 0x28; op[1] = 0xB5; op[2] = 0x2F; op[3] = 0xFD; op += 4;
    {   const u32 fcs = (n >= 256) + (n >= 65536 + 256);
        *op++ = (u8)((1 << 5) | (fcs << 6));
        if (fcs == 0) *op++ = (u8)n;
        else if (fcs == 1) { const u32 v = (u32)n - 256; *op++ = (u8)v; *op++ = (u8)(v >> 8); }
        else { const u32 v = (u32)n; memcpy(op, &v, 4); op += 4; }
    }
    if (n == 0) { op[0] = 1; op[1] = 0; op[2] = 0; return (size_t)(op + 3 - dst); }   /* ZSTD_writeEpilogue */
    {   size_t cSize = 0;
        if (n >= 7) {                                                 /* ZSTD_buildSeqStore :3268 (MIN_CBLOCK_SIZE+3+1+1) */
            size_t litSize, nbSeq;
            zseq* seqs = (zseq*)malloc(sizeof(zseq) * (n / 3 + 8));
            u8* lits = (u8*)malloc(n + 8);
            u8* tmp = (u8*)malloc(2 * n + 1024);
            g_strategy = (u32)strategy;
            nbSeq = strategy == 1 ? fast_block(src, (u32)n, hashLog, minMatch, seqs, lits, &litSize)
                                  : dfast_block(src, (u32)n, hashLog, chainLog, minMatch, seqs, lits, &litSize);
            {   huf_state hprev, hnext; memset(&hprev, 0, sizeof(hprev));
                cSize = entropy_compress(tmp, lits, litSize, seqs, nbSeq, &hprev, &hnext); }
            g_strategy = 1;
            if (cSize >= n - ((n >> 6) + 2)) cSize = 0;               /* ZSTD_minGain, zstd_compress.c:3055-3057 */
            if (cSize) memcpy(op + 3, tmp, cSize);
            free(seqs); free(lits); free(tmp);
        }
        if (cSize == 0) {                                             /* ZSTD_noCompressBlock */
            const u32 h = 1 + (0 << 1) + ((u32)n << 3);
            op[0] = (u8)h; op[1] = (u8)(h >> 8); op[2] = (u8)(h >> 16);
            memcpy(op + 3, src, n); op += 3 + n;
        } else {
            const u32 h = 1 + (2 << 1) + ((u32)cSize << 3);
            op[0] = (u8)h; op[1] = (u8)(h >> 8); op[2] = (u8)(h >> 16); op += 3 + cSize;
        }
    }
    return (size_t)(op - dst);
}

/* A whole frame of any size with the "fast" strategy, as ZSTD_compress_advanced makes it (contentSizeFlag 1, no checksum):
 * ZSTD_compress_frameChunk (zstd_compress.c:4504-4583) cuts the input into 128 KiB blocks that share the hash table, the
 * repeat offsets and the Huffman table of the last block written compressed; ZSTD_compressBlock_internal (:4333-4395)
 * turns a block that is one repeated byte into an RLE block unless it is the first; a block that ends raw or RLE leaves
 * the repeat offsets and the entropy tables as they were (:4385-4387).  FSE tables are never repeated at strategies below
 * "lazy" without a dictionary (ZSTD_selectEncodingType, zstd_compress_sequences.c:166-196: set_repeat needs
 * FSE_repeat_valid, which only a dictionary sets).  Returns the frame size, 0 on failure. */
size_t orc_zstd_compress_frame_cp(const u8* src, size_t n, u8* dst, size_t dstCap, int strategy, u32 windowLog, u32 hashLog, u32 chainLog,
                                  u32 minMatch, u32 targetLength)
{
    u8* op = dst;
    /* blockSize = MIN(ZSTD_BLOCKSIZE_MAX, windowSize) (ZSTD_resetCCtx_internal, zstd_compress.c:2129-2131) */
    const size_t blockMax = windowLog < 17 ? (size_t)1 << windowLog : 128u << 10;
    if (n >= (1u << 31) || dstCap < orc_zstd_compress_bound(n) || dstCap < 18 || windowLog < 10 || windowLog > 31) return 0;
    if (strategy != 1 && strategy != 2) return 0;
    if (strategy == 1 && targetLength > (1u << 17)) return 0;
    /* ZSTD_writeFrameHeader (zstd_compress.c:4586-4632) */
    op[0] = 0x28; op[1] = 0xB5; op[2] = 0x2F; op[3] = 0xFD; op += 4;
    {   const u32 single = ((u64)1 << windowLog) >= n;
        const u32 fcs = (n >= 256) + (n >= 65536 + 256);
        *op++ = (u8)((single << 5) | (fcs << 6));
        if (!single) *op++ = (u8)((windowLog - 10) << 3);
        if (fcs == 0) { if (single) *op++ = (u8)n; }
        else if (fcs == 1) { const u32 v = (u32)n - 256; *op++ = (u8)v; *op++ = (u8)(v >> 8); }
        else { const u32 v = (u32)n; memcpy(op, &v, 4); op += 4; }
    }
    if (n == 0) { op[0] = 1; op[1] = 0; op[2] = 0; return (size_t)(op + 3 - dst); }   /* ZSTD_writeEpilogue */
    {   u32* table = (u32*)calloc((size_t)1 << hashLog, sizeof(u32));
        u32* tableS = strategy == 2 ? (u32*)calloc((size_t)1 << chainLog, sizeof(u32)) : NULL;
        zseq* seqs = (zseq*)malloc(sizeof(zseq) * (blockMax / 3 + 8));
        u8* lits = (u8*)malloc(blockMax + 8);
        u8* tmp = (u8*)malloc(2 * blockMax + 1024);
        zwin w = { 2, 2, 1u << windowLog };
        u32 rep[2] = { 1, 4 };
        huf_state hprev, hnext;
        size_t bs;
        memset(&hprev, 0, sizeof(hprev));
        hprev.disabled = strategy == 1 && targetLength > 0;           /* fast with a target length: literals stay raw (zstd_compress_internal.h:609-622) */
        g_strategy = (u32)strategy;
        g_fast_step = strategy == 1 ? targetLength + !targetLength + 1 : 2;
        for (bs = 0; bs < n; bs += blockMax) {
            const size_t bn = n - bs < blockMax ? n - bs : blockMax;
            const u32 last = bs + bn == n;
            size_t cSize = 0;
            if ((u32)bs + 2 > w.maxDist) {                            /* ZSTD_window_enforceMaxDist at the block's start (:4532) */
                const u32 low = (u32)bs + 2 - w.maxDist;
                if (w.lowLimit < low) w.lowLimit = low;
                if (w.dictLimit < w.lowLimit) w.dictLimit = w.lowLimit;
            }
            if (bn >= 7) {                                            /* ZSTD_buildSeqStore :3268 */
                size_t litSize, nbSeq; u32 nrep[2] = { rep[0], rep[1] };
                nbSeq = strategy == 1 ? fast_block_at(src, (u32)bs, (u32)(bs + bn), table, hashLog, minMatch, &w, nrep, seqs, lits, &litSize)
                                      : dfast_block_at(src, (u32)bs, (u32)(bs + bn), table, tableS, hashLog, chainLog, minMatch, &w, nrep, seqs, lits, &litSize);
                cSize = entropy_compress(tmp, lits, litSize, seqs, nbSeq, &hprev, &hnext);
                if (cSize >= bn - ((bn >> 6) + 2)) cSize = 0;         /* ZSTD_minGain, zstd_compress.c:3055-3057 */
                if (bs != 0 && cSize < 25) {                          /* ZSTD_isRLE, :4372-4383 */
                    size_t i = 1; while (i < bn && src[bs + i] == src[bs]) i++;
                    if (i == bn) { cSize = 1; tmp[0] = src[bs]; }
                }
                if (cSize > 1) { rep[0] = nrep[0]; rep[1] = nrep[1]; hprev = hnext; }   /* confirmRepcodesAndEntropyTables */
            }
            if (cSize == 0) {                                         /* ZSTD_noCompressBlock */
                const u32 h = last + (0 << 1) + ((u32)bn << 3);
                op[0] = (u8)h; op[1] = (u8)(h >> 8); op[2] = (u8)(h >> 16);
                memcpy(op + 3, src + bs, bn); op += 3 + bn;
            } else if (cSize == 1) {
                const u32 h = last + (1 << 1) + ((u32)bn << 3);
                op[0] = (u8)h; op[1] = (u8)(h >> 8); op[2] = (u8)(h >> 16); op[3] = tmp[0]; op += 4;
            } else {
                const u32 h = last + (2 << 1) + ((u32)cSize << 3);
                op[0] = (u8)h; op[1] = (u8)(h >> 8); op[2] = (u8)(h >> 16);
                memcpy(op + 3, tmp, cSize); op += 3 + cSize;
            }
        }
        g_strategy = 1;
        free(table); free(tableS); free(seqs); free(lits); free(tmp);
        g_fast_step = 2;
    }
    return (size_t)(op - dst);
}
size_t orc_zstd_compress_frame_tl(const u8* src, size_t n, u8* dst, size_t dstCap, u32 windowLog, u32 hashLog, u32 minMatch, u32 targetLength)
{
    return orc_zstd_compress_frame_cp(src, n, dst, dstCap, 1, windowLog, hashLog, 0, minMatch, targetLength);
}

size_t orc_zstd_compress_frame(const u8* src, size_t n, u8* dst, size_t dstCap, u32 windowLog, u32 hashLog, u32 minMatch)
{
    return orc_zstd_compress_frame_tl(src, n, dst, dstCap, windowLog, hashLog, minMatch, 0);
}

/* level 1 or 2 (the "fast" rows of clevels.h for inputs <= 128 KiB) */
size_t orc_zstd_compress_level(const u8* src, size_t n, u8* dst, size_t dstCap, int level)
{
    u32 hlog = 0, mls = 0;
    if (level != 1 && level != 2) return 0;
    if (n >= 7 && n <= (128u << 10)) level_params(level, n, &hlog, &mls);
    return orc_zstd_compress_cparams(src, n, dst, dstCap, 1, hlog, 0, mls);
}

/* ==========================================================================================
 * Decoder: ZSTD_decompress (zstd_decompress.c:1252-1389, :1005-1111; zstd_decompress_block.c:162-364, :497-707,
 * :1339-1460, :1007-1210, :2444-2552; huf_decompress.c; fse_decompress.c; entropy_common.c:40-200,234-310).
 * Restated from the frame format; errors return (size_t)-1.
 * ========================================================================================== */
#define ZERR ((size_t)-1)
/* backward bit reader (bitstream.h:262-420): the stream ends with a 1-bit end mark in its last byte */
typedef struct { const u8* p; int64_t pos; } br_t;     /* pos = number of unread bits (may go negative) */
static int br_init(br_t* b, const u8* p, size_t n)
{
    if (n == 0 || p[n - 1] == 0) return -1;
    b->p = p; b->pos = (int64_t)(n - 1) * 8 + z_hb(p[n - 1]);
    return 0;
}
static u32 br_read(br_t* b, int nb)
{   /* bits [pos-nb, pos); bits before the start of the buffer read as 0 (the reference's "overflow" state) */
    u32 v = 0; int i;
    for (i = 0; i < nb; i++) {
        const int64_t bit = b->pos - nb + i;
        if (bit >= 0) v |= (u32)((b->p[bit >> 3] >> (bit & 7)) & 1) << i;
    }
    b->pos -= nb;
    return v;
}

typedef struct { u16 newState; u8 symbol; u8 nbBits; } fse_dt;

/* FSE_readNCount (entropy_common.c:40-186); returns bytes consumed or ZERR */
static size_t fse_read_ncount(s16* norm, u32* maxSym, u32* tableLog, const u8* src, size_t n, u32 maxLog)
{
    u64 bp = 0; u32 charnum = 0, nbBits, previous0 = 0; int remaining, threshold;
    const u32 maxSV1 = *maxSym + 1;
#define NC_PEEK() ({ u32 _v = 0; int _i; for (_i = 0; _i < 32; _i++) { const u64 _b = bp + _i; if ((_b >> 3) < n) _v |= (u32)((src[_b >> 3] >> (_b & 7)) & 1) << _i; } _v; })
    memset(norm, 0, maxSV1 * sizeof(s16));
    if (n < 1) return ZERR;
    nbBits = (NC_PEEK() & 0xF) + FSE_MINLOG; bp += 4;
    if (nbBits > 15 || nbBits > maxLog) return ZERR;
    *tableLog = nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
    for (;;) {
        if (previous0) {
            for (;;) {
                const u32 c = NC_PEEK() & 3; bp += 2;
                if (c == 3) { charnum += 3; continue; }
                charnum += c; break;
            }
            if (charnum >= maxSV1) break;
        }
        {   const u32 bits = NC_PEEK();
            const int max = (2 * threshold - 1) - remaining; int count;
            if ((bits & (u32)(threshold - 1)) < (u32)max) { count = (int)(bits & (u32)(threshold - 1)); bp += nbBits - 1; }
            else { count = (int)(bits & (u32)(2 * threshold - 1)); if (count >= threshold) count -= max; bp += nbBits; }
            count--;
            remaining -= count < 0 ? -count : count;
            norm[charnum++] = (s16)count;
            previous0 = !count;
            if (remaining < threshold) {
                if (remaining <= 1) break;
                nbBits = (u32)z_hb((u32)remaining) + 1; threshold = 1 << (nbBits - 1);
            }
            if (charnum >= maxSV1) break;
        }
    }
#undef NC_PEEK
    if (remaining != 1 || charnum > maxSV1) return ZERR;
    *maxSym = charnum - 1;
    {   const size_t used = (size_t)((bp + 7) >> 3);
        if (used > n) return ZERR;
        return used;
    }
}
/* FSE decoding table (zstd_decompress_block.c:497-615 / fse_decompress.c:58-140) */
static int fse_build_dtable(fse_dt* dt, const s16* norm, u32 maxSym, u32 tableLog)
{
    const u32 size = 1u << tableLog, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    u16 next[256]; u32 high = size - 1, s, pos = 0, u;
    for (s = 0; s <= maxSym; s++) {
        if (norm[s] == -1) { dt[high--].symbol = (u8)s; next[s] = 1; }
        else next[s] = (u16)norm[s];
    }
    for (s = 0; s <= maxSym; s++) {
        int i;
        for (i = 0; i < norm[s]; i++) {
            dt[pos].symbol = (u8)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    if (pos != 0) return -1;
    for (u = 0; u < size; u++) {
        const u8 sym = dt[u].symbol; const u32 ns = next[sym]++;
        dt[u].nbBits = (u8)(tableLog - (u32)z_hb(ns));
        dt[u].newState = (u16)((ns << dt[u].nbBits) - size);
    }
    return 0;
}

/* Huffman literals table: weights -> (symbol, nbBits) lookup of 2^tableLog cells (huf_decompress.c:380-470) */
typedef struct { u8 symbol[4096]; u8 nbBits[4096]; u32 tableLog; } huf_dt;

static size_t huf_read_table(huf_dt* dt, const u8* src, size_t n)
{   /* HUF_readStats (entropy_common.c:234-310) + X1 table fill */
    u8 w[256]; u32 nsym, total = 0, i, tableLog, rank[14] = {0}; size_t used;
    if (!n) return ZERR;
    if (src[0] >= 128) {
        nsym = src[0] - 127; used = 1 + (nsym + 1) / 2;
        if (used > n) return ZERR;
        for (i = 0; i < nsym; i += 2) { w[i] = src[1 + i / 2] >> 4; w[i + 1] = src[1 + i / 2] & 15; }
    } else {
        /* FSE_decompress_wksp (fse_decompress.c:175-300): two interleaved states, max table log 6 */
        s16 norm[256]; u32 maxSym = 255, tl; fse_dt fdt[64]; br_t b; size_t h; u32 s1, s2;
        const size_t isz = src[0];
        used = 1 + isz;
        if (used > n || isz < 2) return ZERR;
        h = fse_read_ncount(norm, &maxSym, &tl, src + 1, isz, 6);
        if (h == ZERR || h >= isz) return ZERR;
        if (fse_build_dtable(fdt, norm, maxSym, tl)) return ZERR;
        if (br_init(&b, src + 1 + h, isz - h)) return ZERR;
        s1 = br_read(&b, (int)tl); s2 = br_read(&b, (int)tl);
        nsym = 0;
        for (;;) {
            if (nsym > 253) return ZERR;
            w[nsym++] = fdt[s1].symbol; s1 = fdt[s1].newState + br_read(&b, fdt[s1].nbBits);
            if (b.pos < 0) { w[nsym++] = fdt[s2].symbol; break; }
            if (nsym > 253) return ZERR;
            w[nsym++] = fdt[s2].symbol; s2 = fdt[s2].newState + br_read(&b, fdt[s2].nbBits);
            if (b.pos < 0) { w[nsym++] = fdt[s1].symbol; break; }
        }
    }
    for (i = 0; i < nsym; i++) { if (w[i] > 12) return ZERR; rank[w[i]]++; total += (1u << w[i]) >> 1; }
    if (total == 0) return ZERR;
    tableLog = (u32)z_hb(total) + 1;
    if (tableLog > 12) return ZERR;
    {   const u32 rest = (1u << tableLog) - total;
        if ((1u << z_hb(rest)) != rest) return ZERR;
        w[nsym] = (u8)(z_hb(rest) + 1); rank[w[nsym]]++; nsym++;
    }
    if (rank[1] < 2 || (rank[1] & 1)) return ZERR;
    {   u32 start[14], wgt, cur = 0;
        for (wgt = 1; wgt <= tableLog; wgt++) { start[wgt] = cur; cur += rank[wgt] << (wgt - 1); }
        for (i = 0; i < nsym; i++) if (w[i]) {
            const u32 len = 1u << (w[i] - 1), nb = tableLog + 1 - w[i]; u32 k;
            for (k = 0; k < len; k++) { dt->symbol[start[w[i]] + k] = (u8)i; dt->nbBits[start[w[i]] + k] = (u8)nb; }
            start[w[i]] += len;
        }
    }
    dt->tableLog = tableLog;
    return used;
}
/* lenient: the stream is one of four that the reference decodes with its fast loops (huf_decompress.c
 * HUF_decompress4X{1,2}_usingDTable_internal_fast, :790-840 / :1600-1655): those make no end-of-stream check - a stream that
 * holds MORE bits than its symbols need decodes to the same symbols and is accepted (bits left over), and a last byte of zero
 * is no error there (HUF_initFastDStream :167-176: all eight bits are data).  A stream that holds FEWER bits than its symbols
 * need is still refused here: the reference goes on reading the bytes in front of it, which this restatement does not follow. */
static int huf_decode_stream(u8* out, size_t nout, const u8* src, size_t n, const huf_dt* dt, int lenient)
{
    br_t b; size_t i; const int tl = (int)dt->tableLog;
    if (lenient && n != 0 && src[n - 1] == 0) { b.p = src; b.pos = (int64_t)n * 8; }
    else if (br_init(&b, src, n)) return -1;
    for (i = 0; i < nout; i++) {
        u32 idx; int k;
        /* peek tableLog bits (zero-padded below the start of the stream) */
        idx = 0;
        for (k = 0; k < tl; k++) { const int64_t bit = b.pos - tl + k; if (bit >= 0) idx |= (u32)((b.p[bit >> 3] >> (bit & 7)) & 1) << k; }
        out[i] = dt->symbol[idx]; b.pos -= dt->nbBits[idx];
    }
    return (b.pos == 0 || (lenient && b.pos > 0)) ? 0 : -1;   /* strict decoder (BIT_endOfDStream): every stream must be consumed exactly */
}

typedef struct {
    fse_dt ll[512], of[256], ml[512]; u32 llLog, ofLog, mlLog; int haveLL, haveOF, haveML;
    huf_dt huf; int haveHuf;
    u32 rep[3];
} zdec_state;

static const u32 kLLbase[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000 };
static const u32 kMLbase[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,
                                 35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003 };

static int seq_table(fse_dt* dt, u32* log, int* have, int mode, const u8** ip, const u8* iend, u32 maxSym, u32 maxLog,
                     const s16* defNorm, u32 defMax, u32 defLog)
{   /* ZSTD_buildSeqTable (zstd_decompress_block.c:659-705) */
    if (mode == 0) { fse_build_dtable(dt, defNorm, defMax, defLog); *log = defLog; *have = 1; return 0; }
    if (mode == 1) {
        if (*ip >= iend || **ip > maxSym) return -1;
        dt[0].symbol = **ip; dt[0].nbBits = 0; dt[0].newState = 0; *log = 0; *have = 1; (*ip)++; return 0;
    }
    if (mode == 2) {
        s16 norm[64]; u32 ms = maxSym, tl; const size_t h = fse_read_ncount(norm, &ms, &tl, *ip, (size_t)(iend - *ip), maxLog);
        if (h == ZERR) return -1;
        if (fse_build_dtable(dt, norm, ms, tl)) return -1;
        *log = tl; *have = 1; *ip += h; return 0;
    }
    return *have ? 0 : -1;                           /* repeat */
}

/* one compressed block (zstd_decompress_block.c:2444-2552); returns decoded size or ZERR */
static size_t decode_block(zdec_state* st, const u8* src, size_t n, u8* dst, size_t opos, size_t cap, size_t blockMax)
{
    const u8* ip = src; const u8* const iend = src + n;
    u8* lits = NULL; const u8* litPtr; size_t litSize, out = opos;
    if (n < 1) return ZERR;                          /* MIN_CBLOCK_SIZE */
    {   /* literals (ZSTD_decodeLiteralsBlock :162-364) */
        const u32 type = ip[0] & 3, fmt = (ip[0] >> 2) & 3;
        if (type < 2) {
            size_t lh;
            if (fmt == 0 || fmt == 2) { lh = 1; litSize = ip[0] >> 3; }
            else if (fmt == 1) { if (n < 2) return ZERR; lh = 2; litSize = (ip[0] | ((u32)ip[1] << 8)) >> 4; }
            else { if (n < 3) return ZERR; lh = 3; litSize = (ip[0] | ((u32)ip[1] << 8) | ((u32)ip[2] << 16)) >> 4; }
            if (litSize > blockMax) return ZERR;
            if (type == 0) { if (lh + litSize > n) return ZERR; litPtr = ip + lh; ip += lh + litSize; }
            else {
                if (lh + 1 > n) return ZERR;
                lits = (u8*)malloc(litSize + 1); memset(lits, ip[lh], litSize); litPtr = lits; ip += lh + 1;
            }
        } else {
            size_t lh, csize; int streams4 = 1;
            if (n < 5 && 0) return ZERR;
            if (fmt == 0 || fmt == 1) { if (n < 3) return ZERR; { const u32 v = ip[0] | ((u32)ip[1] << 8) | ((u32)ip[2] << 16); lh = 3; streams4 = (fmt == 1); litSize = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; } }
            else if (fmt == 2) { if (n < 4) return ZERR; { const u32 v = z_rd32(ip); lh = 4; litSize = (v >> 4) & 0x3FFF; csize = v >> 18; } }
            else { if (n < 5) return ZERR; { const u32 v = z_rd32(ip); lh = 5; litSize = (v >> 4) & 0x3FFFF; csize = (v >> 22) + ((size_t)ip[4] << 10); } }
            if (litSize > blockMax || litSize == 0) return ZERR;
            if (!streams4 && 0) return ZERR;
            if (lh + csize > n) return ZERR;
            {   const u8* hp = ip + lh; size_t hsz = csize;
                if (type == 2) {
                    const size_t used = huf_read_table(&st->huf, hp, hsz);
                    if (used == ZERR) return ZERR;
                    st->haveHuf = 1; hp += used; hsz -= used;
                } else if (!st->haveHuf) return ZERR;
                lits = (u8*)malloc(litSize + 8);
                if (!streams4) { if (huf_decode_stream(lits, litSize, hp, hsz, &st->huf, 0)) { free(lits); return ZERR; } }
                else {
                    size_t s1, s2, s3, s4; const size_t seg = (litSize + 3) / 4;
                    if (litSize < 6 || hsz < 10) { free(lits); return ZERR; }   /* huf_decompress.c: 4X needs >= 10 bytes, zstd :251 */
                    s1 = hp[0] | ((size_t)hp[1] << 8); s2 = hp[2] | ((size_t)hp[3] << 8); s3 = hp[4] | ((size_t)hp[5] << 8);
                    if (6 + s1 + s2 + s3 > hsz) { free(lits); return ZERR; }
                    s4 = hsz - 6 - s1 - s2 - s3;
                    if (3 * seg > litSize) { free(lits); return ZERR; }
                    {   /* the reference's fast loops run when every decoding table has log 11 - they all do: HUF_rescaleStats :320-347,
                         * HUF_readDTableX2_wksp :1157 - the streams have at least 16 / 8 / 8 / 8 bytes and the fourth segment is not empty
                         * (HUF_DecompressFastArgs_init :184-270); otherwise the strict decoder with its end-of-stream check */
                        const int len = s1 >= 16 && s2 >= 8 && s3 >= 8 && s4 >= 8 && 3 * seg < litSize;
                        if (huf_decode_stream(lits, seg, hp + 6, s1, &st->huf, len) || huf_decode_stream(lits + seg, seg, hp + 6 + s1, s2, &st->huf, len)
                         || huf_decode_stream(lits + 2 * seg, seg, hp + 6 + s1 + s2, s3, &st->huf, len)
                         || huf_decode_stream(lits + 3 * seg, litSize - 3 * seg, hp + 6 + s1 + s2 + s3, s4, &st->huf, len)) { free(lits); return ZERR; }
                    }
                }
                litPtr = lits; ip += lh + csize;
            }
        }
    }
    {   /* sequences header (ZSTD_decodeSeqHeaders :707-790) */
        size_t nbSeq, litPos = 0;
        if (ip >= iend) { free(lits); return ZERR; }
        nbSeq = *ip++;
        if (nbSeq > 0x7F) {
            if (nbSeq == 0xFF) { if (ip + 2 > iend) { free(lits); return ZERR; } nbSeq = (ip[0] | ((size_t)ip[1] << 8)) + 0x7F00; ip += 2; }
            else { if (ip >= iend) { free(lits); return ZERR; } nbSeq = ((nbSeq - 0x80) << 8) + *ip++; }
        }
        if (nbSeq == 0) {
            if (ip != iend) { free(lits); return ZERR; }
        } else {
            br_t b; u32 sLL, sOF, sML; size_t i; u8 modes;
            if (ip >= iend) { free(lits); return ZERR; }
            modes = *ip++;
            /* (the two reserved bits of the modes byte are not looked at: ZSTD_decodeSeqHeaders, zstd_decompress_block.c:739-743) */
            if (seq_table(st->ll, &st->llLog, &st->haveLL, modes >> 6, &ip, iend, 35, 9, kLLnorm, 35, 6)
             || seq_table(st->of, &st->ofLog, &st->haveOF, (modes >> 4) & 3, &ip, iend, 31, 8, kOFnorm, 28, 5)
             || seq_table(st->ml, &st->mlLog, &st->haveML, (modes >> 2) & 3, &ip, iend, 52, 9, kMLnorm, 52, 6)) { free(lits); return ZERR; }
            if (br_init(&b, ip, (size_t)(iend - ip))) { free(lits); return ZERR; }
            sLL = br_read(&b, (int)st->llLog); sOF = br_read(&b, (int)st->ofLog); sML = br_read(&b, (int)st->mlLog);
            if (b.pos < 0) { free(lits); return ZERR; }
            for (i = 0; i < nbSeq; i++) {           /* ZSTD_decodeSequence :1339-1460 + execSequence :1007-1108 */
                const u32 llc = st->ll[sLL].symbol, ofc = st->of[sOF].symbol, mlc = st->ml[sML].symbol;
                u32 ofv, ml, ll; size_t offset;
                if (ofc > 31) { free(lits); return ZERR; }
                ofv = (1u << ofc) + br_read(&b, (int)ofc);
                ml = kMLbase[mlc] + br_read(&b, kMLbits[mlc]);
                ll = kLLbase[llc] + br_read(&b, kLLbits[llc]);
                if (ofv > 3) { offset = ofv - 3; st->rep[2] = st->rep[1]; st->rep[1] = st->rep[0]; st->rep[0] = (u32)offset; }
                else {
                    const u32 idx = ofv - 1 + (ll == 0);
                    if (idx == 0) offset = st->rep[0];
                    else {
                        u32 t = (idx == 3) ? st->rep[0] - 1 : st->rep[idx];
                        t += !t;
                        if (idx != 1) st->rep[2] = st->rep[1];
                        st->rep[1] = st->rep[0]; st->rep[0] = t; offset = t;
                    }
                }
                if (i + 1 < nbSeq) {
                    sLL = st->ll[sLL].newState + br_read(&b, st->ll[sLL].nbBits);
                    sML = st->ml[sML].newState + br_read(&b, st->ml[sML].nbBits);
                    sOF = st->of[sOF].newState + br_read(&b, st->of[sOF].nbBits);
                }
                if (ll > litSize - litPos) { free(lits); return ZERR; }
                if (out + ll + ml > cap || (out - opos) + ll + ml > blockMax) { free(lits); return ZERR; }
                memcpy(dst + out, litPtr + litPos, ll); out += ll; litPos += ll;
                if (offset > out) { free(lits); return ZERR; }
                { size_t k; for (k = 0; k < ml; k++) dst[out + k] = dst[out - offset + k]; }
                out += ml;
            }
            /* The stream must be consumed exactly.  More bits than the sequences need: the reference refuses too
             * (BIT_reloadDStream < BIT_DStream_completed, zstd_decompress_block.c:1583).  FEWER bits: the reference does not notice -
             * a read below the stream's first bit wraps around its 64-bit container (bitstream.h:343-351, bitsConsumed & 63) and
             * the block decodes to whatever that gives, or fails later; this restatement refuses the block (round 6; until then
             * it read zeros there and, like the GPU decoder, could return other bytes than the reference's with a success value). */
            if (b.pos != 0) { free(lits); return ZERR; }
        }
        if (out + (litSize - litPos) > cap || (out - opos) + (litSize - litPos) > blockMax) { free(lits); return ZERR; }
        memcpy(dst + out, litPtr + litPos, litSize - litPos); out += litSize - litPos;
    }
    free(lits);
    return out - opos;
}

size_t orc_zstd_decompress(const u8* src, size_t n, u8* dst, size_t dstCap)
{
    size_t ip = 0, op = 0;
    if (n == 0) return 0;                             /* ZSTD_decompressMultiFrame (zstd_decompress.c:1123-1197): the frame loop is not entered, nothing is left over */
    while (ip < n) {
        u32 magic;
        if (n - ip < 4) return ZERR;
        magic = z_rd32(src + ip);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                         /* skippable frame */
            u32 sz;
            if (n - ip < 8) return ZERR;
            sz = z_rd32(src + ip + 4);
            if ((size_t)sz + 8 > n - ip) return ZERR;
            ip += 8 + sz; continue;
        }
        if (magic != 0xFD2FB528u) return ZERR;
        {   /* frame header (ZSTD_getFrameHeader_advanced, zstd_decompress.c:438-560) */
            u8 fhd; u32 fcsId, single, dictIdSz, hasCk; u64 windowSize = 0, fcs = 0; int haveFcs = 0; size_t frameStart;
            zdec_state* st;
            if (n - ip < 5) return ZERR;
            fhd = src[ip + 4]; ip += 5;
            fcsId = fhd >> 6; single = (fhd >> 5) & 1; hasCk = (fhd >> 2) & 1; dictIdSz = fhd & 3;
            if (fhd & 0x08) return ZERR;
            if (!single) {
                u32 wl;
                if (ip >= n) return ZERR;
                wl = (src[ip] >> 3) + 10;
                if (wl > 31) return ZERR;                                    /* ZSTD_WINDOWLOG_MAX (64-bit) */
                windowSize = 1ull << wl; windowSize += (windowSize >> 3) * (src[ip] & 7); ip++;
            }
            {   static const u32 dsz[4] = { 0, 1, 2, 4 }; u32 id = 0, k;
                if (n - ip < dsz[dictIdSz]) return ZERR;
                for (k = 0; k < dsz[dictIdSz]; k++) id |= (u32)src[ip + k] << (8 * k);
                ip += dsz[dictIdSz];
                if (id != 0) return ZERR;                                    /* dictionary_wrong: no dictionary loaded */
            }
            if (fcsId == 0) { if (single) { if (ip >= n) return ZERR; fcs = src[ip++]; haveFcs = 1; } }
            else if (fcsId == 1) { if (n - ip < 2) return ZERR; fcs = (u64)(src[ip] | ((u32)src[ip + 1] << 8)) + 256; ip += 2; haveFcs = 1; }
            else if (fcsId == 2) { if (n - ip < 4) return ZERR; fcs = z_rd32(src + ip); ip += 4; haveFcs = 1; }
            else { if (n - ip < 8) return ZERR; fcs = z_rd64(src + ip); ip += 8; haveFcs = 1; }
            if (single) windowSize = fcs;
            frameStart = op;
            st = (zdec_state*)calloc(1, sizeof(zdec_state));
            st->rep[0] = 1; st->rep[1] = 4; st->rep[2] = 8;
            {   const size_t blockMax = windowSize < (128u << 10) ? (size_t)windowSize : (128u << 10);
                for (;;) {                                                   /* ZSTD_decompressFrame :1005-1111 */
                    u32 bh, last, type, bsz;
                    if (n - ip < 3) { free(st); return ZERR; }
                    bh = src[ip] | ((u32)src[ip + 1] << 8) | ((u32)src[ip + 2] << 16); ip += 3;
                    last = bh & 1; type = (bh >> 1) & 3; bsz = bh >> 3;
                    if (type == 3) { free(st); return ZERR; }
                    if (type == 1) {
                        if (ip >= n || bsz > dstCap - op) { free(st); return ZERR; }            /* (no block-size limit on RLE / raw blocks: ZSTD_setRleBlock, ZSTD_copyRawBlock) */
                        memset(dst + op, src[ip], bsz); ip += 1; op += bsz;
                    } else if (type == 0) {
                        if (bsz > n - ip || bsz > dstCap - op) { free(st); return ZERR; }
                        memcpy(dst + op, src + ip, bsz); ip += bsz; op += bsz;
                    } else {
                        size_t r;
                        if (bsz > n - ip || bsz > blockMax) { free(st); return ZERR; }            /* ZSTD_decompressBlock_internal: srcSize > blockSizeMax */
                        r = decode_block(st, src + ip, bsz, dst + frameStart, op - frameStart, dstCap - frameStart, blockMax);
                        if (r == ZERR) { free(st); return ZERR; }
                        ip += bsz; op += r;
                    }
                    if (last) break;
                }
            }
            free(st);
            if (haveFcs && (u64)(op - frameStart) != fcs) return ZERR;
            if (hasCk) { if (n - ip < 4) return ZERR; ip += 4; }              /* content checksum: present but not verified here */
        }
    }
    return op;
}

/* ---- debug hooks: let tests fuzz the building blocks against the reference's exported functions ---- */
int orc_dbg_fse_normalize(int16_t* norm, unsigned tableLog, const unsigned* count, size_t total, unsigned maxSym, int useLowProb)
{ return fse_normalize(norm, tableLog, count, total, maxSym, useLowProb); }
unsigned orc_dbg_huf_build(uint8_t* nbBitsOut, const unsigned* count, unsigned maxSym, unsigned maxNbBits)
{ huf_ct ct; unsigned r = huf_build(&ct, count, maxSym, maxNbBits); memcpy(nbBitsOut, ct.nbBits, 256); return r; }
size_t orc_dbg_write_ncount(uint8_t* out, const int16_t* norm, unsigned maxSym, unsigned tableLog)
{ return fse_write_ncount(out, norm, maxSym, tableLog); }
/* the fast parser's sequences of every block of a frame (debugging aid and reference of tests/test_zstd_chain_model.py): rows
 * {litLength, offBase, matchLength, 0}, and {0xFFFFFFFF, block, last literals, 0} after every block; every block counts as
 * confirmed (repeat offsets carried on).  Returns the number of rows. */
size_t orc_dbg_zstd_frame_sequences(const uint8_t* src, size_t n, unsigned windowLog, unsigned hashLog, unsigned minMatch, uint32_t* out)
{
    const size_t blockMax = windowLog < 17 ? (size_t)1 << windowLog : 128u << 10;
    u32* table = (u32*)calloc((size_t)1 << hashLog, sizeof(u32));
    zseq* seqs = (zseq*)malloc(sizeof(zseq) * (blockMax / 3 + 8));
    u8* lits = (u8*)malloc(blockMax + 8);
    zwin w = { 2, 2, 1u << windowLog };
    u32 rep[2] = { 1, 4 }, blk = 0; size_t rows = 0, bs, i;
    for (bs = 0; bs < n; bs += blockMax, blk++) {
        const size_t bn = n - bs < blockMax ? n - bs : blockMax;
        size_t litSize = bn, nbSeq = 0, sum = 0;
        if ((u32)bs + 2 > w.maxDist) {
            const u32 low = (u32)bs + 2 - w.maxDist;
            if (w.lowLimit < low) w.lowLimit = low;
            if (w.dictLimit < w.lowLimit) w.dictLimit = w.lowLimit;
        }
        if (bn >= 7) nbSeq = fast_block_at(src, (u32)bs, (u32)(bs + bn), table, hashLog, minMatch, &w, rep, seqs, lits, &litSize);
        for (i = 0; i < nbSeq; i++) { out[4 * rows] = seqs[i].litLength; out[4 * rows + 1] = seqs[i].offBase; out[4 * rows + 2] = seqs[i].matchLength; out[4 * rows + 3] = 0; rows++; sum += seqs[i].litLength; }
        out[4 * rows] = 0xFFFFFFFFu; out[4 * rows + 1] = blk; out[4 * rows + 2] = (u32)(litSize - sum); out[4 * rows + 3] = 0; rows++;
    }
    free(table); free(seqs); free(lits);
    return rows;
}

/* code length per symbol of a Huffman table description (debugging aid of tools/debug): returns the table log, 0 on error */
unsigned orc_dbg_huf_lengths(const uint8_t* desc, size_t n, uint8_t* lens256)
{
    huf_dt* dt = (huf_dt*)malloc(sizeof(huf_dt)); u32 i, tl;
    memset(lens256, 0, 256);
    if (!dt || huf_read_table(dt, desc, n) == ZERR) { free(dt); return 0; }
    for (i = 0; i < (1u << dt->tableLog); i++) lens256[dt->symbol[i]] = dt->nbBits[i];
    tl = dt->tableLog; free(dt);
    return tl;
}
size_t orc_dbg_huf_write(uint8_t* out, const unsigned* count, unsigned maxSym, unsigned maxNbBits)
{ huf_ct ct; unsigned r = huf_build(&ct, count, maxSym, maxNbBits); return huf_write_ctable(out, &ct, maxSym, r); }
/* sequences of the match finder: out[4*i] = {litLength, offBase, matchLength, 0}; returns nbSeq */
size_t orc_dbg_zstd_sequences(const uint8_t* src, size_t n, uint32_t* out, size_t* litSize)
{
    u32 hlog, mls; zseq* seqs = (zseq*)malloc(sizeof(zseq) * (n / 3 + 8)); u8* lits = (u8*)malloc(n + 8); size_t i, nb;
    level1_params(n, &hlog, &mls);
    nb = fast_block(src, (u32)n, hlog, mls, seqs, lits, litSize);
    for (i = 0; i < nb; i++) { out[4*i] = seqs[i].litLength; out[4*i+1] = seqs[i].offBase; out[4*i+2] = seqs[i].matchLength; out[4*i+3] = 0; }
    free(seqs); free(lits); return nb;
}
