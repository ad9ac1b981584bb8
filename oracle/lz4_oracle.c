/* oracle/lz4_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the reference's LZ4 level-1 block path, written index-based (positions are
 * offsets from the chunk start) so it reads like the kernel's specification:
 *
 *   compress   : LZ4_compress_default -> LZ4_compress_fast(acc=1) -> AOCL_LZ4_compress_fast_extState
 *                (algos/lz4/lz4.c:2967-2975, 2937-2965, 2513-2541) -> AOCL_LZ4_compress_generic (:2401-2436)
 *                -> AOCL_LZ4_compress_generic_validated (:1371-1845), noDict / noDictIssue, tableType byU16
 *                (srcSize < 65547) or byU32, outputDirective notLimited or limitedOutput.
 *   decompress : LZ4_decompress_safe (:4898-4923) -> AOCL_LZ4_decompress_generic(endOnInputSize,
 *                decode_full_block, noDict) (:3367-3803).  Only the accept/reject behaviour and the decoded
 *                bytes are restated (the reference's wild-copy tricks are an x86 implementation detail).
 *
 * Constants: lz4.c:231-252,694-695; lz4.h:163,252,261.
 */
#include "oracle.h"
#include <string.h>
#include <stdlib.h>

enum {
    LZ4_MINMATCH = 4, LZ4_LASTLITERALS = 5, LZ4_MFLIMIT = 12, LZ4_MINLENGTH = LZ4_MFLIMIT + 1,
    LZ4_ML_BITS = 4, LZ4_ML_MASK = 15, LZ4_RUN_MASK = 15, LZ4_DIST_MAX = 65535,
    LZ4_64KLIMIT = 65536 + (LZ4_MFLIMIT - 1), LZ4_SKIPTRIGGER = 6, LZ4_HASHLOG = 12,
    LZ4_MAX_INPUT = 0x7E000000
};

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

int orc_lz4_compress_bound(int n)
{   /* lz4.h:261 */
    return ((unsigned)n > (unsigned)LZ4_MAX_INPUT) ? 0 : n + n / 255 + 16;
}

/* lz4.c:759-823.  u16 mode: 13-bit hash of 4 bytes; u32 mode: 12-bit hash of the low 5 bytes. */
static uint32_t lz4_hash(const uint8_t* p, int u16mode)
{
    if (u16mode) return (rd32(p) * 2654435761U) >> (32 - (LZ4_HASHLOG + 1));
    return (uint32_t)(((rd64(p) << 24) * 889523592379ULL) >> (64 - LZ4_HASHLOG));
}

/* common prefix length of src[a..] and src[b..] with a bounded by limit (lz4.c:656-679) */
static uint32_t lz4_count(const uint8_t* s, uint32_t a, uint32_t b, uint32_t limit)
{
    uint32_t n = 0;
    while (a + n < limit && s[a + n] == s[b + n]) n++;
    return n;
}

int orc_lz4_compress(const uint8_t* src, int srcSize, uint8_t* dst, int dstCap)
{
    return orc_lz4_compress_fast(src, srcSize, dst, dstCap, 1);
}

/* == LZ4_compress_fast (lz4.c:2937-2965): acceleration < 1 -> 1, > 65537 -> 65537 (lz4.c:2521-2522) */
int orc_lz4_compress_fast(const uint8_t* src, int srcSize, uint8_t* dst, int dstCap, int acceleration)
{
    if (acceleration < 1) acceleration = 1;
    if (acceleration > 65537) acceleration = 65537;
    if (dst == NULL || (src == NULL && srcSize != 0)) return 0;              /* lz4.c:2516 */
    if ((uint32_t)srcSize > (uint32_t)LZ4_MAX_INPUT) return 0;               /* lz4.c:2417 */
    const int limited = dstCap < orc_lz4_compress_bound(srcSize);            /* lz4.c:2522 */
    if (srcSize == 0) {                                                      /* lz4.c:2418-2429 */
        if (limited && dstCap <= 0) return 0;
        dst[0] = 0; return 1;
    }
    const int u16mode = srcSize < LZ4_64KLIMIT;                              /* lz4.c:2523 */
    uint32_t* table = (uint32_t*)calloc(u16mode ? 8192 : 4096, sizeof(uint32_t));  /* LZ4_initStream zero-fill */
    if (!table) return 0;

    const uint32_t iend = (uint32_t)srcSize;
    const uint32_t mflimitPlusOne = iend - LZ4_MFLIMIT + 1;   /* may wrap when iend < 11; guarded by MINLENGTH test */
    const uint32_t matchlimit = iend - LZ4_LASTLITERALS;
    uint32_t ip = 0, anchor = 0;
    int64_t op = 0;
    const int64_t olimit = dstCap;
    int result = 0;

    if (srcSize < LZ4_MINLENGTH) goto last_literals;                          /* lz4.c:1440 */

    table[lz4_hash(src, u16mode)] = 0;                                        /* first byte, lz4.c:1443 */
    ip = 1;
    for (;;) {
        uint32_t match, token_pos;
        /* ---- find a match (lz4.c:1483-1608) ---- */
        {
            uint32_t fwd = ip, step = 1, nb = (uint32_t)acceleration << LZ4_SKIPTRIGGER;     /* lz4.c:1486-1490 */
            uint32_t fwdH = lz4_hash(src + fwd, u16mode);
            for (;;) {
                const uint32_t h = fwdH, cur = fwd;
                const uint32_t cand = table[h];
                ip = fwd;
                fwd += step;
                step = nb++ >> LZ4_SKIPTRIGGER;
                if (fwd > mflimitPlusOne) goto last_literals;
                fwdH = lz4_hash(src + fwd, u16mode);
                table[h] = cur;
                if (rd32(src + cand) == rd32(src + ip)) {
                    if (!u16mode && cand + LZ4_DIST_MAX < cur) continue;      /* too far (byU32 only) */
                    match = cand;
                    break;
                }
            }
        }
        /* ---- catch up (lz4.c:1611-1623) ---- */
        while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1]) { ip--; match--; }

        /* ---- literals (lz4.c:1626-1650) ---- */
        {
            const uint32_t lit = ip - anchor;
            token_pos = (uint32_t)op++;
            if (limited && op + lit + (2 + 1 + LZ4_LASTLITERALS) + lit / 255 > olimit) goto fail;
            if (lit >= LZ4_RUN_MASK) {
                uint32_t len = lit - LZ4_RUN_MASK;
                dst[token_pos] = LZ4_RUN_MASK << LZ4_ML_BITS;
                for (; len >= 255; len -= 255) dst[op++] = 255;
                dst[op++] = (uint8_t)len;
            } else dst[token_pos] = (uint8_t)(lit << LZ4_ML_BITS);
            memcpy(dst + op, src + anchor, lit);
            op += lit;
        }
next_match:
        /* ---- offset + match length (lz4.c:1667-1740) ---- */
        dst[op++] = (uint8_t)(ip - match);
        dst[op++] = (uint8_t)((ip - match) >> 8);
        {
            uint32_t mc = lz4_count(src, ip + LZ4_MINMATCH, match + LZ4_MINMATCH, matchlimit);
            ip += mc + LZ4_MINMATCH;
            if (limited && op + (1 + LZ4_LASTLITERALS) + (mc + 240) / 255 > olimit) goto fail;
            if (mc >= LZ4_ML_MASK) {
                dst[token_pos] += LZ4_ML_MASK;
                mc -= LZ4_ML_MASK;
                for (; mc >= 255; mc -= 255) dst[op++] = 255;
                dst[op++] = (uint8_t)mc;
            } else dst[token_pos] += (uint8_t)mc;
        }
        anchor = ip;
        if (ip >= mflimitPlusOne) break;                                      /* lz4.c:1747 */
        table[lz4_hash(src + ip - 2, u16mode)] = ip - 2;                      /* lz4.c:1750 */
        {   /* test next position (lz4.c:1763-1800) */
            const uint32_t h = lz4_hash(src + ip, u16mode);
            const uint32_t cand = table[h];
            table[h] = ip;
            if ((u16mode || cand + LZ4_DIST_MAX >= ip) && rd32(src + cand) == rd32(src + ip)) {
                token_pos = (uint32_t)op++;
                dst[token_pos] = 0;
                match = cand;
                goto next_match;
            }
        }
        ip++;                                                                 /* lz4.c:1804 */
    }

last_literals:                                                                /* lz4.c:1808-1834 */
    {
        const uint32_t lastRun = iend - anchor;
        if (limited && op + lastRun + 1 + (lastRun + 255 - LZ4_RUN_MASK) / 255 > olimit) goto fail;
        if (lastRun >= LZ4_RUN_MASK) {
            uint32_t acc = lastRun - LZ4_RUN_MASK;
            dst[op++] = LZ4_RUN_MASK << LZ4_ML_BITS;
            for (; acc >= 255; acc -= 255) dst[op++] = 255;
            dst[op++] = (uint8_t)acc;
        } else dst[op++] = (uint8_t)(lastRun << LZ4_ML_BITS);
        memcpy(dst + op, src + anchor, lastRun);
        op += lastRun;
    }
    result = (int)op;
fail:
    free(table);
    return result;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Linked blocks of an LZ4 frame: LZ4F_compressFrame with LZ4F_blockLinked (lz4frame.c:373-419 -> LZ4F_compressUpdate
 * :825-929 -> LZ4F_makeBlock :740-763 -> LZ4F_compressBlock_continue :777-782 -> LZ4_compress_fast_continue lz4.c:3179-3256).
 * The blocks of a one-shot frame lie back to back in the caller's buffer, so every block after the first runs in "prefix
 * mode" (:3217-3223): LZ4_compress_generic_validated (lz4.c:972-1360) with tableType byU32, withPrefix64k, noDictIssue
 * (the prefix is never shorter than 64 KiB: blocks are 64 KiB - 4 MiB), limitedOutput with room for blockSize - 1 bytes.
 * The context carries over: table indices are offsets from the frame's start (currentOffset accumulates, :1033), matches
 * reach at most 65535 bytes back (:1124-1127) - into the previous blocks too - and the catch-up stops at the frame's start
 * (lowLimit = source - dictSize, :1029).  A block that does not fit is stored raw by the caller; its parse stopped where the
 * output ran out and the table keeps what was written until then (:1139).
 * dst receives the blocks back to back; blockLen[k] = compressed size, or size | 0x80000000 for a stored block.
 * Returns the number of blocks, -1 on bad arguments. */
int orc_lz4_compress_linked_blocks(const uint8_t* src, size_t n, size_t blockSize, int acceleration, uint8_t* dst, size_t dstCap,
                                   uint32_t* blockLen)
{
    uint32_t* table;
    size_t bs, out = 0; int nb = 0;
    if (acceleration < 1) acceleration = 1;
    if (acceleration > 65537) acceleration = 65537;
    if (!src || !dst || blockSize < 65536 || n >= 0x7E000000u || dstCap < n) return -1;
    table = (uint32_t*)calloc(4096, sizeof(uint32_t));
    if (!table) return -1;
    for (bs = 0; bs < n; bs += blockSize) {
        const uint32_t b0 = (uint32_t)bs, bn = (uint32_t)(n - bs < blockSize ? n - bs : blockSize), iend = b0 + bn;
        const uint32_t mflimitPlusOne = iend - LZ4_MFLIMIT + 1, matchlimit = iend - LZ4_LASTLITERALS;
        uint8_t* const d = dst + out;
        const int64_t olimit = (int64_t)bn - 1;
        uint32_t ip = b0, anchor = b0;
        int64_t op = 0; int ok = 0;
        if (bn < LZ4_MINLENGTH) goto last_literals;
        table[lz4_hash(src + ip, 0)] = ip;                                    /* first byte of the block */
        ip++;
        for (;;) {
            uint32_t match, token_pos;
            {
                uint32_t fwd = ip, step = 1, nbp = (uint32_t)acceleration << LZ4_SKIPTRIGGER;
                uint32_t fwdH = lz4_hash(src + fwd, 0);
                for (;;) {
                    const uint32_t h = fwdH, cur = fwd;
                    const uint32_t cand = table[h];
                    ip = fwd;
                    fwd += step;
                    step = nbp++ >> LZ4_SKIPTRIGGER;
                    if (fwd > mflimitPlusOne) goto last_literals;
                    fwdH = lz4_hash(src + fwd, 0);
                    table[h] = cur;
                    if (cand + LZ4_DIST_MAX < cur) continue;                  /* too far */
                    if (rd32(src + cand) == rd32(src + ip)) { match = cand; break; }
                }
            }
            while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1]) { ip--; match--; }
            {
                const uint32_t lit = ip - anchor;
                token_pos = (uint32_t)op++;
                if (op + lit + (2 + 1 + LZ4_LASTLITERALS) + lit / 255 > olimit) goto done;
                if (lit >= LZ4_RUN_MASK) {
                    uint32_t len = lit - LZ4_RUN_MASK;
                    d[token_pos] = LZ4_RUN_MASK << LZ4_ML_BITS;
                    for (; len >= 255; len -= 255) d[op++] = 255;
                    d[op++] = (uint8_t)len;
                } else d[token_pos] = (uint8_t)(lit << LZ4_ML_BITS);
                memcpy(d + op, src + anchor, lit);
                op += lit;
            }
next_match:
            d[op++] = (uint8_t)(ip - match);
            d[op++] = (uint8_t)((ip - match) >> 8);
            {
                uint32_t mc = lz4_count(src, ip + LZ4_MINMATCH, match + LZ4_MINMATCH, matchlimit);
                ip += mc + LZ4_MINMATCH;
                if (op + (1 + LZ4_LASTLITERALS) + (mc + 240) / 255 > olimit) goto done;
                if (mc >= LZ4_ML_MASK) {
                    d[token_pos] += LZ4_ML_MASK;
                    mc -= LZ4_ML_MASK;
                    for (; mc >= 255; mc -= 255) d[op++] = 255;
                    d[op++] = (uint8_t)mc;
                } else d[token_pos] += (uint8_t)mc;
            }
            anchor = ip;
            if (ip >= mflimitPlusOne) break;
            table[lz4_hash(src + ip - 2, 0)] = ip - 2;
            {
                const uint32_t h = lz4_hash(src + ip, 0);
                const uint32_t cand = table[h];
                table[h] = ip;
                if (cand + LZ4_DIST_MAX >= ip && rd32(src + cand) == rd32(src + ip)) {
                    token_pos = (uint32_t)op++;
                    d[token_pos] = 0;
                    match = cand;
                    goto next_match;
                }
            }
            ip++;
        }
last_literals:
        {
            const uint32_t lastRun = iend - anchor;
            if (op + lastRun + 1 + (lastRun + 255 - LZ4_RUN_MASK) / 255 > olimit) goto done;
            if (lastRun >= LZ4_RUN_MASK) {
                uint32_t acc = lastRun - LZ4_RUN_MASK;
                d[op++] = LZ4_RUN_MASK << LZ4_ML_BITS;
                for (; acc >= 255; acc -= 255) d[op++] = 255;
                d[op++] = (uint8_t)acc;
            } else d[op++] = (uint8_t)(lastRun << LZ4_ML_BITS);
            memcpy(d + op, src + anchor, lastRun);
            op += lastRun;
            ok = 1;
        }
done:
        if (ok) { blockLen[nb++] = (uint32_t)op; out += (size_t)op; }
        else { memcpy(d, src + b0, bn); blockLen[nb++] = bn | 0x80000000u; out += bn; }   /* LZ4F_makeBlock :750-755 */
    }
    free(table);
    return nb;
}

/* LZ4_decompress_safe: decode_full_block + endOnInputSize rules of the reference's safe loop
 * (lz4.c:3563-3790): literal run either obeys the parsing restrictions (ends <= oend-MFLIMIT and
 * <= iend-8) and is followed by a match, or is the last run and must end exactly at iend; a match must
 * start inside the output produced so far and end <= oend-LASTLITERALS; offset 0 yields a run of
 * zeros (lz4.c:3745-3753: LZ4_write32(op,0) then self-copy).  The reference's x86 shortcut/fast loop
 * (:3420-3561, :3581-3612) accepts a few end-of-block violations this strict restatement rejects; on
 * every stream a conforming encoder can emit the two agree byte for byte.
 * Returns the decoded size, or a negative value on malformed input. */
int orc_lz4_decompress(const uint8_t* src, int srcSize, uint8_t* dst, int dstCap)
{
    if (src == NULL || dst == NULL || dstCap < 0) return -1;                  /* lz4.c:3393 */
    if (dstCap == 0) return (srcSize == 1 && src[0] == 0) ? 0 : -1;           /* lz4.c:3412-3416 */
    if (srcSize <= 0) return -1;                                              /* lz4.c:3418 */
    const int64_t iend = srcSize, oend = dstCap;
    int64_t ip = 0, op = 0;
    for (;;) {
        const uint32_t token = src[ip++];
        int64_t len = token >> LZ4_ML_BITS;
        if (len == LZ4_RUN_MASK) {                                            /* read_variable_length :3331 */
            uint32_t s;
            if (ip >= iend - LZ4_RUN_MASK) return -(int)ip - 1;               /* initial_error */
            do {
                s = src[ip++]; len += s;
                if (ip >= iend - LZ4_RUN_MASK) break;                         /* loop_error: stop reading */
            } while (s == 255);
        }
        if (op + len > oend - LZ4_MFLIMIT || ip + len > iend - (2 + 1 + LZ4_LASTLITERALS)) {
            /* must be the last sequence (lz4.c:3654-3664) */
            if (ip + len != iend || op + len > oend) return -(int)ip - 1;
            memmove(dst + op, src + ip, (size_t)len);
            op += len;
            break;
        }
        memcpy(dst + op, src + ip, (size_t)len);
        ip += len; op += len;
        const uint32_t offset = src[ip] | ((uint32_t)src[ip + 1] << 8);
        ip += 2;
        int64_t ml = token & LZ4_ML_MASK;
        if (ml == LZ4_ML_MASK) {
            uint32_t s;
            do {
                s = src[ip++]; ml += s;
                if (ip >= iend - LZ4_LASTLITERALS + 1) return -(int)ip - 1;   /* loop_error is fatal here */
            } while (s == 255);
        }
        ml += LZ4_MINMATCH;
        if ((int64_t)offset > op) return -(int)ip - 1;                        /* offset outside buffer */
        if (op + ml > oend - LZ4_LASTLITERALS) return -(int)ip - 1;           /* last 5 bytes are literals */
        if (offset == 0) memset(dst + op, 0, (size_t)ml);
        else for (int64_t i = 0; i < ml; i++) dst[op + i] = dst[op - offset + i];
        op += ml;
    }
    return (int)op;
}

